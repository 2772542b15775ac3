# headline-size bench of DecodeMethods 5, 1 and 2 at 3.0 / 3.6 dB (quick A/B of changes that touch the syndrome stage); run via gpurun
cd $GRAFT_REPO_ROOT
for m in 5 1 2; do
  for eb in 3.0 3.6; do
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-points --no-cpu --method $m --eb-n0 $eb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('method $m eb $eb:', d['value'], 'Gb/s', d['ms_per_step'], 'ms/step')"
  done
done
