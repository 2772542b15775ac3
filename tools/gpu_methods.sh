# kernel time per DecodeMethod, with and without the bit-flipping stage (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
for cfg in "2" "2 --max-bf 0" "5" "5 --max-bf 0" "1" "4" "3" "0"; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu --no-points --method $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('method $cfg:', d['value'], 'Gb/s; kernel avg', d['roofline']['avg_launch_ms'], 'ms')
" || exit 1
done
