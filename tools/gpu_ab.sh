# A/B of the two decode kernels (four / two rows per lane) per DecodeMethod and Eb/N0; run on the GPU box via gpurun
cd $GRAFT_REPO_ROOT
for m in 2 1 5 4 3; do
  for eb in 3.0 3.6 4.2; do
    for k in 4 2; do
      if [ $k = 2 ]; then export LNSFAID_ROWS_PER_LANE=2; else unset LNSFAID_ROWS_PER_LANE; fi
      timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-points --no-cpu --method $m --eb-n0 $eb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('method $m eb $eb rows/lane $k:', d['value'], 'Gb/s', d['ms_per_step'], 'ms/step, launches', d['roofline']['launches'], 'I', round(d['config']['mean_layered_iterations'],2), 'J', round(d['config']['mean_bf_iterations'],2))"
    done
  done
done
