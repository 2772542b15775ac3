# where the time of a launch goes: headline batch at 3.0 dB (nothing converges: every codeword runs exactly --max-iter layered and
# --max-bf bit-flipping iterations) with the iteration limits varied; run on the GPU box through gpurun
cd $GRAFT_REPO_ROOT
for v in "10 10" "10 0" "5 10" "5 0" "2 0" "1 0" "20 0" "10 5"; do
  set -- $v
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --no-points --max-iter $1 --max-bf $2 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('MaxIteration $1 maxBFiter $2:', d['roofline']['avg_launch_ms'], 'ms per launch,', d['ms_per_step'], 'ms per step')
" || exit 1
done
