# full evidence run for a round tag, in two gpurun calls (each within the 1200 s limit):
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh <tag> tests'      GPU tests, bench (unprofiled), the DecodeMethod 5 / 16-QAM line
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh <tag> profiles'   rocprofv3 kernel-trace stats, HBM and SQ counters (separate --pmc passes)
# Afterwards, here: python tools/parse_pmc.py gpurun_out/pmc_<tag> <tag>; python tools/parse_pmc_sq.py gpurun_out/pmc_sq_<tag> <tag>;
# python tools/collect_profiles.py <tag>
set -x
TAG=${1:-r02}
STAGE=${2:-all}
cd $GRAFT_REPO_ROOT
if [ $STAGE = tests ] || [ $STAGE = all ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_$TAG.log 2>&1; tail -3 gpurun_out/gputests_$TAG.log
  timeout -k 10 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json || exit 1
  python -c "import json; d=json.load(open('gpurun_out/bench_$TAG.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('points'), d.get('cpu_baseline'))"
  timeout -k 10 300 python bench.py --method 5 --mod-type 4 --scale 12.5 --eb-n0 8.1 --no-cpu 2>/dev/null | tail -1 > gpurun_out/bench_cfg5_$TAG.json || exit 1
fi
if [ $STAGE = profiles ] || [ $STAGE = all ]; then
  bash tools/gpu_profile.sh $TAG && bash tools/gpu_pmc.sh $TAG && bash tools/gpu_pmc_sq.sh $TAG
fi
