# full evidence run for a round tag: GPU tests, bench (unprofiled), rocprofv3 kernel-trace stats, PMC traffic
set -x
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_$TAG.json
python -c "import json; d=json.load(open('gpurun_out/bench_$TAG.json')); print(d['value'], d['ms_per_step'], d['roofline'], d.get('points'), d.get('cpu_baseline'))"
bash tools/gpu_profile.sh $TAG
bash tools/gpu_pmc.sh $TAG
