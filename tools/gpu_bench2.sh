set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r01.json
cat gpurun_out/bench_r01.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('points'), d['cpu_baseline'])"
bash tools/gpu_profile.sh r01
