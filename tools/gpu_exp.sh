cd $GRAFT_REPO_ROOT
for args in "--method 1" "--method 2 --max-bf 0" "--method 2 --max-iter 0" "--method 2 --max-iter 1 --max-bf 0" "--method 5"; do
  echo "== $args"
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-points --no-cpu $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'Gb/s', d['ms_per_step'], 'ms/step', d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['config']['mean_layered_iterations'], d['config']['mean_bf_iterations'])"
done
