# quick iteration loop on the GPU box: parity tests (fast subset first) then a short bench
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest ${TESTS:-tests} -m gpu -x -q 2>&1 | tail -15 || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('HEAD', d['value'], 'Gb/s', d['ms_per_step'], 'ms/step; kernel avg', d['roofline']['avg_launch_ms'], 'ms; frac', d['roofline']['frac'])
for p in d.get('points', []): print('POINT', p['eb_n0_db'], p['value'], 'Gb/s launches/step', p['launches_per_step'])
"
