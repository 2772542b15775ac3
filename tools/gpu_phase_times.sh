#!/bin/bash
# Launch time of the decode kernel against the numbers of layered / bit-flipping iterations (3.0 dB: nothing converges, every
# iteration runs): T(I, J) = staging + I x (syndrome + layers) + J x bit flipping.  Output: gpurun_out/phase_times.txt
set -o pipefail
out=gpurun_out/phase_times.txt
mkdir -p gpurun_out; : > $out
for cfg in "10 10" "10 0" "5 10" "5 0" "2 0" "1 0" "20 0"; do
  set -- $cfg
  python bench.py --no-cpu --no-dropin --max-iter $1 --max-bf $2 > gpurun_out/pt.json 2> gpurun_out/pt.err || { tail -3 gpurun_out/pt.err; exit 1; }
  python - "$1" "$2" <<'PY' | tee -a $out
import json, sys
d = json.load(open('gpurun_out/pt.json'))
r = d['roofline']
print("max_iter %2s max_bf %2s  launches %d  avg_launch_ms %.4f  ms_per_step %.4f  I %.2f J %.2f" % (sys.argv[1], sys.argv[2], r['launches'], r['avg_launch_ms'], d['ms_per_step'], d['config']['mean_layered_iterations'], d['config']['mean_bf_iterations']))
PY
done
