# usage: tools/gpu_variants.sh "name1:-DFLAG1 -DFLAG2" "name2:..."   (run on the GPU box through gpurun)
# POINTS=" " adds the 3.6 / 4.2 dB side measurements.  Rebuilds csrc/liblnsfaid.so with EXTRA flags per variant (the Makefile's flags
# stamp makes every object follow), prints the headline bench of each, and ALWAYS ends on the default build again - also when a
# variant fails to build or to run - so that nothing after it in the same tree uses a variant library by accident.
cd $GRAFT_REPO_ROOT
CSRC=mod-interleaveavx_multithreads-faid_amd/csrc
trap 'make -s -C $CSRC EXTRA= > gpurun_out/variant_restore.build.log 2>&1' EXIT
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  make -s -C $CSRC EXTRA="$flags" > gpurun_out/variant_$name.build.log 2>&1 || { echo "$name: build failed"; continue; }
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu ${POINTS:---no-points} $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', d['value'], 'Gb/s; kernel avg', d['roofline']['avg_launch_ms'], 'ms', ' '.join('%s dB %.1f' % (p['eb_n0_db'], p['value']) for p in d.get('points', [])), '|', d['roofline'].get('library'))
" || exit 1
done
