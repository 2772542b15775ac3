# usage: tools/gpu_variants.sh "name1:-DFLAG1 -DFLAG2" "name2:..."   (run on the GPU box through gpurun)
# POINTS=" " adds the 3.6 / 4.2 dB side measurements.  Rebuilds csrc/liblnsfaid.so with extra HIPFLAGS per variant and prints the headline bench of each.
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  rm -f mod-interleaveavx_multithreads-faid_amd/csrc/lnsfaid_kernels.o mod-interleaveavx_multithreads-faid_amd/csrc/lnsfaid_kernel4.o mod-interleaveavx_multithreads-faid_amd/csrc/lnsfaid_capi.o
  make -s -C mod-interleaveavx_multithreads-faid_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. $flags" > gpurun_out/variant_$name.build.log 2>&1 || { echo "$name: build failed"; continue; }
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu ${POINTS:---no-points} $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', d['value'], 'Gb/s; kernel avg', d['roofline']['avg_launch_ms'], 'ms', ' '.join('%s dB %.1f' % (p['eb_n0_db'], p['value']) for p in d.get('points', [])))
" || exit 1
done
