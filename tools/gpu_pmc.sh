# HBM traffic counters of the decode kernel: two separate --pmc passes (FETCH_SIZE needs 3 TCC slots,
# WRITE_SIZE 2: /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run on the GPU box via gpurun.
set -x
TAG=${1:-r01}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
(cd $REPO && python3 -c "import bench; print(bench.kernel_source_hash())") > $OUT/kernel_source_hash.txt
(cd $REPO && python3 -c "import bench; print(bench.load_pkg_module('pyabi').load().lnsfaid_version().decode())") > $OUT/library_version.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu > $OUT/l2.json 2> $OUT/l2.err
ls -R $OUT | head
