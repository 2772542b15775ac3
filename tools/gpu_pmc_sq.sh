# SQ-level counters of the decode kernel (two passes), run on the GPU box via gpurun
set -x
TAG=${1:-r02}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
(cd $REPO && python3 -c "import bench; print(bench.kernel_source_hash())") > $OUT/kernel_source_hash.txt
(cd $REPO && python3 -c "import bench; print(bench.load_pkg_module('pyabi').load().lnsfaid_version().decode())") > $OUT/library_version.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu > $OUT/p2.json 2> $OUT/p2.err
ls -R $OUT | head -30
