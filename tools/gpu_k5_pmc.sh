#!/bin/bash
# SQ counters of the experimental two-waves-per-codeword kernel next to the default one (same passes as tools/gpu_pmc_sq.sh), plus
# where the dispatcher puts the waves.  Output: gpurun_out/two_waves/
set -x
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/two_waves
mkdir -p $OUT
[ -x $REPO/tools/ubench/hwid ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o $REPO/tools/ubench/hwid $REPO/tools/ubench/hwid.hip
(cd $REPO && timeout -k 10 60 ./tools/ubench/hwid > $OUT/hwid.txt 2>&1)
cd /tmp && export TMPDIR=/tmp
export LNSFAID_WAVES_PER_CODEWORD=2
python3 $REPO/bench.py --no-cpu --no-dropin > $OUT/bench_two_waves.json 2> $OUT/bench_two_waves.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu --no-dropin > $OUT/p1.json 2> $OUT/p1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu --no-dropin > $OUT/p2.json 2> $OUT/p2.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES_EQ_64 SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES --output-format csv -d $OUT/p3 -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-points --no-cpu --no-dropin > $OUT/p3.json 2> $OUT/p3.err
ls -R $OUT | head -30
