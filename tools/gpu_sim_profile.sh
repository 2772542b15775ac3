#!/bin/bash
# Where a FER sweep of the host driver spends its time: kernel-trace statistics of lnsfaid_sim --device-frontend, one round per
# Eb/N0 point, with 256 and with 2048 streams.  Output: gpurun_out/sim_profile/
set -x
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/sim_profile
mkdir -p $OUT
cd $REPO/mod-interleaveavx_multithreads-faid_amd/host || exit 1
export TMPDIR=/tmp
for S in 256 2048; do
  t0=$(date +%s%N)
  ./lnsfaid_sim --streams $S --gpus 1 --device-frontend --max-rounds 1 > $OUT/plain_$S.txt 2> $OUT/plain_$S.err || exit 1
  echo "streams $S: $(( ($(date +%s%N) - t0) / 1000000 )) ms wall for one round per Eb/N0 point (6 points, start-up included)" | tee $OUT/wall_$S.txt
  cp Result.txt $OUT/Result_$S.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$S -o stats -- ./lnsfaid_sim --streams $S --gpus 1 --device-frontend --max-rounds 1 > $OUT/prof_$S.txt 2> $OUT/prof_$S.err || exit 1
done
ls -R $OUT | head -40
