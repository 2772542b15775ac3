# rocprofv3 kernel-trace of the host driver with the device front-end (one Eb/N0 point, one round); run on the GPU box
set -x
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_sim
mkdir -p $OUT /tmp/simrun && cd /tmp/simrun && rm -f *.txt
sed -e 's/StartSNR: 3.3/StartSNR: 3.6/' -e 's/EndSNR: 3.85/EndSNR: 3.65/' $REPO/mod-interleaveavx_multithreads-faid_amd/host/Profile.txt > Profile.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sim -- $REPO/mod-interleaveavx_multithreads-faid_amd/host/lnsfaid_sim --streams 256 --gpus 1 --max-rounds 1 --device-frontend > $OUT/stdout.txt 2> $OUT/stderr.txt
tail -2 $OUT/stdout.txt
cut -c1-150 $OUT/sim_kernel_stats.csv | head -6
