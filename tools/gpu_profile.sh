# usage: bash tools/gpu_profile.sh <tag>   (runs on the GPU box via gpurun)
set -x
TAG=${1:-r01}
REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o stats -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-points --no-cpu > $REPO/gpurun_out/prof_$TAG/bench_under_rocprof.json 2> $REPO/gpurun_out/prof_$TAG/stderr.log
ls -R $REPO/gpurun_out/prof_$TAG | head -30
