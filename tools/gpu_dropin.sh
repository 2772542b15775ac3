# The drop-in call shape on the GPU box: T host threads x one context each x ONE group of 32 frames per lnsfaid_decode call
# (host/dropin_bench.cpp; reference CSimulate.cpp:136-164, main.cpp:164-172), pageable and registered host buffers.
# usage: bash tools/gpu_dropin.sh <tag> ["T list"] ["Eb/N0 list"] [calls]
TAG=${1:-r03}
TS=${2:-"1 8 32 64"}
EBS=${3:-"3.0 4.2"}
CALLS=${4:-50}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/dropin_$TAG.jsonl
: > $OUT
for eb in $EBS; do for t in $TS; do for reg in "" "--register"; do
  timeout -k 10 120 mod-interleaveavx_multithreads-faid_amd/host/dropin_bench --threads $t --calls $CALLS --eb-n0 $eb $reg >> $OUT || { echo "dropin_bench failed (T=$t eb=$eb $reg)"; exit 1; }
  tail -1 $OUT
done; done; done
