# A/B on the GPU box: compressed messages in registers against streamed through HBM (lnsfaid_select_message_store),
# same build, alternating processes; bench.py's three Eb/N0 points.   usage: bash tools/gpu_ab_msgstore.sh <tag>
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
for run in 1 2; do for mode in regs hbm; do
  LNSFAID_MSG_STORE=$mode timeout -k 10 300 python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/ab_msg_${mode}_$TAG.json || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_msg_${mode}_$TAG.json"))
print("$mode", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], [(p["eb_n0_db"], p["value"]) for p in d.get("points", [])])
PY
done; done | tee gpurun_out/ab_msgstore_$TAG.txt
