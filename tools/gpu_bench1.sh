set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 800 python bench.py --steps 3 --warmup 1 2>&1 | tail -5
