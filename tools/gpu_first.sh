set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -m2 -E "gfx|Marketing" 
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -40
