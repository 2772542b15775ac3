# per-launch kernel times of one decode at several Eb/N0 points (LNSFAID_TRACE=1 prints them); run on the GPU box via gpurun
cd $GRAFT_REPO_ROOT
for eb in 3.6 4.2; do
  echo "== Eb/N0 $eb"
  LNSFAID_TRACE=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-points --no-cpu --eb-n0 $eb 2>&1 | grep -E "lnsfaid\]|\"value\"" | cut -c1-160 | tail -8
done
