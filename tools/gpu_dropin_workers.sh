# The reference's call shape (host/dropin_bench: T threads x one context x one group per call) against the call combiner's number of
# workers (batches in flight, LNSFAID_COMB_WORKERS) and of batches the members' calls are cut into (LNSFAID_COMB_BATCHES)
cd $GRAFT_REPO_ROOT
B=mod-interleaveavx_multithreads-faid_amd/host/dropin_bench
EB=${1:-3.0}
for w in 2 3 4; do for mult in 1 2 4; do for t in 64 128; do
  b=$((w * mult))
  echo -n "workers $w batches $b T $t: "; LNSFAID_COMB_WORKERS=$w LNSFAID_COMB_BATCHES=$b timeout -k 10 100 $B --threads $t --calls 60 --eb-n0 $EB | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['aggregate_Gbps'], 'Gb/s  per call', d['per_call_ms_mean'], 'p95', d['per_call_ms_p95'])"
done; done; done
