cd $GRAFT_REPO_ROOT
B=mod-interleaveavx_multithreads-faid_amd/host/dropin_bench
for w in 1 2 3 4; do for t in 32 64 128; do
  echo -n "workers $w T $t: "; LNSFAID_COMB_WORKERS=$w timeout -k 10 100 $B --threads $t --calls 60 --eb-n0 3.0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['aggregate_Gbps'], 'Gb/s  per call', d['per_call_ms_mean'], 'p95', d['per_call_ms_p95'])"
done; done
for t in 64; do echo -n "registered workers 2 T $t: "; timeout -k 10 100 $B --threads $t --calls 60 --eb-n0 3.0 --register | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['aggregate_Gbps'], d['per_call_ms_mean'])"; done
echo -n "copy mode T 64: "; LNSFAID_COMB_COPY=1 timeout -k 10 100 $B --threads 64 --calls 60 --eb-n0 3.0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['aggregate_Gbps'], d['per_call_ms_mean'])"
