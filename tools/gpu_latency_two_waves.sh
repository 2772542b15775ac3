cd $GRAFT_REPO_ROOT
B=mod-interleaveavx_multithreads-faid_amd/host/dropin_bench
for eb in 3.0 4.2; do for t in 1 2 3; do for w in 1 2; do
  echo -n "eb $eb T $t waves $w: "; LNSFAID_WAVES_PER_CODEWORD=$w timeout -k 10 100 $B --threads $t --calls 200 --eb-n0 $eb --register | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['aggregate_Gbps'], 'Gb/s  per call', d['per_call_ms_mean'], 'p50', d['per_call_ms_p50'])"
done; done; done
for g in 4 16 32; do for w in 1 2; do
  echo -n "one thread, $g groups per call, waves $w: "; LNSFAID_WAVES_PER_CODEWORD=$w timeout -k 10 100 $B --threads 1 --calls 100 --eb-n0 3.0 --register --groups-per-call $g | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['aggregate_Gbps'], 'Gb/s  per call', d['per_call_ms_mean'])"
done; done
