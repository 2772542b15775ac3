#!/bin/bash
# lnsfaid_kernel5.hip (two waves per codeword): where its waves land, parity, A/B timing against the one-wave kernel
set -o pipefail
mkdir -p gpurun_out/k5
timeout -k 10 60 ./tools/ubench/hwid > gpurun_out/k5/hwid.txt 2>&1; tail -12 gpurun_out/k5/hwid.txt
export LNSFAID_WAVES_PER_CODEWORD=2
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_decode_matches_oracle or test_iteration_caps" > gpurun_out/k5/first.log 2>&1
echo "first: $?" | tee -a gpurun_out/k5/first.log
tail -3 gpurun_out/k5/first.log
grep -q "first: 0" gpurun_out/k5/first.log || exit 1
python bench.py --no-cpu --no-dropin > gpurun_out/k5/bench_k5.json 2> gpurun_out/k5/bench_k5.err && python -c "
import json; d=json.load(open('gpurun_out/k5/bench_k5.json')); print('k5', d['value'], d['roofline']['avg_launch_ms'], [p['value'] for p in d['points']])"
unset LNSFAID_WAVES_PER_CODEWORD
python bench.py --no-cpu --no-dropin > gpurun_out/k5/bench_k4.json 2> gpurun_out/k5/bench_k4.err && python -c "
import json; d=json.load(open('gpurun_out/k5/bench_k4.json')); print('k4', d['value'], d['roofline']['avg_launch_ms'], [p['value'] for p in d['points']])"
