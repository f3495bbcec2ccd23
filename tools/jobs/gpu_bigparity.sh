#!/bin/bash
# full-size oracle parity: the whole 10 M-pair corpus, 12 queries, grammar files of the HIP path == the CPU oracle's (≈ 15 min, almost all of it the single-thread oracle)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(while true; do date >> gpurun_out/bigparity_tick.log; sleep 60; done) & TICK=$!
CGX_BIG_PARITY=${1:-10000000,200000,12} timeout -k 10 1150 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "synthetic and ${2:-10000000}" > gpurun_out/r2_bigparity.log 2>&1; rc=$?
kill $TICK
tail -5 gpurun_out/r2_bigparity.log; echo "rc=$rc"; exit $rc
