#!/bin/bash
# round 4: the full-size property tests (with the .gz and hit-order legs) and the other tests touched since the last full run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_bruteforce.py tests/test_gpu_parity.py -m gpu -x -q --durations=6 -k "batching or thin or gzip or optional" > gpurun_out/r4u_pytest.log 2>&1; rc=$?
tail -14 gpurun_out/r4u_pytest.log; echo "pytest rc=$rc"; exit $rc
