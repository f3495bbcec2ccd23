#!/bin/bash
# round 4: two contexts over one index (cgx_share_index) working at the same time: the test, then the rest of the parity file (the allocator changed)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_contexts" > gpurun_out/r4aq_pytest1.log 2>&1; rc=$?
tail -15 gpurun_out/r4aq_pytest1.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not two_contexts_share" > gpurun_out/r4aq_pytest2.log 2>&1; rc=$?
tail -5 gpurun_out/r4aq_pytest2.log; echo "pytest rc=$rc"; exit $rc
