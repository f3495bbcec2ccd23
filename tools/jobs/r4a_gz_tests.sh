#!/bin/bash
# round 4: the gzip-member tests first, then the formatter-dependent parity tests (the formatter moved into cgx_fmt.h)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 -k "gzip or staged or bit_exact or edge or async" > gpurun_out/r4a_pytest.log 2>&1; rc=$?
tail -30 gpurun_out/r4a_pytest.log; echo "pytest rc=$rc"; exit $rc
