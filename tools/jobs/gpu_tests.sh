#!/bin/bash
# the GPU parity suite, as the driver runs it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/${1:-gpu}_pytest.log 2>&1; rc=$?
tail -25 gpurun_out/${1:-gpu}_pytest.log; echo "pytest rc=$rc"; exit $rc
