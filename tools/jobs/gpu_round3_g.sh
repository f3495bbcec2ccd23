#!/bin/bash
# Parity of the build in place (stage-by-stage tests + the definition-level checks), then the A/B of the library builds named on
# the command line.  usage: tools/jobs/gpu_round3_g.sh <tag> <lib> [<lib> ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=$1; shift
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_bruteforce.py tests/test_gpu_stress.py -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/${TAG}_pytest.log; echo "pytest rc=$rc"
[ $rc -gt 1 ] && exit $rc            # a timeout or a fault: no further GPU step
tools/jobs/gpu_lib_variants.sh $TAG "$@"
