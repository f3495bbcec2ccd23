#!/bin/bash
# round 4: selection with register-resident buckets: parity subset, then per-kernel times of the chain without files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "every_stage or definitions or sizing or id_level" > gpurun_out/r4k_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r4k_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
bash tools/jobs/r4j_kstats_nowrite.sh
