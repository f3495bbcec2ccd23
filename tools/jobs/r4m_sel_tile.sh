#!/bin/bash
# round 4: selection with a wave per rank for huge lists; then tile order A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "every_stage or definitions or sizing or id_level or bit_exact" > gpurun_out/r4m_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r4m_pytest.log; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
bash tools/jobs/r4l_tile_order_ab.sh
