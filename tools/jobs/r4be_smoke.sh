#!/bin/bash
# round 4: the driver's entry points on a fresh box: build() is what ran in the container; smoke() here
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4be_smoke.log 2>&1; rc=$?
tail -5 gpurun_out/r4be_smoke.log; exit $rc
