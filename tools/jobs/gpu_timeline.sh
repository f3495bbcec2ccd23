#!/bin/bash
# kernel timeline of one batch (launch order, merged runs, idle gaps).  usage: tools/jobs/gpu_timeline.sh <tag>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-write > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_run.log 2>&1; echo "rc=$?"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/tl_tmp -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py "$f" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_timeline.txt; rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl_tmp
head -5 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_timeline.txt
