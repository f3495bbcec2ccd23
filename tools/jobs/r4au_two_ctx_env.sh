#!/bin/bash
# round 4: tools/two_contexts.py under stream / queue settings of the HIP runtime
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "CGX_COMPUTE_PRIORITY=normal" "CGX_COMPUTE_PRIORITY=normal GPU_MAX_HW_QUEUES=8" "CGX_COMPUTE_PRIORITY=normal HIP_FORCE_DEV_KERNARG=1 DEBUG_HIP_GRAPH_DOT_PRINT=0" "GPU_MAX_HW_QUEUES=2"; do
  echo "$v"; env $v timeout -k 10 300 python tools/two_contexts.py --batches 8 2>gpurun_out/r4au.err | tail -1 || tail -3 gpurun_out/r4au.err
done
