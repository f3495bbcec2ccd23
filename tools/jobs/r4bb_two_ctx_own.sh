#!/bin/bash
# round 4: two contexts in one process, each with an index of its own: is it the shared index that limits their overlap?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 500 python tools/two_contexts.py --batches 8 --own-index 2>gpurun_out/r4bb.err | tail -1; tail -2 gpurun_out/r4bb.err
