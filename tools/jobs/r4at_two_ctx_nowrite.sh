#!/bin/bash
# round 4: two contexts over one index against one, the GPU chain alone (no text, no files): tools/two_contexts.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python tools/two_contexts.py --batches 8 > gpurun_out/r4at.log 2>gpurun_out/r4at.err; rc=$?
tail -3 gpurun_out/r4at.log; tail -3 gpurun_out/r4at.err; exit $rc
