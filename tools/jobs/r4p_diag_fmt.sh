#!/bin/bash
# round 4: lines and bytes per emission group of one cfg3 batch written as .gz (dynamic and fixed codes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for dyn in 1 0; do
CGX_DIAG_FMT=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --fresh-steps 0 --gz-steps 0 --query-sets 1 --option gz_level=1 --option gz_dynamic=$dyn > gpurun_out/r4p_diag$dyn.log 2>gpurun_out/r4p_diag$dyn.err
echo "gz_dynamic=$dyn"; grep "fmt groups" gpurun_out/r4p_diag$dyn.err | tail -8
done
