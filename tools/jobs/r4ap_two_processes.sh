#!/bin/bash
# round 4: would two batches in flight on one card pay?  Two bench processes (each with its own index: --no-write, 100 GB each) side by side
# against one alone: the aggregate rate of the GPU chain
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-write > gpurun_out/r4ap_alone.log 2>gpurun_out/r4ap_alone.err || { tail -5 gpurun_out/r4ap_alone.err; exit 1; }
timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-write --seed 1234 > gpurun_out/r4ap_a.log 2>gpurun_out/r4ap_a.err &
pa=$!
timeout -k 10 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-write --seed 1234 > gpurun_out/r4ap_b.log 2>gpurun_out/r4ap_b.err &
pb=$!
wait $pa; ra=$?; wait $pb; rb=$?
echo "exit codes $ra $rb"
python - <<P
import json
def ms(f):
    try: return json.loads([l for l in open(f) if l.startswith("{")][-1])["ms_per_step"]
    except Exception as e: return None
a, x, y = ms("gpurun_out/r4ap_alone.log"), ms("gpurun_out/r4ap_a.log"), ms("gpurun_out/r4ap_b.log")
print("alone: %s ms per step; side by side: %s and %s ms per step" % (a, x, y))
if a and x and y: print("aggregate rate side by side / alone = %.3f" % ((1.0 / x + 1.0 / y) * a))
P
