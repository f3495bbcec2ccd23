#!/bin/bash
# round 4: stage counters read back through page-locked memory: parity, one context / two contexts (tools/two_contexts.py), then the chain
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4ax_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r4ax_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/two_contexts.py --batches 8 2>gpurun_out/r4ax.err | tail -1
timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write > gpurun_out/r4ax_b.log 2>gpurun_out/r4ax_b.err || { tail -5 gpurun_out/r4ax_b.err; exit 1; }
python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4ax_b.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("no-write:", j["ms_per_step"], "gappy", s["gappy"], "extract", s["extract"], "lexicon", s["lexicon"])
P
