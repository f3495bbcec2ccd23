#!/bin/bash
# round 4: the lookups compiled for the default layouts (MODE 2) against the general build (CGX_LOOK_GENERIC=1): parity, then the stages on one box, no files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bruteforce.py -m gpu -x -q > gpurun_out/r4ac_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r4ac_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
for o in 1 0 1 0; do
  if [ $o = 1 ]; then export CGX_LOOK_GENERIC=1; else unset CGX_LOOK_GENERIC; fi
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write > gpurun_out/r4ac_g$o.log 2>gpurun_out/r4ac_g$o.err || { tail -20 gpurun_out/r4ac_g$o.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4ac_g$o.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("generic $o:", j["ms_per_step"], "gappy", s["gappy"], "look1", s["look1_kernel"], "look2", s["look2_kernel"], "extract", s["extract"], "lexicon", s["lexicon"])
P
done
