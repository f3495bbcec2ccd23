#!/bin/bash
# round 4: the lookups with the next round's occurrence fetched a round ahead, against a build without (CGX_PIPE=0): parity of the head, then the A/B on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bruteforce.py -m gpu -x -q -k "every_stage or bit_exact or definitions or sizing" > gpurun_out/r4aj_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r4aj_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
for n in 0 1 0 1; do
  if [ $n = 1 ]; then unset CGX_LIB; else export CGX_LIB=$GRAFT_REPO_ROOT/cgx_amd/libcgx_nopipe.so; fi
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write > gpurun_out/r4aj_p$n.log 2>gpurun_out/r4aj_p$n.err || { tail -5 gpurun_out/r4aj_p$n.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4aj_p$n.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("fetched ahead $n:", j["ms_per_step"], "look1", s["look1_kernel"], "look2", s["look2_kernel"], "gappy", s["gappy"])
P
done
