#!/bin/bash
# round 4: the window table (128-byte row per corpus position) against windows read from tok8: parity, then the lookup / extraction stages on one box, no files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bruteforce.py -m gpu -x -q > gpurun_out/r4aa_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r4aa_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
for o in 0 1 0 1; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write --option win_table=$o > gpurun_out/r4aa_w$o.log 2>gpurun_out/r4aa_w$o.err || { tail -20 gpurun_out/r4aa_w$o.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4aa_w$o.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("win_table $o:", j["ms_per_step"], "gappy", s["gappy"], "look1", s["look1_kernel"], "look2", s["look2_kernel"], "extract", s["extract"], "lexicon", s["lexicon"], "hbm", j["hbm_in_use_gb"])
P
done
