#!/bin/bash
# corpus-order occurrence lists in k_look1 (A/B), parity subset, then the multi-writer rehearsal
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r3h}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_bruteforce.py -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/${T}_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for o in 1 0; do
  timeout -k 10 400 python3 bench.py --steps 4 --warmup 1 --no-write --no-cpu-baseline --option occ_order=$o > gpurun_out/${T}_nowrite_occ$o.json 2> gpurun_out/${T}_nowrite_occ$o.err
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/${T}_nowrite_occ$o.json").read().strip().splitlines()[-1])
s=d["stages_ms_per_step"]; print("occ_order=$o", {k:s[k] for k in ("gappy","look1_kernel","look2_kernel","extract","lexicon")}, d["value"])
PY
done
timeout -k 10 900 python3 tools/rehearse_writers.py --queries 2500 --ranks 1,2,4,8 > gpurun_out/${T}_rehearse.txt 2> gpurun_out/${T}_rehearse.err; echo "rehearse rc=$?"; grep -v "^{" gpurun_out/${T}_rehearse.txt | tail -8; tail -3 gpurun_out/${T}_rehearse.err
exit 0
