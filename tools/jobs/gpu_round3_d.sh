#!/bin/bash
# parity subset, a --no-write bench for the stage timers, then the multi-writer rehearsal
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r3i}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_bruteforce.py -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/${T}_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python3 bench.py --steps 4 --warmup 1 --no-write --no-cpu-baseline > gpurun_out/${T}_nowrite.json 2> gpurun_out/${T}_nowrite.err
python3 - <<PY
import json
d=json.loads(open("gpurun_out/${T}_nowrite.json").read().strip().splitlines()[-1])
s=d["stages_ms_per_step"]; print({k:s[k] for k in ("gappy","look1_kernel","look2_kernel","extract","lexicon","format")}, d["value"], d["counts"])
PY
if [ "$2" != "norehearse" ]; then
timeout -k 10 900 python3 tools/rehearse_writers.py --queries 2500 --ranks 1,2,4,8 > gpurun_out/${T}_rehearse.txt 2> gpurun_out/${T}_rehearse.err; echo "rehearse rc=$?"; grep -v "^{" gpurun_out/${T}_rehearse.txt | tail -8; grep -v amdgpu.ids gpurun_out/${T}_rehearse.err | tail -5
fi
exit 0
