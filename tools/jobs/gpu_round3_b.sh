#!/bin/bash
# quick check of the SA-lookup kernel: its parity tests, then a short bench for the kernel time, then the look1 group diagnostics
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=${1:-r3d}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "ngram or every_stage or intervals or smoke or toy" > gpurun_out/${T}_pytest.log 2>&1; rc=$?
tail -8 gpurun_out/${T}_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python3 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --fresh-steps 0 > gpurun_out/${T}_bench_cfg3.json 2> gpurun_out/${T}_bench_cfg3.err; rc=$?
echo "bench rc=$rc"; python3 - <<PY
import json
d=json.loads(open("gpurun_out/${T}_bench_cfg3.json").read().strip().splitlines()[-1])
r=d["roofline"]; print({k:r[k] for k in ("achieved","frac","kernel_ms","kernel_ms_single_relaunch_after_run","units_per_launch","algorithmic_bytes_per_launch")})
print(d["value"], d["index"], d["hbm_in_use_gb"])
PY
CGX_DIAG_GROUPS=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-write --no-cpu-baseline > /dev/null 2> gpurun_out/${T}_groups.txt; grep -v amdgpu.ids gpurun_out/${T}_groups.txt | tail -22
exit $rc
