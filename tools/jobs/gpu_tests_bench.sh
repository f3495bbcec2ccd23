#!/bin/bash
# GPU parity suite, then bench.py with the driver's arguments.  usage: tools/jobs/gpu_tests_bench.sh <tag> [extra bench args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-x}; shift
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/${TAG}_pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python3 - <<P
import json
for line in open("gpurun_out/${TAG}_bench.log"):
    if line.startswith("{"):
        d = json.loads(line)
        print("value", d["value"], "ms/step", d["ms_per_step"])
        print({k: v for k, v in d["stages_ms_per_step"].items() if v})
        print("roofline", d["roofline"]["frac"], d["roofline"]["kernel_ms"], "dominant", d["roofline_dominant"]["frac"], d["roofline_dominant"]["ms_per_launch"], [ (k["kernel"][:8], k["frac"], k["ms_per_launch"]) for k in d["roofline_other_kernels"]])
P
