#!/bin/bash
# round 3, step f: parity suite, the bench line, then the same bench with one option switched (A/B).  usage: gpu_round3_f.sh <tag> [name=value]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; TAG=${1:-r3x}; AB=${2:-src_blocks=0}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/${TAG}_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
show() { python3 - "$1" <<'P'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("value", d["value"], "ms/step", d["ms_per_step"], "gpu_chain", d.get("value_gpu_chain"), "fresh", d.get("value_fresh_files"))
        print({k: round(v, 2) for k, v in d["stages_ms_per_step"].items() if v})
P
}
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err; rc=$?; echo "bench rc=$rc"
if [ $rc -ne 0 ]; then tail -5 gpurun_out/${TAG}_bench.err; exit $rc; fi
show gpurun_out/${TAG}_bench.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --fresh-steps 0 --option $AB > gpurun_out/${TAG}_bench_nosrc.log 2> gpurun_out/${TAG}_bench_nosrc.err; rc=$?; echo "bench($AB) rc=$rc"
if [ $rc -ne 0 ]; then tail -5 gpurun_out/${TAG}_bench_nosrc.err; exit $rc; fi
show gpurun_out/${TAG}_bench_nosrc.log
