#!/bin/bash
# round 4: presence bits in front of the pair keys of MaxLex (lex_bits 1) against none (0): parity first, then the lexicon stage on one box, no files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bruteforce.py -m gpu -x -q -k "every_stage or sizing or golden or maxlex or lex" > gpurun_out/r4an_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/r4an_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
for o in 0 1 0 1; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write --option lex_bits=$o > gpurun_out/r4an_lf$o.log 2>gpurun_out/r4an_lf$o.err || { tail -20 gpurun_out/r4an_lf$o.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4an_lf$o.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("lex_bits $o:", j["ms_per_step"], "gappy", s["gappy"], "extract", s["extract"], "lexicon", s["lexicon"])
P
done
