#!/bin/bash
# round 4: k_lex_finish_flat with the key-array probes: parity, A/B of the lexicon stage, then per-kernel times (no files)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_bruteforce.py -m gpu -x -q -k "every_stage or sizing or golden or maxlex or lex" > gpurun_out/r4w_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/r4w_pytest.log; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
for o in 0 1; do
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-write --option lex_flat=$o > gpurun_out/r4w_lf$o.log 2>gpurun_out/r4w_lf$o.err || { tail -20 gpurun_out/r4w_lf$o.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4w_lf$o.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("lex_flat $o:", j["ms_per_step"], "gappy", s["gappy"], "extract", s["extract"], "lexicon", s["lexicon"])
P
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-write > $GRAFT_REPO_ROOT/gpurun_out/r4w_kstats.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4w_kstats_nowrite.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tmp
python3 - <<P
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r4w_kstats_nowrite.csv")))
for r in rows[:30]:
    print(r['Name'].split('(')[0].replace('void ','')[:70].ljust(70), r['Calls'], round(float(r["TotalDurationNs"])/1e6/8,2), round(float(r['AverageNs'])/1e6,3))
P
