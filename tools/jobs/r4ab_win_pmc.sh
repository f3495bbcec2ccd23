#!/bin/bash
# round 4: why the window table buys nothing: DRAM / L2 request counts of the lookups with and without it (cfg3), and the same A/B on a quarter-size corpus (8 GB table)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for o in 0 1; do
  PMC_KER='k_look1|k_look2|k_extract1|k_extract2' PMC_ONLY="3 5" bash tools/pmc_passes.sh gpurun_out/r4ab_pmc$o --option win_table=$o > gpurun_out/r4ab_pmc$o.log 2>&1
  cat gpurun_out/r4ab_pmc$o/p*.sum.txt > gpurun_out/r4ab_pmc_win_table_$o.txt 2>/dev/null
  echo "win_table=$o"; grep -E "k_look1|k_look2" gpurun_out/r4ab_pmc_win_table_$o.txt | grep -E "TCC_EA0_RDREQ_sum|TCC_REQ_sum|TCC_HIT|TCP_TCC_READ_REQ_sum|TCP_PENDING" | awk '{printf "%-12s %-34s %s\n", $1, $2, $8}'
done
for o in 0 1 0 1; do
  timeout -k 10 300 python bench.py --pairs 2500000 --steps 4 --warmup 2 --no-cpu-baseline --no-write --option win_table=$o > gpurun_out/r4ab_q$o.log 2>gpurun_out/r4ab_q$o.err || { tail -20 gpurun_out/r4ab_q$o.err; exit 1; }
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/r4ab_q$o.log") if l.startswith("{")][-1])
s=j["stages_ms_per_step"]
print("2.5M pairs, win_table $o:", j["ms_per_step"], "gappy", s["gappy"], "look1", s["look1_kernel"], "look2", s["look2_kernel"], "extract", s["extract"])
P
done
