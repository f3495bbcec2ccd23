#!/bin/bash
# HBM traffic of one k_sa_lookup launch: separate rocprofv3 --pmc passes (never combined with tracing), summed per kernel.
# Usage on the GPU box: tools/pmc_sa_lookup.sh <outdir>     -> <outdir>/pmc_k_sa_lookup.txt
set -e
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
: > "$OUT/pmc_k_sa_lookup.txt"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "k_sa_lookup" --output-format csv -d "$OUT/s$i" -- python3 "$REPO/bench.py" --no-write --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$OUT/s$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/fail.txt"
  f=$(find "$OUT/s$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 "$REPO/tools/pmc_sum.py" "$f" >> "$OUT/pmc_k_sa_lookup.txt"
  rm -rf "$OUT/s$i"
  echo "pass $i done"
done
