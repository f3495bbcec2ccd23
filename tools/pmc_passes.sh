#!/bin/bash
# Hardware-counter passes over the per-batch kernels (one rocprofv3 --pmc run per counter group;
# never combined with tracing).  Usage on the GPU box:  tools/pmc_passes.sh <outdir> [bench args...]
set -e
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
KER=${PMC_KER:-'k_look1|k_look2|k_lex_finish|k_fmt_lines|k_extract|k_sa_lookup|k_runsort'}
NOWRITE=--no-write; [ -n "$PMC_WRITE" ] && NOWRITE=--fresh-steps=0      # PMC_WRITE=1: lay out the text and write the files too (k_fmt_lines runs)
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
  "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi     # PMC_ONLY="1 2": just those passes
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "$KER" --output-format csv -d "$OUT/p$i" -- python3 "$REPO/bench.py" $NOWRITE --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/fail.txt"
  f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 "$REPO/tools/pmc_sum.py" "$f" > "$OUT/p$i.sum.txt"
  rm -rf "$OUT/p$i"
  echo "pass $i done"
done
