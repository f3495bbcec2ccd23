#!/bin/bash
# round 4: counter passes over the .gz formatter kernels, the new MaxLex kernel and the lexicon's hash / head kernels (separate --pmc runs, no tracing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PMC_WRITE=1 PMC_KER='k_fmt_lines|k_lex_finish|k_rule_hash|k_lex_heads|k_gz' PMC_ONLY="1 2 3" bash tools/pmc_passes.sh gpurun_out/r4x_pmc --gz-steps 0 --query-sets 1 --option gz_level=1 > gpurun_out/r4x_pmc.log 2>&1
cat gpurun_out/r4x_pmc/p*.sum.txt > gpurun_out/r4x_pmc_fmt_gz_lexicon_kernels.txt 2>/dev/null
cat gpurun_out/r4x_pmc_fmt_gz_lexicon_kernels.txt | grep -E "k_fmt_lines_gz|k_lex_finish" | head -80
