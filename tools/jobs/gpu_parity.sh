#!/bin/bash
# First-contact GPU check: oracle CLI vs product CLI on generated fixtures, byte compare.
set -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/parity; mkdir -p $OUT
make -C oracle liboracle.so strmatch_oracle > $OUT/oracle_build.log 2>&1 || { echo "oracle build failed"; cat $OUT/oracle_build.log; exit 1; }
rc=0
for spec in "tiny 400 160 7 7" "toy 20000 160 7 11" "mid 5000 300 40 3"; do
  set -- $spec; name=$1
  python tools/gen_fixture.py /tmp/fx_$name --pairs $2 --vocab $3 --queries $4 --seed $5 --long-query
  F=/tmp/fx_$name; mkdir -p $OUT/${name}_o $OUT/${name}_p; rm -f $OUT/${name}_o/* $OUT/${name}_p/*
  oracle/strmatch_oracle $F/corpus.f $F/query.f $F/corpus.e $F/corpus.a $F/lex.txt $OUT/${name}_o > $OUT/${name}_o.log 2>&1 || { echo "oracle failed on $name"; rc=1; continue; }
  timeout -k 10 300 bin/strmatchcuda $F/corpus.f $F/query.f $F/corpus.e $F/corpus.a $F/lex.txt $OUT/${name}_p > $OUT/${name}_p.log 2>&1
  prc=$?
  tail -2 $OUT/${name}_o.log; tail -3 $OUT/${name}_p.log
  if [ $prc -ne 0 ]; then echo "product exit $prc on $name"; rc=1; continue; fi
  nd=0
  for f in $OUT/${name}_o/grammar.*.s; do b=$(basename $f); cmp -s $f $OUT/${name}_p/$b || { nd=$((nd+1)); echo "DIFF $name $b: $(wc -l < $f) vs $(wc -l < $OUT/${name}_p/$b 2>/dev/null) lines"; }; done
  echo "== $name: $nd files differ"
  [ $nd -eq 0 ] && rm -rf $OUT/${name}_o $OUT/${name}_p
  [ $nd -ne 0 ] && rc=1
  rm -f $OUT/${name}_o/grammar.*.s.keep
done
exit $rc
