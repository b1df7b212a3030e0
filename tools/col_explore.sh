#!/bin/bash
# GPU-side exploration of k_stepc_col (run from the repo root on the GPU box): parity check, walks / band widths, counters.
OUT=gpurun_out/r03/explore_$1
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 ./tools/colbench --prec f64 > $OUT/f64.log 2>&1 || exit 1
for b in 16 24 37; do timeout -k 10 120 ./tools/colbench --prec f64 --no-check --band $b --filter walk2 > $OUT/f64_band$b.log 2>&1; done
timeout -k 10 300 ./tools/colbench --prec f32 > $OUT/f32.log 2>&1
for f in "D=6 walk1" "D=5 walk1" "lds tile"; do
  tag=$(echo "$f" | tr -d ' =')
  tools/pmc_passes.sh $OUT/pmc_$tag -- ./tools/colbench --prec f64 --no-check --reps 30 --rounds 1 --filter "$f" > $OUT/pmc_$tag.log 2>&1
  rm -rf $OUT/pmc_$tag/stats $OUT/pmc_$tag/sq1 $OUT/pmc_$tag/sq2 $OUT/pmc_$tag/sq3 $OUT/pmc_$tag/fetch $OUT/pmc_$tag/write $OUT/pmc_$tag/tcc
done
grep -h -A8 "^==" $OUT/f64.log $OUT/f64_band*.log $OUT/f32.log
