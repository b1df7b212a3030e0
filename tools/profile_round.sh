#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_round.sh OUTDIR PART
# The per-configuration passes behind profiles/rNN/<config>/ and profiles/traffic.json: for every BASELINE.json grid the
# plan the measurement picks (c*) and the other candidates forced with --set (alt_*), four rocprofv3 passes each
# (tools/pmc_traffic.sh). PART 1: the fp64 grids; PART 2: the fp32 grids.
set -e
OUT=$1; PART=$2
B="python3 bench.py --no-cpu-baseline --no-other-arith --no-f32-variant --no-sustained"
run() { name=$1; shift; tools/pmc_traffic.sh "$OUT/$name" -- $B "$@"; cp "$OUT/$name"/stats/*kernel_stats.csv "$OUT/$name/kernel_stats.csv" 2>/dev/null || true; rm -rf "$OUT/$name/stats" "$OUT/$name/fetch" "$OUT/$name/write" "$OUT/$name/sq" "$OUT/$name"/*.log; echo "$name: $(cut -c1-90 $OUT/$name/bench_line.json)"; }
C5="--set tune=0 --set layout=1 --set variant=1 --set nt=1 --set xcd=1 --set pair_ty=12 --set deep=6"
C6="--set tune=0 --set layout=1 --set variant=1 --set nt=1 --set xcd=1 --set pair_ty=12 --set deep=7"
D6="--set tune=0 --set layout=1 --set variant=1 --set nt=1 --set xcd=1 --set pair_ty=12 --set deep=1"
D8="--set tune=0 --set layout=1 --set variant=1 --set nt=1 --set xcd=1 --set pair_ty=12 --set deep=3"
T3="--set tune=0 --set layout=1 --set variant=1 --set nt=1 --set xcd=1 --set pair_ty=12 --set fuse=3"
T4="--set tune=0 --set layout=1 --set variant=1 --set nt=1 --set xcd=1 --set pair_ty=8 --set fuse=4"
C6N="--set tune=0 --set layout=1 --set variant=1 --set nt=0 --set xcd=1 --set pair_ty=12 --set deep=7"
if [ "$PART" = 1 ]; then
  S="--steps 6000 --warmup 600"
  run c3_f64_contracted $S
  run alt_c3_f64_col5 $S $C5
  run alt_c3_f64_col6 $S $C6
  run alt_c3_f64_col6_nt0 $S $C6N
  run alt_c3_f64_6step_64x16 $S $D6
  run alt_c3_f64_3step $S $T3
  run alt_c3_f64_4step $S $T4
  run c3_f64_strict $S --arith strict
  run alt_c3_f64_strict_col5 $S --arith strict $C5
  run alt_c3_f64_strict_col6 $S --arith strict $C6
  run alt_c3_f64_strict_3step $S --arith strict $T3
  S="--steps 12000 --warmup 1200 --nx 1024 --ny 256 --re 100"
  run c2_1024x256_f64 $S
  run alt_c2_f64_col6 $S $C6
  run alt_c2_f64_6step_64x16 $S $D6
  run alt_c2_f64_8step_32x32 $S $D8
  run alt_c2_f64_4step $S $T4
  S="--steps 1500 --warmup 150 --nx 8192 --ny 2048"
  run c4_8192x2048_f64 $S
  run alt_c4_f64_col5 $S $C5
  run alt_c4_f64_col6 $S $C6
  run alt_c4_f64_col6_nt0 $S $C6N
  run alt_c4_f64_3step $S $T3
else
  S="--steps 6000 --warmup 600 --precision f32"
  run c3_f32 $S
  run alt_c3_f32_col5 $S $C5
  run alt_c3_f32_col6 $S $C6
  run alt_c3_f32_col6_nt0 $S $C6N
  run alt_c3_f32_6step_64x16 $S $D6
  run alt_c3_f32_4step $S $T4
  S="--steps 600 --warmup 60 --precision f32 --nx 16384 --ny 4096"
  run c5_16384x4096_f32 $S
  run alt_c5_f32_col5 $S $C5
  run alt_c5_f32_col6 $S $C6
  run alt_c5_f32_col6_nt0 $S $C6N
  run alt_c5_f32_6step_64x16 $S $D6
  run alt_c5_f32_3step $S --set tune=0 --set layout=1 --set variant=1 --set nt=1 --set xcd=1 --set pair_ty=8 --set fuse=3
fi
