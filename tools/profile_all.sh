#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_all.sh OUTDIR PART   — PART 1: fp64 grids; PART 2: fp32 grids, the
# all-counter pass of the headline run, `python bench.py` with default flags and a driver-style 20-step run.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$1; PART=$2; mkdir -p "$O"
timeout -k 10 1100 tools/profile_round.sh "$O" "$PART" > "$O/part$PART.log" 2>&1
cut -c1-100 "$O/part$PART.log"
if [ "$PART" = 2 ]; then
  timeout -k 10 300 tools/pmc_passes.sh "$O/c3_f64_contracted_sq" -- python3 bench.py --no-cpu-baseline --no-other-arith --no-f32-variant --no-sustained --steps 600 --warmup 60 > /dev/null 2>&1
  rm -rf "$O"/c3_f64_contracted_sq/stats "$O"/c3_f64_contracted_sq/sq? "$O"/c3_f64_contracted_sq/fetch "$O"/c3_f64_contracted_sq/write "$O"/c3_f64_contracted_sq/tcc "$O"/c3_f64_contracted_sq/*.log
  timeout -k 10 300 python3 bench.py > "$O/bench_default.json" 2>/dev/null; cut -c1-150 "$O/bench_default.json"
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$O/bench_driver_style.json" 2>/dev/null; cut -c1-150 "$O/bench_driver_style.json"
fi
