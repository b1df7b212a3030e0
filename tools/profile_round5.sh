#!/bin/bash
# Usage (on the GPU box, from the repo root): [ONLY="names"] tools/profile_round5.sh OUTDIR [skip-traffic]
# Round-5 record of the binary in the tree, in the order that keeps every committed line reproducible (round 4
# wrote the per-grid lines with --no-live-pmc BEFORE the new traffic table was in place, so all six carried `stale: true`):
#  (1) profiles/traffic.json FIRST — live counter passes of every fused candidate of every BASELINE.json grid (tools/collect_live_traffic.py)
#      — and installed into the tree on the box, so that any fallback lookup below already sees this build's table;
#  (2) per grid: the bench line of the measured plan WITH ITS OWN LIVE PASSES (a plain run), then the rocprofv3 --kernel-trace --stats
#      summary of the same command (a second run, whose line is dropped: only its kernel_stats.csv is kept);
#  (3) the default and the driver-style bench lines (with their own live passes).
# Every step appends a line to OUTDIR/progress.log at once (a silent harness kill must leave the last step on record).
set -e
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
export LBM_TRACE="$PWD/$OUT/lbm_trace.txt"
say() { echo "$(date +%T) $*" >> "$OUT/progress.log"; }
if [ "$2" != "skip-traffic" ]; then
  say "traffic table: start"
  python3 tools/collect_live_traffic.py --out "$OUT/traffic.json" > "$OUT/collect_traffic.log" 2>&1
  cp "$OUT/traffic.json" profiles/traffic.json
  say "traffic table: done"
fi
B="bench.py --no-cpu-baseline --no-other-arith --no-f32-variant"
run() {
  name=$1; shift
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return 0; fi      # ONLY="c3_f64_contracted c2_1024x256_f64": a part of the record per call
  mkdir -p "$OUT/$name"
  say "$name: bench line with live passes"
  python3 $B "$@" > "$OUT/$name/bench_line.json" 2> "$OUT/$name/bench_line.err"
  # the traced run is pinned to the plan the line above measured (finalists 1-2 % apart flip between two runs: round 5's first record
  # paired a line on the nt-load plan with the trace of an nt-store run)
  PIN=$(python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(' '.join('--set ' + kv for kv in ['tune=0'] + d['config']['plan_options'].split()))" "$OUT/$name/bench_line.json")
  say "$name: kernel trace ($PIN)"
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/$name/stats" -o s --output-format csv -- python3 "$GRAFT_REPO_ROOT/"$B --no-live-pmc $PIN "$@" > "$GRAFT_REPO_ROOT/$OUT/$name/stats.log" 2>&1 )
  cp "$OUT/$name"/stats/*kernel_stats.csv "$OUT/$name/kernel_stats.csv"
  rm -rf "$OUT/$name/stats" "$OUT/$name/stats.log"
  say "$name: $(cut -c1-90 $OUT/$name/bench_line.json)"
}
run c3_f64_contracted --steps 6000 --warmup 600
run c3_f64_strict --steps 6000 --warmup 600 --arith strict
run c2_1024x256_f64 --steps 12000 --warmup 1200 --nx 1024 --ny 256 --re 100
run c4_8192x2048_f64 --steps 1500 --warmup 150 --nx 8192 --ny 2048
run c3_f32 --steps 6000 --warmup 600 --precision f32
run c5_16384x4096_f32 --steps 600 --warmup 60 --precision f32 --nx 16384 --ny 4096
if [ -z "$ONLY" ] || echo " $ONLY " | grep -q " bench "; then
  say "default bench line"
  python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
  say "driver-style bench line"
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
fi
say done
