#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_round4.sh OUTDIR
# Round-4 record of the binary in the tree: (1) profiles/traffic.json from live counter passes of every fused candidate of every
# BASELINE.json grid (tools/collect_live_traffic.py: FETCH_SIZE / WRITE_SIZE / SQ passes + an unprofiled timing each), (2) per grid the
# bench line of the measured plan with the rocprofv3 --kernel-trace --stats summary of the same command, (3) the default and the
# driver-style bench lines (with their own live passes).
set -e
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/collect_live_traffic.py --out "$OUT/traffic.json" > "$OUT/collect_traffic.log" 2>&1
B="bench.py --no-cpu-baseline --no-other-arith --no-f32-variant --no-live-pmc"
run() { name=$1; shift; mkdir -p "$OUT/$name"; ( cd /tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/$name/stats" -o s --output-format csv -- python3 "$GRAFT_REPO_ROOT/"$B "$@" > "$GRAFT_REPO_ROOT/$OUT/$name/stats.log" 2>&1 ); cp "$OUT/$name"/stats/*kernel_stats.csv "$OUT/$name/kernel_stats.csv"; grep -h '"metric"' "$OUT/$name/stats.log" | tail -1 > "$OUT/$name/bench_line.json"; rm -rf "$OUT/$name/stats" "$OUT/$name/stats.log"; echo "$name: $(cut -c1-80 $OUT/$name/bench_line.json)"; }
run c3_f64_contracted --steps 6000 --warmup 600
run c3_f64_strict --steps 6000 --warmup 600 --arith strict
run c2_1024x256_f64 --steps 12000 --warmup 1200 --nx 1024 --ny 256 --re 100
run c4_8192x2048_f64 --steps 1500 --warmup 150 --nx 8192 --ny 2048
run c3_f32 --steps 6000 --warmup 600 --precision f32
run c5_16384x4096_f32 --steps 600 --warmup 60 --precision f32 --nx 16384 --ny 4096
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
echo done
