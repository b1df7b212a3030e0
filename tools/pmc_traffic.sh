#!/bin/bash
# Usage: tools/pmc_traffic.sh OUTDIR -- python3 bench.py [args]      (run from the repo root on the GPU box)
# The passes bench.py's roofline needs for one configuration: kernel stats, FETCH_SIZE, WRITE_SIZE (separate passes: the two
# do not fit the TCC counter slots together) and one SQ pass (vector / LDS instruction counts, LDS bank conflicts and stalls,
# busy and wait cycles). The program follows `--` directly.
set -e
OUT=$1; shift; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s --output-format csv -- "$@" > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o p --output-format csv -- "$@" > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$OUT/sq" -o p --output-format csv -- "$@" > "$OUT/sq.log" 2>&1
python3 tools/pmc_summary.py "$OUT" --csv "$OUT/pmc_summary.csv" > "$OUT/kernel_stats.txt"
grep -h '"metric"' "$OUT/stats.log" | tail -1 > "$OUT/bench_line.json" || true
