#!/bin/bash
# Usage: tools/pmc_passes.sh OUTDIR -- python3 <program> [args]   (run from the repo root on the GPU box)
# One rocprofv3 pass per counter group (SQ has 8 slots; FETCH_SIZE and WRITE_SIZE do not fit one pass), plus a
# --kernel-trace --stats pass. The program follows `--` directly (no env/bash hop under rocprofv3).
set -e
OUT=$1; shift; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s --output-format csv -- "$@" > "$OUT/stats.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d "$OUT/sq1" -o p --output-format csv -- "$@" > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM -d "$OUT/sq2" -o p --output-format csv -- "$@" > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE -d "$OUT/sq3" -o p --output-format csv -- "$@" > "$OUT/sq3.log" 2>&1 || true
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o p --output-format csv -- "$@" > "$OUT/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/tcc" -o p --output-format csv -- "$@" > "$OUT/tcc.log" 2>&1 || true
python3 tools/pmc_summary.py "$OUT" --csv "$OUT/pmc_summary.csv"
