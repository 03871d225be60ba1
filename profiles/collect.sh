#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers (run ON the GPU box, from the
# repo root):   bash profiles/collect.sh <round-tag>
# Output lands in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
# Counter passes are separate runs with --kernel-trace only (no sys/runtime/hip traces).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --single-stream ${BENCH_ARGS:-}"   # (one stream: a kernel's duration is its own)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- $BENCH > "$OUT/kt.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 \
    -d "$OUT/pmc_mfma" -o bench -- $BENCH > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -40
python3 profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
