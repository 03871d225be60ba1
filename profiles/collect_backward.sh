#!/bin/bash
# rocprofv3 evidence for the fused backward kernels (run ON the GPU box, from the repo root):  bash profiles/collect_backward.sh <tag>
# kernel trace + two separate --pmc passes (each with --kernel-trace only) of scratch/netbwd_time.py (whole-network backward, 786 432 points).
set -u
TAG=${1:-r03_backward}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PROG="python3 scratch/netbwd_time.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- $PROG > "$OUT/kt.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 \
    -d "$OUT/pmc_mfma" -o bench -- $PROG > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d "$OUT/pmc_sq" -o bench -- $PROG > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $PROG > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $PROG > "$OUT/pmc_write.log" 2>&1
python3 profiles/summarize.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
