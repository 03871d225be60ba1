#!/bin/bash
# counters for the dev MX kernel (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export IBLNERF_LIB=scratch/lib_mxdev.so
OUT=gpurun_out/mxprof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $OUT/sq -o b -- python3 scratch/mxbench.py > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/mf -o b -- python3 scratch/mxbench.py > $OUT/mf.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("sq", "mf"):
    f = glob.glob("gpurun_out/mxprof/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print("no csv for", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "mlp_kernel" not in k: continue
        k = "mx" if "mxk" in k else "bf16x3"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        print(d, k, "  ".join("%s=%.3g" % (c, x) for c, x in sorted(v.items())))
PY
