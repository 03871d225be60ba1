#!/bin/bash
# PMC breakdown of the MLP kernel on the microbench (run on the GPU box from repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_mlp; rm -rf $OUT; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --output-format csv --pmc $2 -d $OUT/$1 -o b -- python3 scratch/mlpbench.py > $OUT/$1.log 2>&1; }
run g1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run g2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH"
run g3 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE"
python3 - <<'PY'
import csv, glob, collections
for g in ("g1","g2","g3"):
    f = glob.glob("gpurun_out/pmc_mlp/%s/**/*counter_collection.csv" % g, recursive=True)
    if not f: print(g, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:28]
        if "mlp_kernel<1>" not in r["Kernel_Name"]: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        print(g, k, "launches", len(n[k]), " ".join("%s=%.4g" % (c, v/len(n[k])) for c, v in sorted(agg[k].items())))
PY
