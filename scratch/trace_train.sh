#!/bin/bash
# bash scratch/trace_train.sh <tag> [n_rays]: rocprofv3 kernel trace + stats of `bench.py --train` -> gpurun_out/trace_<tag>/stats.txt
TAG=$1; N=${2:-4096}
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o t -- python3 bench.py --train --train-rays $N --train-headline $N --steps 20 --warmup 3 > $OUT/log.txt 2>&1
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - "$F" > $OUT/stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:32]:
    print("%6.2f %%  %9.2f ms  calls %5s  avg %8.3f ms  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e6, r["Name"][:120]))
print("total %.1f ms in %d kernels" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
PY
cat $OUT/stats.txt; tail -1 $OUT/log.txt | cut -c1-300
