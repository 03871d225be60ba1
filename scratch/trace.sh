#!/bin/bash
# bash scratch/trace.sh <tag> <args of trace_frame.py...>: kernel trace + stats of two frames -> gpurun_out/trace_<tag>/stats.txt
TAG=$1; shift
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o t -- python3 scratch/trace_frame.py "$@" > $OUT/log.txt 2>&1
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - "$F" > $OUT/stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%6.2f %%  %9.1f ms  calls %5s  avg %8.3f ms  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e6, r["Name"][:110]))
print("total %.1f ms" % (tot / 1e6))
PY
cat $OUT/stats.txt
