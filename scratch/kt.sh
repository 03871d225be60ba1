#!/bin/bash
# kernel-trace stats of one bench run: bash scratch/kt.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$OUT/kt.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:14]:
    print("%-60s calls %5s total %9.2f avg %8.4f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
tail -1 "$OUT/kt.log" | cut -c1-160
