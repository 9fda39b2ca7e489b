#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (run on the GPU box through gpurun).
# usage: bash scripts/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ + summary text
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import csv, glob, os
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
with open("$OUT/summary.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python bench.py $*\n")
    o.write("%-70s %8s %14s %14s %8s\n" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in rows:
        o.write("%-70s %8s %14.3f %14.3f %8s\n" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
    o.write("\n# bench.py output line\n" + open("$OUT/bench.json").read())
print(open("$OUT/summary.txt").read())
PY
