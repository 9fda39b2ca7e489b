#!/bin/bash
# kernel timeline of the bench step with several trace groups (rocprofv3 --kernel-trace): do the launches of different groups overlap?
# usage: bash scripts/prof_overlap.sh <tag> "<PNR_BENCH_OPTS>" [bench args...]
set -e
TAG=$1; OPTS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/overlap_$TAG
rm -rf $OUT && mkdir -p $OUT
export PNR_BENCH_OPTS=$OPTS
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows if "ph_" in r["Kernel_Name"]]
ks.sort()
# second half only (the timed step after the warm-up)
ks = ks[len(ks) // 2:]
t0, t1 = ks[0][0], max(k[1] for k in ks)
busy = 0; cur_s, cur_e = ks[0][0], ks[0][1]; sum_d = 0
for s, e, n, q in ks:
    sum_d += e - s
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in ks:
    k = n.split("(")[0].split("::")[-1][:24]; per[k][0] += e - s; per[k][1] += 1
print("span %.1f ms, union busy %.1f ms, sum of kernel durations %.1f ms (overlap factor %.2f), queues %s" % ((t1 - t0) / 1e6, busy / 1e6, sum_d / 1e6, sum_d / max(busy, 1), sorted(set(k[3] for k in ks))))
for k, (d, n) in sorted(per.items()): print("  %-26s n %5d  avg %.3f ms  total %.1f ms" % (k, n, d / n / 1e6, d / 1e6))
print(open("$OUT/bench.json").read()[:400])
PY
