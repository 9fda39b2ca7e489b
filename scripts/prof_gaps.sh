#!/bin/bash
# What sits between the kernels of a trace group's chain (rocprofv3 --kernel-trace, sorted per queue): the idle time of each stream by
# the pair (kernel that ended, kernel that started), and the share of every stream's span that is gaps.
# usage: bash scripts/prof_gaps.sh <tag> "<PNR_BENCH_OPTS>" [bench args...]
set -e
TAG=$1; OPTS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/gaps_$TAG
rm -rf $OUT && mkdir -p $OUT
export PNR_BENCH_OPTS=$OPTS
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
def short(n):
    for k in ("ph_predict", "ph_cube", "ph_sample", "ph_sums", "ph_update", "ph_poll", "copyBuf", "fillBuffer"):
        if k in n: return k
    return n.split("(")[0][-20:]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
ks.sort()
first = [i for i, k in enumerate(ks) if k[2] == "ph_predict"]
ks = ks[first[len(first) // 2]:]   # the timed step (after the warm-up)
ks = [k for k in ks if k[2].startswith("ph_") or k[2] == "copyBuf"]
byq = collections.defaultdict(list)
for k in ks: byq[k[3]].append(k)
out = []
for q, L in sorted(byq.items()):
    if sum(1 for k in L if k[2] == "ph_sample") < 50: continue
    span = L[-1][1] - L[0][0]; busy = sum(e - s for s, e, n, _ in L)
    gaps = collections.defaultdict(lambda: [0, 0])
    for a, b in zip(L, L[1:]):
        g = b[0] - a[1]
        gaps[(a[2], b[2])][0] += max(g, 0); gaps[(a[2], b[2])][1] += 1
    out.append("queue %s: %d dispatches, span %.1f ms, kernels %.1f ms, gaps %.1f ms (%.1f %% of the span)" % (q, len(L), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
    for (a, b), (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0]):
        out.append("    %-11s -> %-11s n %5d  avg %7.1f us  total %6.2f ms" % (a, b, n, g / n / 1e3, g / 1e6))
print("\n".join(out))
open("$OUT/summary.txt", "w").write("\n".join(out) + "\n" + open("$OUT/bench.json").read()[:300] + "\n")
PY
