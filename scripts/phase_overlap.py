"""Which kernels of the two trace groups run at the same time?  Reads a rocprofv3 --kernel-trace CSV of one bench step and prints, for the
tracing span, the share of time in which the two groups' queues run (sample, sample), (sums, sums), (sample, sums), ... together.
usage: phase_overlap.py <dir with *_kernel_trace.csv> [label]"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
def short(n):
    for k in ("ph_predict", "ph_cube", "ph_sample", "ph_sums", "ph_update", "ph_poll", "ph_snapshot"):
        if k in n: return k[3:]
    return None
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
ks = [k for k in ks if k[2]]
ks.sort()
first = [i for i, k in enumerate(ks) if k[2] == "predict"]
ks = ks[first[len(first) // 2]:]  # the timed step (after the warm-up)
byq = collections.defaultdict(list)
for k in ks: byq[k[3]].append(k)
qs = [q for q, L in byq.items() if sum(1 for k in L if k[2] == "sample") > 50]
assert len(qs) == 2, qs
ev = []
for qi, q in enumerate(qs):
    for s, e, n, _ in byq[q]:
        ev.append((s, 1, qi, n)); ev.append((e, 0, qi, ""))  # (an end sorts in front of a start at the same time)
ev.sort()
cur = ["idle", "idle"]
t_prev = ev[0][0]
acc = collections.Counter()
for t, kind, qi, n in ev:
    acc[tuple(sorted(cur))] += t - t_prev
    t_prev = t
    cur[qi] = n if kind == 1 else "idle"
span = ev[-1][0] - ev[0][0]
lab = sys.argv[2] if len(sys.argv) > 2 else ""
print(lab, "span %.1f ms:" % (span / 1e6), ", ".join("%s+%s %.1f%%" % (a, b, 100.0 * v / span) for (a, b), v in acc.most_common(8)))
