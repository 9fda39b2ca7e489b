"""how much do the phased SMC kernels of different streams overlap?  usage: overlap.py <kernel_trace.csv>"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ph_" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
ev = []
names = ["ph_sample", "ph_sums", "ph_update", "ph_predict"]
tot = {n: 0 for n in names}
for r in rows:
    n = next(k for k in names if k in r["Kernel_Name"])
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    tot[n] += b - a
    ev.append((a, 1, n)); ev.append((b, -1, n))
ev.sort()
cur = {n: 0 for n in names}
last = 0
busy = 0; both = 0; two_sample = 0
for t, d, n in ev:
    dt = t - last
    if sum(cur.values()) > 0: busy += dt
    if cur["ph_sample"] > 0 and cur["ph_sums"] > 0: both += dt
    if cur["ph_sample"] > 1: two_sample += dt
    cur[n] += d; last = t
span = max(int(r["End_Timestamp"]) for r in rows) - t0
print("span ms %.1f busy %.1f sample&sums overlap %.1f two samples %.1f" % (span / 1e6, busy / 1e6, both / 1e6, two_sample / 1e6))
print({k: round(v / 1e6, 1) for k, v in tot.items()}, "queues", sorted(set(r["Queue_Id"] for r in rows)))
