"""per-launch durations of the phased SMC kernels binned by the number of active traces.
usage: phased_bins.py <kernel_trace.csv> [skip_fraction]   (run the profile with PNR_BENCH_OPTS=groups=1)"""
import csv, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
get = lambda k: [r for r in rows if k in r["Kernel_Name"]]
sam, sums, upd = get("ph_sample"), get("ph_sums"), get("ph_update")
n0 = int(len(sam) * skip)
sam, sums, upd = sam[n0:], sums[n0:], upd[n0:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
d = np.array([dur(r) for r in sam]); ds = np.array([dur(r) for r in sums]); du = np.array([dur(r) for r in upd])
act = np.array([int(r["Grid_Size_X"]) // 256 for r in upd])
ns = np.array([int(r["Grid_Size_X"]) // 768 for r in sam]) // np.maximum(act, 1)
st = np.array([int(r["Start_Timestamp"]) for r in sam]); en = np.array([int(r["End_Timestamp"]) for r in upd])
print("launches", len(d), "sample %.1f sums %.1f update %.1f ms; span %.1f ms" % (d.sum(), ds.sum(), du.sum(), (en[-1] - st[0]) / 1e6))
period = np.diff(st) / 1e6
bins = [0, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 1 << 20]
for lo, hi in zip(bins[:-1], bins[1:]):
    m = (act > lo) & (act <= hi)
    if m.sum() == 0: continue
    mp = m[:-1]
    print(f"active ({lo},{hi}] launches {m.sum():4d} nsplit~{np.median(ns[m]):4.0f} sample avg {d[m].mean():.3f} sums avg {ds[m].mean():.3f} update avg {du[m].mean():.3f} "
          f"period avg {period[mp].mean() if mp.sum() else 0:.3f} total period {period[mp].sum():.1f} | per 256 traces: sample {(d[m] / act[m] * 256).mean():.3f} sums {(ds[m] / act[m] * 256).mean():.3f}")
