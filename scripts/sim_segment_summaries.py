"""Host model for "exact segment summaries" of ZNCC pass 1 (EXPERIMENTS.md round 4 (v), tracker.cpp:1929-1940).

The ordered mean of a chain (one particle, one sigma) is an f32 running sum S of M non-negative samples.  While S stays inside one
binade, a segment of adds is the exact two-case map "S/ulp even -> S + A, odd -> S + B", and A, B fall out of two surrogate chains a
sampling lane could run beside the interpolation -- pass 1 of ph_sums would then read 2 x C floats per segment and lane (C candidate
binades) instead of the segment's samples, and fall back to the stash where the map does not apply.  What decides whether that pays:

  (a) how often a segment can be summarised at all: the binade of S at its start must be among the C candidates the sampler was
      given BEFORE it sampled (here: the binades the trace's chains had at this segment in the previous SMC iteration), and S must
      not cross into the next binade inside the segment;
  (b) the fallback is per WAVE, not per lane: ph_sums walks 64 chains in lock-step through [sample][lane] rows of 256 B, so one
      lane that needs the stash costs the wave the segment's rows (a masked load still moves 64-B sectors, 16 lanes each).

This script measures (a) and (b) on real chains: oracle traces (np = 200, scales {2,4,6}) on a synthetic stack, every particle of
every iteration re-sampled in numpy with the oracle's template tables and f32 arithmetic, np.cumsum(float32) as the sequential sum.
It prints, per segment length (the 125-sample work item of ph_sample, its 25-sample rows), the share of (lane, segment) pairs and of
(64-lane wave, segment) pairs that can be summarised, with chains in particle order and sorted by their previous-iteration sum, and
the pass-1 bytes and the end-to-end estimate that follow.

  python scripts/sim_segment_summaries.py [size=96] [seeds=6] [iters=25]  > profiles/r05_segment_summaries_model.txt   (CPU only)
"""
import os
import sys
import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import orc  # noqa: E402
import synth  # noqa: E402

f32 = np.float32


def frames(P):
    """make_frame of smc_device.h (tracker.cpp:1893-1917), vectorised over particles P[:, 0:6]"""
    x, y, z, vx, vy, vz = [P[:, k].astype(f32) for k in range(6)]
    nrm = np.sqrt(vx.astype(np.float64) ** 2 + vy.astype(np.float64) ** 2).astype(f32)
    ok = nrm > 0.0001
    sg = np.where(vy < 0, -1.0, 1.0).astype(f32)
    safe = np.where(ok, nrm, f32(1))
    ux = np.where(ok, sg * (vy / safe), f32(1)).astype(f32)
    uy = np.where(ok, -sg * (vx / safe), f32(0)).astype(f32)
    uz = np.zeros_like(ux)
    wx = (uy * vz - uz * vy).astype(f32)
    wy = (-ux * vz + uz * vx).astype(f32)
    wz = (ux * vy - uy * vx).astype(f32)
    return (x, y, z), (-vx, -vy, -vz), (ux, uy, uz), (wx, wy, wz)


def interp(img, X, Y, Z):
    """Tracker::interp 3-D (tracker.cpp:2178-2213) in f32"""
    l, h, w = img.shape
    xc = np.clip(X, f32(0), f32(w - 1.001)); yc = np.clip(Y, f32(0), f32(h - 1.001)); zc = np.clip(Z, f32(0), f32(l - 1.001))
    x1 = xc.astype(np.int64); y1 = yc.astype(np.int64); z1 = zc.astype(np.int64)
    fx = (xc - x1.astype(f32)).astype(f32); fy = (yc - y1.astype(f32)).astype(f32); fz = (zc - z1.astype(f32)).astype(f32)
    g = lambda dz, dy, dx: img[z1 + dz, y1 + dy, x1 + dx].astype(f32)
    one = f32(1)
    a = ((one - fy) * ((one - fx) * g(0, 0, 0) + fx * g(0, 0, 1)) + fy * ((one - fx) * g(0, 1, 0) + fx * g(0, 1, 1))).astype(f32)
    b = ((one - fy) * ((one - fx) * g(1, 0, 0) + fx * g(1, 0, 1)) + fy * ((one - fx) * g(1, 1, 0) + fx * g(1, 1, 1))).astype(f32)
    return ((one - fz) * a + fz * b).astype(f32)


def chains(img, P, vuw):
    """samples [np][M] of every particle for one sigma"""
    p, nv, u, w = frames(P)
    out = []
    for a in range(3):
        t = (p[a][:, None] + vuw[None, :, 0] * nv[a][:, None]).astype(f32)
        t = (t + vuw[None, :, 1] * u[a][:, None]).astype(f32)
        t = (t + vuw[None, :, 2] * w[a][:, None]).astype(f32)
        out.append(t)
    return interp(img, out[0], out[1], out[2])


def binade(S):
    with np.errstate(divide="ignore"):
        return np.where(S > 0, np.floor(np.log2(np.maximum(S, f32(1e-30)))), -200).astype(np.int32)


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    L = orc.load_oracle()
    img = synth.synth(size, size, max(32, size // 2), seed=3)
    sigs = [2.0, 4.0, 6.0]
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, img, sigs, 2.0)
    s = orc.extract_seeds(L, 5, orc.j8(L, J, jmin, jmax), Vx, Vy, Vz)
    T = orc.Tracker(L, sigs, 2, 200, iters, 3.0, 0.3, zdist=2.0)
    corr, _ = T.zncc(img, s[:, :6])
    order = np.argsort(-corr, kind="stable")[:nseeds]
    models = {si: T.model(si)[0].astype(f32) for si in (1, 2)}  # the two long templates (M = 5625)
    print(f"# stack {img.shape[::-1]}, {nseeds} best seeds, {iters} iterations at most, np = 200, scales {sigs}; chains of sigma = 4 and 6 (M = {len(models[1])})")
    # statistics per segment length
    SEG = (125, 25)
    keys = ("lane", "lane_in", "pred_same", "pred_2", "pred_3", "wave", "wave_2", "wave_3", "wave_2_sorted", "wave_3_sorted", "wave_in", "wave_in_sorted")
    tot = {sl: dict.fromkeys(keys, 0) for sl in SEG}
    nch = 0
    for q, si in enumerate(order):
        Tn, stop, xc, xf, idx, neff = T.trace(img, s[si, :6].copy(), max_dbg=iters)
        prev = {}  # sigma -> {seglen: binades [np][nseg], "final": sums [np]} of the previous iteration
        for it in range(min(Tn, iters)):
            P = xf[it][:, :6]
            # the particle's parent in the previous iteration (tracker.cpp:1108): the resampled index where that step resampled, else itself
            par = np.arange(len(P))
            if it > 0 and neff[it - 1] / len(P) < 0.8:
                par = idx[it - 1].astype(np.int64)
            for sg, vuw in models.items():
                V = chains(img, P, vuw)                      # [200][5625]
                Srun = np.cumsum(V, axis=1, dtype=f32)       # sequential f32 adds
                nch += len(V)
                cur = {"final": Srun[:, -1].copy()}
                for sl in SEG:
                    nseg = V.shape[1] // sl
                    starts = np.concatenate([np.zeros((len(V), 1), f32), Srun[:, sl - 1:nseg * sl - 1:sl]], axis=1)  # S at the start of segment k
                    ends = Srun[:, sl - 1:nseg * sl:sl]
                    e0, e1 = binade(starts), binade(ends)
                    inside = (e0 == e1) & (starts > 0)           # the map applies: no crossing inside the segment (and S > 0)
                    cur[sl] = e0
                    t = tot[sl]
                    t["lane"] += inside.size
                    t["lane_in"] += int(inside.sum())
                    if sg not in prev:
                        continue
                    pe = prev[sg][sl][par]                        # the parent chain's binade at this segment, one iteration ago
                    d = e0 - pe
                    h1 = inside & (d == 0)                        # one candidate: the parent's binade (2 surrogate adds per sample)
                    h2 = inside & ((d == 0) | (d == 1))           # two candidates: the parent's and the next (4 adds)
                    h3 = inside & (np.abs(d) <= 1)                # three (6 adds)
                    t["pred_same"] += int(h1.sum()); t["pred_2"] += int(h2.sum()); t["pred_3"] += int(h3.sum())
                    o = np.argsort(prev[sg]["final"][par], kind="stable")  # chains dealt to the waves by their parent's final sum
                    for base in range(0, len(V), 64):
                        t["wave"] += nseg
                        t["wave_in"] += int(inside[base:base + 64].all(axis=0).sum())
                        t["wave_2"] += int(h2[base:base + 64].all(axis=0).sum())
                        t["wave_3"] += int(h3[base:base + 64].all(axis=0).sum())
                        t["wave_in_sorted"] += int(inside[o][base:base + 64].all(axis=0).sum())
                        t["wave_2_sorted"] += int(h2[o][base:base + 64].all(axis=0).sum())
                        t["wave_3_sorted"] += int(h3[o][base:base + 64].all(axis=0).sum())
                prev[sg] = cur
        print(f"# seed {q}: T = {Tn}, stop {stop}", flush=True)
    print(f"# {nch} chains of 5625 samples")
    print("\nper lane and segment: S stays inside one binade | ... and that binade is the parent chain's of the previous iteration | ... or the next one up (2 candidates) | ... or one either side (3)")
    for sl in SEG:
        t = tot[sl]
        print(f"segment {sl:3d}: {t['lane_in'] / t['lane']:.3f} | {t['pred_same'] / t['lane']:.3f} | {t['pred_2'] / t['lane']:.3f} | {t['pred_3'] / t['lane']:.3f}")
    print("\nper WAVE (64 chains in lock-step) and segment -- the unit the fallback is paid in: no lane crosses (an oracle that knew every binade) | all 64 lanes summarised with 2 candidates | with 3;"
          "  chains in particle order / dealt to the waves by their parent's final sum")
    for sl in SEG:
        t = tot[sl]
        w = max(t["wave"], 1)
        print(f"segment {sl:3d}: {t['wave_in'] / w:.3f} / {t['wave_in_sorted'] / w:.3f} | {t['wave_2'] / w:.3f} / {t['wave_2_sorted'] / w:.3f} | {t['wave_3'] / w:.3f} / {t['wave_3_sorted'] / w:.3f}")
    # what it buys: pass 1 reads, per wave and segment, either the summaries (2 floats per candidate binade and lane) or the segment's
    # rows; the stash round trip is 1 write + 2 reads of every sample, pass 1 is one of the three; the summaries are written and read once
    print("\nmodel: pass-1 bytes = share x (2 C / segment length) + (1 - share); stash traffic = (2 + pass 1 + 2 C / segment length) / 3; "
          "sampler VALU x (37 + 2 C) / 37 (37 vector instructions per sample today)")
    for sl in SEG:
        t = tot[sl]
        w = max(t["wave"], 1)
        for name, key, Cn in (("oracle binades, particle order", "wave_in", 1), ("2 candidates, particle order", "wave_2", 2), ("2 candidates, sorted", "wave_2_sorted", 2),
                              ("3 candidates, sorted", "wave_3_sorted", 3)):
            share = t[key] / w
            p1 = share * (2.0 * Cn / sl) + (1 - share)
            total = (2 + p1 + 2.0 * Cn / sl) / 3.0
            print(f"segment {sl:3d}, {name:32s}: waves summarised {share:.3f} -> pass-1 bytes x {p1:.3f}, stash traffic x {total:.3f}, sampler VALU x {(37 + 2 * Cn) / 37:.3f}")


if __name__ == "__main__":
    main()
