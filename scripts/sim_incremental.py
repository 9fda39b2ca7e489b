"""Offline study of the streaming trace scheduler on the recorded map-free traces of the bench workload (scripts/dump_traces.py):
what would ITERATION-granular replay buy?

Today (stream_sched.h) the replay advances seed by seed: the records of a seed are applied when BOTH its traces have stopped, so
the nodes of a long trace at the frontier reach the GPU's density map only when it ends.  Here the replay pointer is
(seed, direction, iteration): every iteration of the frontier trace is replayed at the first poll after it was recorded, its
nodes go to the map at once, and the trace itself is ended by the replay the moment its own DENSITY stop is known.  The result of
the replay is the same by construction (same records, same order); only when the map learns of them changes.

Cost model (measured): step = A_MS + C_MS x active traces; a poll every `poll` steps.
  python scripts/sim_incremental.py [gpurun_out/traces_1024_s2000.npz]
"""
import sys

import numpy as np

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traces_1024_s2000.npz")
T, pos, seeds = d["T"], d["pos"], d["seeds"]
NI = pos.shape[1]
n = len(seeds)
S = 1024
NPV = 4
r = lambda a: np.floor(a + 0.5).astype(np.int64)
vox = ((r(pos[..., 2]) * S + r(pos[..., 1])) * S + r(pos[..., 0]))
svox = ((r(seeds[:, 2]) * S + r(seeds[:, 1])) * S + r(seeds[:, 0]))
voxl = [row.tolist() for row in vox]
Tl = T.tolist()
A_MS, C_MS = 0.30, 0.0075


def simulate(window=768, look0=128, look_pct=50, poll=4, incremental=False, a_ms=A_MS, c_ms=C_MS, kill_lag=0):
    den = {}
    it = [0] * (2 * n)        # iterations executed
    state = [0] * (2 * n)     # 0 not admitted, 1 running, 2 stopped (by itself, by the map or by the replay), 4 skipped
    frontier = 0              # first seed not completely replayed
    rp_dir, rp_it = 0, 0      # replay pointer inside the frontier seed (incremental mode)
    nxt = 0
    steps = iters = nodes = 0
    ms = 0.0
    active = []
    seed_checked = False
    while frontier < n:
        lim = frontier + max(look0, frontier * look_pct // 100)
        while nxt < n and nxt < lim and len(active) + 2 <= window:
            if den.get(svox[nxt], 0) >= NPV:
                state[2 * nxt] = state[2 * nxt + 1] = 4
            else:
                for g in (2 * nxt, 2 * nxt + 1):
                    state[g] = 1
                    active.append(g)
            nxt += 1
        for _ in range(poll):
            if not active:
                break
            steps += 1
            ms += a_ms + c_ms * len(active)
            iters += len(active)
            keep = []
            for g in active:
                if state[g] != 1:
                    continue
                i = it[g]
                it[g] = i + 1
                if i >= Tl[g] or den.get(voxl[g][i], 0) >= NPV:  # its own end, or the map's DENSITY stop
                    state[g] = 2
                    continue
                keep.append(g)
            active = keep
        # ---- replay
        while frontier < n:
            a = 2 * frontier
            if state[a] == 0:
                break
            if state[a] == 4:
                frontier += 1
                rp_dir = rp_it = 0
                seed_checked = False
                continue
            if not incremental:
                if state[a] == 1 or state[a + 1] == 1:
                    break
                if den.get(svox[frontier], 0) < NPV:
                    for g in (a, a + 1):
                        for i in range(min(it[g], Tl[g])):
                            v = voxl[g][i]
                            if den.get(v, 0) >= NPV:
                                break
                            den[v] = den.get(v, 0) + 1
                            nodes += 1
                frontier += 1
                continue
            # incremental: the seed's skip test once, then iteration by iteration as far as the records reach
            if not seed_checked:
                seed_checked = True
                if den.get(svox[frontier], 0) >= NPV:  # the reference would not trace it at all
                    for g in (a, a + 1):
                        state[g] = 2
                    frontier += 1
                    rp_dir = rp_it = 0
                    seed_checked = False
                    continue
            blocked = False
            while rp_dir < 2:
                g = a + rp_dir
                done = False
                while True:
                    if rp_it >= Tl[g]:
                        done = True
                        break
                    if rp_it >= it[g]:  # not recorded yet
                        if state[g] != 1:
                            done = True  # stopped by the map at an iteration the replay has already passed: cannot happen; be safe
                        break
                    v = voxl[g][rp_it]
                    if den.get(v, 0) >= NPV:
                        done = True
                        break
                    den[v] = den.get(v, 0) + 1
                    nodes += 1
                    rp_it += 1
                if not done:
                    blocked = True
                    break
                state[g] = 2  # the replay knows the trace's end: the GPU trace is ended at the next poll
                rp_dir += 1
                rp_it = 0
            if blocked:
                break
            frontier += 1
            rp_dir = rp_it = 0
            seed_checked = False
    return dict(steps=steps, iters=iters, ms=round(ms), nodes=nodes)


if __name__ == "__main__":
    print("recorded traces:", 2 * n, "map-free iterations", int(T.sum()))
    for inc in (False, True):
        for look0, pct in ((128, 50), (128, 100), (256, 100), (256, 200), (512, 400), (64, 25), (32, 25)):
            print("incremental" if inc else "seed-granular", "lookahead max(%d, %d%%)" % (look0, pct), simulate(look0=look0, look_pct=pct, incremental=inc), flush=True)
