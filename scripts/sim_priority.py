"""Offline study on the recorded traces (scripts/dump_traces.py): a RANK-GRADED stepping rate.

The streaming scheduler steps every running trace at every step.  Here only the `kfull` lowest-ranked running traces step at every
step; the others step at every `stride`-th step only -- the pool of running traces is larger (`target`), the launch size about the
same, and the traces most likely to be cut by the replay (the highest-ranked) spend fewer iterations before the verdict arrives.
Everything else (tentative replay at every poll, admission by a target of running traces) is sim_tentative.simulate().

  python scripts/sim_priority.py [gpurun_out/traces_1024_s2000.npz]
"""
import sys

import numpy as np

d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traces_1024_s2000.npz")
T, pos, seeds = d["T"], d["pos"], d["seeds"]
n = len(seeds)
S = 1024
NPV = 4
r = lambda a: np.floor(a + 0.5).astype(np.int64)
vox = ((r(pos[..., 2]) * S + r(pos[..., 1])) * S + r(pos[..., 0]))
svox = ((r(seeds[:, 2]) * S + r(seeds[:, 1])) * S + r(seeds[:, 0])).tolist()
voxl = [row.tolist() for row in vox]
Tl = T.tolist()
A_MS, C_MS = 0.30, 0.0075
INF = 1 << 30


BLK = 8
def block_of(v):
    x = v % S; y = (v // S) % S; z = v // (S * S)
    return ((z // BLK) * (S // BLK) + (y // BLK)) * (S // BLK) + (x // BLK)


def simulate(kfull=0, stride=1, window=768, look0=128, look_pct=50, poll=4, tentative=False, every=1, a_ms=A_MS, c_ms=C_MS, margin=0, hold_seeds=True, far=0, far_max=0, detail=None, target=0):
    den = {}
    it = [0] * (2 * n)
    state = [0] * (2 * n)  # 0 not admitted, 1 running, 2 stopped, 3 paused, 4 skipped
    frontier = nxt = 0
    steps = iters = nodes = polls = 0
    ms = 0.0
    active = []
    paused = set()
    held = set()          # seeds not admitted because their voxel is tentatively saturated
    npause = nresume = 0
    tent_work = 0
    blocks = set()       # 8^3 blocks that hold a node of any admitted trace (maintained from the records at every poll)
    far_started = set()
    seen = [0] * (2 * n)
    launches = []
    while frontier < n:
        lim = frontier + max(look0, frontier * look_pct // 100)
        # admission: rank order, seeds on tentatively saturated voxels wait (they are looked at again at every poll)
        s = nxt
        while s < n and s < lim and len(active) + len(paused) + 2 <= window and (not target or len(active) + 2 <= target):
            if state[2 * s] == 0 and s not in held:
                if den.get(svox[s], 0) >= NPV:
                    state[2 * s] = state[2 * s + 1] = 4
                else:
                    for g in (2 * s, 2 * s + 1):
                        state[g] = 1
                        active.append(g)
            s += 1
        nxt = max(nxt, s)
        # far admission: seeds beyond the lookahead whose 8^3 block holds no node yet (final, tentative or of a started seed): nobody in
        # front of them has been there, so their traces are likely to be needed in full -- start them early (at most `far` per poll)
        if far and tentative:
            got = 0
            s2 = nxt
            while got < far and s2 < min(n, frontier + far_max) and len(active) + len(paused) + 2 <= window:
                if state[2 * s2] == 0 and s2 not in held:
                    b = block_of(svox[s2])
                    if b not in blocks:
                        for g in (2 * s2, 2 * s2 + 1):
                            state[g] = 1
                            active.append(g)
                        blocks.add(b)
                        far_started.add(s2)
                        got += 1
                s2 += 1
        for _ in range(poll):
            if not active:
                break
            steps += 1
            if kfull and stride > 1 and steps % stride:
                run = sorted(active)[:kfull]
                runset = set(run)
                rest = [g for g in active if g not in runset]
            else:
                run, rest = active, []
            ms += a_ms + c_ms * len(run)
            iters += len(run)
            launches.append(len(run))
            keep = rest
            for g in run:
                i = it[g]
                it[g] = i + 1
                if i >= Tl[g] or den.get(voxl[g][i], 0) >= NPV:
                    state[g] = 2
                    continue
                keep.append(g)
            active = keep
        if not active and not paused and not held and nxt >= n and all(state[2 * q] != 1 for q in range(frontier, n)):
            pass
        polls += 1
        if far:
            for g in range(2 * frontier, 2 * n):
                if state[g] in (1, 2, 3) and seen[g] < min(it[g], Tl[g]):
                    row = voxl[g]
                    for i in range(seen[g], min(it[g], Tl[g])):
                        blocks.add(block_of(row[i]))
                    seen[g] = min(it[g], Tl[g])
        # ---- final replay, seed-granular as in stream_sched.h
        while frontier < n:
            a = 2 * frontier
            if frontier in held:
                break  # (a held seed at the frontier is released below)
            if state[a] == 0:
                break
            if state[a] == 4:
                frontier += 1
                continue
            if state[a] in (1, 3) or state[a + 1] in (1, 3):
                # a paused trace at the frontier: its tentative cut is now final knowledge -- check it against the final map
                stuck = False
                for g in (a, a + 1):
                    if state[g] == 1:
                        stuck = True
                    elif state[g] == 3:
                        cut = False
                        if den.get(svox[frontier], 0) >= NPV:
                            cut = True
                        else:
                            dd = {}
                            for g2 in ((a,) if g == a else (a, a + 1)):
                                for i in range(min(it[g2], Tl[g2])):
                                    v = voxl[g2][i]
                                    if den.get(v, 0) + dd.get(v, 0) >= NPV:
                                        if g2 == g:
                                            cut = True
                                        break
                                    dd[v] = dd.get(v, 0) + 1
                        if cut:
                            state[g] = 2
                            paused.discard(g)
                        else:
                            state[g] = 1
                            paused.discard(g)
                            active.append(g)
                            nresume += 1
                            stuck = True
                if stuck or state[a] in (1, 3) or state[a + 1] in (1, 3):
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
        # a held seed that became the frontier: the final map decides
        while frontier < n and frontier in held:
            held.discard(frontier)
            if den.get(svox[frontier], 0) >= NPV:
                state[2 * frontier] = state[2 * frontier + 1] = 4
                frontier += 1
            else:
                for g in (2 * frontier, 2 * frontier + 1):
                    state[g] = 1
                    active.append(g)
                break
        if not tentative or polls % every:
            continue
        # ---- tentative replay of everything admitted and unreplayed, in rank order, on top of the final map
        tden = {}
        hi = max(nxt, max(far_started) + 1) if far_started else nxt
        for s in range(frontier, hi):
            a = 2 * s
            if state[a] == 4:
                continue
            if state[a] == 0:  # held or not yet admitted
                if hold_seeds and s < lim:
                    if den.get(svox[s], 0) + tden.get(svox[s], 0) >= NPV:
                        held.add(s)
                    elif s in held:
                        held.discard(s)
                continue
            sat = den.get(svox[s], 0) + tden.get(svox[s], 0) >= NPV
            for g in (a, a + 1):
                cut = 0 if sat else INF
                if not sat:
                    row = voxl[g]
                    for i in range(min(it[g], Tl[g])):
                        v = row[i]
                        if den.get(v, 0) + tden.get(v, 0) >= NPV:
                            cut = i
                            break
                        tden[v] = tden.get(v, 0) + 1
                        tent_work += 1
                if state[g] == 1 and cut + margin < it[g]:
                    state[g] = 3
                    paused.add(g)
                    npause += 1
                elif state[g] == 3 and cut == INF:
                    state[g] = 1
                    paused.discard(g)
                    active.append(g)
                    nresume += 1
        active = [g for g in active if state[g] == 1]
    if detail is not None:
        detail['it'] = list(it)
    return dict(mean_launch=round(sum(launches) / max(len(launches), 1), 1), steps=steps, iters=iters, ms=round(ms), nodes=nodes, pauses=npause, resumes=nresume, tentative_nodes_per_poll=tent_work // max(polls, 1))



if __name__ == "__main__":
    base = dict(window=1536, look0=4096, look_pct=1000, tentative=True, poll=2, a_ms=0.05, c_ms=0.0065)
    for tgt in (96, 120, 160, 200):
        print("target", tgt, "all traces at every step:", simulate(target=tgt, **base), flush=True)
    for tgt, kfull, stride in ((160, 80, 2), (160, 100, 2), (200, 80, 2), (200, 60, 2), (200, 100, 3), (240, 80, 3), (240, 60, 2), (320, 80, 4)):
        print("target", tgt, "kfull", kfull, "stride", stride, simulate(kfull=kfull, stride=stride, target=tgt, **base), flush=True)
