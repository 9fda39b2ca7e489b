"""Offline study (recorded map-free traces of the bench workload, scripts/dump_traces.py): pausing on a TENTATIVE REPLAY.

The streaming scheduler speculates: seeds beyond the replay frontier are traced without the nodes of the unreplayed seeds in front
of them, and 45 % of the SMC iterations it runs are cut away by the replay later.  Pausing a trace when the nodes recorded so far
by lower-ranked seeds (finished or running) saturate its voxel was modelled in sim_pause.py and does not pay: four pauses in five
are lifted again, because most of those nodes are speculation themselves and get cut.

Here the prediction is a full tentative replay at every poll: all records of the admitted, unreplayed seeds are replayed in rank
order on top of the final map, exactly as the final replay would if nothing more were recorded -- so a trace that is cut takes its
own later nodes out of the picture.  A running trace whose tentative cut lies inside what it has recorded is paused (it costs no
step time but keeps its slot); it resumes when a later tentative replay no longer cuts it; seeds on tentatively saturated voxels
wait.  The final result is untouched (the final replay decides).  [Result: 177 k -> 119 k iterations at the old window, 1764 -> 1338 ms in
the cost model; with a window of 1536 slots and a lookahead of max(256, 100 %) 1255 ms -- built in stream_sched.h, measured on the GPU:
1544 -> 1179 ms.]

`target`: admit while fewer than `target` traces are RUNNING instead of by a lookahead in ranks (late seeds need 8 iterations per trace,
early ones 67: a rank window starts too many traces at the beginning and too few later).  [Result: 1255 -> 1165 ms in the model at 200
running traces (160 ... 256: within 1 %), with the rank lookahead max(512, 200 %) as a bound that no longer binds; a target that ramps up
with the frontier is worse (1172 - 1217 ms); built in stream_sched.h (option `target`), measured on the GPU: 1172 -> 1131 ms.]  `detail`:
a dict that receives the iterations run per trace (waste by rank: 11.8 k of the 35 k wasted iterations belong to the first 200 seeds).

`far` / `far_max`: also start, at most `far` per poll, seeds up to `far_max` ranks beyond the frontier whose 8^3 block no admitted trace
has touched yet (nobody in front of them has been there, so their traces are probably needed in full): 1255 -> 1223 ms in the model at
best, not built.

  python scripts/sim_tentative.py [gpurun_out/traces_1024_s2000.npz]
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


def simulate(window=768, look0=128, look_pct=50, poll=4, tentative=False, every=1, a_ms=A_MS, c_ms=C_MS, margin=0, hold_seeds=True, far=0, far_max=0, detail=None, target=0):
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
            ms += a_ms + c_ms * len(active)
            iters += len(active)
            keep = []
            for g in active:
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
    return dict(steps=steps, iters=iters, ms=round(ms), nodes=nodes, pauses=npause, resumes=nresume, tentative_nodes_per_poll=tent_work // max(polls, 1))


if __name__ == "__main__":
    print("base", simulate())
    for far, fmax in ((4, 2000), (8, 2000), (16, 2000), (8, 800)):
        print("far", far, fmax, simulate(window=1536, look0=256, look_pct=100, tentative=True, far=far, far_max=fmax), flush=True)
    for look0, pct, win in ((128, 50, 768), (256, 100, 1536), (512, 200, 1536), (1024, 400, 3072), (2048, 1000, 4096)):
        for every in (1, 4):
            print("tentative look max(%d, %d%%) window %d every %d:" % (look0, pct, win, every), simulate(window=win, look0=look0, look_pct=pct, tentative=True, every=every), flush=True)
