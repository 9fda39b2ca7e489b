import sys, numpy as np, collections
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traces_1024_s2000.npz")
T, pos, seeds = d["T"], d["pos"], d["seeds"]
NI = pos.shape[1]; n = len(seeds); S = 1024; NPV = 4
r = lambda a: np.floor(a + 0.5).astype(np.int64)
vox = (r(pos[..., 2]) * S + r(pos[..., 1])) * S + r(pos[..., 0])
svox = (r(seeds[:, 2]) * S + r(seeds[:, 1])) * S + r(seeds[:, 0])
A_MS = 0.3
step_ms = lambda a: A_MS + 0.0075 * a

def simulate(window=768, look0=128, look_pct=50, poll=4, pause=False, near=32, extra=0, verbose=False):
    den = {}
    prov = collections.defaultdict(list)   # voxel -> ranks of provisional nodes (unreplayed seeds), as known at the last poll
    it = np.zeros(2 * n, np.int32)
    seen = np.zeros(2 * n, np.int32)      # iterations whose nodes are in prov
    state = np.zeros(2 * n, np.int8)      # 0 not admitted, 1 active, 2 finished, 3 paused, 4 skipped
    cutT = np.full(2 * n, -1, np.int32)
    pv = np.full(2 * n, -1, np.int64)
    frontier = 0; nxt = 0; steps = 0; iters = 0; ms = 0.0; nodes = 0
    active = []; paused = set()
    npause = 0; nresume_near = 0; nresume_drop = 0; nconfirm = 0; stall_steps = 0
    def lower(v, s): return sum(1 for q in prov.get(v, ()) if q < s)
    while frontier < n:
        lim = frontier + max(look0, frontier * look_pct // 100)
        while nxt < n and nxt < lim and len(active) + len(paused) + 2 <= window:
            if den.get(svox[nxt], 0) >= NPV:
                state[2 * nxt] = state[2 * nxt + 1] = 4
            else:
                for g in (2 * nxt, 2 * nxt + 1):
                    state[g] = 1; active.append(g)
            nxt += 1
        if active:
            for _ in range(poll):
                if not active: break
                steps += 1; ms += step_ms(len(active)); iters += len(active)
                keep = []
                for g in active:
                    i = it[g]
                    if i >= T[g]:
                        it[g] = i + 1; state[g] = 2; cutT[g] = T[g]; continue
                    v = vox[g, i]
                    if den.get(v, 0) >= NPV:
                        it[g] = i + 1; state[g] = 2; cutT[g] = i + 1; continue
                    it[g] = i + 1
                    keep.append(g)
                active = keep
        elif paused:
            stall_steps += 1
        # ---- poll: host learns the new nodes, decides pauses
        if pause:
            for g in list(active) + [g for g in range(0)]:
                pass
            # new nodes of every admitted, unreplayed trace
            for s in range(frontier, nxt):
                for g in (2 * s, 2 * s + 1):
                    if state[g] in (1, 2, 3):
                        for i in range(seen[g], min(it[g], T[g])):
                            prov[vox[g, i]].append(s)
                        seen[g] = max(seen[g], min(it[g], T[g]))
            keep = []
            for g in active:
                s = g // 2
                hit = -1
                if s >= frontier + near:
                    for i in range(max(0, it[g] - poll), min(it[g], T[g])):
                        if lower(vox[g, i], s) >= NPV + extra: hit = i; break
                if hit >= 0:
                    state[g] = 3; pv[g] = vox[g, hit]; paused.add(g); npause += 1
                else:
                    keep.append(g)
            active = keep
        # ---- replay
        while frontier < n:
            a, b = 2 * frontier, 2 * frontier + 1
            if state[a] == 0: break
            if state[a] == 4: frontier += 1; continue
            if state[a] in (1, 3) or state[b] in (1, 3): break
            if den.get(svox[frontier], 0) < NPV:
                for g in (a, b):
                    for i in range(cutT[g]):
                        v = vox[g, i]
                        if den.get(v, 0) >= NPV: break
                        den[v] = den.get(v, 0) + 1; nodes += 1
            if pause:
                for g in (a, b):
                    for i in range(seen[g]):
                        lst = prov.get(vox[g, i])
                        if lst and frontier in lst: lst.remove(frontier)
            frontier += 1
        # ---- resolve paused traces
        if pause and paused:
            for g in list(paused):
                s = g // 2
                if den.get(pv[g], 0) >= NPV:           # confirmed: a definite stop
                    state[g] = 2; cutT[g] = it[g]; paused.discard(g); nconfirm += 1
                elif s < frontier + near:
                    state[g] = 1; active.append(g); paused.discard(g); nresume_near += 1
                elif lower(pv[g], s) < NPV + extra:
                    state[g] = 1; active.append(g); paused.discard(g); nresume_drop += 1
    return dict(steps=steps, iters=iters, ms=round(ms), nodes=nodes, pauses=npause, confirmed=nconfirm, resume_near=nresume_near, resume_drop=nresume_drop, stall=stall_steps)

if __name__ == "__main__":
    print("base", simulate())
    for near in (16, 48, 128):
        for extra in (0, 1):
            for lp in (50, 100):
                print("pause near", near, "extra", extra, "look_pct", lp, simulate(pause=True, near=near, extra=extra, look_pct=lp), flush=True)
