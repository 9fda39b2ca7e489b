"""offline study of trace scheduling on recorded one-shot traces (scripts/dump_traces.py): executed iterations and steps of
(a) the streaming scheduler as built (confirmed map only) and (b) provisional pausing with resume at the replay frontier."""
import sys, numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traces_1024_s2000.npz")
T, pos, seeds = d["T"], d["pos"], d["seeds"]
NI = pos.shape[1]; n = len(seeds); S = 1024; NPV = 4
r = lambda a: np.floor(a + 0.5).astype(np.int64)  # round half away for non-negative coords
vox = (r(pos[..., 2]) * S + r(pos[..., 1])) * S + r(pos[..., 0])  # [2n][ni]
svox = (r(seeds[:, 2]) * S + r(seeds[:, 1])) * S + r(seeds[:, 0])
step_ms = lambda a: 0.55 + 0.0075 * a

def simulate(window=768, look0=128, look_pct=100, poll=4, provisional=False, nopause_ahead=16, rank_aware=False, extra=0, verbose=False):
    den = {}                 # confirmed map
    prov = {}                # provisional map (rank-unaware)
    it = np.zeros(2 * n, np.int32)        # iterations executed so far
    state = np.zeros(2 * n, np.int8)      # 0 not admitted, 1 active, 2 finished (complete), 3 paused (provisional), 4 skipped
    cutT = np.full(2 * n, -1, np.int32)   # records available when finished/paused
    nopause = np.zeros(2 * n, bool)
    frontier = 0; nxt = 0; steps = 0; iters = 0; ms = 0.0; resumes = 0; stalls = 0
    nodes = 0
    pauses = [0]
    active = []
    def finished(g): return state[g] in (2, 3, 4)
    while frontier < n:
        # admission
        lim = frontier + max(look0, frontier * look_pct // 100)
        while nxt < n and nxt < lim and len(active) + 2 <= window:
            if den.get(svox[nxt], 0) >= NPV:
                state[2 * nxt] = state[2 * nxt + 1] = 4
            else:
                for g in (2 * nxt, 2 * nxt + 1):
                    state[g] = 1; active.append(g)
                    nopause[g] = nxt < frontier + nopause_ahead
            nxt += 1
        # steps
        if active:
            for _ in range(poll):
                if not active: break
                steps += 1; ms += step_ms(len(active)); iters += len(active)
                keep = []
                for g in active:
                    i = it[g]
                    if i >= T[g]:            # this iteration fails (corr / out of volume / ni reached): trace complete
                        it[g] = i + 1; state[g] = 2; cutT[g] = T[g]; continue
                    v = vox[g, i]
                    if den.get(v, 0) >= NPV:  # confirmed DENSITY stop
                        it[g] = i + 1; state[g] = 2; cutT[g] = i + 1; continue
                    if provisional and not nopause[g]:
                        pv = prov.get(v)
                        if pv is not None and ((sum(1 for q in pv if q < g // 2) >= NPV + extra) if rank_aware else (len(pv) >= NPV + extra)):
                            it[g] = i + 1; state[g] = 3; cutT[g] = i + 1; pauses[0] += 1; continue
                    if provisional: prov.setdefault(v, []).append(g // 2)
                    it[g] = i + 1
                    keep.append(g)
                active = keep
        # replay frontier
        while frontier < n:
            a, b = 2 * frontier, 2 * frontier + 1
            if state[a] == 0: break
            if state[a] == 4: frontier += 1; continue
            if state[a] == 1 or state[b] == 1: break
            if den.get(svox[frontier], 0) >= NPV: frontier += 1; continue
            # transactional replay
            log = []; ok = True; need = None
            for g in (a, b):
                cut = False
                for i in range(cutT[g]):
                    v = vox[g, i]
                    if den.get(v, 0) >= NPV: cut = True; break
                    log.append(v); den[v] = den.get(v, 0) + 1
                if not cut and state[g] == 3:   # ran out of records without a stop: must resume
                    ok = False; need = g; break
            if not ok:
                for v in log: den[v] -= 1
                state[need] = 1; nopause[need] = True; active.append(need); resumes += 1
                # (it[need] continues from where it paused)
                break
            nodes += len(log)
            frontier += 1
        # nopause for seeds close to the frontier
        if provisional:
            for s in range(frontier, min(n, frontier + nopause_ahead)):
                nopause[2 * s] = nopause[2 * s + 1] = True
        if not active and nxt >= n and frontier < n and state[2 * frontier] not in (0,):
            pass
    return dict(steps=steps, iters=iters, ms=round(ms), nodes=nodes, resumes=resumes, pauses=pauses[0])

best = []
for window in (512, 768, 1024, 2048):
    for look0 in (48, 64, 96, 128, 192):
        for look_pct in (30, 50, 70, 100, 150):
            for poll in (2, 4, 8):
                res = simulate(window=window, look0=look0, look_pct=look_pct, poll=poll)
                best.append((res["ms"], window, look0, look_pct, poll, res["steps"], res["iters"]))
best.sort()
for b in best[:12]: print(b)
print("...")
for b in best[-3:]: print(b)
