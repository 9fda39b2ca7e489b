"""offline study of the SHARDED streaming scheduler on the recorded one-shot traces of the bench workload
(gpurun_out/traces_1024_s2000.npz, scripts/dump_traces.py): G ranks, sorted seeds dealt round-robin, every rank keeps its own
window of traces, one exchange of finished records per poll, every rank replays the same records in seed order.  Cost model of
a rank's SMC step: a + c * active (measured on the GPU, DESIGN.md 4: a = 0.30 ms, c = 7.5 us), an exchange costs e ms and is a
barrier.  Prints time / steps / iterations for lookahead policies  max(look0, frontier * pct / 100)."""
import sys, numpy as np
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/traces_1024_s2000.npz")
T, pos, seeds = d["T"], d["pos"], d["seeds"]
n = len(seeds); S = 1024; NPV = 4
r = lambda a: np.floor(a + 0.5).astype(np.int64)
vox = (r(pos[..., 2]) * S + r(pos[..., 1])) * S + r(pos[..., 0])
svox = (r(seeds[:, 2]) * S + r(seeds[:, 1])) * S + r(seeds[:, 0])
A_MS, C_MS = 0.30, 0.0075

def simulate(G=1, window=768, look0=128, look_pct=50, poll=4, e_ms=0.15):
    den = {}
    it = np.zeros(2 * n, np.int32)
    state = np.zeros(2 * n, np.int8)  # 0 not admitted, 1 active, 2 finished, 4 skipped
    cutT = np.full(2 * n, -1, np.int32)
    frontier = 0
    nxt = list(range(G))  # next own seed per rank
    active = [[] for _ in range(G)]
    ms = 0.0; iters = 0; steps = 0; polls = 0
    wr = max(2, (window // G) & ~1) if G > 1 else window
    while frontier < n:
        lim = frontier + max(look0, frontier * look_pct // 100)
        for g in range(G):
            while nxt[g] < n and nxt[g] < lim and len(active[g]) + 2 <= wr:
                s = nxt[g]
                if den.get(svox[s], 0) >= NPV:
                    state[2 * s] = state[2 * s + 1] = 4
                else:
                    for q in (2 * s, 2 * s + 1):
                        state[q] = 1; active[g].append(q)
                nxt[g] += G
        tmax = 0.0
        for g in range(G):
            t = 0.0
            for _ in range(poll):
                if not active[g]: break
                t += A_MS + C_MS * len(active[g]); iters += len(active[g])
                if g == 0: steps += 1
                keep = []
                for q in active[g]:
                    i = it[q]
                    if i >= T[q]:
                        it[q] = i + 1; state[q] = 2; cutT[q] = T[q]; continue
                    if den.get(vox[q, i], 0) >= NPV:
                        it[q] = i + 1; state[q] = 2; cutT[q] = i + 1; continue
                    it[q] = i + 1; keep.append(q)
                active[g] = keep
            tmax = max(tmax, t)
        ms += tmax + (e_ms if G > 1 else 0.0); polls += 1
        while frontier < n:
            a, b = 2 * frontier, 2 * frontier + 1
            if state[a] == 0: break
            if state[a] == 4: frontier += 1; continue
            if state[a] == 1 or state[b] == 1: break
            if den.get(svox[frontier], 0) >= NPV: frontier += 1; continue
            for q in (a, b):
                for i in range(cutT[q]):
                    v = vox[q, i]
                    if den.get(v, 0) >= NPV: break
                    den[v] = den.get(v, 0) + 1
            frontier += 1
    return dict(ms=round(ms), steps0=steps, polls=polls, iters=iters)

if __name__ == "__main__":
    for G in (1, 2, 4, 8):
        rows = []
        for window in (768, 768 * G):
            for look0 in (128, 64 * G, 128 * G):
                for pct in (50, 100, 200, 400):
                    for poll in (2, 4):
                        res = simulate(G, window, look0, pct, poll)
                        rows.append((res["ms"], window, look0, pct, poll, res["steps0"], res["iters"]))
        rows = sorted(set(rows))
        print("G =", G)
        for b in rows[:6]: print("   ", b)
        base = [b for b in rows if b[1] == 768 and b[2] == 128 and b[3] == 50 and b[4] == 4]
        print("    default:", base[:1])
