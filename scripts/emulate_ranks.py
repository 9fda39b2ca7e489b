"""What would ONE rank of an N-GPU run of the bench workload (BASELINE configs[3]) cost?  Measured on the one GPU of the test box:
rank 0 is the real thing -- its z-slab of Frangi + seeds + scores and its share of the sorted seeds through pnr_trace_replay_sharded
on the GPU --, ranks 1..N-1 are host threads that play back the recorded map-free traces of their seeds through the same scheduler
(pnr_sched_playback: instantaneous "GPUs"), all joined by the library's shared-memory all-gather.  The ranks are symmetric (seeds
dealt round-robin), so rank 0's wall time is the step time an N-GPU node would show, up to the imbalance between ranks and the
RCCL collectives of the front half (a 2-float all-reduce and one seed all-gather).  Every run is checked against the one-GPU graph.

  python scripts/emulate_ranks.py [--worlds 1,2,4,8] [--opts "look0=512,look_pct=400;groups=1"] [--size 1024] [--seeds 2000]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import synth  # noqa: E402
import pnr_amd  # noqa: E402
from pnr_amd import lib, multigpu  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--opts", default="", help="';'-separated option sets for rank 0 and the played-back ranks, e.g. 'look0=512,look_pct=400;groups=1'")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--seeds", type=int, default=2000)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    S = a.size
    sigs, zdist = (2.0, 4.0, 6.0), 2.0
    img = synth.synth_torch(S, S, S, seed=3, device="cuda:0")
    p = pnr_amd.make_params(sigmas=sigs, np_=200, ni=200, zdist=zdist)
    ctx = pnr_amd.Context(p, 0)
    shape = (S, S, S)
    ctx.set_volume_device(img.data_ptr(), shape, keepalive=img)
    ctx.set_profiling(True)
    ctx.frangi()
    s0 = ctx.extract_seeds()
    seeds = ctx.score_filter_sort(s0)[:a.seeds]
    n1, l1, nt1, it1 = ctx.trace_replay(seeds)
    t0 = time.perf_counter()
    n1, l1, nt1, it1 = ctx.trace_replay(seeds)
    t_one = 1e3 * (time.perf_counter() - t0)
    print(f"one GPU: {len(seeds)} seeds, tracing {t_one:.0f} ms, {it1} iterations, {len(n1) - 1} nodes", flush=True)
    T, stop, xc, _ = ctx.trace_batch(seeds)  # every trace to its map-free end: what the played-back ranks replay
    ni = p.ni
    xcf = np.ascontiguousarray(xc).view(np.float32).reshape(2 * len(seeds), ni, 8)
    traces = {}
    for i, sd in enumerate(seeds):
        for d, sgn in enumerate((1.0, -1.0)):
            q6 = np.array([sd["x"], sd["y"], sd["z"], sgn * sd["vx"], sgn * sd["vy"], sgn * sd["vz"]], np.float32)
            traces[q6.tobytes()] = (int(T[2 * i + d]), xcf[2 * i + d])
    lookup = lambda q6: traces[np.asarray(q6, np.float32).tobytes()]
    halo = multigpu.frangi_halo(p)

    for world in [int(w) for w in a.worlds.split(",")]:
        # ---- front half of rank 0: its slab (with halo) of Frangi, its layers' seeds, their scores
        front = []
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mine, jmin, jmax = multigpu.frangi_seeds_sharded(ctx, img.data_ptr(), shape, None, 0, world, reduce_fn=lambda x, y: (x, y))
            mine = ctx.score_filter(mine)
            front.append(1e3 * (time.perf_counter() - t0))
        for spec in (a.opts.split(";") if a.opts else [""]):
            kv = dict(x.split("=") for x in spec.split(",") if x)
            kv = {k: int(v) for k, v in kv.items()}
            for k in ("look0", "look_pct", "groups", "window", "poll", "tentative", "target", "lag"):
                ctx.set_option(k, kv.get(k, {"look0": 0, "look_pct": -1, "groups": 0, "window": 0, "poll": 4, "tentative": 1, "target": -1, "lag": -1}[k]))
            best = None
            for rep in range(a.reps):
                name = f"pnr_emu_{os.getpid()}_{world}_{rep}_{abs(hash(spec)) % 100000}"
                out = [None] * world
                X = [None] * world

                def run(r):
                    try:
                        if world > 1:
                            X[r] = lib.ShmExchange(name, r, world, 1 << 20)
                        if r == 0:
                            ctx.reset_kernel_ms()
                            t0 = time.perf_counter()
                            res = ctx.trace_replay_sharded(seeds, 0, world, X[0]) if world > 1 else ctx.trace_replay(seeds)
                            out[0] = (res, 1e3 * (time.perf_counter() - t0))
                        else:
                            out[r] = (lib.sched_playback(p, shape, seeds, lookup, r, world, X[r], window=kv.get("window", 768), groups=kv.get("groups", 0) or (1 if world > 1 else 2),
                                                         poll=kv.get("poll", 4), look0=kv.get("look0", 0), look_pct=kv.get("look_pct", -1), tentative=bool(kv.get("tentative", 1)), target=kv.get("target", -1), lag=kv.get("lag", -1)), 0.0)
                    except Exception as e:  # noqa: BLE001
                        out[r] = e

                th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
                for t in th:
                    t.start()
                for t in th:
                    t.join(timeout=600)
                for x in X:
                    if x is not None:
                        x.close()
                for r in range(world):
                    if isinstance(out[r], Exception) or out[r] is None:
                        raise SystemExit(f"rank {r} of {world}: {out[r]}")
                (nodes, links, nt, it0), ms = out[0]
                same = len(nodes) == len(n1) and np.array_equal(links, l1) and all(np.array_equal(nodes[k], n1[k], equal_nan=True) for k in n1.dtype.names)
                its = [it0] + [out[r][0][3] for r in range(1, world)]
                km = {g: ctx.kernel_ms(g) for g in ("smc_predict", "smc", "smc_sums", "smc_update")}
                if best is None or ms < best[0]:
                    best = (ms, its, same, km)
            ms, its, same, km = best
            fr = min(front)
            print(f"world {world} [{spec or 'defaults'}]: front half (rank 0's slab) {fr:.1f} ms, tracing {ms:.0f} ms, step ~{fr + ms:.0f} ms -> "
                  f"{S ** 3 / (fr + ms) / 1e3:.0f} Mvox/s; iterations rank 0 {its[0]}, all ranks {sum(its)} ({sum(its) / it1:.2f} x one GPU); "
                  f"graph {'identical' if same else 'DIFFERENT'}; rank 0: {km['smc'][1]} SMC steps, per launch predict / sample / sums / update "
                  + " / ".join(f"{km[g][0] / max(km[g][1], 1):.3f}" for g in ("smc_predict", "smc", "smc_sums", "smc_update")) + " ms", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
