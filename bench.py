#!/usr/bin/env python3
"""bench.py -- Mvox/s traced (Frangi + SMC step) on a synthetic stack, MI355X.

One "step" = one pass of the hot path over one stack: Frangi (all scales) -> J8 -> seed
extraction -> ZNCC seed scoring/filter/sort -> SMC tracing of the first `--seeds` sorted seeds
(both directions) -> host replay.  The input stack is resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

N>1 (weak scaling): every rank owns an independent synthetic stack (seed 3+rank) and runs the
whole path on it -- the unit of sharding is the stack, no data-path collective; the only
collective is the final gather of the node graphs to rank 0 (RCCL over xGMI), inside the timed
region.  `--mode shard` instead shards the sorted seeds of ONE replicated stack round-robin over
the ranks (BASELINE configs[3]; strong scaling of the tracing stage).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel, HIP-event time measured live on the kernel's stream) and `cpu_baseline` (oracle C
restatement, 1 core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="cubic stack edge (BASELINE configs[2]: 1024)")
    ap.add_argument("--seeds", type=int, default=2000, help="sorted seeds traced per stack (configs[3]: 2000)")
    ap.add_argument("--np", type=int, default=200)
    ap.add_argument("--ni", type=int, default=200)
    ap.add_argument("--mode", choices=["stacks", "shard"], default="stacks")
    ap.add_argument("--one-shot", action="store_true", help="trace every seed to its map-free end + one replay (no early DENSITY stops)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed full-occupancy measurement of the sampling kernel")
    ap.add_argument("--driver", choices=["phased", "persistent"], default="phased",
                    help="SMC scheduler (pnr_set_smc_driver): one launch per phase (default) or one persistent work-group per trace")
    return ap.parse_args()


def cpu_baseline(img_dev, sigs, zdist, np_, n_iters_gpu, nvox, nseed_init):
    """Oracle (C restatement, 1 thread) on a bounded sample of the same stack: Frangi + J8 + seeds
    on a 160x160x80 crop around the stack centre, SMC on the crop's best seeds for ~150 iterations.
    Scaled to the metric's unit with the GPU step's own work counts."""
    import orc
    L = orc.load_oracle()
    S = img_dev.shape[0]
    cw, ch, cl = min(160, S), min(160, S), min(80, S)
    z0, y0, x0 = (S - cl) // 2, int(0.62 * S) - ch // 2, (S - cw) // 2
    y0 = max(0, min(S - ch, y0))
    crop = img_dev[z0:z0 + cl, y0:y0 + ch, x0:x0 + cw].contiguous().cpu().numpy()
    t0 = time.perf_counter()
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, crop, sigs, zdist)
    J8 = orc.j8(L, J, jmin, jmax)
    t1 = time.perf_counter()
    seeds = orc.extract_seeds(L, 5, J8, Vx, Vy, Vz)
    t2 = time.perf_counter()
    T = orc.Tracker(L, sigs, 2, np_, 25, 3.0, 0.3, zdist=zdist)
    corr, _ = T.zncc(crop, seeds[:, :6]) if len(seeds) else (np.zeros(0), None)
    t3 = time.perf_counter()
    order = np.argsort(-corr, kind="stable")[:3]
    iters = 0
    for i in order:
        for sgn in (1, -1):
            q = seeds[i, :6].copy()
            q[3:] *= sgn
            Tn, stop, *_ = T.trace(crop, q)
            iters += min(Tn + 1, 25)
    t4 = time.perf_counter()
    vs = crop.size
    t_frangi_vox = (t1 - t0) / vs
    t_seed_vox = (t2 - t1) / vs
    t_eval = (t3 - t2) / max(len(seeds), 1)
    t_iter = (t4 - t3) / max(iters, 1)
    total = nvox * (t_frangi_vox + t_seed_vox) + nseed_init * t_eval + n_iters_gpu * t_iter
    return {
        "value": nvox / total / 1e6, "unit": "Mvox/s", "cores": 1, "kind": "port",
        "sample": (f"oracle/pnr_oracle.c, 1 thread: Frangi+J8 {t1 - t0:.2f}s and seeds {t2 - t1:.3f}s on a {cw}x{ch}x{cl} crop of "
                   f"the same stack; {len(seeds)} znccBBB evals {t3 - t2:.2f}s; {iters} SMC iterations (np={np_}) {t4 - t3:.2f}s; "
                   f"scaled to the step's {nvox} voxels, {nseed_init} seed scores and {n_iters_gpu} SMC iterations"),
        "frangi_Mvox_s": 1e-6 / t_frangi_vox, "smc_ms_per_iter": 1e3 * t_iter,
    }


def stash_row_floats(np_):
    """f32 values per template sample and trace in the phased driver's stash: 64 per full group of chains (np particles
    + the pending centroid) plus the last group's chains rounded up to 16 (pnr_amd/csrc/smc_phased.hip)"""
    n = np_ + 1
    return 64 * (n // 64) + (((n % 64) + 15) // 16) * 16


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no CPU path)", file=sys.stderr)
        sys.exit(2)
    # rehearsal on a box with fewer GPUs than ranks (never the measured configuration): PNR_BENCH_BACKEND=gloo puts the collectives
    # on CPU tensors and lets the ranks share the visible GPUs
    backend = os.environ.get("PNR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    coll_dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1 or os.environ.get("PNR_BENCH_FORCE_DIST"):  # FORCE_DIST: one-rank rehearsal of the RCCL code path (RANK / WORLD_SIZE / MASTER_* set by hand)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    import synth
    import pnr_amd
    from pnr_amd import multigpu

    S = a.size
    sigs, zdist = (2.0, 4.0, 6.0), 2.0
    stack_seed = 3 + (rank if a.mode == "stacks" else 0)
    img = synth.synth_torch(S, S, S, seed=stack_seed, device=f"cuda:{local}")
    torch.cuda.synchronize()
    p = pnr_amd.make_params(sigmas=sigs, np_=a.np, ni=a.ni, zdist=zdist)
    ctx = pnr_amd.Context(p, local)
    ctx.set_smc_driver(a.driver)
    ctx.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
    ctx.set_profiling(not os.environ.get("PNR_BENCH_NOPROF"))  # NOPROF: how much do the HIP events of the kernel timers cost? (diagnostic; no roofline then)
    nvox = S * S * S

    def step():
        st = {}
        t0 = time.perf_counter()
        if a.mode == "shard" and dist is not None:  # one stack: z-slabs of Frangi + seeds per rank, 2-float all-reduce, seed all-gather
            mine, _, _ = multigpu.frangi_seeds_sharded(ctx, img.data_ptr(), (S, S, S), dist, rank, world, device=coll_dev)
            t1 = time.perf_counter()
            s0 = multigpu.gather_seeds(mine, dist, rank, world, coll_dev)
        else:
            ctx.frangi()
            t1 = time.perf_counter()
            s0 = ctx.extract_seeds()
        t2 = time.perf_counter()
        s = ctx.score_filter_sort(s0)[:a.seeds]
        t3 = time.perf_counter()
        if a.mode == "shard" and dist is not None:
            nodes, links, T = multigpu.trace_sharded(ctx, s, dist, rank, world, device=coll_dev)
            iters = int((T + (T < a.ni)).sum())
        else:
            if a.one_shot:
                T, stop, xc, _ = ctx.trace_batch(s)
                nodes, links, _ = ctx.replay(s, T, xc)
                iters = int((T + (T < a.ni)).sum())
            else:
                nodes, links, _, iters = ctx.trace_replay(s)
            if dist is not None:
                multigpu.gather_graphs(nodes, links, dist, rank, world, coll_dev)
        t5 = time.perf_counter()
        st.update(frangi_ms=1e3 * (t1 - t0), seeds_ms=1e3 * (t2 - t1), score_ms=1e3 * (t3 - t2), trace_replay_gather_ms=1e3 * (t5 - t3),
                  n_seeds_init=len(s0), n_seeds=len(s), iters=iters, nodes=len(nodes) - 1)
        return st

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    ctx.reset_kernel_ms()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        it = torch.tensor([st["iters"]], device=coll_dev, dtype=torch.int64)
        dist.all_reduce(it)
        iters_all = int(it.item())
    else:
        iters_all = st["iters"]

    if rank == 0:
        units = nvox * (world if a.mode == "stacks" else 1)
        ms_step = 1e3 * dt / a.steps
        value = units / (dt / a.steps) / 1e6
        if os.environ.get("PNR_BENCH_NOPROF"):
            print(json.dumps({"ms_per_step": ms_step, "value": value, "note": "kernel timers off: no roofline"}))
            return
        km = {g: ctx.kernel_ms(g) for g in ("gauss", "hessian_eigen", "j8", "seed_maxima", "zncc", "smc", "smc_sums", "smc_predict", "smc_update")}
        # dominant kernel: the sampling kernel of the particle filter -- ph_sample (one launch per SMC step over all
        # active traces; the "smc" timer group) with the phased driver, smc_trace (one launch per batch, sampling +
        # sums + update) with the persistent one.  Algorithmic bytes: 8 corner bytes x sum(M_sigma)
        # samples per particle evaluation, (np+1) evaluations per SMC iteration; ph_sample performs every one of them.
        kname = "ph_sample<54, false>" if a.driver == "phased" else "smc_trace"
        Mtot = sum(len(ctx.table(f"model_wgt{s}")) for s in range(len(sigs)))
        smc_ms, smc_n = km["smc"]
        smc_all_ms = sum(km[g][0] for g in ("smc", "smc_sums", "smc_predict", "smc_update"))
        evals = st["iters"] * (a.np + 1)
        # several launches per step (seed-rank batches): bytes per launch / average launch duration
        # = total bytes of the timed region / total smc_trace device time
        bytes_launch = 8.0 * Mtot * evals * a.steps / max(smc_n, 1)
        achieved = bytes_launch / (smc_ms / max(smc_n, 1) * 1e-3) / 1e9 if smc_ms > 0 else 0.0
        fr_ms = (km["gauss"][0] + km["hessian_eigen"][0] + km["j8"][0]) / a.steps
        # HBM traffic per launch: PMC counters (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, calibrated on
        # a known byte count in the same access pattern: scripts/prof_traffic.sh), collected for THIS workload and
        # committed under profiles/; null for any other workload
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_1024_s2000.json" if a.driver == "persistent" else "r01h_traffic_1024_s2000.json")
        if S == 1024 and a.seeds == 2000 and a.np == 200 and a.ni == 200 and not a.one_shot and a.mode == "stacks" and os.path.exists(tpath):
            tj = json.load(open(tpath)).get("smc_trace" if a.driver == "persistent" else "ph_sample", {})
            if "FETCH_SIZE" in tj and "WRITE_SIZE" in tj:
                traffic = tj["FETCH_SIZE"]["bytes_per_launch"] + tj["WRITE_SIZE"]["bytes_per_launch"]
        out = {
            "metric": "Mvox/s traced (Frangi+SMC step) on 1024^3 synthetic stack; % HBM roofline" if S == 1024 else f"Mvox/s traced (Frangi+SMC step) on {S}^3 synthetic stack; % HBM roofline",
            "value": value, "unit": "Mvox/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak" if a.mode == "stacks" else "strong", "vs_baseline": None,
            "dtype": "f32 (+f64 3x3 eigen-solver)", "data": "synthetic",
            "config": {"workload": f"{S}^3 synthetic u8 stack (tests/synth.py seed {stack_seed}), scales={{2,4,6}}, zdist=2, np={a.np}, ni={a.ni}, "
                                   f"first {a.seeds} sorted seeds traced in both directions per stack, tolerance=5, znccth=0.3, step=2, kappa=3",
                       "parallelism": ("1 GPU" if world == 1 else (f"{world} independent stacks, one per GPU; RCCL gather of node graphs" if a.mode == "stacks"
                                       else f"one stack: Frangi + seeds in z-slabs, sorted seeds round-robin over {world} GPUs; RCCL all-reduce(min,max), all-gather of seeds and trace records"))},
            "roofline": {"kernel": kname, "bound": "hbm", "bytes_per_launch": bytes_launch, "avg_launch_ms": smc_ms / max(smc_n, 1), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "algorithmic gather bytes 8*sum(M_sigma)=%d B per particle evaluation; served from L1/L2, see DESIGN.md" % (8 * Mtot)},
            "roofline_frangi": {"kernels": "gauss_x_u8+gauss_axis(y,z)+hessian_eigen+j8", "bound": "hbm", "achieved": (len(sigs) + 12) * nvox / (fr_ms * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (len(sigs) + 12) * nvox / (fr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "(S+12) B/voxel compulsory bytes over the Frangi kernel group; fp64 eigen-solver is the limiter"},
            "roofline_sums": None if a.driver != "phased" or km["smc_sums"][0] <= 0 else {
                "kernel": "ph_sums", "bound": "hbm", "launches": km["smc_sums"][1], "avg_launch_ms": km["smc_sums"][0] / max(km["smc_sums"][1], 1),
                "achieved": 2.0 * 4 * Mtot * stash_row_floats(a.np) * st["iters"] * a.steps / (km["smc_sums"][0] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": 2.0 * 4 * Mtot * stash_row_floats(a.np) * st["iters"] * a.steps / (km["smc_sums"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "real bytes: the ordered sums stream every stashed f32 sample twice (mean, then corr): 2 x 4 x sum(M) x %d B per SMC iteration" % stash_row_floats(a.np)},
            "stages_ms": {k: v for k, v in st.items() if k.endswith("_ms")},
            "kernel_ms_per_step": {g: km[g][0] / a.steps for g in km},
            "counts": {k: v for k, v in st.items() if not k.endswith("_ms")},
            # particle evaluations / time of the whole SMC kernel group (sampling + sums + predict + update)
            "Mevals_per_s_smc": evals * a.steps / smc_all_ms / 1e3 if smc_all_ms > 0 else None,
            "smc_group_GBs": 8.0 * Mtot * evals * a.steps / (smc_all_ms * 1e-3) / 1e9 if smc_all_ms > 0 else None,
            "smc_launches_per_step": smc_n / a.steps,
            "Mvox_per_s_frangi": nvox / (fr_ms * 1e-3) / 1e6,
        }
        if world == 1 and not a.one_shot and not a.no_extra:
            # the same kernel with every CU busy: ONE launch over all traces (outside the timed region)
            ctx.reset_kernel_ms()
            s_all = ctx.score_filter_sort(ctx.extract_seeds())[:a.seeds]
            T1, _, _, _ = ctx.trace_batch(s_all)
            ms1, n1 = ctx.kernel_ms("smc")
            ev1 = int((T1 + (T1 < a.ni)).sum()) * (a.np + 1)
            out["roofline_full_occupancy"] = {
                "kernel": kname, "launches": n1, "avg_launch_ms": ms1 / max(n1, 1), "achieved": 8.0 * Mtot * ev1 / (ms1 * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8.0 * Mtot * ev1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "Mevals_per_s": ev1 / ms1 / 1e3,
                "note": "all %d traces started together, traced to their map-free end (no early DENSITY stops): %d SMC iterations; measured after the timed region" % (len(T1), ev1 // (a.np + 1))}
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(img, list(sigs), zdist, a.np, st["iters"], nvox, st["n_seeds_init"])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
