#!/usr/bin/env python3
"""bench.py -- Mvox/s traced (Frangi + SMC step) on a synthetic stack, MI355X.

One "step" = one pass of the hot path over one stack: Frangi (all scales) -> J8 -> seed
extraction -> ZNCC seed scoring/filter/sort -> SMC tracing of the first `--seeds` sorted seeds
(both directions) -> host replay.  The input stack is resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 without WORLD_SIZE in the environment: this process starts N ranks of itself with
torch.distributed.run (one per GPU, RCCL) BEFORE anything touches a GPU, waits for them and exits
with their code; launched by torch.distributed.run it is one of the ranks.

N > 1, default `--mode shard` = BASELINE configs[3] (strong scaling of ONE 1024^3 stack): every rank
holds a replica of the u8 stack; Frangi + seed extraction run in z-slabs (2-float all-reduce of
Jmin / Jmax, all-gather of the scored seeds), the sorted seeds are dealt round-robin and traced by
pnr_trace_replay_sharded (per-poll all-gather of finished trace records, every rank replays them in
global seed order: early DENSITY stops as on one GPU, every rank ends with the same node graph).
`--mode stacks` is the replica mode: every rank owns an independent stack (weak scaling, no data-path
collective, one final gather of the node graphs).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event time measured live on the kernel's stream), `roofline_smc_group` (the whole particle-filter
kernel group), `roofline_frangi`, and `cpu_baseline` (oracle C restatement, 1 core, bounded sample).
Scheduler options for experiments: PNR_BENCH_OPTS="groups=2,window=1024" (pnr_set_option keys).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

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
    ap.add_argument("--mode", choices=["shard", "stacks"], default="shard", help="N > 1: one stack sharded over the ranks (configs[3]) or one stack per rank")
    ap.add_argument("--one-shot", action="store_true", help="trace every seed to its map-free end + one replay (no early DENSITY stops; 1 GPU)")
    ap.add_argument("--cpu-baseline", choices=["default", "survey", "off"], default="default",
                    help="oracle on one host core: a 32-plane slab + 10 seeds (~1 min), SURVEY 8(d)'s 256-plane slab + 50 seeds (~10 min), or none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed full-occupancy measurement of the sampling kernel")
    ap.add_argument("--driver", choices=["phased", "persistent"], default="phased",
                    help="SMC scheduler (pnr_set_smc_driver): one launch per phase (default) or one persistent work-group per trace")
    return ap.parse_args()


def launch_ranks(a):
    """--gpus N given to a plain `python bench.py`: become the launcher of N ranks (nothing in this process has touched a GPU)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    return subprocess.run(cmd, env=env).returncode


def cpu_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


class heartbeat:
    """a line on stderr every minute while a long, silent host computation runs (the oracle's C calls release the GIL)"""

    def __init__(self, what):
        import threading
        self.what, self.stop, self.t0 = what, threading.Event(), time.perf_counter()
        self.th = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop.wait(60.0):
            print(f"[bench] {self.what}: {time.perf_counter() - self.t0:.0f} s ...", file=sys.stderr, flush=True)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join()


def cpu_baseline(img_dev, sigs, zdist, np_, ni, counts, nvox, kind):
    """Oracle (oracle/pnr_oracle.c, the C restatement of the reference's scalar loops; 1 thread) on a bounded sample of the SAME
    stack: Frangi + J8 + seed extraction on a z-slab of full xy extent (so the strided y / z passes see the real
    row and plane pitches), seed scores on the slab's seeds, full-depth traces of the slab's best seeds until the time cap.
    Scaled to the step: voxels x (Frangi + seeds per voxel) + seed scores + the SMC iterations the sequential reference would
    run for the same graph (one per node + the stopping iteration of every trace -- no speculation)."""
    import numpy as np
    import orc
    L = orc.load_oracle()
    S = img_dev.shape[0]
    planes, nseed, cap_s = (256, 50, 300.0) if kind == "survey" else (32, 10, 25.0)
    planes = min(planes, S)
    z0 = (S - planes) // 2
    slab = img_dev[z0:z0 + planes].contiguous().cpu().numpy()
    t0 = time.perf_counter()
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, slab, sigs, zdist)
    J8 = orc.j8(L, J, jmin, jmax)
    t1 = time.perf_counter()
    seeds = orc.extract_seeds(L, 5, J8, Vx, Vy, Vz)
    t2 = time.perf_counter()
    del J, J8, Vx, Vy, Vz
    T = orc.Tracker(L, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    nsc = min(len(seeds), 2000)
    corr, _ = T.zncc(slab, seeds[:nsc, :6]) if nsc else (np.zeros(0), None)
    t3 = time.perf_counter()
    order = np.argsort(-corr, kind="stable")[:nseed]
    iters = ntr = 0
    for i in order:
        for sgn in (1, -1):
            if time.perf_counter() - t3 > cap_s:
                break
            q = seeds[i, :6].copy()
            q[3:] *= sgn
            Tn, stop, *_ = T.trace(slab, q)
            iters += min(Tn + 1, ni)
            ntr += 1
    t4 = time.perf_counter()
    vs = slab.size
    t_frangi_vox = (t1 - t0) / vs
    t_seed_vox = (t2 - t1) / vs
    t_eval = (t3 - t2) / max(nsc, 1)
    t_iter = (t4 - t3) / max(iters, 1)
    seq_iters = counts["nodes"] + 2 * counts["traces_used"]
    total = nvox * (t_frangi_vox + t_seed_vox) + counts["n_seeds_init"] * t_eval + seq_iters * t_iter
    model, ncpu = cpu_info()
    return {
        "value": nvox / total / 1e6, "unit": "Mvox/s", "cores": 1, "cores_total": ncpu, "cpu_model": model, "kind": "port",
        "sample": (f"oracle/pnr_oracle.c, 1 thread of {ncpu} ({model}): Frangi+J8 {t1 - t0:.1f}s and seeds {t2 - t1:.2f}s on the {planes}x{S}x{S} "
                   f"slab z={z0}..{z0 + planes} of the same stack; {nsc} znccBBB seed scores {t3 - t2:.2f}s; {ntr} full-depth traces of the slab's best "
                   f"seeds, {iters} SMC iterations (np={np_}) {t4 - t3:.1f}s (cap {cap_s:.0f}s); scaled to the step's {nvox} voxels, "
                   f"{counts['n_seeds_init']} seed scores and {seq_iters} sequential SMC iterations (nodes + 2 per trace; the GPU ran {counts['iters_all']})"),
        "frangi_Mvox_s": 1e-6 / t_frangi_vox, "seeds_Mvox_s": 1e-6 / max(t_seed_vox, 1e-12), "smc_ms_per_iter": 1e3 * t_iter,
        "Mevals_per_s_smc": (np_ + 1) / t_iter / 1e6,
    }


def stash_row_floats(np_):
    """f32 values per template sample and trace in the phased driver's stash: 64 per full group of chains (np particles
    + the pending centroid) plus the last group's chains rounded up to 16 (pnr_amd/csrc/smc_phased.hip)"""
    n = np_ + 1
    return 64 * (n // 64) + (((n % 64) + 15) // 16) * 16


def main():
    a = parse()
    if a.no_cpu_baseline:
        a.cpu_baseline = "off"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))  # before torch / HIP are even imported
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no CPU path)", file=sys.stderr)
        sys.exit(2)
    # rehearsal on a box with fewer GPUs than ranks (never the measured configuration): PNR_BENCH_BACKEND=gloo puts the collectives
    # on CPU tensors and lets the ranks share the visible GPUs
    backend = os.environ.get("PNR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        local = local % max(1, ndev)
    elif local >= ndev:
        print(f"bench.py: rank {rank} needs GPU {local} but only {ndev} are visible (PNR_BENCH_BACKEND=gloo rehearses N ranks on fewer GPUs)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    coll_dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1 or os.environ.get("PNR_BENCH_FORCE_DIST"):  # FORCE_DIST: one-rank rehearsal of the RCCL code path (RANK / WORLD_SIZE / MASTER_* set by hand)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()  # the ranks the backend actually joined
    import synth
    import pnr_amd
    from pnr_amd import multigpu

    S = a.size
    sigs, zdist = (2.0, 4.0, 6.0), 2.0
    # FORCE_SHARD (with FORCE_DIST): the sharded code path -- its RCCL collectives included -- on a world of one (rehearsal)
    shard = (world > 1 or bool(os.environ.get("PNR_BENCH_FORCE_SHARD")) and dist is not None) and a.mode == "shard"
    stack_seed = 3 + (rank if (world > 1 and a.mode == "stacks") else 0)
    img = synth.synth_torch(S, S, S, seed=stack_seed, device=f"cuda:{local}")
    torch.cuda.synchronize()
    p = pnr_amd.make_params(sigmas=sigs, np_=a.np, ni=a.ni, zdist=zdist)
    ctx = pnr_amd.Context(p, local)
    ctx.set_smc_driver(a.driver)
    ctx.set_option("local_ranks", local_world)  # the host threads of the seed flood fill are shared between the ranks of this node
    opts = ctx.set_options(os.environ.get("PNR_BENCH_OPTS"))
    ctx.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
    ctx.set_profiling(not os.environ.get("PNR_BENCH_NOPROF"))  # NOPROF: how much do the HIP events of the kernel timers cost? (diagnostic; no roofline then)
    nvox = S * S * S
    # the per-poll exchange of finished trace records: host data on both ends (pinned records in, host replay out), so on ONE node
    # it goes through the library's shared-memory all-gather (a few microseconds, no Python in the loop); across nodes, or with
    # PNR_BENCH_EXCHANGE=rccl, through RCCL (torch.distributed.all_gather_into_tensor with pinned staging)
    exchange, exchange_kind = None, None
    if shard:
        want = os.environ.get("PNR_BENCH_EXCHANGE", "shm" if local_world == world else "rccl")
        if want == "shm":
            # rank 0 probes /dev/shm and picks the segment's name (its pid + clock: never the name of a crashed earlier job)
            ok = torch.tensor([1, os.getpid(), time.time_ns() % (1 << 40)], dtype=torch.int64, device=coll_dev)
            if rank == 0:
                try:
                    probe = pnr_amd.lib.ShmExchange(f"pnr_probe_{os.getpid()}", 0, 1, 64)
                    probe.close()
                except Exception:  # noqa: BLE001 -- no usable /dev/shm: every rank falls back together
                    ok[0] = 0
            dist.broadcast(ok, 0)
            if int(ok[0].item()):
                exchange = pnr_amd.lib.ShmExchange(f"pnr_bench_{os.getuid()}_{int(ok[1].item())}_{int(ok[2].item())}", rank, world, 1 << 20)
                exchange_kind = "shared memory (one node)"
        if exchange is None:
            exchange = multigpu.make_exchange(dist, world, coll_dev)
            exchange_kind = "RCCL all_gather_into_tensor" if backend == "nccl" else backend

    def step():
        st = {}
        t0 = time.perf_counter()
        if shard:  # one stack: z-slabs of Frangi + seeds per rank, 2-float all-reduce, scored seeds all-gathered
            mine, _, _ = multigpu.frangi_seeds_sharded(ctx, img.data_ptr(), (S, S, S), dist, rank, world, device=coll_dev)
            t1 = time.perf_counter()
            n_init = len(mine)
            mine = ctx.score_filter(mine)  # znccBBB of this slab's seeds on the whole stack, threshold
            t2 = time.perf_counter()
            s0 = multigpu.gather_seeds(mine, dist, rank, world, coll_dev)
            s = ctx.sort_seeds(s0)[:a.seeds]
            t3 = time.perf_counter()
            nodes, links, ntr, iters = ctx.trace_replay_sharded(s, rank, world, exchange)
        else:
            ctx.frangi()
            t1 = time.perf_counter()
            s0 = ctx.extract_seeds()
            n_init = len(s0)
            t2 = time.perf_counter()
            s = ctx.score_filter_sort(s0)[:a.seeds]
            t3 = time.perf_counter()
            if a.one_shot:
                T, stop, xc, _ = ctx.trace_batch(s)
                nodes, links, ntr = ctx.replay(s, T, xc)
                iters = int((T + (T < a.ni)).sum())
            else:
                nodes, links, ntr, iters = ctx.trace_replay(s)
            if dist is not None and a.mode == "stacks":
                multigpu.gather_graphs(nodes, links, dist, rank, world, coll_dev)
        t5 = time.perf_counter()
        if shard:
            st.update(frangi_seeds_slab_ms=1e3 * (t1 - t0), score_slab_ms=1e3 * (t2 - t1), gather_sort_ms=1e3 * (t3 - t2), trace_replay_exchange_ms=1e3 * (t5 - t3))
        else:
            st.update(frangi_ms=1e3 * (t1 - t0), seeds_ms=1e3 * (t2 - t1), score_ms=1e3 * (t3 - t2), trace_replay_gather_ms=1e3 * (t5 - t3))
        st.update(n_seeds_init=n_init, n_seeds=len(s), iters=iters, nodes=len(nodes) - 1, traces_used=int(ntr))
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
        it = torch.tensor([st["iters"], st["n_seeds_init"]], device=coll_dev, dtype=torch.int64)
        dist.all_reduce(it)
        iters_all = int(it[0].item())
        if shard:
            st["n_seeds_init"] = int(it[1].item())
    else:
        iters_all = st["iters"]
    st["iters_all"] = iters_all

    if rank == 0:
        units = nvox * (world if (world > 1 and a.mode == "stacks") else 1)
        ms_step = 1e3 * dt / a.steps
        value = units / (dt / a.steps) / 1e6
        if os.environ.get("PNR_BENCH_NOPROF"):
            print(json.dumps({"ms_per_step": ms_step, "value": value, "n_gpus": world, "note": "kernel timers off: no roofline"}))
            if dist is not None:
                dist.barrier()
                dist.destroy_process_group()
            return
        km = {g: ctx.kernel_ms(g) for g in ("gauss", "hessian_tile", "hessian_eigen", "j8", "seed_maxima", "zncc", "smc", "smc_sums", "smc_predict", "smc_update")}
        # dominant kernel: the sampling kernel of the particle filter -- ph_sample (one launch per SMC step over all
        # active traces; the "smc" timer group) with the phased driver, smc_trace (one launch per batch, sampling +
        # sums + update) with the persistent one.  Algorithmic bytes: 8 corner bytes x sum(M_sigma)
        # samples per particle evaluation, (np+1) evaluations per SMC iteration; ph_sample performs every one of them
        # (rank 0's kernels and rank 0's iterations when the seeds are sharded).
        kname = "ph_sample<54, false>" if a.driver == "phased" else "smc_trace"
        Mtot = sum(len(ctx.table(f"model_wgt{s}")) for s in range(len(sigs)))
        smc_ms, smc_n = km["smc"]
        smc_all_ms = sum(km[g][0] for g in ("smc", "smc_sums", "smc_predict", "smc_update"))
        evals = st["iters"] * (a.np + 1)
        # several launches per step: bytes per launch / average launch duration = total bytes of the timed region / total device time
        bytes_launch = 8.0 * Mtot * evals * a.steps / max(smc_n, 1)
        achieved = bytes_launch / (smc_ms / max(smc_n, 1) * 1e-3) / 1e9 if smc_ms > 0 else 0.0
        fr_ms = (km["gauss"][0] + km["hessian_tile"][0] + km["hessian_eigen"][0] + km["j8"][0]) / a.steps
        fr_vox = nvox if not shard else None  # a rank's slab + halo when sharded: no per-stack Frangi figure then
        # HBM traffic per launch: PMC counters (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, calibrated on
        # a known byte count in the same access pattern: scripts/prof_traffic.sh), collected for THIS workload and
        # committed under profiles/ -- read from that file, not measured in this run; null for any other workload
        traffic, traffic_src = None, None
        for cand in ("r02_traffic_1024_s2000.json", "r01h_traffic_1024_s2000.json"):
            tpath = os.path.join(ROOT, "profiles", cand if a.driver == "phased" else "r01_traffic_1024_s2000.json")
            if S == 1024 and a.seeds == 2000 and a.np == 200 and a.ni == 200 and not a.one_shot and world == 1 and os.path.exists(tpath):
                tj = json.load(open(tpath)).get("smc_trace" if a.driver == "persistent" else "ph_sample", {})
                if "FETCH_SIZE" in tj and "WRITE_SIZE" in tj:
                    traffic = tj["FETCH_SIZE"]["bytes_per_launch"] + tj["WRITE_SIZE"]["bytes_per_launch"]
                    traffic_src = "profiles/" + os.path.basename(tpath)
                    break
        group_GBs = 8.0 * Mtot * evals * a.steps / (smc_all_ms * 1e-3) / 1e9 if smc_all_ms > 0 else None
        if world == 1:
            par = "1 GPU"
        elif shard:
            par = (f"one stack on {world} GPUs: Frangi + seeds + seed scores in z-slabs, sorted seeds round-robin; RCCL all-reduce(min,max), all-gather of "
                   "seeds; finished trace records all-gathered once per rotation of the trace groups (host data: shared memory on one node, RCCL otherwise); "
                   "every rank replays in global seed order")
        else:
            par = f"{world} independent stacks, one per GPU; RCCL gather of node graphs"
        out = {
            "metric": "Mvox/s traced (Frangi+SMC step) on 1024^3 synthetic stack; % HBM roofline" if S == 1024 else f"Mvox/s traced (Frangi+SMC step) on {S}^3 synthetic stack; % HBM roofline",
            "value": value, "unit": "Mvox/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak" if (world > 1 and a.mode == "stacks") else "strong", "vs_baseline": None,
            "dtype": "f32 (+f64 3x3 eigen-solver)", "data": "synthetic",
            "config": {"workload": f"{S}^3 synthetic u8 stack (tests/synth.py seed {stack_seed}), scales={{2,4,6}}, zdist=2, np={a.np}, ni={a.ni}, "
                                   f"first {a.seeds} sorted seeds traced in both directions per stack, tolerance=5, znccth=0.3, step=2, kappa=3",
                       "parallelism": par, "backend": backend if world > 1 else None, "record_exchange": exchange_kind, "options": opts or None},
            "roofline": {"kernel": kname, "bound": "lds-gather/valu", "bytes_per_launch": bytes_launch, "avg_launch_ms": smc_ms / max(smc_n, 1), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "note": "ALGORITHMIC gather bytes, 8*sum(M_sigma)=%d B per particle evaluation, over the launch time of the sampling kernel alone, "
                                 "priced against the HBM peak as SURVEY 8(d) prescribes; the gather is served from the LDS cube (PMC: VALU issue ~80 %%, LDS "
                                 "bank conflicts ~60 %% of LDS cycles), so the kernel is VALU / LDS-gather bound, not HBM bound; `traffic` (real HBM bytes per "
                                 "launch, PMC) is read from the committed profile named in traffic_source, not measured in this run; the ordered sums of "
                                 "the same evaluations are the separate kernel ph_sums -- roofline_smc_group prices the whole evaluation" % (8 * Mtot)},
            "roofline_smc_group": None if smc_all_ms <= 0 else {
                "kernels": "ph_predict+ph_sample+ph_sums+ph_update" if a.driver == "phased" else "smc_trace", "bound": "lds-gather/valu + hbm (stash)",
                "achieved": group_GBs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": group_GBs / HBM_PEAK_GBS,
                "device_ms_per_step": smc_all_ms / a.steps, "Mevals_per_s": evals * a.steps / smc_all_ms / 1e3,
                "note": "SURVEY 8(d): 8*sum(M) algorithmic bytes x particle evaluations / device time of the whole SMC kernel group (t_smc)"},
            "roofline_frangi": None if not fr_vox or fr_ms <= 0 else {
                "kernels": "gauss_x_u8_t+gauss_axis_t(y,z)+hessian_tile+eigen_queue+j8", "bound": "hbm", "achieved": (len(sigs) + 12) * fr_vox / (fr_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (len(sigs) + 12) * fr_vox / (fr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "device_ms_per_step": fr_ms,
                "note": "(S+12) B/voxel compulsory bytes over the Frangi kernel group; VALU-bound: three Gaussian passes with separate multiply and add (the reference's rounding) and the Hessian stencil with its zero-response tests; the fp64 JAMA eigen-solver runs only where the response can reach J8 > 0"},
            "roofline_sums": None if a.driver != "phased" or km["smc_sums"][0] <= 0 else {
                "kernel": "ph_sums", "bound": "hbm", "launches": km["smc_sums"][1], "avg_launch_ms": km["smc_sums"][0] / max(km["smc_sums"][1], 1),
                "achieved": 2.0 * 4 * Mtot * stash_row_floats(a.np) * st["iters"] * a.steps / (km["smc_sums"][0] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": 2.0 * 4 * Mtot * stash_row_floats(a.np) * st["iters"] * a.steps / (km["smc_sums"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "stash bytes of np + 1 chains: the ordered sums stream every stashed f32 sample twice (mean, then corr): 2 x 4 x sum(M) x %d B per SMC iteration; an upper bound since exact duplicate poses are evaluated once (PMC: 15 %% fewer bytes on this workload)" % stash_row_floats(a.np)},
            "stages_ms": {k: v for k, v in st.items() if k.endswith("_ms")},
            "kernel_ms_per_step": {g: km[g][0] / a.steps for g in km},
            "counts": {k: v for k, v in st.items() if not k.endswith("_ms")},
            "smc_launches_per_step": smc_n / a.steps,
            "Mvox_per_s_frangi": (nvox / (fr_ms * 1e-3) / 1e6) if (fr_vox and fr_ms > 0) else None,
        }
        groups = ctx.get_option("groups")
        out["config"]["trace_groups"] = groups
        if groups > 1 and a.driver == "phased":
            # launches of different trace groups overlap: a kernel's duration includes the time it shares the CUs, and the sum of the
            # durations counts that time twice -- the group figure is taken over the wall time of the tracing stage instead
            tr_ms = st.get("trace_replay_gather_ms", st.get("trace_replay_exchange_ms"))
            g2 = 8.0 * Mtot * evals / (tr_ms * 1e-3) / 1e9
            out["roofline_smc_group"].update(achieved=g2, frac=g2 / HBM_PEAK_GBS, device_ms_per_step=None, wall_ms_per_step=tr_ms, Mevals_per_s=evals / tr_ms / 1e3,
                                             note=f"SURVEY 8(d): 8*sum(M) algorithmic bytes x particle evaluations / WALL time of the tracing stage (kernels of the {groups} trace "
                                                  "groups overlap on their streams, host replay and polls included)")
            out["roofline"]["note"] += (f"; {groups} trace groups: this kernel's launches overlap the other group's ph_sums / ph_predict / ph_update launches, so its average "
                                        "duration includes shared time -- roofline_isolated is the same kernel with one trace group (untimed extra step)")
        if world == 1 and not shard and not a.one_shot and not a.no_extra:
            s_all = ctx.score_filter_sort(ctx.extract_seeds())[:a.seeds]
            if groups > 1 and a.driver == "phased":
                # the same step with ONE trace group (launches never overlap): what each kernel needs alone
                ctx.set_option("groups", 1)
                ctx.reset_kernel_ms()
                t0i = time.perf_counter()
                _, _, _, it_iso = ctx.trace_replay(s_all)
                t_iso = 1e3 * (time.perf_counter() - t0i)
                ctx.set_option("groups", groups)
                ki = {g: ctx.kernel_ms(g) for g in ("smc", "smc_sums", "smc_predict", "smc_update")}
                ev_i = it_iso * (a.np + 1)
                all_i = sum(v[0] for v in ki.values())
                out["roofline_isolated"] = {
                    "kernel": kname, "bound": "lds-gather/valu", "launches": ki["smc"][1], "avg_launch_ms": ki["smc"][0] / max(ki["smc"][1], 1),
                    "achieved": 8.0 * Mtot * ev_i / (ki["smc"][0] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8.0 * Mtot * ev_i / (ki["smc"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "smc_group_frac": 8.0 * Mtot * ev_i / (all_i * 1e-3) / 1e9 / HBM_PEAK_GBS, "smc_group_device_ms": all_i, "trace_wall_ms": t_iso,
                    "sums_avg_launch_ms": ki["smc_sums"][0] / max(ki["smc_sums"][1], 1),
                    "sums_GBs": 2.0 * 4 * Mtot * stash_row_floats(a.np) * it_iso / (ki["smc_sums"][0] * 1e-3) / 1e9,
                    "note": "one trace group, same seeds, after the timed region: per-kernel durations without overlap (what rocprofv3 shows with option groups=1)"}
            # the same kernel with every CU busy: ONE launch over all traces (outside the timed region)
            ctx.reset_kernel_ms()
            T1, _, _, _ = ctx.trace_batch(s_all)
            ms1, n1 = ctx.kernel_ms("smc")
            ev1 = int((T1 + (T1 < a.ni)).sum()) * (a.np + 1)
            out["roofline_full_occupancy"] = {
                "kernel": kname, "launches": n1, "avg_launch_ms": ms1 / max(n1, 1), "achieved": 8.0 * Mtot * ev1 / (ms1 * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8.0 * Mtot * ev1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "Mevals_per_s": ev1 / ms1 / 1e3,
                "note": "all %d traces started together, traced to their map-free end (no early DENSITY stops): %d SMC iterations; measured after the timed region" % (len(T1), ev1 // (a.np + 1))}
        if a.cpu_baseline != "off" and world == 1 and not shard:
            with heartbeat("cpu baseline (oracle, one host core)"):
                out["cpu_baseline"] = cpu_baseline(img, list(sigs), zdist, a.np, a.ni, st, nvox, a.cpu_baseline)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
