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

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the whole particle
evaluation: ph_predict + ph_sample + ph_sums + ph_update, HIP-event times measured live on the kernels'
streams), `roofline_sample` / `roofline_sums` (its two halves alone), `roofline_frangi`,
`full_trace_loop` (all sorted seeds until MAX_TRACE_COUNT, untimed extra) and `cpu_baseline` (oracle C
restatement, 1 core, bounded sample).
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
    ap.add_argument("--seeds", type=int, default=2000, help="sorted seeds traced per stack (configs[3]: 2000); 0 = ALL sorted seeds until MAX_TRACE_COUNT traces (the reference's full trace loop)")
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (a rank started by someone else's torch.distributed.run: before HIP initialises)
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
    # the event pairs around every SMC launch cost 1.2 % of the step (~9000 launches per stack): the streaming tracer times every 4th poll
    # of a trace group and counts it fourfold (the per-launch averages are unbiased, the totals are estimates; PNR_BENCH_OPTS=profile_every=1: all)
    ctx.set_option("profile_every", 4)
    opts = ctx.set_options(os.environ.get("PNR_BENCH_OPTS"))
    pe = max(1, ctx.get_option("profile_every"))
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
            # (a rehearsal over gloo that asks for the RCCL exchange gets its shape -- pinned staging, one all_gather_into_tensor -- on host tensors)
            exchange = multigpu.make_exchange(dist, world, coll_dev, staged=(backend != "nccl" and want == "rccl"))
            exchange_kind = "RCCL all_gather_into_tensor" if backend == "nccl" else (backend + (" all_gather_into_tensor, pinned staging (the RCCL exchange's shape)" if want == "rccl" else ""))

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
            s = ctx.sort_seeds(s0)
            s = s[:a.seeds] if a.seeds > 0 else s
            t3 = time.perf_counter()
            nodes, links, ntr, iters = ctx.trace_replay_sharded(s, rank, world, exchange)
        else:
            ctx.frangi()
            t1 = time.perf_counter()
            s0 = ctx.extract_seeds()
            n_init = len(s0)
            t2 = time.perf_counter()
            s = ctx.score_filter_sort(s0)
            s = s[:a.seeds] if a.seeds > 0 else s
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
        km = {g: ctx.kernel_ms(g) for g in ("gauss", "hessian_tile", "hessian_eigen", "j8", "seed_maxima", "zncc", "smc", "smc_sums", "smc_predict", "smc_update", "smc_cube")}
        # The dominant kernels are the particle evaluation of the SMC step (tracker.cpp:1891-1964 is ONE evaluation: the gather and
        # the ordered ZNCC sums): ph_predict + ph_sample + ph_sums + ph_update with the phased driver (one launch of each per SMC
        # step over all active traces), smc_trace with the persistent one.  Algorithmic bytes: 8 corner bytes x
        # sum(M_sigma) samples per particle evaluation, (np + 1) evaluations per SMC iteration (rank 0's when the seeds are sharded).
        # `roofline` prices the WHOLE evaluation: bytes of a step / duration of a step, where the duration is the summed device
        # time of the step's four launches with one trace group, and the wall time of the tracing stage per step with several
        # (their launches overlap on separate streams, so summed durations would count shared time twice).
        Mtot = sum(len(ctx.table(f"model_wgt{s}")) for s in range(len(sigs)))
        groups_opt = ctx.get_option("groups")
        groups = groups_opt if groups_opt > 0 else (1 if world > 1 and a.mode == "shard" else 2)  # 0 = automatic (pnr_hip.h)
        # (ph_cube: the traces' cubes fetched from the image once per step; absent -- zero launches -- with option cube_copy = 0)
        EV = ("smc_predict", "smc_cube", "smc", "smc_sums", "smc_update") if (a.driver == "phased" and km["smc_cube"][1] > 0) else ("smc_predict", "smc", "smc_sums", "smc_update")
        KNAME = {"smc_predict": "ph_predict", "smc_cube": "ph_cube", "smc": "ph_sample<54, false, true>", "smc_sums": "ph_sums", "smc_update": "ph_update"} if a.driver == "phased" else {"smc": "smc_trace"}
        smc_n = km["smc"][1]
        smc_all_ms = sum(km[g][0] for g in EV)
        evals = st["iters"] * (a.np + 1)
        bytes_total = 8.0 * Mtot * evals * a.steps            # the timed region
        steps_smc = max(smc_n, 1)                              # SMC steps (= launches of each kernel) in the timed region
        tr_ms = st.get("trace_replay_gather_ms", st.get("trace_replay_exchange_ms"))
        overlapped = groups > 1 and a.driver == "phased"
        step_ms = (tr_ms * a.steps / steps_smc) if overlapped else (smc_all_ms / steps_smc)
        achieved = bytes_total / steps_smc / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
        fr_ms = (km["gauss"][0] + km["hessian_tile"][0] + km["hessian_eigen"][0] + km["j8"][0]) / a.steps
        fr_vox = nvox if not shard else None  # a rank's slab + halo when sharded: no per-stack Frangi figure then
        # HBM traffic per SMC step: PMC counters (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, calibrated on a known
        # byte count in the same access pattern: scripts/prof_traffic.sh), collected for THIS workload with one trace group and
        # committed under profiles/ -- read from that file, not measured in this run; null for any other workload
        traffic, traffic_src, traffic_k, traffic_note = None, None, {}, None
        src_hash = pnr_amd.lib.kernel_source_hash()
        if S == 1024 and a.seeds == 2000 and a.np == 200 and a.ni == 200 and not a.one_shot and world == 1:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_1024_s2000.json")), reverse=True)
            stale = []
            for tpath in cands:
                tj = json.load(open(tpath))
                wl = tj.get("workload", {})
                if wl.get("driver", "phased") != a.driver:
                    continue
                if wl.get("kernel_source_hash") != src_hash:  # taken from other kernels than the ones that run here: never quoted
                    stale.append(os.path.basename(tpath))
                    continue
                # options that select another kernel form or split than this run's (anything but the number of trace groups, which the
                # profile fixes at one so that no two kernels overlap under the counters): not this run's bytes either
                po = dict((wl.get("config") or {}).get("options") or {})
                ro = dict(opts or {})
                for o_ in (po, ro):
                    o_.pop("groups", None)
                    o_.pop("profile_every", None)
                if po != ro:
                    stale.append(os.path.basename(tpath) + " (options %s)" % json.dumps(po, sort_keys=True))
                    continue
                # the profile was taken with one trace group (no two kernels overlap while the counters run): its launches hold more
                # traces than this run's when the groups differ -- scale by the SMC iterations per launch (bytes per trace-iteration are
                # what the kernels' traffic is made of)
                it_prof, st_prof = wl.get("smc_iterations"), wl.get("smc_steps")
                scale = ((st["iters"] * a.steps / steps_smc) / (it_prof / st_prof)) if (it_prof and st_prof) else 1.0
                for key in (("ph_predict", "ph_cube", "ph_sample", "ph_sums", "ph_update") if a.driver == "phased" else ("smc_trace",)):
                    e = tj.get(key, {})
                    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                        traffic_k[key] = (e["FETCH_SIZE"]["bytes_per_launch"] + e["WRITE_SIZE"]["bytes_per_launch"]) * scale
                if ("ph_sample" in traffic_k and "ph_sums" in traffic_k) or "smc_trace" in traffic_k:
                    traffic = sum(traffic_k.values())
                    traffic_src = "profiles/" + os.path.basename(tpath)
                    break
            if traffic is None:
                traffic_note = ("no committed PMC profile was taken from these kernel sources (hash %s; found: %s): run scripts/prof_traffic.sh and commit its JSON"
                                % (src_hash, ", ".join(stale) or "none"))
        if world == 1:
            par = "1 GPU"
        elif shard:
            par = (f"one stack on {world} GPUs: Frangi + seeds + seed scores in z-slabs, sorted seeds round-robin; RCCL all-reduce(min,max), all-gather of "
                   "seeds; finished trace records all-gathered once per rotation of the trace groups (host data: shared memory on one node, RCCL otherwise); "
                   "every rank replays in global seed order")
        else:
            par = f"{world} independent stacks, one per GPU; RCCL gather of node graphs"

        def kernel_block(g, bytes_tot, note):  # one kernel of the evaluation against the bytes named in `note`
            ms, n = km[g]
            if ms <= 0 or n <= 0:
                return None
            gb = bytes_tot / (ms * 1e-3) / 1e9
            return {"kernel": KNAME[g], "launches": n, "avg_launch_ms": ms / n, "bytes_per_launch": bytes_tot / n, "achieved": gb, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": gb / HBM_PEAK_GBS, "share_of_evaluation_device_time": ms / smc_all_ms if smc_all_ms > 0 else None, "note": note}

        stash_bytes = 2.0 * 4 * Mtot * stash_row_floats(a.np) * st["iters"] * a.steps
        dominant = max((g for g in EV if g in KNAME), key=lambda g: km[g][0])  # by measured device time, not by hand
        out = {
            "metric": "Mvox/s traced (Frangi+SMC step) on 1024^3 synthetic stack; % HBM roofline" if S == 1024 else f"Mvox/s traced (Frangi+SMC step) on {S}^3 synthetic stack; % HBM roofline",
            "value": value, "unit": "Mvox/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak" if (world > 1 and a.mode == "stacks") else "strong", "vs_baseline": None,
            "dtype": "f32 (+f64 3x3 eigen-solver)", "data": "synthetic",
            "config": {"workload": f"{S}^3 synthetic u8 stack (tests/synth.py seed {stack_seed}), scales={{2,4,6}}, zdist=2, np={a.np}, ni={a.ni}, "
                                   + (f"first {a.seeds} sorted seeds" if a.seeds > 0 else "ALL sorted seeds until MAX_TRACE_COUNT = 5000 traces (the reference's full trace loop)")
                                   + " traced in both directions per stack, tolerance=5, znccth=0.3, step=2, kappa=3",
                       "parallelism": par, "backend": backend if world > 1 else None, "record_exchange": exchange_kind, "options": opts or None,
                       "trace_groups": groups},
            "roofline": {
                "kernel": "+".join(KNAME[g] for g in EV if g in KNAME), "dominant_by_device_time": KNAME[dominant],
                "bound": "hbm",
                "bound_measured": ("hbm: with two trace groups overlapping, the evaluation as built moves its REAL bytes (traffic: the f32 sample stash written once and read twice, "
                                   "~1.5 x the algorithmic bytes) at traffic_rate.frac_of_mixed_rw_ceiling_5500 of what HBM delivers to a concurrent writer and reader (scripts/probes/mall_wr); "
                                   "taken alone ph_sample is bound by LDS gather + VALU issue (roofline_full_occupancy) and a small ph_sums launch by one wave's serial f32 / f64 chain"),
                "bound_detail": "priced against the 8 TB/s HBM peak on ALGORITHMIC bytes as SURVEY 8(d) prescribes; `traffic` / `traffic_rate` give the real bytes and their rate",
                "bytes_per_launch": bytes_total / steps_smc, "avg_launch_ms": step_ms, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_time_base": "tracing wall time" if overlapped else "summed device time",
                "traffic": traffic, "traffic_source": traffic_src, "traffic_per_kernel": traffic_k or None, "traffic_note": traffic_note, "kernel_source_hash": src_hash,
                # the REAL bytes over the same time base: how close the evaluation as built runs to what HBM delivers under mixed
                # reads and writes (~5.5 TB/s measured, scripts/probes; the 8 TB/s peak is what `frac` is priced against)
                "traffic_rate": ({"achieved": traffic / (step_ms * 1e-3) / 1e9, "unit": "GB/s", "frac_of_peak": traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "frac_of_mixed_rw_ceiling_5500": traffic / (step_ms * 1e-3) / 1e9 / 5500.0} if traffic and step_ms > 0 else None),
                "launch": "one SMC step = one launch of each kernel over all active traces of a trace group",
                "time_base": (f"wall time of the tracing stage / SMC steps ({groups} trace groups on separate streams: launches overlap, host replay and polls included)"
                              if overlapped else "summed device time of the four launches of a step (HIP events on the launching stream)"),
                "device_ms_per_stack": smc_all_ms / a.steps, "Mevals_per_s": bytes_total / (8.0 * Mtot) / (step_ms * steps_smc) / 1e3 if step_ms > 0 else None,
                # speculation: the streaming scheduler runs more SMC iterations than the sequential reference needs for the same graph
                # (one per node + the stopping iteration of every trace); `frac` prices every iteration that was run
                "iterations_run": st["iters"], "iterations_needed_sequentially": st["nodes"] + 2 * st["traces_used"],
                "frac_of_needed_work": achieved / HBM_PEAK_GBS * min(1.0, (st["nodes"] + 2 * st["traces_used"]) / max(st["iters"], 1)),
                "note": "ALGORITHMIC bytes of the whole particle evaluation (znccBBB, tracker.cpp:1891-1964: gather AND ordered sums): 8*sum(M_sigma)=%d B per "
                        "evaluation x (np+1) evaluations x the trace-iterations of a step, over the duration of the step; `traffic` = real HBM bytes per step (PMC, "
                        "sum over the kernels) read from the committed profile named in traffic_source (scaled to this run's trace-iterations per launch), not measured in this run" % (8 * Mtot)},
            "roofline_sample": kernel_block("smc", bytes_total, "the gather half alone: the evaluation's algorithmic gather bytes over the sampling kernel's own launch time (served from the LDS cube: VALU / LDS bound, not HBM)"),
            "roofline_sums": None if a.driver != "phased" else kernel_block("smc_sums", stash_bytes, "ordered sums alone against their REAL stash bytes: every stashed f32 sample is streamed twice (mean, then corr), 2 x 4 x sum(M) x %d B per SMC iteration -- an upper bound, exact duplicate poses are evaluated once" % stash_row_floats(a.np)),
            "roofline_frangi": None if not fr_vox or fr_ms <= 0 else {
                "kernels": "gauss_xy_u8_m(x,y)+gauss_axis_t(z)+hessian_tile+eigen_queue+j8", "bound": "hbm", "achieved": (len(sigs) + 12) * fr_vox / (fr_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (len(sigs) + 12) * fr_vox / (fr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "device_ms_per_step": fr_ms,
                "note": "(S+12) B/voxel compulsory bytes over the Frangi kernel group; VALU-bound: three Gaussian passes with separate multiply and add (the reference's rounding) and the Hessian stencil with its zero-response tests; the fp64 JAMA eigen-solver runs only where the response can reach J8 > 0"},
            "stages_ms": {k: v for k, v in st.items() if k.endswith("_ms")},
            "kernel_ms_per_step": {g: km[g][0] / a.steps for g in km},
            "kernel_timers": ("HIP events on the launching stream; the streaming tracer times every %d-th poll of a trace group (always its first) and counts it "
                              "%d-fold: per-launch averages unbiased, totals and launch counts of the SMC kernels are estimates (PNR_BENCH_OPTS=profile_every=1 times "
                              "every launch); the untimed one-group pass (roofline_isolated) times every launch" % (pe, pe)) if pe > 1 else "HIP events around every launch",
            "counts": {k: v for k, v in st.items() if not k.endswith("_ms")},
            "smc_launches_per_step": smc_n / a.steps,
            "Mvox_per_s_frangi": (nvox / (fr_ms * 1e-3) / 1e6) if (fr_vox and fr_ms > 0) else None,
        }
        if world == 1 and not shard and not a.one_shot and not a.no_extra:
            s_sorted = ctx.score_filter_sort(ctx.extract_seeds())
            s_all = s_sorted[:a.seeds] if a.seeds > 0 else s_sorted
            if overlapped:
                # the same step with ONE trace group (launches never overlap): what each kernel needs alone, and the evaluation
                # over summed device time -- what rocprofv3 --kernel-trace --stats shows with option groups=1
                ctx.set_option("groups", 1)
                ctx.set_option("profile_every", 1)  # outside the timed region: every launch is timed
                ctx.reset_kernel_ms()
                t0i = time.perf_counter()
                _, _, _, it_iso = ctx.trace_replay(s_all)
                t_iso = 1e3 * (time.perf_counter() - t0i)
                ctx.set_option("groups", groups_opt)
                ctx.set_option("profile_every", pe)
                ki = {g: ctx.kernel_ms(g) for g in EV}
                ev_i = it_iso * (a.np + 1)
                all_i = sum(v[0] for v in ki.values())
                n_i = max(ki["smc"][1], 1)
                out["roofline_isolated"] = {
                    "kernel": out["roofline"]["kernel"], "bound": "hbm", "launches": ki["smc"][1], "avg_launch_ms": all_i / n_i, "bytes_per_launch": 8.0 * Mtot * ev_i / n_i,
                    "achieved": 8.0 * Mtot * ev_i / (all_i * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8.0 * Mtot * ev_i / (all_i * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "per_kernel_avg_launch_ms": {KNAME[g]: ki[g][0] / max(ki[g][1], 1) for g in EV}, "trace_wall_ms": t_iso,
                    "sample_frac": 8.0 * Mtot * ev_i / (ki["smc"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "sums_GBs": 2.0 * 4 * Mtot * stash_row_floats(a.np) * it_iso / (ki["smc_sums"][0] * 1e-3) / 1e9,
                    "note": "one trace group, same seeds, after the timed region: the whole evaluation over the summed device time of its four kernels (no overlap)"}
                out["roofline"]["frac_summed_device_time_one_group"] = out["roofline_isolated"]["frac"]
            if a.seeds > 0:
                # the same sampling kernel with every CU busy: ONE launch over all traces (outside the timed region)
                ctx.reset_kernel_ms()
                T1, _, _, _ = ctx.trace_batch(s_all)
                ms1, n1 = ctx.kernel_ms("smc")
                ev1 = int((T1 + (T1 < a.ni)).sum()) * (a.np + 1)
                out["roofline_full_occupancy"] = {
                    "kernel": KNAME["smc"], "launches": n1, "avg_launch_ms": ms1 / max(n1, 1), "achieved": 8.0 * Mtot * ev1 / (ms1 * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 8.0 * Mtot * ev1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "Mevals_per_s": ev1 / ms1 / 1e3,
                    "note": "sampling kernel alone, all %d traces started together, traced to their map-free end (no early DENSITY stops): %d SMC iterations; measured after the timed region" % (len(T1), ev1 // (a.np + 1))}
                # the reference's own trace loop (Advantra_plugin.cpp:2658-2710): EVERY sorted seed until MAX_TRACE_COUNT = 5000
                # traces (:72, :2702) -- what a user of advantra_func sees; once, after the timed region
                t0f = time.perf_counter()
                nodes_f, _, ntr_f, it_f = ctx.trace_replay(s_sorted)
                t_full = 1e3 * (time.perf_counter() - t0f)
                front_ms = sum(v for k, v in st.items() if k in ("frangi_ms", "seeds_ms", "score_ms"))
                out["full_trace_loop"] = {
                    "seeds_sorted": len(s_sorted), "traces_used": int(ntr_f), "nodes": len(nodes_f) - 1, "smc_iterations": int(it_f), "trace_replay_ms": t_full,
                    "step_ms": front_ms + t_full, "Mvox_per_s": nvox / ((front_ms + t_full) * 1e-3) / 1e6,
                    "note": "all sorted seeds until MAX_TRACE_COUNT = 5000 traces are used (Advantra_plugin.cpp:72, 2702): the trace loop as the reference runs it; "
                            "Frangi + seeds + scores of the timed region + this tracing; one run after the timed region (bench.py --seeds 0 times exactly this)"}
        if a.cpu_baseline != "off" and world == 1 and not shard:
            with heartbeat("cpu baseline (oracle, one host core)"):
                out["cpu_baseline"] = cpu_baseline(img, list(sigs), zdist, a.np, a.ni, st, nvox, a.cpu_baseline)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
