"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).  The reference has no distributed code at all
(SURVEY.md 2.1); the decomposition is new:

  * independent seed traces are dealt round-robin over the ranks (rank r takes sorted seeds
    r, r+G, ...: every GPU gets the same mix of strong and weak seeds); every rank holds a replica
    of the u8 stack, so there is NO data-path collective while tracing;
  * one collective at the end: the (fixed-stride, padded) trace records are all-gathered and
    rank 0 replays the sequential bookkeeping in global seed order -- the replay itself must not
    be sharded (it is order-dependent by definition).  Records are tiny (<= 2*ni*32 B per seed):
    latency-bound over xGMI, not per-link-bandwidth-bound.
"""
import numpy as np
import torch

from .lib import SEED_DT, XEST_DT, NODE_DT, replay as _replay


def shard_indices(n, rank, world):
    return np.arange(rank, n, world)


def _all_gather(dist, t, world):
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return out


def trace_sharded(ctx, seeds, dist, rank, world, trace_fn=None, device=None, params=None, shape=None):
    """Trace `seeds` (sorted, identical on every rank) sharded round-robin; returns
    (nodes, links, T_all) on every rank (rank 0's replay result is authoritative; the others run the
    same deterministic replay on the same gathered records)."""
    params = params if params is not None else ctx.p
    shape = shape if shape is not None else ctx.shape
    ni = params.ni
    trace_fn = trace_fn or (lambda s: ctx.trace_batch(s)[:3])
    n = len(seeds)
    mine = shard_indices(n, rank, world)
    T, stop, xc = trace_fn(seeds[mine])
    m = (n + world - 1) // world  # padded share
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    Tp = torch.zeros(2 * m, dtype=torch.int32)
    Tp[:len(T)] = torch.from_numpy(np.ascontiguousarray(T, np.int32))
    xp = torch.zeros((2 * m, ni, 8), dtype=torch.float32)
    if len(T):
        xp[:len(T)] = torch.from_numpy(np.ascontiguousarray(xc).view(np.float32).reshape(len(T), ni, 8))
    Tg = _all_gather(dist, Tp.to(dev), world)
    xg = _all_gather(dist, xp.to(dev), world)
    T_all = np.zeros(2 * n, np.int32)
    xc_all = np.zeros((2 * n, ni, 8), np.float32)
    for r in range(world):
        idx = shard_indices(n, r, world)
        k = len(idx)
        if k == 0:
            continue
        Tr = Tg[r].cpu().numpy()[:2 * k]
        xr = xg[r].cpu().numpy()[:2 * k]
        T_all[np.repeat(2 * idx, 2) + np.tile([0, 1], k)] = Tr
        xc_all[np.repeat(2 * idx, 2) + np.tile([0, 1], k)] = xr
    nodes, links, _ = _replay(params, shape, seeds, T_all, xc_all.view(XEST_DT).reshape(2 * n, ni))
    return nodes, links, T_all


def slab_bounds(l, rank, world, halo):
    """planes [z0, z1) owned by `rank` and the slab [zlo, zhi) it has to filter to get them exactly"""
    z0, z1 = (l * rank) // world, (l * (rank + 1)) // world
    return z0, z1, max(0, z0 - halo), min(l, z1 + halo)


def frangi_halo(params):
    """planes a cut side needs: radius of the z pass of the widest Gaussian (sigma / zdist) + the radius-2 Hessian stencil"""
    import math
    smax = max(params.sig[i] for i in range(params.nsig))
    return int(math.ceil(3 * (smax / params.zdist))) + 2


def frangi_seeds_sharded(ctx, img_ptr, shape, dist, rank, world, device=None, reduce_fn=None):
    """Frangi + seed extraction of ONE replicated stack split into z-slabs: each rank filters its slab (+ halo) from its own copy
    of the image -- no halo exchange --, the ranks all-reduce Jmin / Jmax (2 floats), quantise J8 with the global extremes and
    extract the seeds of their own layers (MaximumFinder works per layer, seed.cpp:574).  Returns this rank's seeds with global
    z; the caller all-gathers them (concatenated by rank they are in the z-major order of the unsharded extraction).
    `img_ptr` is the device pointer of the whole u8 stack; the context is left pointing at the whole stack again."""
    l, h, w = shape
    halo = frangi_halo(ctx.p)
    z0, z1, zlo, zhi = slab_bounds(l, rank, world, halo)
    seeds = np.zeros(0, SEED_DT)
    jmin, jmax = np.float32(np.inf), np.float32(-np.inf)
    if z1 > z0:
        ctx.set_volume_device(img_ptr + zlo * h * w, (zhi - zlo, h, w))
        jmin, jmax = ctx.frangi_slab(z0 - zlo, z1 - zlo)
    if reduce_fn is not None:  # tests: emulate the all-reduce without a process group
        jmin, jmax = reduce_fn(jmin, jmax)
    elif world > 1:
        dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
        t = torch.tensor([-float(jmin), float(jmax)], dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        jmin, jmax = -float(t[0].item()), float(t[1].item())
    if z1 > z0:
        ctx.quantise_j8(jmin, jmax)
        seeds = ctx.extract_seeds(z0 - zlo, z1 - zlo)
        seeds["z"] += np.float32(zlo)
    ctx.set_volume_device(img_ptr, (l, h, w))
    return seeds, float(jmin), float(jmax)


def gather_seeds(seeds, dist, rank, world, device):
    """variable-length all-gather of the per-slab seed lists (counts first, then padded payloads); concatenated by rank they are
    in the z-major order of the unsharded extraction"""
    cnt = torch.tensor([len(seeds)], dtype=torch.int64, device=device)
    cnts = torch.stack(_all_gather(dist, cnt, world)).cpu().numpy().reshape(-1)
    m = int(cnts.max()) if len(cnts) else 0
    buf = torch.zeros((max(m, 1), SEED_DT.itemsize // 4), dtype=torch.float32)
    if len(seeds):
        buf[:len(seeds)] = torch.from_numpy(np.ascontiguousarray(seeds).view(np.float32).reshape(len(seeds), -1))
    parts = _all_gather(dist, buf.to(device), world)
    return np.concatenate([parts[r].cpu().numpy()[:cnts[r]].copy().view(SEED_DT).reshape(-1) for r in range(world)])


def gather_graphs(nodes, links, dist, rank, world, device):
    """Final node-graph gather for independent stacks: variable-length node / link lists of every
    rank to rank 0 (counts first, then padded payloads).  Returns [(nodes, links)] * world on
    rank 0, None elsewhere."""
    cnt = torch.tensor([len(nodes), len(links)], dtype=torch.int64, device=device)
    cnts = torch.stack(_all_gather(dist, cnt, world)).cpu().numpy()
    mn, ml = int(cnts[:, 0].max()), int(cnts[:, 1].max())
    nb = torch.zeros((mn, NODE_DT.itemsize // 4), dtype=torch.int32)
    nb[:len(nodes)] = torch.from_numpy(np.ascontiguousarray(nodes).view(np.int32).reshape(len(nodes), -1))
    lb = torch.zeros((ml, 2), dtype=torch.int32)
    lb[:len(links)] = torch.from_numpy(np.ascontiguousarray(links, np.int32).reshape(-1, 2))
    nb, lb = nb.to(device), lb.to(device)
    gn = [torch.empty_like(nb) for _ in range(world)] if rank == 0 else None
    gl = [torch.empty_like(lb) for _ in range(world)] if rank == 0 else None
    dist.gather(nb, gn, dst=0)
    dist.gather(lb, gl, dst=0)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        a = gn[r].cpu().numpy()[:cnts[r, 0]].copy().view(NODE_DT).reshape(-1)
        b = gl[r].cpu().numpy()[:cnts[r, 1]].copy()
        out.append((a, b))
    return out
