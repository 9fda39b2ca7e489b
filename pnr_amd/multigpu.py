"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests).  The reference has no distributed code at all
(SURVEY.md 2.1); the decomposition of ONE stack (BASELINE configs[3]) is new:

  * Frangi + seed extraction in z-slabs cut from every rank's replica of the u8 stack (no halo
    exchange), one 2-float all-reduce for Jmin / Jmax, one variable-length all-gather of the seeds;
  * tracing: the sorted seeds are dealt round-robin (rank r takes r, r+G, ...: every GPU gets the
    same mix of strong and weak seeds) and every rank streams its share through its own window of
    trace slots (pnr_trace_replay_sharded).  After every poll the ranks all-gather the records of
    the traces that finished (one fixed-size block per rank: latency-bound over xGMI, never
    per-link-bandwidth-bound) and every rank replays them in global seed order -- the replay must
    not be sharded, it is order-dependent by definition -- so every GPU's density map holds the
    replayed nodes of all ranks: early DENSITY stops and the seed skip rule work as on one GPU, and
    every rank ends with the same node graph (no final gather is needed).

The all-gather is handed to the C ABI as a callback (pnr_allgather_fn): make_exchange() wraps
torch.distributed, ThreadExchange joins several contexts of one process (tests).
"""
import ctypes as C
import threading

import numpy as np
import torch

from .lib import SEED_DT, NODE_DT, ALLGATHER_FN


def shard_indices(n, rank, world):
    return np.arange(rank, n, world)


def _all_gather(dist, t, world):
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return out


def make_exchange(dist, world, device=None, staged=False):
    """pnr_allgather_fn over torch.distributed: `bytes` bytes per rank -> world x bytes in rank order.  With a CUDA `device` (backend
    nccl = RCCL) the block goes through pinned staging buffers and the GPU into ONE all_gather_into_tensor; otherwise (gloo) it stays
    in host memory.  `staged` gives the host form the shape of the RCCL one -- pinned staging buffers and a single
    all_gather_into_tensor -- so that a rehearsal of N ranks on fewer GPUs (gloo cannot gather device tensors) walks the same code.
    Keep the returned object alive while the C call runs."""
    cuda = device is not None and torch.device(device).type == "cuda"
    st = {"nb": -1}

    def fn(user, send, recv, nbytes):
        try:
            nb = int(nbytes)
            if st["nb"] != nb:
                st["nb"] = nb
                st["inp"] = torch.empty(nb, dtype=torch.uint8)
                st["out"] = torch.empty(world * nb, dtype=torch.uint8)
                if cuda or (staged and torch.cuda.is_available()):
                    st["inp"], st["out"] = st["inp"].pin_memory(), st["out"].pin_memory()
                if cuda:
                    st["ginp"] = torch.empty(nb, dtype=torch.uint8, device=device)
                    st["gout"] = torch.empty(world * nb, dtype=torch.uint8, device=device)
            C.memmove(st["inp"].data_ptr(), send, nb)
            if cuda:
                st["ginp"].copy_(st["inp"], non_blocking=True)
                dist.all_gather_into_tensor(st["gout"], st["ginp"])
                st["out"].copy_(st["gout"])  # blocking: the block is on the host when this returns
            elif staged:
                dist.all_gather_into_tensor(st["out"], st["inp"])
            else:
                dist.all_gather(list(st["out"].view(world, nb).unbind(0)), st["inp"])
            C.memmove(recv, st["out"].data_ptr(), world * nb)
            return 0
        except Exception:  # noqa: BLE001 -- must not propagate through the C frames
            import traceback
            traceback.print_exc()
            return 1

    return ALLGATHER_FN(fn)


class ThreadExchange:
    """all-gather between `world` threads of ONE process, each driving its own context (tests: logical ranks on one GPU)"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.blocks = [b""] * world

    def callback(self, rank):
        def fn(user, send, recv, nbytes):
            try:
                self.blocks[rank] = C.string_at(send, int(nbytes))
                self.barrier.wait(timeout=300)
                data = b"".join(self.blocks)
                C.memmove(recv, data, len(data))
                self.barrier.wait(timeout=300)  # nobody overwrites its block before everyone has read it
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                self.barrier.abort()
                return 1
        return ALLGATHER_FN(fn)


def trace_sharded(ctx, seeds, dist, rank, world, device=None):
    """This rank's share of tracing the sorted `seeds` (identical on every rank): (nodes, links, iterations run here).  Every rank
    returns the same graph -- the graph ctx.trace_replay(seeds) gives on one GPU."""
    if ctx.p.somaradius > 0 and not ctx.have_soma():
        raise RuntimeError("somaradius > 0: run ctx.soma() on every rank before tracing")
    if world == 1:
        nodes, links, _, iters = ctx.trace_replay(seeds)
        return nodes, links, iters
    ex = make_exchange(dist, world, device)
    nodes, links, _, iters = ctx.trace_replay_sharded(seeds, rank, world, ex)
    return nodes, links, iters


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
    keep = ctx._keep  # the caller's keep-alive reference of the whole stack survives the slab views
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
    ctx.set_volume_device(img_ptr, (l, h, w), keepalive=keep)
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
