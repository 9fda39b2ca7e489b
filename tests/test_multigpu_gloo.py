"""CPU, world_size 2, gloo: the multi-GPU host logic (round-robin seed sharding, padded
all-gather of trace records, replay on the merged records, variable-length graph gather) gives
exactly the unsharded result.  The GPU trace kernel is replaced here by the oracle tracer (the
only place a CPU tracer may stand in: this is a test of the host logic, not a product path)."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    import synth
    import pnr_amd
    from pnr_amd import lib, multigpu
    L = orc.load_oracle()
    img = synth.synth(48, 40, 24, seed=1)
    ni, np_ = 20, 24
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, img, [2.0], 2.0)
    so = orc.extract_seeds(L, 5, orc.j8(L, J, jmin, jmax), Vx, Vy, Vz)
    T = orc.Tracker(L, [2.0], 2, np_, ni, 3.0, 0.3, zdist=2.0)
    corr, _ = T.zncc(img, so[:, :6])
    so[:, 7] = corr
    so = so[corr >= 0.3]
    so = so[np.argsort(-so[:, 7], kind="stable")][:9]  # odd count: ragged shares
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]

    def trace_fn(s):
        Ts, xcs = [], []
        for sd in s:
            for sgn in (1, -1):
                q6 = np.array([sd["x"], sd["y"], sd["z"], sgn * sd["vx"], sgn * sd["vy"], sgn * sd["vz"]], np.float32)
                Tn, st, xc, *_ = T.trace(img, q6)
                Ts.append(Tn)
                xcs.append(xc)
        return np.array(Ts, np.int32), None, (np.stack(xcs) if xcs else np.zeros((0, ni, 8), np.float32))

    p = pnr_amd.make_params(sigmas=[2.0], np_=np_, ni=ni, nodepervol=4, vol=5)
    nodes, links, T_all = multigpu.trace_sharded(None, seeds, dist, rank, world, trace_fn=trace_fn, device=torch.device("cpu"),
                                                 params=p, shape=img.shape)
    graphs = multigpu.gather_graphs(nodes, links, dist, rank, world, torch.device("cpu"))
    # ragged seed all-gather (z-slab sharding of the seed extraction): rank r contributes seeds[r::world]... as contiguous slices
    cut = [0, len(seeds) // 3, len(seeds)][: world + 1] if world == 2 else None
    mine = seeds[cut[rank]:cut[rank + 1]]
    allseeds = multigpu.gather_seeds(mine, dist, rank, world, torch.device("cpu"))
    seeds_ok = len(allseeds) == len(seeds) and all(np.array_equal(allseeds[k], seeds[k], equal_nan=True) for k in seeds.dtype.names)
    z0, z1, zlo, zhi = multigpu.slab_bounds(24, rank, world, multigpu.frangi_halo(p))
    seeds_ok = seeds_ok and (z0, z1) == ((0, 12) if rank == 0 else (12, 24)) and zlo == max(0, z0 - 5) and zhi == min(24, z1 + 5)
    if rank == 0:
        Tf, _, xcf = trace_fn(seeds)  # unsharded
        n1, l1, _ = lib.replay(p, img.shape, seeds, Tf, xcf.view(lib.XEST_DT).reshape(len(Tf), ni))
        ok = np.array_equal(T_all, Tf) and np.array_equal(links, l1) and all(np.array_equal(nodes[k], n1[k]) for k in nodes.dtype.names)
        ok = ok and seeds_ok and len(graphs) == world and all(np.array_equal(g[1], links) and np.array_equal(g[0]["x"], nodes["x"]) for g in graphs)
        q.put((ok, len(nodes), int(Tf.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_trace_equals_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, nn, tsum = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and nn > 10 and tsum > 10
