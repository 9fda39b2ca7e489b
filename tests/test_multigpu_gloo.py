"""CPU tests of the multi-GPU host logic.  The GPU trace engine is replaced by the library's host play-back engine
(pnr_sched_playback) fed with the oracle's map-free traces -- the only place a CPU tracer may stand in: these are tests of the
host logic (round-robin seed sharding, the streaming window with early DENSITY stops, the per-poll all-gather of finished trace
records with carry-over, the replay in global seed order on every rank, the variable-length seed / graph gathers), not a product
path.  What is checked: every rank ends with exactly the node graph of the unsharded map-free trace + replay.

  * world_size 2 over torch.distributed gloo (two processes),
  * 1..4 logical ranks as threads of one process (ThreadExchange), several windows / polls / block sizes.
"""
import os
import sys
import threading
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _workload(n_seeds=9, ni=20, np_=24):
    """small stack, its best seeds and the oracle's map-free traces of both directions of every seed"""
    import orc
    import synth
    import pnr_amd
    from pnr_amd import lib
    L = orc.load_oracle()
    img = synth.synth(48, 40, 24, seed=1)
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, img, [2.0], 2.0)
    so = orc.extract_seeds(L, 5, orc.j8(L, J, jmin, jmax), Vx, Vy, Vz)
    T = orc.Tracker(L, [2.0], 2, np_, ni, 3.0, 0.3, zdist=2.0)
    corr, _ = T.zncc(img, so[:, :6])
    so[:, 7] = corr
    so = so[corr >= 0.3]
    so = so[np.argsort(-so[:, 7], kind="stable")][:n_seeds]
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    traces = {}
    Ts, xcs = [], []
    for sd in seeds:
        for sgn in (1, -1):
            q6 = np.array([sd["x"], sd["y"], sd["z"], sgn * sd["vx"], sgn * sd["vy"], sgn * sd["vz"]], np.float32)
            Tn, st, xc, *_ = T.trace(img, q6)
            traces[q6.tobytes()] = (int(Tn), np.asarray(xc, np.float32).reshape(ni, 8))
            Ts.append(Tn)
            xcs.append(xc)
    p = pnr_amd.make_params(sigmas=[2.0], np_=np_, ni=ni, nodepervol=4, vol=5)
    Tf = np.array(Ts, np.int32)
    xcf = np.stack(xcs).astype(np.float32)
    n1, l1, _ = lib.replay(p, img.shape, seeds, Tf, xcf.view(lib.XEST_DT).reshape(len(Tf), ni))
    total_free = int(np.minimum(Tf + 1, ni).sum())
    return dict(img=img, seeds=seeds, traces=traces, p=p, nodes=n1, links=l1, ni=ni, total_free=total_free)


def _same_graph(nodes, links, W):
    return (len(nodes) == len(W["nodes"]) and np.array_equal(links, W["links"])
            and all(np.array_equal(nodes[k], W["nodes"][k]) for k in nodes.dtype.names))


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pnr_amd import lib, multigpu
    W = _workload()
    seeds, p, img = W["seeds"], W["p"], W["img"]
    lookup = lambda q6: W["traces"][np.asarray(q6, np.float32).tobytes()]
    ex = multigpu.make_exchange(dist, world, torch.device("cpu"))
    # block of 1 KiB: a record of 20 rows is 656 B, so finished traces queue up and are carried over several polls
    nodes, links, nt, iters = lib.sched_playback(p, img.shape, seeds, lookup, rank, world, ex, block_bytes=1024, window=6, poll=2)
    ok = _same_graph(nodes, links, W)
    # the same through the exchange in the shape it has over RCCL: staging buffers and ONE all_gather_into_tensor per exchange
    ex2 = multigpu.make_exchange(dist, world, torch.device("cpu"), staged=True)
    nodes2, links2, _, _ = lib.sched_playback(p, img.shape, seeds, lookup, rank, world, ex2, block_bytes=2048, window=6, poll=2)
    ok = ok and _same_graph(nodes2, links2, W)
    it = torch.tensor([iters], dtype=torch.int64)
    dist.all_reduce(it)
    graphs = multigpu.gather_graphs(nodes, links, dist, rank, world, torch.device("cpu"))
    # ragged seed all-gather (z-slab sharding of the seed extraction): contiguous, unequal slices
    cut = [0, len(seeds) // 3, len(seeds)]
    mine = seeds[cut[rank]:cut[rank + 1]]
    allseeds = multigpu.gather_seeds(mine, dist, rank, world, torch.device("cpu"))
    seeds_ok = len(allseeds) == len(seeds) and all(np.array_equal(allseeds[k], seeds[k], equal_nan=True) for k in seeds.dtype.names)
    z0, z1, zlo, zhi = multigpu.slab_bounds(24, rank, world, multigpu.frangi_halo(p))
    seeds_ok = seeds_ok and (z0, z1) == ((0, 12) if rank == 0 else (12, 24)) and zlo == max(0, z0 - 5) and zhi == min(24, z1 + 5)
    oks = torch.tensor([int(ok and seeds_ok)], dtype=torch.int64)
    dist.all_reduce(oks)  # every rank must hold the same, correct graph
    if rank == 0:
        ok_all = int(oks.item()) == world and len(graphs) == world and all(np.array_equal(g[1], links) and np.array_equal(g[0]["x"], nodes["x"]) for g in graphs)
        q.put((ok_all, len(nodes), int(it.item()), W["total_free"]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_trace_equals_unsharded_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, nn, iters, total_free = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and nn > 10
    assert 10 < iters <= total_free  # early DENSITY stops never add iterations


@pytest.fixture(scope="module")
def workload():
    sys.path.insert(0, HERE)
    return _workload(n_seeds=14)


@pytest.mark.parametrize("tentative", [True, False])
@pytest.mark.parametrize("world,window,poll,block,groups", [(1, 768, 4, 0, 1), (1, 4, 1, 0, 1), (1, 8, 2, 0, 2), (2, 6, 2, 1024, 1), (3, 4, 3, 700, 1),
                                                            (4, 768, 4, 0, 1), (2, 8, 1, 4096, 2), (1, 768, 1, 0, 2), (2, 768, 2, 0, 3)])
def test_playback_logical_ranks(workload, world, window, poll, block, groups, tentative):
    from pnr_amd import lib, multigpu
    W = workload
    lookup = lambda q6: W["traces"][np.asarray(q6, np.float32).tobytes()]
    X = multigpu.ThreadExchange(world)
    out = [None] * world

    def run(r):
        try:
            out[r] = lib.sched_playback(W["p"], W["img"].shape, W["seeds"], lookup, r, world, X.callback(r) if world > 1 else None,
                                        block_bytes=block, window=window, poll=poll, groups=groups, tentative=tentative)
        except Exception as e:  # noqa: BLE001
            out[r] = e
            X.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        nodes, links, nt, iters = out[r]
        assert _same_graph(nodes, links, W), f"rank {r} of {world}"
    assert sum(o[3] for o in out) <= W["total_free"]


def test_failing_rank_aborts_the_others(workload):
    """a rank whose engine fails says so in one last exchange: the other rank returns an error instead of waiting for ever"""
    from pnr_amd import lib, multigpu
    W = workload
    X = multigpu.ThreadExchange(2)
    out = [None, None]
    calls = [0]

    def lookup_ok(q6):
        return W["traces"][np.asarray(q6, np.float32).tobytes()]

    def lookup_bad(q6):
        calls[0] += 1
        if calls[0] > 3:
            raise RuntimeError("injected failure")
        return lookup_ok(q6)

    def run(r):
        try:
            out[r] = lib.sched_playback(W["p"], W["img"].shape, W["seeds"], lookup_bad if r == 1 else lookup_ok, r, 2, X.callback(r), window=4, poll=1)
        except lib.PnrError as e:
            out[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    assert isinstance(out[0], lib.PnrError) and "aborted" in str(out[0])
    assert isinstance(out[1], lib.PnrError)


@pytest.mark.parametrize("groups_bad", [1, 2])
def test_rank_failing_before_its_scheduler_runs_aborts_the_others(workload, groups_bad):
    """a rank whose engine cannot even be set up (on a GPU: no memory for the sample stash) never enters the scheduler -- it still
    owes the others the exchange they are about to enter (one block with the abort word), or they would wait for ever.  The
    surviving rank runs another number of trace groups than the failing one would have."""
    from pnr_amd import lib, multigpu
    W = workload
    X = multigpu.ThreadExchange(2)
    out = [None, None]
    lookup = lambda q6: W["traces"][np.asarray(q6, np.float32).tobytes()]

    def run(r):
        try:
            out[r] = lib.sched_playback(W["p"], W["img"].shape, W["seeds"], lookup, r, 2, X.callback(r), window=0 if r == 1 else 8, poll=1,
                                        groups=groups_bad if r == 1 else 2)
        except lib.PnrError as e:
            out[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    assert isinstance(out[1], lib.PnrError) and "window" in str(out[1])
    assert isinstance(out[0], lib.PnrError) and "aborted" in str(out[0])


def test_ranks_with_different_group_counts_end_together(workload):
    """G is local (a rank whose window is smaller than 4 G slots falls back to one trace group): ranks with different G exchange
    in lock step all the same and end with the same graph"""
    from pnr_amd import lib, multigpu
    W = workload
    X = multigpu.ThreadExchange(2)
    out = [None, None]
    lookup = lambda q6: W["traces"][np.asarray(q6, np.float32).tobytes()]

    def run(r):
        try:
            out[r] = lib.sched_playback(W["p"], W["img"].shape, W["seeds"], lookup, r, 2, X.callback(r), window=4 if r == 1 else 16, poll=2, groups=2)
        except Exception as e:  # noqa: BLE001
            out[r] = e
            X.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    for r in range(2):
        assert not isinstance(out[r], Exception), out[r]
        assert _same_graph(out[r][0], out[r][1], W), f"rank {r}"


def test_shm_exchange_survives_a_stale_segment_under_its_name():
    """a crashed job's segment under the same name, found by a rank that arrives before rank 0: rank 0 marks it dead before it
    removes the name, the early rank opens the name again, and the exchange works (pnr_shm_exchange_open)"""
    import struct
    import time
    from pnr_amd import lib
    name = f"pnr_stale_{os.getpid()}"
    world, cap = 2, 4096
    path = "/dev/shm/" + name
    hdr = struct.pack("<IIQIIIIQ", 0x504e5258, world, cap, 0, 0, 1, 0, 1)  # magic, world, capacity, arrived, phase, attached, failed, stamp
    with open(path, "wb") as f:
        f.write(hdr.ljust(256, b"\0") + b"\xee" * (2 * world * cap))
    out = [None, None]

    def run(r):
        try:
            if r == 0:
                time.sleep(0.5)  # rank 1 has attached to the stale segment by now
            X = lib.ShmExchange(name, r, world, cap)
            out[r] = X.allgather(bytes([65 + r]) * 7)
            X.close()
        except Exception as e:  # noqa: BLE001
            out[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
        assert not t.is_alive()
    assert out[0] == out[1] == [b"A" * 7, b"B" * 7], out
    assert not os.path.exists(path)  # the name is gone once everybody is attached


def _shm_worker(rank, world, name, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from pnr_amd import lib
    W = _workload()
    lookup = lambda q6: W["traces"][np.asarray(q6, np.float32).tobytes()]
    X = lib.ShmExchange(name, rank, world, capacity=4096)
    parts = X.allgather(bytes([rank + 1]) * 5)  # the same segment carries the other small collectives of the C++ host
    nodes, links, nt, iters = lib.sched_playback(W["p"], W["img"].shape, W["seeds"], lookup, rank, world, X, block_bytes=1024, window=6, poll=2, groups=2)
    X.close()
    q.put((rank, _same_graph(nodes, links, W) and parts == [bytes([r + 1]) * 5 for r in range(world)], iters))


def test_sharded_trace_over_the_shared_memory_exchange():
    """three processes joined by the library's own all-gather (pnr_shm_exchange: no Python in the scheduler's loop)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, name = 3, f"pnr_test_{os.getpid()}"
    procs = [ctx.Process(target=_shm_worker, args=(r, world, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and sum(it for _, _, it in res) > 10


def _random_workload(rs, n_seeds, ni, shape, npv, vol):
    """synthetic map-free traces: random walks with ~2-voxel steps in a small volume, so that traces cross, saturate voxels and cut
    each other all the time -- the scheduler's pause / resume / end paths get exercised far more densely than on a real stack"""
    import pnr_amd
    from pnr_amd import lib
    l, h, w = shape
    seeds = np.zeros(n_seeds, lib.SEED_DT)
    seeds["x"] = rs.randint(2, w - 2, n_seeds); seeds["y"] = rs.randint(2, h - 2, n_seeds); seeds["z"] = rs.randint(1, l - 1, n_seeds)
    d = rs.randn(n_seeds, 3).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    seeds["vx"], seeds["vy"], seeds["vz"] = d[:, 0], d[:, 1], d[:, 2]
    seeds["corr"] = np.sort(rs.rand(n_seeds).astype(np.float32))[::-1]
    traces, Ts, xcs = {}, [], []
    for sd in seeds:
        for sgn in (1.0, -1.0):
            q6 = np.array([sd["x"], sd["y"], sd["z"], sgn * sd["vx"], sgn * sd["vy"], sgn * sd["vz"]], np.float32)
            Tn = int(rs.choice([0, 1, 2, ni // 2, ni, ni, ni]))  # ni = "never failed"
            pos = np.array([sd["x"], sd["y"], sd["z"]], np.float32)
            v = q6[3:].copy()
            xc = np.zeros((ni, 8), np.float32)
            for i in range(ni):
                v = v + 0.4 * rs.randn(3).astype(np.float32)
                v /= np.linalg.norm(v)
                pos = np.clip(pos + 2.0 * v, 0, [w - 1, h - 1, l - 1]).astype(np.float32)
                xc[i] = [pos[0], pos[1], pos[2], v[0], v[1], v[2], 2.0, 0.5 + 0.4 * rs.rand()]
            traces[q6.tobytes()] = (Tn, xc)
            Ts.append(Tn)
            xcs.append(xc)
    p = pnr_amd.make_params(sigmas=[2.0], np_=8, ni=ni, nodepervol=npv, vol=vol)
    Tf = np.array(Ts, np.int32)
    xcf = np.stack(xcs).astype(np.float32)
    n1, l1, _ = lib.replay(p, shape, seeds, Tf, xcf.view(lib.XEST_DT).reshape(len(Tf), ni))
    return dict(shape=shape, seeds=seeds, traces=traces, p=p, nodes=n1, links=l1, ni=ni, total_free=int(np.minimum(Tf + 1, ni).sum()))


@pytest.mark.parametrize("case", range(40))
def test_scheduler_random_crossing_traces(case):
    """randomised stress of the streaming scheduler on the host play-back engine: crowded synthetic traces, random window / lookahead /
    polling period / running-trace target / trace groups / world, the tentative replay on and off -- every rank must end with the node graph of the one-shot
    replay of the map-free traces, and with no more iterations than they hold"""
    from pnr_amd import lib, multigpu
    rs = np.random.RandomState(7000 + case)
    ni = int(rs.choice([6, 12, 25]))
    W = _random_workload(rs, n_seeds=int(rs.choice([8, 20, 40])), ni=ni, shape=(int(rs.choice([6, 10])), 16, 20), npv=int(rs.choice([1, 2, 3])), vol=int(rs.choice([1, 5])))
    lookup = lambda q6: W["traces"][np.asarray(q6, np.float32).tobytes()]
    world = int(rs.choice([1, 1, 2, 3]))
    kw = dict(window=int(rs.choice([2, 4, 8, 64, 768])), poll=int(rs.choice([1, 2, 5])), groups=int(rs.choice([1, 2, 3])), look0=int(rs.choice([0, 1, 3, 16])),
              look_pct=int(rs.choice([-1, 0, 50, 400])), block_bytes=int(rs.choice([0, 900, 4096])), target=int(rs.choice([-1, 0, 2, 6, 40])), lag=int(rs.choice([-1, 0, 1, 3])))
    iters = {}
    for tentative in (True, False):
        X = multigpu.ThreadExchange(world)
        out = [None] * world

        def run(r):
            try:
                out[r] = lib.sched_playback(W["p"], W["shape"], W["seeds"], lookup, r, world, X.callback(r) if world > 1 else None, tentative=tentative, **kw)
            except Exception as e:  # noqa: BLE001
                out[r] = e
                X.barrier.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=120)
            assert not t.is_alive(), (case, world, kw, tentative)
        for r in range(world):
            assert not isinstance(out[r], Exception), (case, world, kw, tentative, out[r])
            assert _same_graph(out[r][0], out[r][1], W), (case, world, kw, tentative, r)
        iters[tentative] = sum(o[3] for o in out)
        assert iters[tentative] <= W["total_free"]
