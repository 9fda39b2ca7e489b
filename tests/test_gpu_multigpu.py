"""One stack on several GPUs (BASELINE configs[3]) through the real HIP contexts, on the one GPU of the test box:

  * pnr_trace_replay_sharded with 2 / 3 logical ranks -- one context and one host thread per rank, joined by an in-process
    all-gather (ThreadExchange) -- must end, on EVERY rank, with exactly the node graph of pnr_trace_replay on one GPU, and with
    no more SMC iterations than the map-free tracing needs;
  * the sharded front half (Frangi + seeds per z-slab, scores per slab, merged sort) must give the sorted seed list of one GPU;
  * `bench.py --gpus 2` must start two ranks (here over gloo, sharing the GPU) and report the graph of `--gpus 1`.
The multi-process collectives themselves are covered on CPU (tests/test_multigpu_gloo.py)."""
import json
import os
import subprocess
import sys
import threading
import numpy as np
import pytest
import synth
import pnr_amd
from pnr_amd import lib, multigpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph_equal(a, b):
    return len(a[0]) == len(b[0]) and np.array_equal(a[1], b[1]) and all(np.array_equal(a[0][k], b[0][k], equal_nan=True) for k in a[0].dtype.names)


@pytest.mark.parametrize("world,opts", [(2, {}), (3, dict(window=16, poll=3, exchange_block=2048)), (2, dict(groups=2, window=32)),
                                        (8, {}), (8, dict(window=8, groups=2))])  # 8 = the ranks of one MI355X node
def test_sharded_trace_logical_ranks(world, opts):
    img = synth.synth(96, 80, 40, seed=11)
    p = pnr_amd.make_params(sigmas=[2.0, 3.0], np_=48, ni=40, zdist=2.0, nodepervol=3, vol=5)
    c0 = pnr_amd.Context(p, 0)
    c0.set_volume(img)
    c0.frangi()
    seeds = c0.score_filter_sort(c0.extract_seeds())[:60]
    assert len(seeds) > 30
    T, stop, xc, _ = c0.trace_batch(seeds)
    free_iters = int((T + (T < p.ni)).sum())
    n1, l1, nt1, it1 = c0.trace_replay(seeds)
    X = multigpu.ThreadExchange(world)
    ctxs, out = [], [None] * world
    for r in range(world):
        c = pnr_amd.Context(p, 0)
        c.set_volume(img)
        for k, v in opts.items():
            c.set_option(k, v)
        ctxs.append(c)

    def run(r):
        try:
            out[r] = ctxs[r].trace_replay_sharded(seeds, r, world, X.callback(r))
        except Exception as e:  # noqa: BLE001
            out[r] = e
            X.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
        assert not t.is_alive()
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        assert out[r][2] == nt1 and _graph_equal((out[r][0], out[r][1]), (n1, l1)), f"rank {r} of {world}"
    assert sum(o[3] for o in out) <= free_iters
    for c in ctxs:
        c.close()


def test_sharded_front_half_gives_the_one_gpu_seed_list():
    """Frangi + seeds per z-slab, scores per slab (pnr_score_filter_seeds), merged stable sort (pnr_sort_seeds) == pnr_frangi +
    pnr_extract_seeds + pnr_score_filter_sort_seeds; three logical ranks in sequence on one context (the all-reduce of Jmin / Jmax
    emulated), ragged slabs"""
    import torch
    img = synth.synth_torch(88, 72, 50, seed=5)
    p = pnr_amd.make_params(sigmas=[2.0, 4.0], np_=32, ni=10, zdist=2.0)
    c = pnr_amd.Context(p, 0)
    shape = tuple(img.shape)
    c.set_volume_device(img.data_ptr(), shape, keepalive=img)
    c.frangi()
    want = c.score_filter_sort(c.extract_seeds())
    world = 3
    ext = []
    for r in range(world):  # pass 1: every rank's Jmin / Jmax over its own planes
        z0, z1, zlo, zhi = multigpu.slab_bounds(shape[0], r, world, multigpu.frangi_halo(p))
        c.set_volume_device(img.data_ptr() + zlo * shape[1] * shape[2], (zhi - zlo, shape[1], shape[2]), keepalive=img)
        ext.append(c.frangi_slab(z0 - zlo, z1 - zlo))
    gmin, gmax = min(e[0] for e in ext), max(e[1] for e in ext)
    parts = []
    for r in range(world):
        mine, _, _ = multigpu.frangi_seeds_sharded(c, img.data_ptr(), shape, None, r, world, reduce_fn=lambda a, b: (gmin, gmax))
        assert c._keep is img  # the caller's keep-alive reference survives the slab views
        parts.append(c.score_filter(mine))
    got = c.sort_seeds(np.concatenate(parts))
    assert len(got) == len(want) > 20
    for k in want.dtype.names:
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    torch.cuda.synchronize()


def test_front_half_in_8_slabs_at_bench_size():
    """BASELINE configs[3] as the driver's 8-GPU run cuts it: the 1024^3 bench stack in 8 z-slabs of 128 planes (+ 11 halo planes a
    side), scales {2,4,6} -- the eight ranks in sequence on one context.  Extremes, seed list and scores equal the unsharded ones."""
    import torch
    S = 1024
    img = synth.synth_torch(S, S, S, seed=3)
    p = pnr_amd.make_params(sigmas=[2.0, 4.0, 6.0], np_=200, ni=200, zdist=2.0)
    c = pnr_amd.Context(p, 0)
    shape = (S, S, S)
    c.set_volume_device(img.data_ptr(), shape, keepalive=img)
    jmin1, jmax1 = c.frangi()
    want = c.score_filter_sort(c.extract_seeds())
    world = 8
    assert multigpu.frangi_halo(p) == 11
    ext = []
    for r in range(world):
        z0, z1, zlo, zhi = multigpu.slab_bounds(S, r, world, 11)
        c.set_volume_device(img.data_ptr() + zlo * S * S, (zhi - zlo, S, S), keepalive=img)
        ext.append(c.frangi_slab(z0 - zlo, z1 - zlo))
    gmin, gmax = min(e[0] for e in ext), max(e[1] for e in ext)
    assert (gmin, gmax) == (jmin1, jmax1)
    parts = []
    for r in range(world):
        mine, _, _ = multigpu.frangi_seeds_sharded(c, img.data_ptr(), shape, None, r, world, reduce_fn=lambda a, b: (gmin, gmax))
        parts.append(c.score_filter(mine))
    got = c.sort_seeds(np.concatenate(parts))
    assert len(got) == len(want) > 10000
    for k in want.dtype.names:
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    torch.cuda.synchronize()
    c.close()


def test_sharded_trace_8_ranks_at_bench_size():
    """BASELINE configs[3] at its real size: the 1024^3 bench stack, the first 2000 sorted seeds dealt to the 8 ranks of one node.
    Rank 0 is the real thing -- its share of the seeds through pnr_trace_replay_sharded on the GPU -- and ranks 1..7 are host
    threads that play the recorded map-free traces of their seeds through the same scheduler (pnr_sched_playback), all joined by
    the library's shared-memory all-gather (what scripts/emulate_ranks.py measures).  Every rank must end with the one-GPU node
    graph, and all ranks together with no more SMC iterations than tracing every seed to its map-free end
    (the sequential semantics: Advantra_plugin.cpp:2658-2710, tracker.cpp:855,870-882)."""
    import torch
    S, world = 1024, 8
    img = synth.synth_torch(S, S, S, seed=3)
    p = pnr_amd.make_params(sigmas=[2.0, 4.0, 6.0], np_=200, ni=200, zdist=2.0)
    c = pnr_amd.Context(p, 0)
    shape = (S, S, S)
    c.set_volume_device(img.data_ptr(), shape, keepalive=img)
    c.frangi()
    seeds = c.score_filter_sort(c.extract_seeds())[:2000]
    assert len(seeds) == 2000
    n1, l1, nt1, it1 = c.trace_replay(seeds)
    assert len(n1) > 50000
    T, stop, xc, _ = c.trace_batch(seeds)  # every trace to its map-free end: what the played-back ranks hold
    free_iters = int((T + (T < p.ni)).sum())
    assert it1 <= free_iters
    xcf = np.ascontiguousarray(xc).view(np.float32).reshape(2 * len(seeds), p.ni, 8)
    traces = {}
    for i, sd in enumerate(seeds):
        for d, sgn in enumerate((1.0, -1.0)):
            q6 = np.array([sd["x"], sd["y"], sd["z"], sgn * sd["vx"], sgn * sd["vy"], sgn * sd["vz"]], np.float32)
            traces[q6.tobytes()] = (int(T[2 * i + d]), xcf[2 * i + d])
    lookup = lambda q6: traces[np.asarray(q6, np.float32).tobytes()]
    name = f"pnr_t8_{os.getpid()}"
    out, X = [None] * world, [None] * world

    def run(r):
        try:
            X[r] = lib.ShmExchange(name, r, world, 1 << 20)
            out[r] = c.trace_replay_sharded(seeds, 0, world, X[0]) if r == 0 else lib.sched_playback(p, shape, seeds, lookup, r, world, X[r], groups=2)
        except Exception as e:  # noqa: BLE001
            out[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
        assert not t.is_alive()
    for x in X:
        if x is not None:
            x.close()
    for r in range(world):
        assert not isinstance(out[r], Exception) and out[r] is not None, (r, out[r])
        assert out[r][2] == nt1 and _graph_equal((out[r][0], out[r][1]), (n1, l1)), f"rank {r} of {world}"
    assert sum(o[3] for o in out) <= free_iters
    torch.cuda.synchronize()
    c.close()


def test_rank_without_memory_for_its_stash_aborts_the_others():
    """a rank whose engine cannot be set up (stash budget of 1 MB: not one trace slot fits) fails before its scheduler runs; it still
    sends the abort block (include/pnr_hip.h), so the other rank returns an error instead of waiting in its first exchange"""
    img = synth.synth(96, 80, 40, seed=11)
    p = pnr_amd.make_params(sigmas=[2.0, 3.0], np_=48, ni=40, zdist=2.0)
    ctxs = []
    for r in range(2):
        c = pnr_amd.Context(p, 0)
        c.set_smc_driver("phased")
        c.set_volume(img)
        ctxs.append(c)
    ctxs[0].frangi()
    seeds = ctxs[0].score_filter_sort(ctxs[0].extract_seeds())[:40]
    ctxs[1].set_option("stash_mb", 1)
    X = multigpu.ThreadExchange(2)
    out = [None, None]

    def run(r):
        try:
            out[r] = ctxs[r].trace_replay_sharded(seeds, r, 2, X.callback(r))
        except lib.PnrError as e:
            out[r] = e

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    assert isinstance(out[1], lib.PnrError) and "device memory" in str(out[1]), out[1]
    assert isinstance(out[0], lib.PnrError) and "aborted" in str(out[0]), out[0]
    for c in ctxs:
        c.close()


def _bench(args, env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_and_matches_one_gpu():
    """`python bench.py --gpus 2` (no torchrun around it) becomes the launcher of two ranks; rehearsed on one GPU over gloo.
    Same stack, same 150 sorted seeds: the sharded step reports n_gpus 2, strong scaling and the node count of the 1-GPU step."""
    common = ["--size", "160", "--seeds", "150", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra"]
    one = _bench(["--gpus", "1"] + common, {})
    two = _bench(["--gpus", "2"] + common, {"PNR_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong" and one["scaling"] == "strong"
    assert two["counts"]["nodes"] == one["counts"]["nodes"] > 50
    assert two["counts"]["n_seeds"] == one["counts"]["n_seeds"] and two["counts"]["n_seeds_init"] == one["counts"]["n_seeds_init"]
    assert two["counts"]["traces_used"] == one["counts"]["traces_used"]
    for k in ("roofline", "roofline_sample", "roofline_sums"):
        assert two[k]["frac"] > 0 and one[k]["frac"] > 0
    assert one["roofline"]["kernel"] == "ph_predict+ph_cube+ph_sample<54, false, true>+ph_sums+ph_update" and one["roofline"]["dominant_by_device_time"] in one["roofline"]["kernel"]


def test_bench_gpus_2_with_the_rccl_shaped_exchange():
    """the per-poll exchange of trace records in the form it takes across nodes -- pinned staging buffers and ONE all_gather_into_tensor
    per exchange (pnr_amd/multigpu.py make_exchange) -- with a world of TWO: rehearsed on one GPU over gloo (which cannot gather device
    tensors: the device hop of the RCCL form is covered with a world of one below), against the shared-memory default"""
    common = ["--size", "160", "--seeds", "150", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra", "--gpus", "2"]
    shm = _bench(common, {"PNR_BENCH_BACKEND": "gloo"})
    rc = _bench(common, {"PNR_BENCH_BACKEND": "gloo", "PNR_BENCH_EXCHANGE": "rccl"})
    assert "shared memory" in shm["config"]["record_exchange"] and "all_gather_into_tensor" in rc["config"]["record_exchange"]
    for k in ("nodes", "n_seeds", "n_seeds_init", "traces_used"):
        assert rc["counts"][k] == shm["counts"][k], k
    assert rc["counts"]["nodes"] > 50 and rc["n_gpus"] == 2


def test_exchange_callback_over_rccl_one_rank():
    """the RCCL form of the exchange callback (pinned staging -> device -> all_gather_into_tensor -> host) with a world of one, the
    only RCCL world a one-GPU box can form: the block comes back unchanged, twice, also after the block size changes"""
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.environ["PNR_ROOT"])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["PNR_PORT"], RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
from pnr_amd import multigpu
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ex = multigpu.make_exchange(dist, 1, torch.device("cuda", 0))
for nb in (4096, 4096, 65536):
    send = (C.c_ubyte * nb)(*[(7 * i + nb) % 251 for i in range(nb)])
    recv = (C.c_ubyte * nb)()
    rc = ex(None, C.cast(send, C.c_void_p), C.cast(recv, C.c_void_p), nb)
    assert rc == 0 and bytes(recv) == bytes(send), nb
dist.destroy_process_group()
print("exchange ok")
'''
    env = dict(os.environ, PNR_ROOT=ROOT, PNR_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "exchange ok" in r.stdout, r.stderr[-2000:]


def test_bench_sharded_path_over_rccl_world_of_one():
    """the sharded step of bench.py with its RCCL collectives (broadcast, all-gather of seeds, all-reduce of the timings) on the only
    RCCL world a one-GPU box can form, against the plain one-GPU step: same seeds, traces and nodes"""
    common = ["--size", "160", "--seeds", "150", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extra"]
    one = _bench(common, {})
    env = dict(PNR_BENCH_FORCE_DIST="1", PNR_BENCH_FORCE_SHARD="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200))
    sh = _bench(common, env)
    assert sh["config"]["record_exchange"] and sh["n_gpus"] == 1
    for k in ("nodes", "n_seeds", "n_seeds_init", "traces_used"):
        assert sh["counts"][k] == one["counts"][k], k


def test_c_abi_rccl_exchange_world_of_one():
    """pnr_rccl_exchange (include/pnr_hip.h; rccl_exchange.cpp): the ncclAllGather-backed pnr_allgather_fn and the (min, max)
    ncclAllReduce from a host that is not torch, on the only RCCL world a one-GPU box can form -- in a fresh process, before anything
    else has touched the GPU, as advantra_cli's ranks do: blocks come back unchanged (several sizes, also after the size changes, up
    to the capacity), the all-reduce is the identity, a block beyond the capacity is refused."""
    code = r'''
import os, sys
sys.path.insert(0, os.environ["PNR_ROOT"])
os.environ["PNR_NO_TORCH_PRELOAD"] = "1"
from pnr_amd import lib
x = lib.RcclExchange(lib.RcclExchange.unique_id(), 0, 1, 0, capacity=1 << 18)
for nb in (16, 4096, 4096, 65536, 1 << 18, 8):
    blk = bytes((7 * i + nb) % 251 for i in range(nb))
    got = x.allgather(blk)
    assert len(got) == 1 and got[0] == blk, nb
assert x.minmax(0.25, 3.5) == (0.25, 3.5) and x.minmax(-1.5, -1.0) == (-1.5, -1.0)
try:
    x.allgather(bytes((1 << 18) + 16))
    raise SystemExit("a block beyond the capacity was accepted")
except lib.PnrError as e:
    assert "opened for" in str(e)
x.close()
print("rccl exchange ok")
'''
    env = dict(os.environ, PNR_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl exchange ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_cli_ranks_over_rccl_world_of_one(tmp_path):
    """advantra_cli --ranks 1 --exchange rccl: the C++ host's sharded path (z-slab Frangi, 2-float ncclAllReduce, ncclAllGather of the
    scored seeds, pnr_trace_replay_sharded with pnr_rccl_allgather as its exchange) on a world of one, against the plain CLI and against
    the shared-memory transport's sharded path: the same SWC file, byte for byte"""
    CLI = os.path.join(ROOT, "pnr_amd", "host", "advantra_cli")
    raw = str(tmp_path / "s.raw")
    synth.synth(96, 80, 40, seed=7).tofile(raw)
    paras = "2,3 0 5 0.3 3 2 40 48 2 4 1".split()
    outs = {}
    for name, extra in (("plain", ()), ("rccl", ("--ranks", "1", "--exchange", "rccl"))):
        r = subprocess.run([CLI, "-d", "96,80,40", *extra, "-f", "advantra_func", "-i", raw, "-p", *paras], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, (name, r.stdout[-800:], r.stderr[-2000:])
        outs[name] = open(raw + "_Advantra.swc").read()
        os.remove(raw + "_Advantra.swc")
    body = lambda t: [ln for ln in t.splitlines() if ln and ln[0] != "#"]
    assert body(outs["plain"]) == body(outs["rccl"]) and len(body(outs["plain"])) > 50
