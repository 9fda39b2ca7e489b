"""CPU: the C-ABI library loads and exports every symbol include/pnr_hip.h declares, refuses to
run without a GPU (no CPU path), and its pure-host pieces (tables, replay) match the oracle."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import orc
import synth
import pnr_amd
from pnr_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_header_symbols():
    declared = []
    for name, exports in (("pnr_hip.h", lib.PRODUCT_EXPORTS), ("pnr_hip_test.h", lib.TEST_EXPORTS)):
        hdr = open(os.path.join(ROOT, "include", name)).read()
        here = sorted(set(re.findall(r"^(?:int|void|const char \*)\s*(pnr_[a-z0-9_]+)\s*\(", hdr, re.M)))
        assert set(here) == set(exports), (name, set(here) ^ set(exports))
        declared += here
    L = lib.load()
    for name in declared:
        assert getattr(L, name) is not None


def test_struct_layout_matches_header():
    assert C.sizeof(lib.Params) == 8 * 4 + 18 * 4
    assert lib.SEED_DT.itemsize == 32 and lib.XEST_DT.itemsize == 32 and lib.NODE_DT.itemsize == 36


def test_no_cpu_fallback():
    if _has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(pnr_amd.PnrError, match="no HIP device"):
        pnr_amd.Context(pnr_amd.make_params(), 0)


def test_rccl_exchange_needs_the_ranks_gpu():
    """the C-ABI RCCL transport (pnr_rccl_*): librccl is opened at run time (no link-time dependency of libpnr_hip.so), the unique id
    is 128 bytes that differ from call to call, and opening an exchange without the rank's GPU fails with PNR_E_NODEVICE -- no fallback"""
    import subprocess
    deps = subprocess.run(["ldd", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in deps
    a, b = lib.RcclExchange.unique_id(), lib.RcclExchange.unique_id()
    assert len(a) == 128 and a != b
    if _has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(pnr_amd.PnrError, match="no HIP device"):
        lib.RcclExchange(a, 0, 1, 0)


def test_parameter_validation_messages():
    """range errors of Advantra::dofunc (Advantra_plugin.cpp:317-326) surface before any device work"""
    L = lib.load()
    h = C.c_void_p()
    for kw, msg in [(dict(znccth=1.5), "znccth out of range"), (dict(kappa=6), "kappa out of range"),
                    (dict(step=0), "step out of range"), (dict(ni=0), "ni out of range"), (dict(np_=0), "np out of range"),
                    (dict(zdist=0.5), "zdist out of range"), (dict(nodepervol=2), "nodepervol out of range"),
                    (dict(vol=7), "vol can be 1,5,9,11,19,27"), (dict(tolerance=-1), "tolerance out of range")]:
        p = pnr_amd.make_params(**kw)
        assert L.pnr_create(C.byref(p), 0, C.byref(h)) == -1
        assert L.pnr_last_error().decode() == msg


def test_advantra_func_contract():
    """11 positional parameters or help + False; range error -> 0 (Advantra_plugin.cpp:295-326)"""
    assert pnr_amd.advantra_func(["x.tif"], ["2,4,6", "0", "5"]) is False
    assert pnr_amd.advantra_func([], ["2"] * 11) is False
    assert pnr_amd.advantra_func(["x.tif"], "2,4,6 0 5 0.3 9 2 200 20 2 4 1".split()) == 0  # kappa>5
    assert pnr_amd.advantra_func(["x.tif"], "2,4,6 0 5 0.3 3 2 200 20 2 4 3".split()) == 0  # vol


def _traces_from_oracle(oracle, img, sigs, np_, ni, zdist, nseeds=12):
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, sigs, zdist)
    J8 = orc.j8(oracle, J, jmin, jmax)
    s = orc.extract_seeds(oracle, 5, J8, Vx, Vy, Vz)
    T = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    corr, _ = T.zncc(img, s[:, :6])
    s[:, 7] = corr
    s = s[corr >= 0.3]
    s = s[np.argsort(-s[:, 7], kind="stable")][:nseeds]
    Ts, xcs = [], []
    for sd in s:
        for sgn in (1, -1):
            q = sd[:6].copy()
            q[3:] *= sgn
            Tn, stop, xc, *_ = T.trace(img, q)
            Ts.append(Tn)
            xcs.append(xc)
    return s, np.array(Ts, np.int32), np.stack(xcs)


@pytest.mark.parametrize("vol,nodepervol", [(1, 4), (5, 3), (27, 4)])
def test_replay_matches_oracle(oracle, vol, nodepervol):
    img = synth.synth(48, 40, 24, seed=1)
    s, T, xc = _traces_from_oracle(oracle, img, [2.0], 24, 30, 2.0)
    assert len(s) >= 4 and T.sum() > 20
    nodes_o, links_o, nt_o = orc.replay(oracle, s, T, xc, 30, img.shape, nodepervol, vol)
    p = pnr_amd.make_params(sigmas=[2.0], np_=24, ni=30, nodepervol=nodepervol, vol=vol)
    seeds = np.zeros(len(s), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = s[:, i]
    nodes, links, nt = lib.replay(p, img.shape, seeds, T, xc.view(lib.XEST_DT).reshape(len(T), 30))
    assert nt == nt_o and len(nodes) == len(nodes_o) > 1
    for k in nodes.dtype.names:
        assert np.array_equal(nodes[k], nodes_o[k]), k
    assert np.array_equal(links, links_o)
    assert (nodes["type"][1:] == 6).sum() >= 1  # END markers present


def test_replay_many_voxels_matches_oracle(oracle):
    """the replay's voxel table (Replayer::CellMap, open addressing, grows by doubling from 2^17 slots) against the oracle's dense
    arrays on synthetic random-walk traces that put > 80 000 nodes (each with its four in-plane neighbours: several hundred thousand voxels) and cross each other (DENSITY links, vol = 5 neighbours)"""
    rs = np.random.RandomState(11)
    shape = (96, 160, 192)  # l, h, w
    l, h, w = shape
    ni, nseed = 200, 700
    s = np.zeros((nseed, 8), np.float32)
    s[:, 0] = rs.uniform(4, w - 5, nseed); s[:, 1] = rs.uniform(4, h - 5, nseed); s[:, 2] = rs.uniform(4, l - 5, nseed)
    s[:, 3:6] = rs.randn(nseed, 3); s[:, 6] = 0.9; s[:, 7] = 2.0
    T = rs.randint(60, ni + 1, 2 * nseed).astype(np.int32)
    xc = np.zeros((2 * nseed, ni, 8), np.float32)
    for t in range(2 * nseed):
        d = rs.randn(3); d /= np.linalg.norm(d)
        steps = 1.1 * d + 0.5 * rs.randn(ni, 3)
        pos = s[t // 2, :3] + np.cumsum(steps, 0)
        pos[:, 0] = np.clip(pos[:, 0], 0, w - 1.01); pos[:, 1] = np.clip(pos[:, 1], 0, h - 1.01); pos[:, 2] = np.clip(pos[:, 2], 0, l - 1.01)
        xc[t, :, 0:3] = pos
        xc[t, :, 3:6] = d
        xc[t, :, 6] = 2.0
        xc[t, :, 7] = rs.uniform(0.4, 0.9, ni)
    for npv, vol, least in ((4, 5, 80000), (1, 1, 15000)):
        nodes_o, links_o, nt_o = orc.replay(oracle, s, T, xc, ni, shape, npv, vol)
        p = pnr_amd.make_params(sigmas=[2.0], np_=24, ni=ni, nodepervol=npv, vol=vol)
        seeds = np.zeros(nseed, lib.SEED_DT)
        for i, k in enumerate(lib.SEED_DT.names):
            seeds[k] = s[:, i]
        nodes, links, nt = lib.replay(p, shape, seeds, T, xc.view(lib.XEST_DT).reshape(len(T), ni))
        assert nt == nt_o and len(nodes) == len(nodes_o) > least, (len(nodes), len(nodes_o))
        for k in nodes.dtype.names:
            assert np.array_equal(nodes[k], nodes_o[k]), k
        assert np.array_equal(links, links_o)


@pytest.mark.parametrize("vol,tree_min", [(5, 10), (1, 3)])
def test_reconstruct_matches_oracle(oracle, vol, tree_min):
    """reconstruct() chain (Advantra_plugin.cpp:2096-2181): the grid-accelerated host implementation visits
    neighbours in ascending index, so it must equal the oracle's plain O(n^2) restatement bit for bit."""
    img = synth.synth(48, 40, 24, seed=1)
    s, T, xc = _traces_from_oracle(oracle, img, [2.0], 24, 30, 2.0, nseeds=20)
    nodes, links, _ = orc.replay(oracle, s, T, xc, 30, img.shape, 4, vol)
    assert len(nodes) > 60 and len(links) > 40
    want_n, want_p = orc.reconstruct(oracle, nodes, links, tree_size_min=tree_min)
    got_n, got_p = lib.reconstruct(nodes.astype(lib.NODE_DT), links, tree_size_min=tree_min)
    assert len(got_n) == len(want_n) > 10
    assert np.array_equal(got_p, want_p)
    for k in got_n.dtype.names:
        assert np.array_equal(got_n[k], want_n[k], equal_nan=True), k
    # tree-list invariants: one parent per node, parents are valid, roots exist, all types AXON(2) after the final resampling
    assert got_p[0] == -1 and (got_p[1:] < len(got_n)).all() and (got_p[1:] == -1).sum() >= 1
    assert set(np.unique(got_n["type"][1:])) == {2}


def test_reconstruct_dense_graph_matches_oracle(oracle):
    """a dense graph (random-walk traces that cross each other: balls of the mean-shift hold dozens of nodes out of many grid
    cells, thousands of nodes share a corr value so the grouping order falls back to the node index) through the packed grid,
    the filter-then-order neighbour scan and the keyed corr sort of the host implementation, against the oracle's O(n^2) scan"""
    rs = np.random.RandomState(5)
    shape = (40, 64, 72)
    l, h, w = shape
    ni, nseed = 50, 45
    s = np.zeros((nseed, 8), np.float32)
    s[:, 0] = rs.uniform(20, w - 20, nseed); s[:, 1] = rs.uniform(20, h - 20, nseed); s[:, 2] = rs.uniform(12, l - 12, nseed)
    s[:, 3:6] = rs.randn(nseed, 3); s[:, 6] = 0.9; s[:, 7] = 2.0
    T = rs.randint(30, ni + 1, 2 * nseed).astype(np.int32)
    xc = np.zeros((2 * nseed, ni, 8), np.float32)
    for t in range(2 * nseed):
        d = rs.randn(3); d /= np.linalg.norm(d)
        pos = s[t // 2, :3] + np.cumsum(1.3 * d + 0.4 * rs.randn(ni, 3), 0)
        pos[:, 0] = np.clip(pos[:, 0], 0, w - 1.01); pos[:, 1] = np.clip(pos[:, 1], 0, h - 1.01); pos[:, 2] = np.clip(pos[:, 2], 0, l - 1.01)
        xc[t, :, 0:3] = pos
        xc[t, :, 3:6] = d
        xc[t, :, 6] = rs.choice([2.0, 4.0, 6.0], ni)
        xc[t, :, 7] = np.round(rs.uniform(0.4, 0.9, ni), 1)  # six distinct corr values
    nodes, links, _ = orc.replay(oracle, s, T, xc, ni, shape, 4, 5)
    assert len(nodes) > 2500
    want_n, want_p = orc.reconstruct(oracle, nodes, links, tree_size_min=4)
    got_n, got_p = lib.reconstruct(nodes.astype(lib.NODE_DT), links, tree_size_min=4)
    assert len(got_n) == len(want_n) > 300
    assert np.array_equal(got_p, want_p)
    for k in got_n.dtype.names:
        assert np.array_equal(got_n[k], want_n[k], equal_nan=True), k


def test_replay_and_reconstruct_with_soma_match_oracle(oracle):
    """soma nodes in the node list: the oracle's replay with a soma map supplies a graph whose traces end on
    SOMA nodes; group1 keeps soma nodes as groups of their own (Advantra_plugin.cpp:1580-1588), bfs keeps their type."""
    img = synth.add_somas(synth.synth(64, 56, 32, seed=2), ((20, 28, 16, 6), (48, 20, 14, 5)))
    E8, th, smap, n4 = orc.soma_extract(oracle, img, 3)
    assert len(n4) >= 1
    s, T, xc = _traces_from_oracle(oracle, img, [2.0], 24, 30, 2.0, nseeds=40)
    l, h, w = img.shape
    vox = np.round(s[:, 2]).astype(np.int64) * w * h + np.round(s[:, 1]).astype(np.int64) * w + np.round(s[:, 0]).astype(np.int64)
    keep = smap.reshape(-1)[vox] == 0
    s, T, xc = s[keep], T.reshape(-1, 2)[keep].reshape(-1), xc.reshape(len(keep), 2, 30, 8)[keep].reshape(-1, 30, 8)
    nodes, links, _ = orc.replay(oracle, s, T, xc, 30, img.shape, 4, 1, smap=smap, soma4=n4)
    ns = len(n4)
    assert np.all(nodes["type"][1:1 + ns] == 1) and ((links >= 1) & (links <= ns)).any()
    want_n, want_p = orc.reconstruct(oracle, nodes, links, tree_size_min=3)
    got_n, got_p = lib.reconstruct(nodes.astype(lib.NODE_DT), links, tree_size_min=3)
    assert len(got_n) == len(want_n) > 10 and np.array_equal(got_p, want_p)
    for k in got_n.dtype.names:
        assert np.array_equal(got_n[k], want_n[k], equal_nan=True), k
    assert (got_n["type"] == 1).sum() >= 1  # a soma node survives in the tree list with its type


def test_reconstruct_with_nan_corr_nodes():
    """group1's sort by corr (Advantra_plugin.cpp:1571) must stay defined when some nodes carry a NaN corr (the stop test
    `corr < znccth` lets a NaN through): numbers by decreasing corr, NaNs last, ties by index -- product and oracle alike, on more
    nodes than std::sort's insertion-sort threshold (16) so that an invalid comparator would walk out of bounds"""
    rng = np.random.default_rng(11)
    n = 400
    nodes = np.zeros(n + 1, lib.NODE_DT)
    t = np.arange(n)
    nodes["x"][1:] = 5 + 0.8 * t % 60
    nodes["y"][1:] = 5 + (t // 75) * 1.5 + rng.random(n).astype(np.float32)
    nodes["z"][1:] = 5 + rng.random(n).astype(np.float32)
    nodes["vx"][1:] = 1
    nodes["sig"][1:] = 2.0
    nodes["corr"][1:] = rng.uniform(0.3, 0.95, n).astype(np.float32)
    nodes["corr"][1 + rng.choice(n, 37, replace=False)] = np.nan
    nodes["corr"][[7, 8, 9]] = np.float32(0.5)  # exact ties beside the NaNs
    nodes["type"][1:] = 2
    links = np.array([[i + 1, i] for i in range(1, n) if i % 75], np.int32)
    L = orc.load_oracle()
    want_n, want_p = orc.reconstruct(L, nodes, links, tree_size_min=2)
    for _ in range(2):
        got_n, got_p = lib.reconstruct(nodes, links, tree_size_min=2)
        assert len(got_n) == len(want_n) > 20 and np.array_equal(got_p, want_p)
        for k in got_n.dtype.names:
            assert np.array_equal(got_n[k], want_n[k], equal_nan=True), k


def test_reconstruct_single_tree_and_stage_taps(oracle):
    """the plugin's ENFORCE_SINGLE_TREE branch (Advantra_plugin.cpp:81, :2142-2152; extract_largest_tree :546-589: tree_size_min < 0)
    against the oracle, and the saveMidres taps of reconstruct() (:2098-2141, pnr_reconstruct_stage): the lists behind the stages are
    consistent with each other and with the final result"""
    img = synth.synth(64, 56, 32, seed=2)
    s, T, xc = _traces_from_oracle(oracle, img, [2.0], 24, 30, 2.0, nseeds=40)
    nodes, links, _ = orc.replay(oracle, s, T, xc, 30, img.shape, 4, 1)
    nodes = nodes.astype(lib.NODE_DT)
    want_n, want_p = orc.reconstruct(oracle, nodes, links, tree_size_min=-1)
    got_n, got_p = lib.reconstruct(nodes, links, tree_size_min=-1)
    assert len(got_n) == len(want_n) > 10 and np.array_equal(got_p, want_p)
    for k in got_n.dtype.names:
        assert np.array_equal(got_n[k], want_n[k], equal_nan=True), k
    assert (got_p[1:] == -1).sum() == 1                      # one tree
    # three separate chains of 12 / 20 / 7 nodes: all three survive tree_size_min = 2, only the longest the single-tree branch
    n = 12 + 20 + 7
    ch = np.zeros(n + 1, lib.NODE_DT)
    off = 1
    lk = []
    for c, m in enumerate((12, 20, 7)):
        ch["x"][off:off + m] = 10 + 2.0 * np.arange(m)
        ch["y"][off:off + m] = 10 + 40 * c
        ch["z"][off:off + m] = 8
        lk += [[off + i + 1, off + i] for i in range(m - 1)]
        off += m
    ch["sig"][1:] = 2.0
    ch["corr"][1:] = np.linspace(0.9, 0.4, n)
    ch["type"][1:] = 2
    lk = np.array(lk, np.int32)
    all_n, all_p = lib.reconstruct(ch, lk, tree_size_min=2)
    one_n, one_p = lib.reconstruct(ch, lk, tree_size_min=-1)
    wn, wp = orc.reconstruct(oracle, ch, lk, tree_size_min=-1)
    assert (all_p[1:] == -1).sum() == 3 and (one_p[1:] == -1).sum() == 1 and len(one_n) < len(all_n)
    assert np.array_equal(one_p, wp) and all(np.array_equal(one_n[k], wn[k], equal_nan=True) for k in wn.dtype.names)
    assert np.all(np.abs(one_n["y"][1:] - 50) < 1e-3) and len(one_n) - 1 >= 20  # the 20-node chain (mean-shifted, grouped, resampled)
    # stage taps: 1 resampled links (every link <= TRACE_RSMPL... at most ~1 voxel), 2 mean-shift keeps nodes and links, 3 grouping
    # shrinks the list, 4 the BFS forest keeps every node of a tree of >= 2 nodes with at most one parent
    n0r, l0r = lib.reconstruct_stage(nodes, links, 1)
    n1, l1 = lib.reconstruct_stage(nodes, links, 2)
    n2, l2 = lib.reconstruct_stage(nodes, links, 3)
    n2t, l2t = lib.reconstruct_stage(nodes, links, 4)
    assert len(n0r) >= len(nodes) and len(l0r) >= len(links)
    d = np.sqrt(sum((n0r[k][l0r[:, 0]] - n0r[k][l0r[:, 1]]) ** 2 for k in "xyz"))
    assert d.max() <= 1.0 + 1e-4
    assert len(n1) == len(n0r) and np.array_equal(l1, l0r) and not np.array_equal(n1["x"], n0r["x"])
    assert len(n2) < len(n1) and len(l2) > 0 and l2.max() < len(n2)
    assert len(n2t) <= len(n2) and len(np.unique(l2t[:, 0])) == len(l2t) and l2t.max() < len(n2t)


def test_reconstruct_degenerate_inputs():
    """only the dummy node; isolated nodes; a self-link and duplicate links (what a DENSITY stop can produce)"""
    dummy = np.zeros(1, lib.NODE_DT)
    t, p = lib.reconstruct(dummy, np.zeros((0, 2), np.int32))
    assert len(t) == 1 and p[0] == -1
    nodes = np.zeros(16, lib.NODE_DT)
    nodes["x"][1:] = np.arange(15) * 3.0
    nodes["sig"][1:] = 2.0
    nodes["corr"][1:] = np.linspace(0.9, 0.5, 15)
    nodes["type"][1:] = 2
    links = np.array([[i + 1, i] for i in range(1, 15)] + [[5, 5], [3, 2], [3, 2]], np.int32)
    t1, p1 = lib.reconstruct(nodes, links, tree_size_min=1)
    import orc as _orc
    L = _orc.load_oracle()
    t2, p2 = _orc.reconstruct(L, nodes, links, tree_size_min=1)
    assert np.array_equal(p1, p2) and all(np.array_equal(t1[k], t2[k], equal_nan=True) for k in t1.dtype.names)
    assert len(t1) > 15 and (p1[1:] == -1).sum() == 1  # one chain -> one tree, resampled at unit steps
    iso = nodes.copy()
    t3, p3 = lib.reconstruct(iso, np.zeros((0, 2), np.int32), tree_size_min=1)
    assert len(t3) == 1  # single-node trees are dropped by bfs2
    with pytest.raises(pnr_amd.PnrError, match="link index out of range"):
        lib.reconstruct(nodes, np.array([[1, 99]], np.int32))


def test_swc_diff_tool(tmp_path):
    """scripts/swc_diff.py: ids / types / parents exactly, x / y / z / r within the tolerance, first difference reported"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import swc_diff
    a = tmp_path / "a.swc"
    b = tmp_path / "b.swc"
    a.write_text("# name x\n#param=1\n1 2 1.000 2.000 3.000 1.500 -1\n2 2 2.000 2.500 3.000 1.500 1\n3 6 3.000 3.000 3.250 0.750 2\n")
    b.write_text("#other header\n1 2 1.0004 2.000 3.000 1.5 -1\n2 2 2.000 2.5009 3.000 1.500 1\n\n3 6 3.000 3.000 3.250 0.75 2\n")
    ok, msg = swc_diff.diff(str(a), str(b))
    assert ok and "3 nodes" in msg, msg
    assert not swc_diff.diff(str(a), str(b), tol=5e-4)[0]
    b.write_text("1 2 1.0 2.0 3.0 1.5 -1\n2 2 2.0 2.5 3.0 1.5 1\n3 6 3.0 3.0 3.25 0.75 1\n")
    ok, msg = swc_diff.diff(str(a), str(b))
    assert not ok and "parent differs at node #2" in msg, msg
    b.write_text("3 6 3.0 3.0 3.25 0.75 2\n1 2 1.0 2.0 3.0 1.5 -1\n2 2 2.0 2.5 3.0 1.5 1\n")
    assert not swc_diff.diff(str(a), str(b))[0] and swc_diff.diff(str(a), str(b), unordered=True)[0]
    tool = [sys.executable, os.path.join(ROOT, "scripts", "swc_diff.py")]
    assert subprocess.run(tool + [str(a), str(b), "--unordered", "--quiet"]).returncode == 0
    assert subprocess.run(tool + [str(a), str(b), "--quiet"]).returncode == 1
    assert subprocess.run(tool + [str(a), str(tmp_path / "missing.swc")], capture_output=True).returncode == 2
