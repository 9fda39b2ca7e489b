"""GPU parity of the 2-D mode: a single-slice stack (P == 1) takes frangi2d (Advantra_plugin.cpp:2496-2497) and the
2-D branches of the tracker (30 directions, in-plane prediction offsets, (v, u, 0) templates, bilinear interp, w = 0).
Frangi is compared with the oracle (itself pinned on the reference's frangi.cpp) bit for bit; tracker tables, seed scores,
traces and the end-to-end node graph / tree list with the oracle's 2-D tracker."""
import numpy as np
import pytest
import orc
import synth
import pnr_amd
from pnr_amd import lib

pytestmark = pytest.mark.gpu
# Contract of the particle filter: BIT IDENTITY with the oracle -- every IEEE operation of the reference's scalar loops in its order
# (DESIGN.md 2): particle states, weights, N_eff, estimates, resampling indices and stop reasons are compared with array_equal.


def _slice(w=96, h=80, seed=4):
    vol = synth.synth(w, h, 9, seed=seed)
    return np.ascontiguousarray(vol.max(0, keepdims=True))  # maximum projection: one slice with all the tubes


def mat(a):
    return np.stack([a[k] for k in a.dtype.names], -1)


@pytest.mark.parametrize("shape,sigs", [((1, 80, 96), [2.0, 3.0]), ((1, 33, 21), [2.0]), ((1, 64, 64), [1.0, 2.0, 4.0])])
def test_frangi2d_seeds_vs_oracle(oracle, shape, sigs):
    _, h, w = shape
    img = _slice(w, h)
    Jo, jmin, jmax, Vxo, Vyo, Vzo = orc.frangi2d(oracle, img, sigs)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=20, ni=5), 0)
    c.set_volume(img)
    gmin, gmax = c.frangi()
    g = c.get_frangi()
    J, J8, Vx, Vy, Vz = g["J"], g["J8"], g["Vx"], g["Vy"], g["Vz"]
    assert np.array_equal(J, Jo) and gmin == jmin and gmax == jmax
    assert np.array_equal(Vx, Vxo) and np.array_equal(Vy, Vyo) and np.array_equal(Vz, Vzo)
    J8o = orc.j8(oracle, Jo, jmin, jmax)
    assert np.array_equal(J8, J8o)
    so = orc.extract_seeds(oracle, 5, J8o, Vxo, Vyo, Vzo)
    s = c.extract_seeds()
    assert len(s) == len(so) and np.array_equal(mat(s)[:, :6], so[:, :6], equal_nan=True)
    if w > 40:
        assert len(so) > 10 and np.all(so[:, 2] == 0)


def test_tables_2d_bit_exact(oracle):
    img = _slice()
    sigs = [2.0, 3.0]
    To = orc.Tracker(oracle, sigs, 2, 30, 10, 3.0, 0.3, zdist=2.0, is2d=True)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=30, ni=10, zdist=2.0), 0)
    c.set_volume(img)
    assert To.ndir == 30 and To.sz == 48  # disc of radius 2*step = 4 without the centre
    for nm in ("p", "u", "w0", "w0_cws", "v", "w", "w_cws"):
        assert np.array_equal(c.table(nm).reshape(-1), To.table(nm).reshape(-1)), nm
    for s in range(len(sigs)):
        vuw, wgt, avg = To.model(s)
        assert np.array_equal(c.table(f"model_vuw{s}").reshape(-1, 3), vuw) and np.all(vuw[:, 2] == 0)
        assert np.array_equal(c.table(f"model_wgt{s}"), wgt)
    # switching back to a 3-D stack rebuilds the 3-D tables
    c.set_volume(synth.synth(32, 32, 8, seed=1))
    assert len(c.table("v")) // 3 == 50


@pytest.mark.parametrize("sigs,np_,ni", [([2.0], 40, 25), ([2.0, 3.0], 64, 15)])
def test_trace_2d_vs_oracle(oracle, sigs, np_, ni):
    img = _slice()
    Jo, jmin, jmax, Vxo, Vyo, Vzo = orc.frangi2d(oracle, img, sigs)
    so = orc.extract_seeds(oracle, 5, orc.j8(oracle, Jo, jmin, jmax), Vxo, Vyo, Vzo)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=2.0, is2d=True)
    corr_o, sig_o = To.zncc(img, so[:, :6])
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=2.0), 0)
    c.set_volume(img)
    corr, sig = c.zncc(so[:, :6])
    assert np.array_equal(corr, corr_o) and np.array_equal(sig, sig_o)
    so = so[np.argsort(-corr_o, kind="stable")][:6]
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    T, stop, xc, dbg = c.trace_batch(seeds, dbg_iters=ni)
    for i, sd in enumerate(so):
        for d, sgn in enumerate((1, -1)):
            q = sd[:6].copy(); q[3:] *= sgn
            Tn, st, xco, xf, idx, neff = To.trace(img, q, max_dbg=ni)
            j = 2 * i + d
            assert T[j] == Tn and stop[j] == st, (j, T[j], Tn, stop[j], st)
            rows = min(Tn + 1, ni)
            assert np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True)
            assert np.array_equal(dbg["xfilt"][j, :rows], xf[:rows], equal_nan=True)
    assert T.max() > 3


def test_persistent_driver_rejects_2d():
    img = _slice()
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], np_=20, ni=5), 0)
    c.set_smc_driver("persistent")
    c.set_volume(img)
    c.frangi()
    s = c.score_filter_sort(c.extract_seeds())[:2]
    with pytest.raises(lib.PnrError, match="phased"):
        c.trace_batch(s)


def test_end_to_end_2d_vs_oracle(oracle):
    img = _slice()
    sigs, np_, ni, zdist = [2.0, 3.0], 40, 25, 2.0
    Jo, jmin, jmax, Vxo, Vyo, Vzo = orc.frangi2d(oracle, img, sigs)
    so = orc.extract_seeds(oracle, 5, orc.j8(oracle, Jo, jmin, jmax), Vxo, Vyo, Vzo)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist, is2d=True)
    corr, _ = To.zncc(img, so[:, :6])
    so[:, 7] = corr
    so = so[corr >= 0.3]
    so = so[np.argsort(-so[:, 7], kind="stable")]
    Tn, xcs = [], []
    for sd in so:
        for sgn in (1, -1):
            q = sd[:6].copy(); q[3:] *= sgn
            t, st, xco, *_ = To.trace(img, q)
            Tn.append(t); xcs.append(xco)
    nodes_o, links_o, _ = orc.replay(oracle, so, np.array(Tn, np.int32), np.stack(xcs), ni, img.shape, 4, 1)
    tree_o, par_o = orc.reconstruct(oracle, nodes_o, links_o)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=zdist), 0)
    res = pnr_amd.advantra.run_pipeline(c, img)
    assert len(res["seeds"]) == len(so) > 10
    assert len(res["nodes"]) == len(nodes_o) > 30 and np.array_equal(res["links"], links_o)
    for k in res["nodes"].dtype.names:
        assert np.array_equal(res["nodes"][k], nodes_o[k]), k
    assert len(res["tree"]) == len(tree_o) and np.array_equal(res["parent"], par_o)
    for k in tree_o.dtype.names:
        assert np.array_equal(res["tree"][k], tree_o[k]), k


@pytest.mark.parametrize("name", ["p2d_96x80_s2-3", "p2d_33x21_s2"])
def test_frangi2d_seeds_vs_golden(name):
    """the reference's own frangi2d / extractSeeds outputs (tests/golden/make_golden.py), bit for bit"""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz")))
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[float(v) for v in g["sigs"]], tolerance=float(g["tol"]), np_=20, ni=5), 0)
    c.set_volume(g["img"])
    jmin, jmax = c.frangi()
    f = c.get_frangi()
    assert np.array_equal(f["J"], g["J"]) and jmin == g["Jmin"] and jmax == g["Jmax"]
    assert np.array_equal(f["Vx"], g["Vx"]) and np.array_equal(f["Vy"], g["Vy"]) and np.array_equal(f["Vz"], g["Vz"])
    assert np.array_equal(f["J8"], g["J8_restated"])
    s = c.extract_seeds()
    assert len(s) == len(g["seeds"]) and np.array_equal(mat(s)[:, :6], g["seeds"][:, :6], equal_nan=True)
