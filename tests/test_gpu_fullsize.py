"""BASELINE.json configs at their real sizes and parameters inside `-m gpu` (the builder-run scripts/run_config5.py and
scripts/parity_production.py, trimmed to what the test box does in well under a minute each):

  * configs[0]: 128 x 128 x 64, production parameters (scales {2,4,6}, np = 200, ni = 200, zdist 2): Frangi / J8 / V / seeds / seed
    scores and EVERY iteration of the first sorted seeds' traces, to their full depth, against the oracle byte for byte; then the
    replayed graph and the reconstruct() tree.
  * configs[4]: 2048 x 2048 x 512 (2^31 voxels, past the reference's int indexing), scales {2,4,6,8}, zdist 4, np = 500: properties
    that need no oracle -- seeds appear in every quadrant, the streamed schedule gives the one-shot node graph, also for seeds
    whose voxel index lies beyond 2^30 (cube staging, density map and replay in 64-bit arithmetic).
"""
import numpy as np
import pytest
import orc
import synth
import pnr_amd
from pnr_amd import lib

pytestmark = pytest.mark.gpu
mat = lambda a: np.stack([a[k] for k in a.dtype.names], -1)


def test_config0_production_parameters_vs_oracle(oracle, nseeds=5):
    L = oracle
    sigs, np_, ni, zdist = [2.0, 4.0, 6.0], 200, 200, 2.0
    img = synth.synth(128, 128, 64, seed=1)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=zdist), 0)
    c.set_volume(img)
    c.frangi()
    g = c.get_frangi(J=True, J8=True, V=True)
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, img, sigs, zdist)
    J8 = orc.j8(L, J, jmin, jmax)
    for k, want in (("J", J), ("J8", J8), ("Vx", Vx), ("Vy", Vy), ("Vz", Vz)):
        assert np.array_equal(g[k].reshape(want.shape), want), k
    so = orc.extract_seeds(L, 5, J8, Vx, Vy, Vz)
    sg = c.extract_seeds()
    assert np.array_equal(mat(sg)[:, :6], so[:, :6]) and len(so) > 100
    T = orc.Tracker(L, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    corr, _ = T.zncc(img, so[:, :6])
    ss = c.score_filter_sort(sg)
    keep = corr >= np.float32(0.3)
    assert np.array_equal(ss["corr"], corr[keep][np.argsort(-corr[keep], kind="stable")])
    sel = ss[:nseeds]
    Tg, stop, xc, _ = c.trace_batch(sel)
    its = 0
    for i in range(len(sel)):
        for d_, sgn in enumerate((1, -1)):
            q = np.array([sel[k][i] for k in lib.SEED_DT.names[:6]], np.float32)
            q[3:] *= sgn
            Tn, st, xco, *_ = T.trace(img, q)
            j = 2 * i + d_
            rows = min(Tn + 1, ni)
            assert Tg[j] == Tn and stop[j] == st and np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True), (j, Tg[j], Tn, stop[j], st)
            its += rows
    assert its > 100
    n1, l1, nt1 = c.replay(sel, Tg, xc)
    n2, l2, nt2, _ = c.trace_replay(sel)
    assert nt1 == nt2 and np.array_equal(l1, l2) and all(np.array_equal(n1[k], n2[k], equal_nan=True) for k in n1.dtype.names)
    xcm = np.stack([mat(xc[j]) for j in range(len(Tg))])
    no, lo, nto = orc.replay(L, np.stack([sel[k] for k in lib.SEED_DT.names], -1).astype(np.float32), Tg.astype(np.int32), xcm, ni, img.shape, 4, 1)
    assert len(no) == len(n1) and np.array_equal(lo, l1) and all(np.array_equal(n1[k], no[k], equal_nan=True) for k in n1.dtype.names)
    tg, pg = lib.reconstruct(n1, l1)
    to, po = orc.reconstruct(L, n1, l1)
    assert np.array_equal(pg, po) and all(np.array_equal(tg[k], to[k], equal_nan=True) for k in to.dtype.names)


def test_config4_full_size_properties():
    import torch
    w, h, l = 2048, 2048, 512
    vol = torch.zeros((l, h, w), dtype=torch.uint8, device="cuda")
    # four 1024 x 1024 x 256 synthetic quadrants with different seeds (anisotropic tubes, zdist 4)
    for qi, (z0, y0, x0) in enumerate([(0, 0, 0), (256, 1024, 1024), (0, 1024, 0), (256, 0, 1024)]):
        vol[z0:z0 + 256, y0:y0 + 1024, x0:x0 + 1024] = synth.synth_torch(1024, 1024, 256, seed=5 + qi, zdist=4.0)
    torch.cuda.synchronize()
    p = pnr_amd.make_params(sigmas=(2, 4, 6, 8), np_=500, ni=200, zdist=4)
    c = pnr_amd.Context(p, 0)
    c.set_volume_device(vol.data_ptr(), (l, h, w), keepalive=vol)
    jmin, jmax = c.frangi()
    assert jmin == 0 and 0 < jmax < 1
    s0 = c.extract_seeds()
    assert s0["z"].max() > 255 and s0["y"].max() > 1024 and s0["x"].max() > 1024, "seeds must appear in the far quadrants (voxel index > 2^30)"
    s = c.score_filter_sort(s0)
    assert len(s) > 1000 and np.all(np.diff(s["corr"]) <= 0)
    for name, sb in (("best", s[:80]), ("far", s[s["z"] >= 300][:50])):
        assert len(sb) > 20, name
        T, stop, xc, _ = c.trace_batch(sb)
        n1, l1, nt1 = c.replay(sb, T, xc)
        n2, l2, nt2, iters = c.trace_replay(sb)
        assert nt1 == nt2 and np.array_equal(l1, l2) and all(np.array_equal(n1[k], n2[k], equal_nan=True) for k in n1.dtype.names), name
        assert iters <= int((T + (T < p.ni)).sum()) and len(n1) > 100
        if name == "far":
            far = (n1["z"][1:].astype(np.float64) * w * h + n1["y"][1:] * w + n1["x"][1:]) > 2 ** 30
            assert far.sum() > 0.9 * len(far)
    c.close()
    del vol
    torch.cuda.empty_cache()
