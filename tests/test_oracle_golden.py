"""CPU: pin oracle/pnr_oracle.c against the golden vectors produced by the reference's own
frangi.cpp / seed.cpp (tests/golden/make_golden.py) and, when oracle/_ref is built, against
the reference live on fresh inputs.  Bit-exact: the restatement keeps the reference's
operation order and precision."""
import ctypes as C
import os
import numpy as np
import pytest
import orc
import synth


def test_gaussian_matches_reference(oracle, golden):
    img = golden["img"]; l, h, w = img.shape
    F = np.zeros(img.shape, np.float32)
    oracle.orc_imgaussian3d(img, w, h, l, float(golden["sigs"][0]), float(golden["zdist"]), F)
    assert np.array_equal(F, golden["F_sig0"])


def test_hessian_matches_reference(oracle, golden):
    img = golden["img"]; l, h, w = img.shape
    H = [np.zeros(img.shape, np.float32) for _ in range(6)]
    oracle.orc_hessian3d(golden["F_sig0"], w, h, l, float(golden["sigs"][0]), *H)
    for got, key in zip(H, ("Dzz", "Dyy", "Dyz", "Dxx", "Dxy", "Dxz")):
        assert np.array_equal(got, golden[key]), key


def test_eigen_kat(oracle):
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "eigen_kat.npz"))
    for A, V, d in zip(k["A"], k["V"], k["d"]):
        v = np.zeros((3, 3)); e = np.zeros(3)
        oracle.orc_eigen3(np.ascontiguousarray(A), v, e)
        assert np.array_equal(v, V) and np.array_equal(e, d)


def test_frangi_matches_reference(oracle, golden):
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, golden["img"], golden["sigs"], float(golden["zdist"]))
    assert np.array_equal(J, golden["J"])
    assert jmin == golden["Jmin"] and jmax == golden["Jmax"]
    assert np.array_equal(Vx, golden["Vx"]) and np.array_equal(Vy, golden["Vy"]) and np.array_equal(Vz, golden["Vz"])


def test_j8_rule(oracle, golden):
    J8 = orc.j8(oracle, golden["J"], float(golden["Jmin"]), float(golden["Jmax"]))
    assert np.array_equal(J8, golden["J8_restated"])
    # numpy restatement of Advantra_plugin.cpp:2499-2512 (f32 ratio, round half away, clamp)
    r = ((golden["J"] - golden["Jmin"]) / (golden["Jmax"] - golden["Jmin"]) * np.float32(255)).astype(np.float64)
    exp = np.clip(np.where(r > 0, np.floor(r + 0.5), np.ceil(r - 0.5)), 0, 255).astype(np.uint8)
    assert np.array_equal(J8, exp)
    flat = np.zeros(10, np.float32)
    assert orc.j8(oracle, flat, 0.0, 0.0).sum() == 0  # |Jmax-Jmin| <= FLT_MIN branch


def test_seeds_match_reference(oracle, golden):
    s = orc.extract_seeds(oracle, float(golden["tol"]), golden["J8_restated"], golden["Vx"], golden["Vy"], golden["Vz"])
    assert s.shape == golden["seeds"].shape and len(s) > 0
    assert np.array_equal(s, golden["seeds"])


@pytest.mark.parametrize("shape,sigs,zdist,tol", [((33, 21, 9), [2.0], 2.0, 5.0), ((20, 50, 14), [2.0, 3.0], 1.0, 2.0),
                                                     ((9, 9, 5), [2.0], 2.0, 5.0)])
def test_live_reference_ragged(oracle, ref, shape, sigs, zdist, tol):
    """ragged / smaller-than-kernel extents, live against the reference build"""
    if ref is None:
        pytest.skip("oracle/_ref not built on this box")
    w, h, l = shape
    img = synth.synth(w, h, l, seed=3)
    a = orc.frangi3d(oracle, img, sigs, zdist)
    b = orc.frangi3d(ref, img, sigs, zdist, prefix="ref")
    assert np.array_equal(a[0], b[0]) and a[1:3] == b[1:3]
    for i in (3, 4, 5):
        assert np.array_equal(a[i], b[i])
    J8 = orc.j8(oracle, a[0], a[1], a[2])
    sa = orc.extract_seeds(oracle, tol, J8, a[3], a[4], a[5])
    sb = orc.extract_seeds(ref, tol, J8, b[3], b[4], b[5], prefix="ref")
    assert np.array_equal(sa, sb)


def test_seeds_random_layers_live(oracle, ref):
    """MaximumFinder on random u8 layers: plateaus, ties and tolerance chains."""
    if ref is None:
        pytest.skip("oracle/_ref not built on this box")
    rs = np.random.RandomState(11)
    for tol in (0.0, 1.0, 5.0, 40.0):
        J8 = (rs.randint(0, 6, (4, 31, 37)) * rs.randint(0, 50, (4, 31, 37))).astype(np.uint8)
        J8[1] = 0  # empty layer: globalMax == globalMin
        J8[2] = np.clip(J8[2], 0, 3)
        V = [rs.randint(0, 256, J8.shape).astype(np.uint8) for _ in range(3)]
        sa = orc.extract_seeds(oracle, tol, J8, *V)
        sb = orc.extract_seeds(ref, tol, J8, *V, prefix="ref")
        assert np.array_equal(sa, sb, equal_nan=True), tol


def test_glibc_rand_stream(oracle):
    libc = C.CDLL("libc.so.6")
    for seed in (1, 42, 12345, 0):
        libc.srand(seed)
        want = np.array([libc.rand() for _ in range(300)], np.uint32)
        got = np.zeros(300, np.uint32)
        oracle.orc_glibc_rand(seed, 300, got)
        assert np.array_equal(got, want)
    got = np.zeros(4, np.uint32)
    oracle.orc_glibc_rand(42, 4, got)
    assert got.tolist() == [71876166, 708592740, 1483128881, 907283241]  # SURVEY.md section 5 (reference probe)


def test_tracker_table_shapes(oracle):
    """sizes recorded from the reference in SURVEY.md section 8: sz=256 (step 2), ndir=50,
    M_sigma = 845/5625/5625/5625 for sigma 2/4/6/8."""
    T = orc.Tracker(oracle, [2, 4, 6, 8], 2, 20, 5, 3.0, 0.3, zdist=2.0)
    assert T.sz == 256 and T.ndir == 50
    assert [T.model(s)[0].shape[0] for s in range(4)] == [845, 5625, 5625, 5625]
    assert abs(T.table("w0").sum() - 1) < 1e-5 and np.allclose(T.table("w").sum(1), 1, atol=1e-5)
    v = T.table("v")
    assert np.allclose((v * v).sum(1), 1, atol=1e-6)
    assert np.all(np.diff(T.table("w_cws"), axis=1) >= 0)


# ---- soma path: erosion and the u8 Gaussian are pinned on the reference's own frangi.cpp ----
@pytest.mark.parametrize("shape,rad", [((24, 40, 48), 3), ((5, 7, 9), 4), ((3, 33, 20), 2), ((1, 16, 16), 1)])
def test_imerode_imgaussian_u8_vs_reference(oracle, ref, shape, rad):
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(5)
    l, h, w = shape
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    img[:, h // 4: h // 2, w // 4: w // 2] = 200  # a plateau survives the erosion
    Eo, Er = np.zeros_like(img), np.zeros_like(img)
    oracle.orc_imerode_xy(img, w, h, l, float(rad), Eo)
    ref.ref_imerode_xy(img.copy(), w, h, l, float(rad), Er)
    assert np.array_equal(Eo, Er)
    Go, Gr = Eo.copy(), Er.copy()
    oracle.orc_imgaussian_u8_xy(Go, w, h, l, float(rad))
    ref.ref_imgaussian_u8_xy(Gr, w, h, l, float(rad))
    assert np.array_equal(Go, Gr) and Go.max() > 0


def test_maxentropy_and_conn3d_properties(oracle):
    """maxentropy_th / conn3d are parity-unpinned restatements (toolbox.cpp needs a Vaa3D header): check what the
    published algorithm guarantees -- threshold between two well separated modes, regions = 26-connected components
    numbered in raster order of their first voxel, centroid and mean radius of a ball."""
    rng = np.random.default_rng(2)
    a = np.concatenate([rng.integers(0, 20, 9000), rng.integers(180, 220, 1000)]).astype(np.uint8)
    th = oracle.orc_maxentropy_th(np.ascontiguousarray(a), len(a))
    assert 19 <= th < 180
    hist = np.bincount(a, minlength=256).astype(np.int64)
    assert oracle.orc_maxentropy_hist(hist) == th
    import scipy.ndimage as ndi
    vol = np.zeros((12, 20, 24), np.uint8)
    vol[2:5, 3:6, 4:9] = 255
    vol[5, 6, 9] = 255          # touches the first block by a corner only: same region with diagonal connectivity
    vol[8:11, 12:18, 2:5] = 255
    zz, yy, xx = np.meshgrid(np.arange(12), np.arange(20), np.arange(24), indexing="ij")
    vol[(xx - 17) ** 2 + (yy - 8) ** 2 + (zz - 6) ** 2 <= 9] = 255
    lab = np.zeros(vol.shape, np.int32)
    xc, yc, zc, rc = (np.zeros(16, np.float32) for _ in range(4))
    n = oracle.orc_conn3d(vol, 24, 20, 12, lab.reshape(-1), 1, 0, 1, xc, yc, zc, rc, 16)
    want, nw = ndi.label(vol > 0, structure=np.ones((3, 3, 3)))
    assert n == nw == 3
    first = [np.flatnonzero(lab.reshape(-1) == k)[0] for k in range(1, n + 1)]
    assert first == sorted(first)  # numbered in raster order
    for k in range(1, n + 1):
        m = lab == k
        assert len(np.unique(want[m])) == 1 and (want == want[m][0]).sum() == m.sum()
        assert np.allclose([xc[k - 1], yc[k - 1], zc[k - 1]], [xx[m].mean(), yy[m].mean(), zz[m].mean()], atol=1e-3)
    ball = int(lab[6, 8, 17])
    assert 1.5 < rc[ball - 1] < 3.0  # mean distance to the centre of a radius-3 ball = 3/4 * 3


# ---- 2-D mode: Frangi::frangi2d / hessian2d pinned on the reference's own frangi.cpp ----
@pytest.mark.parametrize("shape,sigs", [((1, 40, 48), [2.0]), ((1, 33, 21), [2.0, 3.0]), ((1, 64, 64), [1.0, 2.0, 4.0]), ((1, 7, 9), [2.0])])
def test_frangi2d_vs_reference(oracle, ref, shape, sigs):
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    _, h, w = shape
    img = synth.synth(w, h, 9, seed=4)[4:5].copy() if min(h, w) > 16 else np.random.default_rng(3).integers(0, 255, shape, dtype=np.uint8)
    for sg in sigs:
        a = [np.zeros(shape, np.float32) for _ in range(3)]
        b = [np.zeros(shape, np.float32) for _ in range(3)]
        oracle.orc_hessian2d(img, w, h, sg, *a)
        ref.ref_hessian2d(img.copy(), w, h, sg, *b)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    Jo, jmo, jMo, Vxo, Vyo, Vzo = orc.frangi2d(oracle, img, sigs)
    Jr, jmr, jMr, Vxr, Vyr, Vzr = orc.frangi2d(ref, img.copy(), sigs, prefix="ref")
    assert np.array_equal(Jo, Jr) and jmo == jmr and jMo == jMr
    assert np.array_equal(Vxo, Vxr) and np.array_equal(Vyo, Vyr) and np.array_equal(Vzo, Vzr) and not Vzo.any()
    if min(h, w) > 16:
        assert jMo > 0.05


# ---- committed fixtures of the reference's 2-D Frangi and soma filters (tests/golden/make_golden.py) ----
def _load(name):
    return dict(np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz")))


@pytest.mark.parametrize("name", ["p2d_96x80_s2-3", "p2d_33x21_s2"])
def test_oracle_2d_matches_golden(oracle, name):
    g = _load(name)
    img = g["img"]; _, h, w = img.shape
    D = [np.zeros(img.shape, np.float32) for _ in range(3)]
    oracle.orc_hessian2d(img, w, h, float(g["sigs"][0]), *D)
    for got, key in zip(D, ("Dyy", "Dxy", "Dxx")):
        assert np.array_equal(got, g[key]), key
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi2d(oracle, img, g["sigs"])
    assert np.array_equal(J, g["J"]) and jmin == g["Jmin"] and jmax == g["Jmax"]
    assert np.array_equal(Vx, g["Vx"]) and np.array_equal(Vy, g["Vy"]) and np.array_equal(Vz, g["Vz"])
    J8 = orc.j8(oracle, J, jmin, jmax)
    assert np.array_equal(J8, g["J8_restated"])
    s = orc.extract_seeds(oracle, float(g["tol"]), J8, Vx, Vy, Vz)
    assert np.array_equal(s, g["seeds"], equal_nan=True)


@pytest.mark.parametrize("name", ["soma_64x56x32_r3", "soma_23x20x9_r2"])
def test_oracle_soma_filters_match_golden(oracle, name):
    g = _load(name)
    img = g["img"]; l, h, w = img.shape
    E = np.zeros_like(img)
    oracle.orc_imerode_xy(img, w, h, l, float(g["rad"]), E)
    assert np.array_equal(E, g["eroded"])
    oracle.orc_imgaussian_u8_xy(E, w, h, l, float(g["rad"]))
    assert np.array_equal(E, g["blurred"])
