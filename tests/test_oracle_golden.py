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
