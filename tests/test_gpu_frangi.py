"""GPU parity: HIP Frangi path (through the C ABI) vs the oracle and the golden vectors from the
reference's own frangi.cpp.  Gaussian / Hessian / eigenvector bytes are expected bit-exact (same
IEEE operations in the same order, FMA contraction off); J may differ only through fp64 exp()
(ocml vs glibc), tolerance stated below; J8 / Vx / Vy / Vz bytes must be identical."""
import numpy as np
import pytest
import orc
import synth
import pnr_amd

pytestmark = pytest.mark.gpu
J_RTOL = 2e-6  # f32 J: at most ~1 f32 ulp (6e-8 rel) expected from a 1-ulp fp64 exp difference; bound stated generously


def ctx_for(sigs, zdist, **kw):
    return pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, zdist=zdist, **kw), 0)


def test_gaussian_bit_exact(golden):
    c = ctx_for(golden["sigs"], float(golden["zdist"]))
    c.set_volume(golden["img"])
    F = c.gaussian(float(golden["sigs"][0]))
    assert np.array_equal(F, golden["F_sig0"])


@pytest.mark.parametrize("w,sig", [(47, 4.0), (111, 4.0), (59, 8.0), (79, 2.0), (48, 4.0)])
def test_gaussian_row_ends_on_exactly_sized_device_buffer(oracle, w, sig):
    """widths at which the dword loads of the x pass's interior path end exactly at the row end (w % 32 = 15 for L = 12, 27 for
    L = 24, ...): the volume is a caller-owned device buffer of exactly w*h*l bytes (pnr_set_volume_device) -- no byte past its
    end may be read -- and F equals the oracle's"""
    import torch
    h, l = 9, 5
    img = synth.synth(w, h, l, seed=9)
    dev = torch.from_numpy(img.copy()).cuda()
    assert dev.numel() == w * h * l
    c = ctx_for([sig], 2.0)
    c.set_volume_device(dev.data_ptr(), img.shape, keepalive=dev)
    F = c.gaussian(sig)
    want = np.empty(img.shape, np.float32)
    oracle.orc_imgaussian3d(img, w, h, l, sig, 2.0, want)
    assert np.array_equal(F, want)
    c.close()


@pytest.mark.parametrize("sig", [2.0, 4.0, 6.0])
def test_gaussian_marching_xy_pass(oracle, sig):
    """gauss_xy_u8_m (the fused x-y pass marching down strips of 512 rows in chunks of 64) on a stack whose slices hold a second,
    partly filled strip (600 rows), an interior tile between two border tiles (150 columns at L = 6: x0 = 64) and a last chunk
    of 24 rows: F equals the oracle's and the tile kernel's (option gauss_march = 0), bit for bit"""
    w, h, l = 150, 600, 3
    img = synth.synth(w, h, l, seed=21)
    want = np.empty(img.shape, np.float32)
    oracle.orc_imgaussian3d(img, w, h, l, sig, 2.0, want)
    got = []
    for march in (1, 0):
        c = ctx_for([sig], 2.0)
        c.set_option("gauss_march", march)
        c.set_volume(img)
        got.append(c.gaussian(sig))
        c.close()
    assert np.array_equal(got[0], want)
    assert np.array_equal(got[1], want)


def test_device_eigen_solver_on_reference_kats():
    """Frangi::eigen_decomposition (frangi.cpp:1269-1493: tred2, tql2, the |lambda| re-sort) as compiled for the DEVICE, fed the 512
    known-answer matrices the reference itself produced (tests/golden/eigen_kat.npz: zero, diagonal, repeated and near-repeated
    eigenvalues, random symmetric): eigenvalues and all three eigenvector columns -- column 0 is the axis the direction bytes and the
    trackPos / trackNeg order come from -- bit for bit including the solver-defined sign, in both forms the pipeline compiles:
    the eigenvalues-only form of the vesselness kernel (eigen_queue) and the full solver of the direction kernel (vdir_points)."""
    import os
    k = np.load(os.path.join(os.path.dirname(__file__), "golden", "eigen_kat.npz"))
    c = ctx_for([2.0], 2.0)
    V, d = c.eigen(k["A"], vectors=True)
    assert np.array_equal(d, k["d"])
    assert np.array_equal(V[:, :, 0], k["V"][:, :, 0])  # the axis column, sign included
    assert np.array_equal(V, k["V"])
    _, d2 = c.eigen(k["A"], vectors=False)
    assert np.array_equal(d2, k["d"])
    # the same through a batch that is not a multiple of the work-group size, in another order
    idx = np.random.default_rng(3).permutation(len(k["A"]))[:301]
    V3, d3 = c.eigen(k["A"][idx], vectors=True)
    assert np.array_equal(d3, k["d"][idx]) and np.array_equal(V3, k["V"][idx])
    c.close()


def test_hessian_bit_exact(golden):
    c = ctx_for(golden["sigs"], float(golden["zdist"]))
    c.set_volume(golden["img"])
    H = c.hessian(float(golden["sigs"][0]))
    for k, v in H.items():
        assert np.array_equal(v, golden[k]), k


def test_frangi_vs_golden(golden):
    c = ctx_for(golden["sigs"], float(golden["zdist"]))
    c.set_volume(golden["img"])
    jmin, jmax = c.frangi()
    # J8 and the extremes as the pipeline uses them: from the run that skips the solver below the first J8 level (option
    # frangi_prune); asking for J or V afterwards recomputes without that shortcut
    fast8 = c.get_frangi(J=False, J8=True, V=False)["J8"]
    assert np.array_equal(fast8, golden["J8_restated"])
    g = c.get_frangi()
    # the contract is J_RTOL (fp64 exp of the device library vs glibc, < 1 ulp each, before the f32 store); what is MEASURED on
    # gfx950 with this ROCm is identity on every golden voxel, and that is what is asserted for these fixed inputs
    assert np.array_equal(g["J"], golden["J"])
    assert jmax == golden["Jmax"] and jmin == golden["Jmin"]
    for k in ("Vx", "Vy", "Vz"):
        assert np.array_equal(g[k], golden[k]), k
    assert np.array_equal(g["J8"], golden["J8_restated"])


@pytest.mark.parametrize("shape,sigs,zdist", [((33, 21, 9), [2.0], 2.0), ((20, 50, 14), [2.0, 3.0], 1.0), ((9, 9, 5), [2.0], 2.0),
                                                ((300, 37, 40), [2.0, 4.0], 2.0), ((70, 130, 35), [2.0, 4.0, 6.0, 8.0], 4.0)])
def test_frangi_vs_oracle_ragged(oracle, shape, sigs, zdist):
    """extents smaller than the kernel, not multiples of any tile, 4 scales / anisotropic"""
    w, h, l = shape
    img = synth.synth(w, h, l, seed=4)
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, sigs, zdist)
    c = ctx_for(sigs, zdist)
    c.set_volume(img)
    gmin, gmax = c.frangi()
    assert np.array_equal(c.get_frangi(J=False, J8=True, V=False)["J8"], orc.j8(oracle, J, jmin, jmax))  # the pruned run's J8
    g = c.get_frangi()
    assert np.allclose(g["J"], J, rtol=J_RTOL, atol=0)
    assert gmin == jmin and abs(gmax - jmax) <= J_RTOL * jmax
    assert np.array_equal(g["Vx"], Vx) and np.array_equal(g["Vy"], Vy) and np.array_equal(g["Vz"], Vz)
    assert np.array_equal(g["J8"], orc.j8(oracle, J, jmin, jmax))


def test_frangi_constant_volume():
    """flat image: every Hessian is zero, vesselness NaN->0, |Jmax-Jmin|<=FLT_MIN -> J8 all zero"""
    img = np.full((12, 20, 24), 37, np.uint8)
    c = ctx_for([2.0], 2.0)
    c.set_volume(img)
    jmin, jmax = c.frangi()
    g = c.get_frangi()
    assert jmin == 0 and jmax == 0 and g["J"].max() == 0 and g["J8"].max() == 0


def test_frangi_properties_larger():
    """size-independent properties at a larger size: (1) re-running gives identical bytes
    (idempotent state, no stale J from the previous run); (2) a y-mirrored input gives the mirrored
    J8 up to rounding of the (order-dependent) f32 tap sums: |diff| <= 1 level on < 1% of voxels;
    (3) the response is on the tubes: maximum 255, mostly-zero background."""
    img = synth.synth(160, 96, 48, seed=6)
    c = ctx_for([2.0, 4.0], 2.0)
    c.set_volume(img)
    c.frangi()
    a = c.get_frangi()
    c.frangi()
    a2 = c.get_frangi()
    for k in a:
        assert np.array_equal(a[k], a2[k]), k
    c.set_volume(np.ascontiguousarray(img[:, ::-1, :]))
    c.frangi()
    b = c.get_frangi()
    d = np.abs(a["J8"].astype(int) - b["J8"][:, ::-1, :].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.01
    assert a["J8"].max() == 255 and (a["J8"] > 0).mean() < 0.5


def test_j8_shortcut_and_foreign_extremes(oracle):
    """the solver is skipped where the response cannot reach J8 = 1 OF THE RUN'S OWN extremes (option frangi_prune); quantising with
    other extremes -- a smaller Jmax, a Jmin above 0 -- needs the exact response and gets it; the option off gives the same bytes"""
    img = synth.synth(96, 80, 40, seed=7)
    sigs, zdist = [2.0, 4.0], 2.0
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, sigs, zdist)
    c = ctx_for(sigs, zdist)
    c.set_volume(img)
    assert c.frangi() == (jmin, jmax) and jmin == 0.0
    fast = c.get_frangi(J=False, J8=True, V=False)["J8"]
    assert np.array_equal(fast, orc.j8(oracle, J, jmin, jmax)) and (fast > 0).mean() < 0.5
    for lo, hi in ((0.0, jmax / 7), (jmax / 300, jmax), (0.0, jmax * 3)):
        c.frangi()
        c.quantise_j8(lo, hi)
        assert np.array_equal(c.get_frangi(J=False, J8=True, V=False)["J8"], orc.j8(oracle, J, np.float32(lo), np.float32(hi))), (lo, hi)
    c.set_option("frangi_prune", 0)
    assert c.frangi() == (jmin, jmax)
    assert np.array_equal(c.get_frangi(J=False, J8=True, V=False)["J8"], fast)
    c.close()
