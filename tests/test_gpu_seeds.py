"""GPU parity: extractSeeds through the C ABI (GPU candidate kernels + host flood-fill) vs the
golden vectors of the reference's seed.cpp and the oracle.  Integer/byte work: bit-exact."""
import numpy as np
import pytest
import orc
import synth
import pnr_amd
from pnr_amd import SeedExtractor

pytestmark = pytest.mark.gpu


def as_mat(s):
    return np.stack([s[k] for k in s.dtype.names], 1)


def test_seeds_vs_golden(golden):
    s = SeedExtractor.extractSeeds(float(golden["tol"]), golden["J8_restated"], golden["Vx"], golden["Vy"], golden["Vz"])
    assert np.array_equal(as_mat(s), golden["seeds"])


def test_seeds_random_layers(oracle):
    rs = np.random.RandomState(11)
    for tol in (0.0, 1.0, 5.0, 40.0):
        J8 = (rs.randint(0, 6, (5, 31, 37)) * rs.randint(0, 50, (5, 31, 37))).astype(np.uint8)
        J8[1] = 0          # empty layer
        J8[2] = np.clip(J8[2], 0, 3)
        J8[3] = 9          # flat non-zero layer
        V = [rs.randint(0, 256, J8.shape).astype(np.uint8) for _ in range(3)]
        want = orc.extract_seeds(oracle, tol, J8, *V)
        got = SeedExtractor.extractSeeds(tol, J8, *V)
        assert np.array_equal(as_mat(got), want, equal_nan=True), tol


def test_seeds_wide_layer_and_ranges(oracle):
    """w > 256 (several x tiles), and the layer-range entry point used for multi-GPU sharding"""
    img = synth.synth(300, 64, 12, seed=8)
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, [2.0], 2.0)
    J8 = orc.j8(oracle, J, jmin, jmax)
    want = orc.extract_seeds(oracle, 5, J8, Vx, Vy, Vz)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0]), 0)
    c.set_volume(img)
    c.set_j8_v(J8, Vx, Vy, Vz)
    assert np.array_equal(as_mat(c.extract_seeds()), want) and len(want) > 0
    parts = [as_mat(c.extract_seeds(z0, z1)) for z0, z1 in ((0, 5), (5, 5), (5, 12))]
    assert np.array_equal(np.concatenate(parts), want)


def test_seeds_require_frangi_first():
    c = pnr_amd.Context(pnr_amd.make_params(), 0)
    c.set_volume(np.zeros((4, 8, 8), np.uint8))
    with pytest.raises(pnr_amd.PnrError, match="pnr_frangi"):
        c.extract_seeds()


@pytest.mark.parametrize("world", [2, 3])
def test_zslab_sharding_equals_whole_stack(world):
    """Frangi + seeds of a stack cut into z-slabs with halo (pnr_frangi_slab / pnr_quantise_j8 / pnr_extract_seeds_range, one
    'rank' after the other on this GPU): same Jmin / Jmax, same seeds in the same order as the unsharded extraction"""
    import torch
    from pnr_amd import multigpu
    img = synth.synth(64, 56, 48, seed=3)
    p = pnr_amd.make_params(sigmas=[2.0, 4.0], zdist=2.0, np_=20, ni=5)
    c = pnr_amd.Context(p, 0)
    c.set_volume(img)
    jmin, jmax = c.frangi()
    want = c.extract_seeds()
    dimg = torch.from_numpy(img).cuda()
    # pass 1: every rank's extremes; pass 2: quantise with the reduced ones (what the all-reduce does)
    ext = []
    for r in range(world):
        ctx = pnr_amd.Context(p, 0)
        _, a, b = multigpu.frangi_seeds_sharded(ctx, dimg.data_ptr(), img.shape, None, r, world, reduce_fn=lambda a, b: (a, b))
        ext.append((a, b))
        ctx.close()
    gmin, gmax = min(e[0] for e in ext), max(e[1] for e in ext)
    assert gmin == jmin and gmax == jmax
    got = []
    for r in range(world):
        ctx = pnr_amd.Context(p, 0)
        s, _, _ = multigpu.frangi_seeds_sharded(ctx, dimg.data_ptr(), img.shape, None, r, world, reduce_fn=lambda a, b: (gmin, gmax))
        got.append(s)
        ctx.close()
    got = np.concatenate(got)
    assert len(got) == len(want) > 20
    for k in want.dtype.names:
        assert np.array_equal(got[k], want[k], equal_nan=True), k
