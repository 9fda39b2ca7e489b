"""GPU parity: tracker tables, znccBBB, the SMC trace kernel and the end-to-end path vs the oracle.
The device code performs the reference's scalar IEEE operations in its order (sequential sums per
chain, FMA off, libm-identical expf), so floats are expected bit-identical; the asserted tolerance
for particle weights / estimates is nevertheless the stated fp32 bound below, while indices (T,
stop, resampling indices, node links) must be exact."""
import numpy as np
import pytest
import orc
import synth
import pnr_amd
from pnr_amd import lib

pytestmark = pytest.mark.gpu
# Contract of the particle filter: BIT IDENTITY with the oracle -- every IEEE operation of the reference's scalar loops in its order
# (DESIGN.md 2): particle states, weights, N_eff, estimates, resampling indices and stop reasons are compared with array_equal.


@pytest.fixture(autouse=True, params=["phased", "persistent"])
def smc_driver(request, monkeypatch):
    """every test of this module runs with both schedulers of the particle filter (include/pnr_hip.h:
    pnr_set_smc_driver); a new Context takes its initial driver from pnr_amd.lib.DEFAULTS"""
    monkeypatch.setitem(lib.DEFAULTS, "smc_driver", request.param)
    return request.param


def mat(a):
    return np.stack([a[k] for k in a.dtype.names], -1)


def test_tables_bit_exact(oracle):
    for sigs, step, kappa, zdist in (([2.0], 2, 3.0, 2.0), ([2.0, 4.0, 6.0], 2, 3.0, 2.0), ([2.0, 3.0], 3, 2.0, 2.0),
                                      ([2.0, 4.0, 6.0, 8.0], 2, 3.0, 4.0), ([2.5, 5.0], 1, 0.5, 1.0)):
        T = orc.Tracker(oracle, sigs, step, 50, 5, kappa, 0.3, zdist=zdist, rng_seed=7)
        c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, step=step, kappa=kappa, zdist=zdist, np_=50, ni=5, rng_seed=7), 0)
        for nm in ("p", "u", "w0", "w0_cws", "v", "w", "w_cws"):
            assert np.array_equal(c.table(nm), T.table(nm).ravel()), nm
        assert np.array_equal(c.table("rng"), T.rng())
        for s in range(len(sigs)):
            vuw, wgt, avg = T.model(s)
            assert np.array_equal(c.table(f"model_vuw{s}"), vuw.ravel())
            assert np.array_equal(c.table(f"model_wgt{s}"), wgt)
            assert c.table("model_avg")[s] == np.float32(avg)


def test_expf_matches_libm():
    c = pnr_amd.Context(pnr_amd.make_params(), 0)
    rs = np.random.RandomState(3)
    x = np.concatenate([rs.uniform(-25, 25, 2_000_000), rs.uniform(-1, 1, 500_000) * 20, [0.0, -0.0, 20.0, -20.0, 1e-30]]).astype(np.float32)
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.expf.restype = C.c_float
    libm.expf.argtypes = [C.c_float]
    y = c.expf(x)
    idx = rs.choice(len(x), 200_000, replace=False)
    want = np.array([libm.expf(float(v)) for v in x[idx]], np.float32)
    assert np.array_equal(y[idx], want)
    assert np.array_equal(y[-5:], np.array([libm.expf(float(v)) for v in x[-5:]], np.float32))


@pytest.mark.parametrize("sigs,zdist", [([2.0], 2.0), ([2.0, 4.0, 6.0], 2.0)])
def test_zncc_vs_oracle(oracle, sigs, zdist):
    img = synth.synth(64, 56, 32, seed=2)
    T = orc.Tracker(oracle, sigs, 2, 20, 5, 3.0, 0.3, zdist=zdist)
    rs = np.random.RandomState(5)
    n = 300
    pos = rs.uniform(-3, 1, (n, 3)) + rs.uniform(0, 1, (n, 3)) * [66, 58, 34]  # includes out-of-volume poses (clamped)
    d = rs.randn(n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:5] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, -1, 0], [1e-5, 1e-5, 1]]  # degenerate xy projection branch
    pd = np.concatenate([pos, d], 1).astype(np.float32)
    want_c, want_s = T.zncc(img, pd)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, zdist=zdist), 0)
    c.set_volume(img)
    got_c, got_s = c.zncc(pd)
    assert np.array_equal(got_c, want_c)
    assert np.array_equal(got_s, want_s)
    assert c.zncc(np.zeros((0, 6), np.float32))[0].shape == (0,)


def test_zncc_in_batches(oracle):
    """pnr_zncc_batch scores at most 32 768 poses per pass over its stash: 70 001 poses (a ragged last group in the third batch) must give, pose by pose, the scores of the same poses evaluated a few hundred at a time -- and those the oracle's"""
    img = synth.synth(48, 40, 24, seed=4)
    sigs, zdist = [2.0, 3.0], 2.0
    rs = np.random.RandomState(9)
    m = 257
    pos = rs.uniform(-2, 1, (m, 3)) + rs.uniform(0, 1, (m, 3)) * [50, 42, 26]
    d = rs.randn(m, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    base = np.concatenate([pos, d], 1).astype(np.float32)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, zdist=zdist), 0)
    c.set_volume(img)
    small_c, small_s = c.zncc(base)
    T = orc.Tracker(oracle, sigs, 2, 20, 5, 3.0, 0.3, zdist=zdist)
    want_c, want_s = T.zncc(img, base)
    assert np.array_equal(small_c, want_c, equal_nan=True) and np.array_equal(small_s, want_s)
    n = 70001
    idx = rs.randint(0, m, n)
    big_c, big_s = c.zncc(base[idx])
    assert np.array_equal(big_c, small_c[idx], equal_nan=True) and np.array_equal(big_s, small_s[idx])
    c.close()


def _seeds_for(oracle, img, sigs, zdist, n):
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, sigs, zdist)
    s = orc.extract_seeds(oracle, 5, orc.j8(oracle, J, jmin, jmax), Vx, Vy, Vz)
    return s[:: max(1, len(s) // n)][:n]


@pytest.mark.parametrize("sigs,np_,ni,zdist", [([2.0], 50, 30, 2.0), ([2.0, 3.0], 64, 20, 2.0), ([2.0, 4.0], 37, 12, 1.0)])
def test_trace_vs_oracle(oracle, sigs, np_, ni, zdist):
    img = synth.synth(64, 56, 32, seed=2)
    so = _seeds_for(oracle, img, sigs, zdist, 6)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=zdist), 0)
    c.set_volume(img)
    T, stop, xc, dbg = c.trace_batch(seeds, dbg_iters=ni)
    nres = 0
    for i, sd in enumerate(so):
        for d, sgn in enumerate((1, -1)):
            q = sd[:6].copy(); q[3:] *= sgn
            Tn, st, xco, xf, idx, neff = To.trace(img, q, max_dbg=ni)
            j = 2 * i + d
            assert T[j] == Tn and stop[j] == st, (j, T[j], Tn, stop[j], st)
            rows = min(Tn + 1, ni)
            assert np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True)
            assert np.array_equal(dbg["xfilt"][j, :rows], xf[:rows], equal_nan=True)  # particle weights incl.
            assert np.array_equal(dbg["neff"][j, :rows], neff[:rows], equal_nan=True)
            res = (neff[:Tn] / np_ < 0.8)
            for it in np.nonzero(res)[0]:
                assert np.array_equal(dbg["idxres"][j, it], idx[it]), (j, it)
                nres += 1
    assert T.max() > 3 and nres > 0


def test_trace_nan_direction_and_border_seed(oracle):
    """NaN seed direction takes u[s] per particle (tracker.cpp:1019-1021); a seed at the volume
    corner exercises the clamp in interp and the out-of-volume stop."""
    img = synth.synth(48, 40, 24, seed=1)
    so = np.array([[24, 12, 11, np.nan, np.nan, np.nan, 0, 0], [0, 0, 0, 0.6, 0.8, 0, 0, 0], [46, 38, 22, 1, 0, 0, 0, 0]], np.float32)
    To = orc.Tracker(oracle, [2.0], 2, 40, 10, 3.0, 0.3, zdist=2.0)
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], np_=40, ni=10), 0)
    c.set_volume(img)
    T, stop, xc, _ = c.trace_batch(seeds)
    for i, sd in enumerate(so):
        for d, sgn in enumerate((1, -1)):
            q = sd[:6].copy(); q[3:] *= sgn
            Tn, st, xco, *_ = To.trace(img, q)
            j = 2 * i + d
            assert T[j] == Tn and stop[j] == st
            rows = min(Tn + 1, 10)
            assert np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True)


def test_trace_leaving_through_a_border(oracle):
    """bright lines running into the +y, +z and -x faces: the trace leaves the volume there, and the record of the failing
    iteration still carries the ZNCC of the out-of-volume centroid, whose samples are clamped onto the two outermost
    rows (Tracker::interp, tracker.cpp:2140-2178) -- rows the cube staging must keep"""
    img = synth.synth(48, 40, 24, seed=1, noise=6)
    img[9:12, :, 19:22] = 200      # along y, through both y faces
    img[:, 28:31, 33:36] = 200     # along z
    img[17:20, 6:9, :] = 200       # along x
    so = np.array([[20, 34, 10, 0, 1, 0, 0, 0], [34, 29, 19, 0, 0, 1, 0, 0], [5, 7, 18, -1, 0, 0, 0, 0], [20, 3, 10, 0, -1, 0, 0, 0]], np.float32)
    To = orc.Tracker(oracle, [2.0], 2, 40, 12, 3.0, 0.3, zdist=1.0)
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], np_=40, ni=12, zdist=1.0), 0)
    c.set_volume(img)
    T, stop, xc, _ = c.trace_batch(seeds)
    left = 0
    for i, sd in enumerate(so):
        for d, sgn in enumerate((1, -1)):
            q = sd[:6].copy(); q[3:] *= sgn
            Tn, st, xco, *_ = To.trace(img, q)
            j = 2 * i + d
            assert T[j] == Tn and stop[j] == st, (j, T[j], Tn, stop[j], st)
            rows = min(Tn + 1, 12)
            assert np.array_equal(mat(xc[j])[:rows], xco[:rows]), j
            left += int(st == 1)
    assert left >= 4  # the outward traces all end outside the volume


def test_end_to_end_vs_oracle(oracle):
    """BASELINE configs[0]-shaped plumbing case, reduced so the oracle finishes in seconds:
    Frangi -> J8 -> seeds -> score/filter/sort -> trace -> replay: seed list, trace lengths, node
    indices, node types and links identical to the oracle's; node floats within fp32 tolerance."""
    img = synth.synth(64, 56, 32, seed=2)
    sigs, np_, ni, zdist = [2.0], 50, 40, 2.0
    p = pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=zdist, nodepervol=4, vol=5)
    c = pnr_amd.Context(p, 0)
    res = pnr_amd.advantra.run_pipeline(c, img, one_shot=True)
    # oracle pipeline
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, sigs, zdist)
    s = orc.extract_seeds(oracle, 5, orc.j8(oracle, J, jmin, jmax), Vx, Vy, Vz)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    corr, _ = To.zncc(img, s[:, :6])
    s[:, 7] = corr
    s = s[corr >= 0.3]
    s = s[np.argsort(-s[:, 7], kind="stable")]
    assert np.array_equal(mat(res["seeds"]), s)
    Ts, xcs = [], []
    for sd in s:
        for sgn in (1, -1):
            q = sd[:6].copy(); q[3:] *= sgn
            Tn, st, xc, *_ = To.trace(img, q)
            Ts.append(Tn); xcs.append(xc)
    Ts = np.array(Ts, np.int32)
    assert np.array_equal(res["T"], Ts)
    nodes_o, links_o, nt = orc.replay(oracle, s, Ts, np.stack(xcs), ni, img.shape, 4, 5)
    assert len(res["nodes"]) == len(nodes_o) and res["ntraces"] == nt
    assert np.array_equal(res["links"], links_o) and np.array_equal(res["nodes"]["type"], nodes_o["type"])
    for k in ("x", "y", "z", "vx", "vy", "vz", "corr", "sig"):
        assert np.array_equal(res["nodes"][k], nodes_o[k], equal_nan=True), k
    assert len(nodes_o) > 50


@pytest.mark.parametrize("first_batch,vol,npv", [(1, 1, 4), (3, 5, 3), (64, 1, 4), (1000, 27, 4)])
def test_batched_trace_replay_equals_one_shot(first_batch, vol, npv):
    """pnr_trace_replay (seed-rank batches, stale density map on the GPU, early DENSITY stops, saturated
    seeds not launched) must give exactly the node graph of tracing everything + one replay."""
    img = synth.synth(96, 80, 40, seed=7)
    p = pnr_amd.make_params(sigmas=[2.0, 3.0], np_=48, ni=60, zdist=2.0, nodepervol=npv, vol=vol)
    c = pnr_amd.Context(p, 0)
    c.set_volume(img)
    c.frangi()
    seeds = c.score_filter_sort(c.extract_seeds())
    assert len(seeds) > 40
    T, stop, xc, _ = c.trace_batch(seeds)
    n1, l1, nt1 = c.replay(seeds, T, xc)
    n2, l2, nt2, iters = c.trace_replay(seeds, first_batch=first_batch)
    assert nt1 == nt2 and len(n1) == len(n2) > 100
    for k in n1.dtype.names:
        assert np.array_equal(n1[k], n2[k], equal_nan=True), k
    assert np.array_equal(l1, l2)
    full = int((T + (T < p.ni)).sum())
    print(f"first_batch={first_batch}: iterations {iters} vs one-shot {full}")
    assert iters <= full


@pytest.mark.parametrize("window,look0,look_pct,poll,maxtr,groups",
                         [(2, 1, 0, 1, 0, 1), (6, 3, 50, 3, 0, 1), (4096, 4096, 100, 7, 0, 1), (64, 16, 100, 4, 12, 1),
                          (64, 16, 50, 4, 0, 2), (48, 8, 100, 2, 0, 3), (4096, 4096, 100, 3, 0, 4), (32, 16, 100, 4, 12, 2)])
def test_streaming_scheduler_edge_cases(monkeypatch, smc_driver, window, look0, look_pct, poll, maxtr, groups):
    """pnr_trace_replay_stream with a window of one seed, a lookahead of one seed, everything admitted at once, odd polling
    periods, the MAX_TRACE_COUNT stop (Advantra_plugin.cpp:2702) and the window split into trace groups that step
    concurrently on their own streams: always the one-shot node graph."""
    if smc_driver != "phased":
        pytest.skip("the streaming scheduler belongs to the phased driver")
    img = synth.synth(80, 64, 32, seed=4)
    kw = dict(max_trace_count=maxtr) if maxtr else {}
    p = pnr_amd.make_params(sigmas=[2.0], np_=32, ni=40, zdist=2.0, nodepervol=3, vol=5, **kw)
    c = pnr_amd.Context(p, 0)
    c.set_volume(img)
    c.frangi()
    seeds = c.score_filter_sort(c.extract_seeds())
    assert len(seeds) > 30
    T, stop, xc, _ = c.trace_batch(seeds)
    n1, l1, nt1 = c.replay(seeds, T, xc)
    for k, v in (("window", window), ("look0", look0), ("look_pct", look_pct), ("poll", poll), ("groups", groups)):
        c.set_option(k, v)
    n2, l2, nt2, iters = c.trace_replay(seeds)
    assert nt1 == nt2 and len(n1) == len(n2) > 20 and np.array_equal(l1, l2)
    for k in n1.dtype.names:
        assert np.array_equal(n1[k], n2[k], equal_nan=True), k
    if maxtr:
        assert nt2 == maxtr + 1  # the loop ends after the trace that exceeds the cap (:2702)


@pytest.mark.parametrize("groups,poll", [(1, 4), (2, 4), (2, 1)])
def test_tentative_replay_saves_iterations_not_results(smc_driver, groups, poll):
    """stream_sched.h, option `tentative`: traces that a tentative replay of everything recorded so far cuts are paused, and ended by
    the host once that verdict is final.  Crowded stack (thick tubes, many seeds on each): the graph is the one-shot graph with the
    option on and off, the option saves SMC iterations, and ends traces itself (the log of trace ends names every replayed trace)."""
    if smc_driver != "phased":
        pytest.skip("the streaming scheduler belongs to the phased driver")
    img = synth.synth(128, 112, 48, seed=8)
    p = pnr_amd.make_params(sigmas=[2.0, 3.0], np_=64, ni=60, zdist=2.0, nodepervol=3, vol=1)
    c = pnr_amd.Context(p, 0)
    c.set_volume(img)
    c.frangi()
    seeds = c.score_filter_sort(c.extract_seeds())[:240]
    assert len(seeds) > 100
    T, stop, xc, _ = c.trace_batch(seeds)
    n1, l1, nt1 = c.replay(seeds, T, xc)
    free_iters = int((T + (T < p.ni)).sum())
    c.set_option("groups", groups)
    c.set_option("poll", poll)
    c.set_option("trace_log", 1)
    its = {}
    for tent in (0, 1):
        c.set_option("tentative", tent)
        n2, l2, nt2, its[tent] = c.trace_replay(seeds)
        assert nt1 == nt2 and len(n1) == len(n2) > 500 and np.array_equal(l1, l2), tent
        for k in n1.dtype.names:
            assert np.array_equal(n1[k], n2[k], equal_nan=True), (tent, k)
        log = c.trace_log()
        assert len(log) == 2 * nt2  # one record per replayed trace, whoever ended it
    assert its[1] < its[0] <= free_iters, its
    print(f"iterations: map-free {free_iters}, streamed {its[0]}, with the tentative replay {its[1]}")
    c.close()


def test_large_sigma_templates_outside_the_cube(oracle):
    """sigma = 12: the templates reach 36 voxels sideways, far beyond the LDS cube -- most corner groups take the HBM fallback"""
    img = synth.synth(96, 80, 40, seed=7)
    sigs, np_, ni = [3.0, 12.0], 40, 5
    so = _seeds_for(oracle, img, [2.0], 2.0, 2)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.1, zdist=2.0)
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=2.0, znccth=0.1), 0)
    c.set_volume(img)
    T, stop, xc, dbg = c.trace_batch(seeds, dbg_iters=ni)
    for i, sd in enumerate(so):
        for d, sgn in enumerate((1, -1)):
            q = sd[:6].copy(); q[3:] *= sgn
            Tn, st, xco, xf, idx, neff = To.trace(img, q, max_dbg=ni)
            j = 2 * i + d
            assert T[j] == Tn and stop[j] == st
            rows = min(Tn + 1, ni)
            assert np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True)
            assert np.array_equal(dbg["xfilt"][j, :rows], xf[:rows], equal_nan=True)


def test_many_particles_vs_oracle(oracle, smc_driver):
    """np = 1200: 19 groups of 64 chains, more than 64 KB of LDS in the update step of the phased driver"""
    if smc_driver != "phased":
        pytest.skip("the persistent kernel keeps all particle state in LDS: np is limited there")
    img = synth.synth(48, 40, 24, seed=1)
    sigs, np_, ni = [2.0], 1200, 4
    so = _seeds_for(oracle, img, sigs, 2.0, 1)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=2.0)
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=2.0), 0)
    c.set_volume(img)
    T, stop, xc, dbg = c.trace_batch(seeds, dbg_iters=ni)
    for d, sgn in enumerate((1, -1)):
        q = so[0][:6].copy(); q[3:] *= sgn
        Tn, st, xco, xf, idx, neff = To.trace(img, q, max_dbg=ni)
        assert T[d] == Tn and stop[d] == st
        rows = min(Tn + 1, ni)
        assert np.array_equal(mat(xc[d])[:rows], xco[:rows], equal_nan=True)
        assert np.array_equal(dbg["xfilt"][d, :rows], xf[:rows], equal_nan=True)


def test_small_stash_budget_gives_same_traces(monkeypatch, smc_driver):
    """a stash budget that holds only a few traces: pnr_trace_batch runs in several waves, pnr_trace_replay with a narrower
    window than asked for -- identical results"""
    if smc_driver != "phased":
        pytest.skip("the stash budget belongs to the phased driver")
    img = synth.synth(80, 64, 32, seed=4)
    p = pnr_amd.make_params(sigmas=[2.0, 3.0], np_=32, ni=30, zdist=2.0)
    a = pnr_amd.Context(p, 0)
    a.set_volume(img)
    a.frangi()
    seeds = a.score_filter_sort(a.extract_seeds())[:40]
    Ta, sa, xa, _ = a.trace_batch(seeds)
    na, la, _, _ = a.trace_replay(seeds)
    b = pnr_amd.Context(p, 0)
    b.set_option("stash_mb", 4)  # room for a handful of traces
    b.set_volume(img)
    Tb, sb, xb, _ = b.trace_batch(seeds)
    nb, lb, _, _ = b.trace_replay(seeds)
    assert np.array_equal(Ta, Tb) and np.array_equal(sa, sb)
    for j in range(len(Ta)):
        rows = min(Ta[j] + 1, 30)
        assert np.array_equal(mat(xa[j])[:rows], mat(xb[j])[:rows])
    assert len(na) == len(nb) and np.array_equal(la, lb) and all(np.array_equal(na[k], nb[k]) for k in na.dtype.names)


def test_trace_config5_shape_vs_oracle(oracle):
    """BASELINE configs[4] parameter shape at a size the oracle finishes in seconds: 4 scales {2,4,6,8},
    zdist=4 (anisotropic), np=500: chains loop over the work-group twice, the LDS cube shrinks to 44^3."""
    img = synth.synth(72, 64, 24, seed=5, zdist=4.0)
    sigs, np_, ni, zdist = [2.0, 4.0, 6.0, 8.0], 500, 6, 4.0
    so = _seeds_for(oracle, img, sigs, zdist, 2)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=zdist), 0)
    c.set_volume(img)
    T, stop, xc, dbg = c.trace_batch(seeds, dbg_iters=ni)
    for i, sd in enumerate(so):
        for d, sgn in enumerate((1, -1)):
            q = sd[:6].copy(); q[3:] *= sgn
            Tn, st, xco, xf, idx, neff = To.trace(img, q, max_dbg=ni)
            j = 2 * i + d
            assert T[j] == Tn and stop[j] == st
            rows = min(Tn + 1, ni)
            assert np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True)
            assert np.array_equal(dbg["xfilt"][j, :rows], xf[:rows], equal_nan=True)


@pytest.mark.parametrize("S,seed,groups", [(512, 2, 1), (1024, 3, 2)])
def test_full_size_properties(monkeypatch, smc_driver, S, seed, groups):
    """BASELINE configs[1] and configs[2] sizes (512^3 / 1024^3 -- the bench stack --, scales {2,4,6}, np=200): properties
    that do not need the oracle -- streamed tracing (one or two trace groups) gives the one-shot node graph; links are pairs
    of valid nodes; every trace start is UNDEFINED(7), ends are END(6); seeds come out sorted by corr; J8 spans 0..255."""
    import torch
    if S > 512 and smc_driver != "phased":
        pytest.skip("once is enough at this size")
    img = synth.synth_torch(S, S, S, seed=seed)
    p = pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2)
    c = pnr_amd.Context(p, 0)
    c.set_option("groups", groups)
    c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
    jmin, jmax = c.frangi()
    assert jmin == 0 and 0 < jmax < 1
    s = c.score_filter_sort(c.extract_seeds())
    assert len(s) > 1000 and np.all(np.diff(s["corr"]) <= 0) and s["corr"].min() >= p.znccth
    sb = s[:160]
    T, stop, xc, _ = c.trace_batch(sb)
    n1, l1, nt1 = c.replay(sb, T, xc)
    n2, l2, nt2, iters = c.trace_replay(sb, first_batch=32)
    assert nt1 == nt2 and np.array_equal(l1, l2) and all(np.array_equal(n1[k], n2[k]) for k in n1.dtype.names)
    assert iters < int((T + (T < p.ni)).sum())
    assert l1.min() >= 1 and l1.max() < len(n1)  # (a DENSITY stop may link a node to itself, as in the reference)
    assert set(np.unique(n1["type"][1:])) <= {2, 6, 7} and (n1["type"][1:] == 7).sum() >= nt1
    assert np.all((n1["x"][1:] >= -0.5) & (n1["x"][1:] < S - 0.5)) and np.all(n1["corr"][1:] >= p.znccth)
    del img
    torch.cuda.empty_cache()


def test_trace_without_stash_matches(oracle, monkeypatch):
    """the in-lane two-pass fallback (no HBM stash slot) must give the same traces as the two-phase form"""
    img = synth.synth(64, 56, 32, seed=2)
    so = _seeds_for(oracle, img, [2.0, 4.0], 2.0, 5)
    seeds = np.zeros(len(so), lib.SEED_DT)
    for i, k in enumerate(lib.SEED_DT.names):
        seeds[k] = so[:, i]
    p = pnr_amd.make_params(sigmas=[2.0, 4.0], np_=70, ni=15, zdist=2.0)
    a = pnr_amd.Context(p, 0)
    a.set_volume(img)
    Ta, sa, xa, _ = a.trace_batch(seeds)
    b = pnr_amd.Context(p, 0)
    b.set_option("no_stash", 1)
    b.set_smc_driver("persistent")  # the fallback belongs to the persistent driver: also a cross-driver comparison
    b.set_volume(img)
    Tb, sb, xb, _ = b.trace_batch(seeds)
    assert np.array_equal(Ta, Tb) and np.array_equal(sa, sb) and Ta.max() > 3
    for j in range(len(Ta)):
        rows = min(Ta[j] + 1, 15)
        assert np.array_equal(mat(xa[j])[:rows], mat(xb[j])[:rows])


def test_end_to_end_tree_vs_oracle(oracle):
    """whole path incl. the reconstruct() chain: final tree list (what _Advantra.swc holds) equals the oracle's"""
    img = synth.synth(64, 56, 32, seed=2)
    sigs, np_, ni, zdist = [2.0], 50, 40, 2.0
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=zdist, nodepervol=4, vol=5), 0)
    res = pnr_amd.advantra.run_pipeline(c, img)
    want_n, want_p = orc.reconstruct(oracle, res["nodes"], res["links"])
    assert len(res["tree"]) == len(want_n) > 50 and np.array_equal(res["parent"], want_p)
    for k in want_n.dtype.names:
        assert np.array_equal(res["tree"][k], want_n[k], equal_nan=True), k


def test_empty_inputs_end_to_end():
    """flat stack: Frangi response 0 everywhere, J8 all zero, no seeds, no traces, empty tree; n=0 entry points"""
    img = np.full((16, 24, 24), 9, np.uint8)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], np_=20, ni=5), 0)
    res = pnr_amd.advantra.run_pipeline(c, img)
    assert len(res["seeds_init"]) == 0 and len(res["seeds"]) == 0 and len(res["nodes"]) == 1 and len(res["links"]) == 0
    assert len(res["tree"]) == 1 and res["ntraces"] == 0 and res["iters"] == 0
    T, stop, xc, _ = c.trace_batch(np.zeros(0, lib.SEED_DT))
    assert len(T) == 0 and xc.shape[0] == 0
    one = pnr_amd.advantra.run_pipeline(c, img, one_shot=True)
    assert len(one["nodes"]) == 1


@pytest.mark.parametrize("opts", ["sums_deep=0", "sums_deep=1", "target=0,look0=64,look_pct=50", "target=40,overfill=0", "target=24,concentrate=0,groups=2",
                                  "target=24,concentrate=1,groups=2,poll=2", "groups=1,target=500", "lag=0", "lag=3,poll=4,groups=2", "groups=1,lag=2,poll=3",
                                  "groups=1,lag=5,poll=6,target=30",
                                  # a host-ended trace's slot must not reach the OTHER trace group before its own group's control() has
                                  # run (advisor finding of round 3): lag = poll - 1 keeps the device stepping it as long as possible,
                                  # the small target and the concentration hand nearly every freed slot to group 0
                                  "target=16,concentrate=1,groups=2,poll=4,lag=3,window=48", "target=12,concentrate=0,groups=3,poll=3,lag=2,window=32"])
def test_scheduler_and_kernel_forms_same_graph(smc_driver, opts):
    """the admission rules of the streaming scheduler (running-trace target, overfill, concentration on one trace group), the lag of
    the host's view behind the device's steps and the two
    forms of the ordered sums (ph_sums<false>: two folded chunk buffers, ph_sums<true>: four buffers in turn) are performance choices:
    the node graph is the one-shot graph whichever is taken"""
    if smc_driver != "phased":
        pytest.skip("the streaming scheduler belongs to the phased driver")
    img = synth.synth(128, 112, 48, seed=8)
    p = pnr_amd.make_params(sigmas=[2.0, 3.0], np_=64, ni=60, zdist=2.0, nodepervol=3, vol=1)
    c = pnr_amd.Context(p, 0)
    c.set_volume(img)
    c.frangi()
    seeds = c.score_filter_sort(c.extract_seeds())[:240]
    T, stop, xc, _ = c.trace_batch(seeds)
    n1, l1, nt1 = c.replay(seeds, T, xc)
    c.set_options(opts)
    n2, l2, nt2, its = c.trace_replay(seeds)
    assert nt1 == nt2 and len(n1) == len(n2) > 500 and np.array_equal(l1, l2), opts
    for k in n1.dtype.names:
        assert np.array_equal(n1[k], n2[k], equal_nan=True), (opts, k)
    assert its <= int((T + (T < p.ni)).sum())
    c.close()

