"""One case of the randomised parity campaign: a random small stack and random parameters, the HIP path against the oracle stage
by stage (Frangi J / J8 / V, seeds, seed scores, every trace iteration, streamed vs one-shot graph, replay, reconstruct, soma).
Every comparison is for equality of bytes.  Used by scripts/fuzz_parity.py (open-ended campaign on the GPU box) and by
tests/test_gpu_fuzz.py (a fixed set of cases inside `-m gpu`).  The oracle is the checker."""
import numpy as np
import orc
import synth
import pnr_amd
from pnr_amd import lib

mat = lambda a: np.stack([a[k] for k in a.dtype.names], -1)


def run_case(L, case, stats, desc, driver=None, big=False):
    """raises AssertionError on the first difference; fills `desc` with the case's parameters, adds its work to `stats`"""
    rs = np.random.RandomState(1000 + case)
    two_d = rs.rand() < 0.12 and driver != "persistent"  # (the persistent driver is 3-D only)
    big = 2 if big else 1  # stacks up to 192 x 160 x 80
    w, h, l = int(rs.randint(40, 97 * big)), int(rs.randint(32, 81 * big)), (1 if two_d else int(rs.randint(12, 41 * big)))
    nsig = int(rs.randint(1, 4))
    sigs = sorted(float(x) for x in rs.choice([1.5, 2.0, 2.5, 3.0, 4.0, 6.0], nsig, replace=False))
    zdist = float(rs.choice([1.0, 2.0, 3.0, 4.0]))
    np_ = int(rs.choice([20, 50, 63, 64, 100, 127, 200]))
    ni = int(rs.randint(5, 41)); step = int(rs.choice([2, 2, 3])); kappa = float(rs.choice([2.0, 3.0, 4.0]))
    tol = float(rs.choice([3, 5, 10])); znccth = float(rs.choice([0.2, 0.3, 0.5])); npv = int(rs.choice([3, 4, 6])); vol = int(rs.choice([1, 5]))
    seed_img = int(rs.randint(1, 10_000))
    if rs.rand() < 0.15:  # BASELINE configs[4]-like parameters at a small size: four scales up to 8 (templates wider than the LDS cube), np 300 / 500
        sigs = sorted(set(sigs) | {8.0}) if rs.rand() < 0.6 else [2.0, 4.0, 6.0, 8.0]
        np_ = int(rs.choice([300, 500])); ni = int(rs.randint(4, 13)); zdist = float(rs.choice([2.0, 4.0]))
    rad = int(rs.choice([2, 3, 4])) if (not two_d and rs.rand() < 0.3) else 0
    groups = int(rs.choice([1, 1, 2, 3]))
    desc.update(case=case, shape=(w, h, l), sigs=sigs, zdist=zdist, np=np_, ni=ni, step=step, kappa=kappa, tol=tol, znccth=znccth, npv=npv, vol=vol, img=seed_img, somaradius=rad, groups=groups)
    img = synth.synth(w, h, l, seed=seed_img) if not two_d else synth.synth(w, h, 3, seed=seed_img)[1:2].copy()
    if rad:
        img = synth.add_somas(img, [(int(rs.randint(8, w - 8)), int(rs.randint(8, h - 8)), int(rs.randint(4, max(5, l - 4))), int(rs.randint(rad + 1, rad + 5)))
                                    for _ in range(int(rs.randint(1, 3)))])
    knobs = dict(groups=groups, window=int(rs.choice([8, 32, 768])), look0=int(rs.choice([2, 16, 128])), poll=int(rs.choice([1, 4, 7])))
    rk = np.random.RandomState(900000 + case)  # (a stream of its own: the cases keep the stacks and parameters they had before these knobs existed)
    knobs.update(target=int(rk.choice([-1, 0, 4, 16, 64])), overfill=int(rk.randint(2)), concentrate=int(rk.randint(2)), sums_deep=int(rk.choice([-1, 0, 1])), lag=int(rk.choice([-1, 0, 1, 2])))
    desc.update(knobs=knobs)
    p = pnr_amd.make_params(sigmas=sigs, somaradius=rad, step=step, kappa=kappa, zdist=zdist, np_=np_, ni=ni, tolerance=tol, znccth=znccth, nodepervol=npv, vol=vol)
    c = pnr_amd.Context(p, 0)
    if driver:
        c.set_smc_driver(driver)
    for k, v in knobs.items():  # the scheduler knobs never change a result
        c.set_option(k, v)
    c.set_volume(img)
    smap, n4 = None, None
    if rad:  # soma extraction (Advantra_plugin.cpp:2426-2486)
        E8o, tho, smap, n4 = orc.soma_extract(L, img, rad)
        sm = c.soma(want_e8=True)
        fg = np.flatnonzero(smap.reshape(-1) > 0)
        assert np.array_equal(sm["E8"], E8o) and sm["threshold"] == tho and np.array_equal(sm["vox"], fg) and np.array_equal(sm["lab"], smap.reshape(-1)[fg]), "soma"
        stats["somas"] = stats.get("somas", 0) + len(n4)
    gext = c.frangi()
    fast8 = c.get_frangi(J=False, J8=True, V=False)["J8"]  # what the pipeline uses: the run that skips the solver below the first J8 level
    fast_seeds = c.extract_seeds()
    g = c.get_frangi(J=True, J8=True, V=True)              # asking for J / V recomputes without that shortcut
    if two_d:
        J, jmin, jmax, Vx, Vy, Vz = orc.frangi2d(L, img, sigs)
    else:
        J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, img, sigs, zdist)
    J8 = orc.j8(L, J, jmin, jmax)
    for k, want in (("J", J), ("J8", J8), ("Vx", Vx), ("Vy", Vy), ("Vz", Vz)):
        assert np.array_equal(g[k].reshape(want.shape), want), f"frangi {k}: {(g[k].reshape(want.shape) != want).sum()} voxels differ"
    assert np.array_equal(fast8.reshape(J8.shape), J8) and gext == (jmin, jmax), "J8 / extremes of the pruned run"
    so = orc.extract_seeds(L, tol, J8, Vx, Vy, Vz)
    sg = c.extract_seeds()
    assert len(fast_seeds) == len(sg) and all(np.array_equal(fast_seeds[k], sg[k], equal_nan=True) for k in sg.dtype.names), "seeds of the pruned run"
    assert len(sg) == len(so) and np.array_equal(mat(sg)[:, :6], so[:, :6]), "seeds"
    if rad and len(so):  # seeds inside a soma are dropped (Advantra_plugin.cpp:2561-2564)
        vx = np.round(so[:, 2]).astype(np.int64) * w * h + np.round(so[:, 1]).astype(np.int64) * w + np.round(so[:, 0]).astype(np.int64)
        so = so[smap.reshape(-1)[vx] == 0]
    T = orc.Tracker(L, sigs, step, np_, ni, kappa, znccth, zdist=zdist, nodespervol=npv, is2d=two_d)
    ss = c.score_filter_sort(sg)
    if len(so):
        corr, sig = T.zncc(img, so[:, :6])
        keep = corr >= np.float32(znccth)
        order = np.argsort(-corr[keep], kind="stable")
        assert len(ss) == keep.sum() and np.array_equal(ss["corr"], corr[keep][order]), "seed scores"
    sel = ss[: int(rs.randint(2, 9))]
    Tg, stop, xc, _ = c.trace_batch(sel)
    for i in range(len(sel)):
        for d_, sgn in enumerate((1, -1)):
            q = np.array([sel[k][i] for k in lib.SEED_DT.names[:6]], np.float32); q[3:] *= sgn
            Tn, st, xco, *_ = T.trace(img, q)
            j = 2 * i + d_
            assert Tg[j] == Tn and stop[j] == st, f"trace {j}: T {Tg[j]} vs {Tn}, stop {stop[j]} vs {st}"
            rows = min(Tn + 1, ni)
            assert np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True), f"trace {j}: xc differs"
            stats["iters"] = stats.get("iters", 0) + rows
        stats["traces"] = stats.get("traces", 0) + 2
    n1, l1, nt1 = c.replay(sel, Tg, xc)
    n2, l2, nt2, _ = c.trace_replay(sel)
    assert nt1 == nt2 and np.array_equal(l1, l2) and all(np.array_equal(n1[k], n2[k], equal_nan=True) for k in n1.dtype.names), "streamed vs one-shot graph"
    if case % 3 == 0 and driver != "persistent" and len(ss) >= 2:
        # BASELINE configs[3]: more of the sorted seeds dealt to 2 / 3 logical ranks (a context and a host thread per rank, joined by an
        # in-process all-gather) -- every rank must end with the one-GPU graph of the same seeds
        import threading
        from pnr_amd import multigpu
        world = 2 + (case // 3) % 2
        many = ss[:24]
        nr, lr, ntr, _ = c.trace_replay(many)
        X = multigpu.ThreadExchange(world)
        ctxs, res = [], [None] * world
        for r in range(world):
            cr = pnr_amd.Context(p, 0)
            for k, v in knobs.items():
                cr.set_option(k, v)
            cr.set_option("exchange_block", [0, 2048, 700][case // 3 % 3])  # small blocks: records queue up and are carried over
            cr.set_volume(img)
            if rad:
                cr.soma()
            ctxs.append(cr)

        def run_rank(r):
            try:
                res[r] = ctxs[r].trace_replay_sharded(many, r, world, X.callback(r))
            except Exception as e:  # noqa: BLE001
                res[r] = e
                X.barrier.abort()

        th = [threading.Thread(target=run_rank, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        for r in range(world):
            assert not isinstance(res[r], Exception), f"sharded rank {r}/{world}: {res[r]}"
            assert res[r][2] == ntr and np.array_equal(res[r][1], lr) and all(np.array_equal(res[r][0][k], nr[k], equal_nan=True) for k in nr.dtype.names), \
                f"sharded graph differs on rank {r} of {world}"
        for cr in ctxs:
            cr.close()
        stats["sharded"] = stats.get("sharded", 0) + 1
    # the sequential bookkeeping (trackPos, trace loop) and the reconstruct chain against the oracle's
    so_sel = np.stack([sel[k] for k in lib.SEED_DT.names], -1).astype(np.float32)
    xcm = np.stack([mat(xc[j]) for j in range(len(Tg))]) if len(Tg) else np.zeros((0, ni, 8), np.float32)
    no, lo, nto = orc.replay(L, so_sel, Tg.astype(np.int32), xcm, ni, img.shape, npv, vol, smap=smap, soma4=n4)
    assert len(no) == len(n1) and np.array_equal(lo, l1) and all(np.array_equal(n1[k], no[k], equal_nan=True) for k in n1.dtype.names), "replay vs oracle"
    tg, pg = lib.reconstruct(n1, l1)
    to, po = orc.reconstruct(L, n1, l1)
    assert np.array_equal(pg, po) and all(np.array_equal(tg[k], to[k], equal_nan=True) for k in to.dtype.names), "reconstruct vs oracle"
    stats["nodes"] = stats.get("nodes", 0) + len(n1) - 1
    stats["seeds"] = stats.get("seeds", 0) + len(so)
    stats["voxels"] = stats.get("voxels", 0) + img.size
