"""Parity at the production parameters (scales {2,4,6}, np=200, ni=200, zdist=2) on the reference's own CPU-runnable size
(BASELINE configs[0], 128x128x64): every trace of the first seeds, to its full depth, HIP path against the oracle, byte for byte;
then the node graph and the tree.  The oracle needs ~25 ms per SMC iteration, so this is minutes of CPU.  usage: [nseeds]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, orc, synth, pnr_amd
from pnr_amd import lib
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = orc.load_oracle()
mat = lambda a: np.stack([a[k] for k in a.dtype.names], -1)
sigs, np_, ni, zdist = [2.0, 4.0, 6.0], 200, 200, 2.0
img = synth.synth(128, 128, 64, seed=1)
c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=zdist), 0)
c.set_volume(img)
t0 = time.time()
c.frangi()
g = c.get_frangi(J=True, J8=True, V=True)
J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(L, img, sigs, zdist)
J8 = orc.j8(L, J, jmin, jmax)
for k, want in (("J", J), ("J8", J8), ("Vx", Vx), ("Vy", Vy), ("Vz", Vz)):
    assert np.array_equal(g[k].reshape(want.shape), want), k
so = orc.extract_seeds(L, 5, J8, Vx, Vy, Vz)
sg = c.extract_seeds()
assert np.array_equal(mat(sg)[:, :6], so[:, :6])
T = orc.Tracker(L, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
corr, _ = T.zncc(img, so[:, :6])
ss = c.score_filter_sort(sg)
keep = corr >= np.float32(0.3)
assert np.array_equal(ss["corr"], corr[keep][np.argsort(-corr[keep], kind="stable")])
print(f"frangi / J8 / V / {len(so)} seeds / {len(ss)} scores identical ({time.time() - t0:.1f} s)", flush=True)
sel = ss[:nseeds]
Tg, stop, xc, _ = c.trace_batch(sel)
its = 0
for i in range(len(sel)):
    for d_, sgn in enumerate((1, -1)):
        q = np.array([sel[k][i] for k in lib.SEED_DT.names[:6]], np.float32); q[3:] *= sgn
        Tn, st, xco, *_ = T.trace(img, q)
        j = 2 * i + d_
        rows = min(Tn + 1, ni)
        assert Tg[j] == Tn and stop[j] == st and np.array_equal(mat(xc[j])[:rows], xco[:rows], equal_nan=True), (j, Tg[j], Tn, stop[j], st)
        its += rows
        print(f"trace {j}: T={Tn} stop={st} identical  ({its} iterations so far, {time.time() - t0:.0f} s)", flush=True)
n1, l1, nt1 = c.replay(sel, Tg, xc)
n2, l2, nt2, _ = c.trace_replay(sel)
assert nt1 == nt2 and np.array_equal(l1, l2) and all(np.array_equal(n1[k], n2[k], equal_nan=True) for k in n1.dtype.names)
xcm = np.stack([mat(xc[j]) for j in range(len(Tg))])
no, lo, nto = orc.replay(L, np.stack([sel[k] for k in lib.SEED_DT.names], -1).astype(np.float32), Tg.astype(np.int32), xcm, ni, img.shape, 4, 1)
assert len(no) == len(n1) and np.array_equal(lo, l1) and all(np.array_equal(n1[k], no[k], equal_nan=True) for k in n1.dtype.names)
tg, pg = lib.reconstruct(n1, l1)
to, po = orc.reconstruct(L, n1, l1)
assert np.array_equal(pg, po) and all(np.array_equal(tg[k], to[k], equal_nan=True) for k in to.dtype.names)
print(f"done: {2 * len(sel)} traces, {its} SMC iterations, {len(n1) - 1} nodes, {len(tg) - 1} tree nodes: all identical")
