import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, orc, synth, pnr_amd
from pnr_amd import lib
oracle = orc.load_oracle()
vol = synth.synth(96, 80, 9, seed=4); img = np.ascontiguousarray(vol.max(0, keepdims=True))
sigs, np_, ni = [2.0], 40, 25
Jo, jmin, jmax, Vxo, Vyo, Vzo = orc.frangi2d(oracle, img, sigs)
so = orc.extract_seeds(oracle, 5, orc.j8(oracle, Jo, jmin, jmax), Vxo, Vyo, Vzo)
To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=2.0, is2d=True)
corr_o, sig_o = To.zncc(img, so[:, :6])
so = so[np.argsort(-corr_o, kind="stable")][:6]
seeds = np.zeros(len(so), lib.SEED_DT)
for i, k in enumerate(lib.SEED_DT.names): seeds[k] = so[:, i]
c = pnr_amd.Context(pnr_amd.make_params(sigmas=sigs, np_=np_, ni=ni, zdist=2.0), 0)
c.set_volume(img)
T, stop, xc, dbg = c.trace_batch(seeds, dbg_iters=ni)
mat = lambda a: np.stack([a[k] for k in a.dtype.names], -1)
for i, sd in enumerate(so):
    for d, sgn in enumerate((1, -1)):
        q = sd[:6].copy(); q[3:] *= sgn
        Tn, st, xco, xf, idx, neff = To.trace(img, q, max_dbg=ni)
        j = 2 * i + d
        rows = min(Tn + 1, ni)
        a = mat(xc[j])[:rows]; b = xco[:rows]
        bad = np.argwhere(~np.isclose(a, b, rtol=1e-5, atol=1e-6))
        xfbad = np.argwhere(~np.isclose(dbg["xfilt"][j, :rows], xf[:rows], rtol=1e-5, atol=1e-6))
        print("trace", j, "T", T[j], Tn, "stop", stop[j], st, "xc bad (row, col):", bad.tolist()[:6], "xfilt bad:", xfbad.tolist()[:4])
        for r, cc in bad[:3]:
            print("    row", r, "gpu", a[r], "orc", b[r], "recomputed zncc", To.zncc(img, b[r:r+1, :6]))
