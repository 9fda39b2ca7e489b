"""Diagnostic: per-phase shader-clock shares of the SMC kernel (needs `make -C pnr_amd/csrc stamps`;
run with PNR_LIB_DIAG=pnr_amd/libpnr_hip_stamps.so).  Never quote this build's run time."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nseed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sigs = tuple(float(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (2, 4, 6)
img = synth.synth_torch(S, S, S, seed=3)
p = pnr_amd.make_params(sigmas=sigs, np_=200, ni=200, zdist=2)
c = pnr_amd.Context(p, 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.frangi(); s = c.score_filter_sort(c.extract_seeds())[:nseed]
T, stop, xc, dbg = c.trace_batch(s, dbg_iters=16)
st = dbg['neff'].view(np.uint64).reshape(len(T), 8).astype(np.float64)
iters = (T + (T < p.ni)).astype(np.float64)
names = ['loop-top', 'P1 predict', 'box stage', 'P2 chains/B', 'P3 pick', 'P4 serial', 'P2 phase A', '-']
tot = st.sum()
for i, nm in enumerate(names[:7]):
    print(f"{nm:12s} share {st[:, i].sum() / tot:6.3f}   cycles/iter {st[:, i].sum() / iters.sum():12.0f}")
print('total cycles/iter', tot / iters.sum())
