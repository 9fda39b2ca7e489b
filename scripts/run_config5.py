"""BASELINE configs[4] shape on one GPU: 2048x2048x512 anisotropic stack (N = 2^31 voxels: past the reference's
int indexing), scales {2,4,6,8}, zdist 4, np 500.  Checks 64-bit indexing end to end and prints stage times."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
w, h, l = 2048, 2048, 512
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.time()
vol = torch.zeros((l, h, w), dtype=torch.uint8, device='cuda')
# four 1024x1024x256 synthetic quadrants with different seeds (generation of the full stack in one piece is slow)
for qi, (z0, y0, x0) in enumerate([(0, 0, 0), (256, 1024, 1024), (0, 1024, 0), (256, 0, 1024)]):
    vol[z0:z0 + 256, y0:y0 + 1024, x0:x0 + 1024] = synth.synth_torch(1024, 1024, 256, seed=5 + qi, zdist=4.0)
torch.cuda.synchronize(); print('synth', round(time.time() - t0, 2), 's', flush=True)
p = pnr_amd.make_params(sigmas=(2, 4, 6, 8), np_=500, ni=200, zdist=4)
c = pnr_amd.Context(p, 0)
c.set_volume_device(vol.data_ptr(), (l, h, w), keepalive=vol)
c.set_profiling(True)
t = [time.time()]
print('frangi', c.frangi()); t.append(time.time())
s0 = c.extract_seeds(); t.append(time.time())
s = c.score_filter_sort(s0); t.append(time.time())
assert s0['z'].max() > 255 and s0['y'].max() > 1024 and s0['x'].max() > 1024, 'seeds must appear in the far quadrants (index > 2^30)'
nodes, links, nt, iters = c.trace_replay(s[:nseed]); t.append(time.time())
# size-independent property at this size: the streamed schedule gives the one-shot node graph (first 100 seeds)
sb = s[:min(100, nseed)]
Tb, stopb, xcb, _ = c.trace_batch(sb)
n1, l1, nt1 = c.replay(sb, Tb, xcb)
n2, l2, nt2, it2 = c.trace_replay(sb)
assert nt1 == nt2 and np.array_equal(l1, l2) and all(np.array_equal(n1[k], n2[k], equal_nan=True) for k in n1.dtype.names)
print('streamed == one-shot graph for', len(sb), 'seeds:', len(n1) - 1, 'nodes;', it2, 'vs', int((Tb + (Tb < p.ni)).sum()), 'iterations')
print('stage wall s:', np.diff(t).round(3).tolist(), 'seeds', len(s0), len(s), 'traced', nseed, 'iterations', iters, 'nodes', len(nodes))
far = (nodes['z'][1:].astype(np.float64) * w * h + nodes['y'][1:] * w + nodes['x'][1:]) > 2 ** 30
print('nodes beyond 2^30 voxels:', int(far.sum()), '| z of the traced seeds: min', float(s['z'][:nseed].min()), 'max', float(s['z'][:nseed].max()))
sf = s[s['z'] >= 300][:60]  # seeds whose voxel index is past 2^30: the tracer's 64-bit addressing (cube staging, density map, replay)
nf, lf, ntf, itf = c.trace_replay(sf)
farf = (nf['z'][1:].astype(np.float64) * w * h + nf['y'][1:] * w + nf['x'][1:]) > 2 ** 30
Tf, stf, xcf, _ = c.trace_batch(sf)
n1f, l1f, _ = c.replay(sf, Tf, xcf)
assert len(sf) > 0 and farf.sum() > 0.9 * len(farf) and np.array_equal(l1f, lf) and all(np.array_equal(n1f[k], nf[k], equal_nan=True) for k in nf.dtype.names)
print('far seeds:', len(sf), 'traced,', len(nf) - 1, 'nodes,', int(farf.sum()), 'beyond 2^30 voxels, streamed == one-shot')
for g in ('gauss', 'hessian_eigen', 'j8', 'seed_maxima', 'zncc', 'smc'):
    print('   ', g, c.kernel_ms(g))
