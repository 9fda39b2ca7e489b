import sys, time, json
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nseed = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sigs = tuple(float(x) for x in sys.argv[3].split(',')) if len(sys.argv) > 3 else (2, 4, 6)
t0 = time.time(); img = synth.synth_torch(S, S, S, seed=3); torch.cuda.synchronize(); print('synth', time.time() - t0, flush=True)
if S <= 128:
    ref = synth.synth(S, S, S, seed=3); print('synth torch==numpy frac', (img.cpu().numpy() == ref).mean())
p = pnr_amd.make_params(sigmas=sigs, np_=200, ni=200, zdist=2)
Mtot = {2.0: 845, 4.0: 5625, 6.0: 5625, 8.0: 5625}
nchain = sum(Mtot.get(float(x), 5625) for x in sigs)
c = pnr_amd.Context(p, 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.set_profiling(True)
for rep in range(int(os.environ.get('REPS', '2'))):
    c.reset_kernel_ms()
    t = [time.time()]
    print(c.frangi()); t.append(time.time())
    s0 = c.extract_seeds(); t.append(time.time())
    s = c.score_filter_sort(s0); t.append(time.time())
    sb = s[:nseed]
    T, stop, xc, _ = c.trace_batch(sb); t.append(time.time())
    nodes, links, nt = c.replay(sb, T, xc); t.append(time.time())
    print('rep', rep, 'wall', np.diff(t).round(4).tolist(), 'seeds', len(s0), len(s), 'T sum', int(T.sum()), 'mean', float(T.mean()), 'nodes', len(nodes), flush=True)
    for g in ('gauss', 'hessian_eigen', 'j8', 'seed_maxima', 'zncc', 'smc'):
        print('   ', g, c.kernel_ms(g))
    steps = int((T + (T < p.ni)).sum())  # iterations executed incl. the failing one
    ms, _ = c.kernel_ms('smc')
    for fb in ([int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [0]):
        t0 = time.time(); n2, l2, nt2, it2 = c.trace_replay(s[:nseed], first_batch=fb); t1 = time.time()
        print('   batched trace_replay first_batch', fb, ': wall', round(t1 - t0, 4), 'iterations', it2, 'nodes', len(n2), 'same graph', len(n2) == len(nodes) and np.array_equal(l2, links))
    t0 = time.time(); tree, par = pnr_amd.lib.reconstruct(nodes, links); print('   reconstruct: wall', round(time.time() - t0, 3), 's;', len(nodes), 'trace nodes ->', len(tree), 'tree nodes,', int((par[1:] == -1).sum()), 'trees')
    print('   trace-iterations', steps, 'evals', steps * 201, 'Mevals/s', steps * 201 / ms / 1e3, 'ms/iter/trace-avg', ms / max(steps, 1), 'Gsamples/s', steps * 201 * nchain / ms / 1e6, 'kernel ms / longest trace iters', ms / max(1, int(T.max()) + 1))
