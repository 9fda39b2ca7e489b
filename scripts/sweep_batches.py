"""batch-schedule sweep of pnr_trace_replay on the bench workload: first batch, growth, largest batch.
usage: sweep_batches.py [size] [nseeds] "fb:growth_pct:max,..." """
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nseed = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cfgs = [tuple(int(v) for v in x.split(':')) for x in (sys.argv[3] if len(sys.argv) > 3 else "128:200:1024").split(',')]
img = synth.synth_torch(S, S, S, seed=3); torch.cuda.synchronize()
p = pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2)
c = pnr_amd.Context(p, 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.frangi()
s = c.score_filter_sort(c.extract_seeds())[:nseed]
c.set_profiling(True)
ref = None
for rep in range(2):
    for fb, gr, mx in cfgs:
        c.set_option("replay_batches", 1); c.set_option("batch_growth", gr); c.set_option("batch_max", mx)
        c.reset_kernel_ms()
        t0 = time.time(); n2, l2, nt2, it2 = c.trace_replay(s, first_batch=fb); t1 = time.time()
        if ref is None: ref = (len(n2), l2.copy())
        same = ref[0] == len(n2) and np.array_equal(ref[1], l2)
        print(f"rep {rep} first {fb} growth {gr}% max {mx}: wall {t1 - t0:.3f} s iterations {it2} nodes {len(n2)} same {same} smc {c.kernel_ms('smc')} sums {c.kernel_ms('smc_sums')}", flush=True)
