"""per-launch times of the phased SMC kernels with a handful of traces (the latency floors).  usage: sums_floor.py [nseeds] [stack edge]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
img = synth.synth_torch(S, S, S, seed=3); torch.cuda.synchronize()
c = pnr_amd.Context(pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2), 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.frangi()
s = c.score_filter_sort(c.extract_seeds())[:int(sys.argv[1]) if len(sys.argv) > 1 else 4]
c.set_profiling(True)
for rep in range(2):
    c.reset_kernel_ms()
    T, stop, xc, _ = c.trace_batch(s)
    print("traces", 2 * len(s), {g: (round(c.kernel_ms(g)[0] / max(c.kernel_ms(g)[1], 1), 4), c.kernel_ms(g)[1]) for g in ("smc", "smc_sums", "smc_predict", "smc_update")}, flush=True)
