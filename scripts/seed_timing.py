import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S = 1024
img = synth.synth_torch(S, S, S, seed=3); torch.cuda.synchronize()
c = pnr_amd.Context(pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2), 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.frangi()
for rep in range(3):
    t0 = time.time(); s = c.extract_seeds(); print("extract_seeds", round(time.time() - t0, 4), len(s), flush=True)
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
