"""dump the one-shot traces (every seed traced to its map-free end) of the bench workload for offline scheduling studies"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S, nseed = 1024, 2000
img = synth.synth_torch(S, S, S, seed=3); torch.cuda.synchronize()
p = pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2)
c = pnr_amd.Context(p, 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.frangi()
s = c.score_filter_sort(c.extract_seeds())[:nseed]
T, stop, xc, _ = c.trace_batch(s)
pos = np.stack([xc["x"], xc["y"], xc["z"]], -1).astype(np.float32)  # [2n][ni][3]
np.savez_compressed(os.path.join(R, "gpurun_out", "traces_1024_s2000.npz"), T=T, stop=stop, pos=pos.astype(np.float16) if False else pos,
                    seeds=np.stack([s["x"], s["y"], s["z"]], -1))
print("saved", T.shape, pos.shape, "sum T", int(T.sum()))
