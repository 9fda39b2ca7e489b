"""time the soma path (somaradius > 0) on a bench-sized stack with a few cell bodies; run on the GPU box.
usage: soma_bench.py [size] [somaradius]"""
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rad = int(sys.argv[2]) if len(sys.argv) > 2 else 5
img = synth.synth_torch(S, S, S, seed=3)
rng = np.random.default_rng(1)
for k in range(12):  # solid bright balls, radius 10..20
    cx, cy, cz = (rng.integers(40, S - 40) for _ in range(3)); r = int(rng.integers(10, 21))
    z0, y0, x0 = cz - r - 1, cy - r - 1, cx - r - 1
    zz = torch.arange(z0, cz + r + 2, device="cuda", dtype=torch.float32)[:, None, None]
    yy = torch.arange(y0, cy + r + 2, device="cuda", dtype=torch.float32)[None, :, None]
    xx = torch.arange(x0, cx + r + 2, device="cuda", dtype=torch.float32)[None, None, :]
    d = torch.sqrt((xx - cx) ** 2 + (yy - cy) ** 2 + (zz - cz) ** 2)
    sub = img[z0:cz + r + 2, y0:cy + r + 2, x0:cx + r + 2]
    sub.copy_(torch.maximum(sub.float(), 230 * (r + 1.0 - d).clamp(0, 1)).floor().to(torch.uint8))
torch.cuda.synchronize()
p = pnr_amd.make_params(sigmas=(2, 4, 6), somaradius=rad, np_=200, ni=200, zdist=2)
c = pnr_amd.Context(p, 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.set_profiling(True)
for rep in range(3):
    c.reset_kernel_ms()
    t0 = time.time(); s = c.soma(); t1 = time.time()
    ms, n = c.kernel_ms("soma")
    N = S ** 3
    print(f"rep {rep}: pnr_soma wall {t1 - t0:.3f} s, kernels {ms:.2f} ms ({n} launches) = {13 * N / (ms * 1e-3) / 1e9:.0f} GB/s of 13 B/voxel compulsory traffic; "
          f"threshold {s['threshold']}, {len(s['nodes'])} soma nodes, {len(s['vox'])} foreground voxels", flush=True)
print("nodes (x, y, z, r):", np.stack([s['nodes'][k] for k in ('x', 'y', 'z', 'sig')], -1).round(1).tolist()[:12])
