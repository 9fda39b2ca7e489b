"""time the Frangi kernel groups on a bench-sized stack (run on the GPU box).  usage: frangi_bench.py [size]"""
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
img = synth.synth_torch(S, S, S, seed=3); torch.cuda.synchronize()
c = pnr_amd.Context(pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2), 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.set_profiling(True)
for rep in range(REPS):
    c.reset_kernel_ms()
    t0 = time.time(); jm = c.frangi(); t1 = time.time()
    print(f"rep {rep}: frangi wall {1e3 * (t1 - t0):.1f} ms Jmax {jm[1]:.6f}", {g: round(c.kernel_ms(g)[0], 1) for g in ("gauss", "hessian_tile", "hessian_eigen", "j8")}, flush=True)
if REPS < 3: sys.exit(0)
g = c.get_frangi(J=True, J8=False, V=True)
import zlib
print("checksums J", zlib.crc32(g["J"].tobytes()), "Vx", zlib.crc32(g["Vx"].tobytes()), "Vz", zlib.crc32(g["Vz"].tobytes()))
