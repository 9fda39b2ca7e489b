"""Diagnostic: where a wave of hessian_tile spends a plane of its march (needs `make -C pnr_amd/csrc variant NAME=hts DEFS=-DPNR_HT_STAMPS`,
run with PNR_LIB_DIAG=$PWD/pnr_amd/libpnr_hip_hts.so).  Never quote this build's run time.  usage: ht_stamps.py [size]"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
from pnr_amd import lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img = synth.synth_torch(S, S, S, seed=3)
c = pnr_amd.Context(pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2), 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.frangi()
L = C.CDLL(lib.LIB_PATH)
st = (C.c_ulonglong * 8)()
L.pnr_debug_ht_stamps(st, 1)
c.frangi()
L.pnr_debug_ht_stamps(st, 0)
n = max(st[4], 1)
tot = sum(st[:4]) / n
print(f"hessian_tile, inner planes: {st[4]} wave-planes, {tot:.0f} cycles per wave-plane: stencil {st[0] / n:.0f} | tests + queue {st[1] / n:.0f} | wait for the prefetched plane + store {st[2] / n:.0f} | barrier {st[3] / n:.0f}")
