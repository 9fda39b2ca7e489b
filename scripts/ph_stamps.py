"""Diagnostic: where a sampling work-group of the phased driver spends its cycles (needs `make -C pnr_amd/csrc stamps`; run with
PNR_LIB_DIAG=pnr_amd/libpnr_hip_stamps.so).  Never quote this build's run time.  usage: [PNR_STAMP_OPTS=target=40] ph_stamps.py [size] [nseeds]"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, synth, pnr_amd
from pnr_amd import lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nseed = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
img = synth.synth_torch(S, S, S, seed=3)
p = pnr_amd.make_params(sigmas=(2, 4, 6), np_=200, ni=200, zdist=2)
c = pnr_amd.Context(p, 0)
c.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
c.set_option("groups", 1)
c.set_options(os.environ.get("PNR_STAMP_OPTS"))  # e.g. target=40: the small launches of a sharded rank
c.frangi(); s = c.score_filter_sort(c.extract_seeds())[:nseed]
L = lib.load()
st = (C.c_ulonglong * 8)()
L.pnr_debug_ph_stamps(st, 1)
c.trace_replay(s)
L.pnr_debug_ph_stamps(st, 0)
stage, items, wgs, nit = st[0], st[1], st[2], st[3]
print(f"ph_sums, full groups of the longest template: pass 1 {st[5] / max(st[7], 1):.0f} cycles, pass 2 {st[6] / max(st[7], 1):.0f} cycles per wave ({st[7]} waves; s_memtime ticks)")
print(f"flags wait {st[4] / wgs:.0f} cycles; work-groups {wgs}, staging {stage / wgs:.0f} cycles per work-group, item loop of wave 0 {items / wgs:.0f} cycles ({nit / wgs:.2f} items), staging share {stage / (stage + items):.3f}")
