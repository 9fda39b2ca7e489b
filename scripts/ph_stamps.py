"""Diagnostic: where a sampling work-group of the phased driver spends its cycles (needs `make -C pnr_amd/csrc stamps`; run with
PNR_LIB_DIAG=pnr_amd/libpnr_hip_stamps.so; through gpurun, which does not send that file: `make -C pnr_amd/csrc variant NAME=st DEFS=-DPNR_SMC_STAMPS`
and PNR_LIB_DIAG=$PWD/pnr_amd/libpnr_hip_st.so).  Never quote this build's run time.  usage: [PNR_STAMP_OPTS=target=40] ph_stamps.py [size] [nseeds]"""
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
pu = (C.c_ulonglong * 16)()
L.pnr_debug_pu_stamps(pu, 1)
fine = (C.c_ulonglong * 8)()
L.pnr_debug_pu_fine(fine, 1)
c.trace_replay(s)
L.pnr_debug_ph_stamps(st, 0)
L.pnr_debug_pu_stamps(pu, 0)
L.pnr_debug_pu_fine(fine, 0)
stage, items, wgs, nit = st[0], st[1], st[2], st[3]
print(f"ph_sums, full groups of the longest template: pass 1 {st[5] / max(st[7], 1):.0f} cycles, pass 2 {st[6] / max(st[7], 1):.0f} cycles per wave ({st[7]} waves; s_memtime ticks)")
print(f"flags wait {st[4] / wgs:.0f} cycles; work-groups {wgs}, staging {stage / wgs:.0f} cycles per work-group, item loop of wave 0 {items / wgs:.0f} cycles ({nit / wgs:.2f} items), staging share {stage / (stage + items):.3f}")
n1, n2 = max(pu[7], 1), max(pu[15], 1)
print("ph_predict, cycles per work-group: init %d | particles + box %d | cube origin %d | per-sigma fit %d | duplicate search %d | chain numbers + map %d (%d work-groups)" % (pu[0] / n1, pu[1] / n1, pu[2] / n1, pu[3] / n1, pu[4] / n1, pu[5] / n1, pu[7]))
print("ph_update, cycles per work-group: loads %d | pending centroid %d | (exit test) %d | likelihood %d | weights, two serial sums %d | N_eff / CDF / centroid %d | decisions %d | resampling + write-back %d (%d)" % (pu[8] / n2, pu[9] / n2, 0, pu[10] / n2, pu[11] / n2, pu[12] / n2, pu[13] / n2, pu[14] / n2, pu[15]))
nf = max(fine[3], 1)
print("ph_predict, thread 0 inside 'particles + box': parent pose there after %d cycles | direction loop %d | CDF search %d" % (fine[0] / nf, fine[1] / nf, fine[2] / nf))
print("ph_predict duplicate search: %d of %d particles fell back to the scan" % (fine[4], fine[5]))
print("ph_sample full-group items: %d on the fast path (no clamp, no range test), %d on the others" % (fine[6], fine[7]))
