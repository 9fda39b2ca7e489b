#!/usr/bin/env python3
"""Dump the trace graph (nodes + links) of the bench stack's full trace loop to an .npz (input for timing pnr_reconstruct on a CPU).
usage (GPU box): python scripts/dump_graph.py gpurun_out/graph_1024.npz [size]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pnr_amd  # noqa: E402
import synth  # noqa: E402

out = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
img = synth.synth_torch(S, S, S, seed=3, device="cuda:0")
p = pnr_amd.make_params(sigmas=(2.0, 4.0, 6.0), np_=200, ni=200, zdist=2.0)
ctx = pnr_amd.Context(p, 0)
ctx.set_volume_device(img.data_ptr(), (S, S, S), keepalive=img)
ctx.frangi()
s = ctx.score_filter_sort(ctx.extract_seeds())
nodes, links, ntr, iters = ctx.trace_replay(s)
np.savez_compressed(out, nodes=nodes, links=np.asarray(links, np.int32))
print(f"{len(nodes) - 1} nodes, {len(links)} links, {ntr} traces, {iters} iterations -> {out}")
