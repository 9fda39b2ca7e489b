"""ctypes binding of libpnr_hip.so (include/pnr_hip.h).  Plumbing only: every compute call goes
through the C ABI into the HIP kernels; there is no Python or CPU implementation behind it, and
loading fails loudly when the extension has not been built."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNR_LIB_DIAG") or os.path.join(HERE, "libpnr_hip.so")  # PNR_LIB_DIAG: diagnostic builds only
PNR_MAX_SIGMAS = 8


class Params(C.Structure):
    _fields_ = [("sig", C.c_float * PNR_MAX_SIGMAS), ("nsig", C.c_int), ("somaradius", C.c_int), ("tolerance", C.c_float),
                ("znccth", C.c_float), ("kappa", C.c_float), ("step", C.c_int), ("ni", C.c_int), ("np", C.c_int),
                ("zdist", C.c_float), ("nodepervol", C.c_int), ("vol", C.c_int), ("Kc", C.c_float),
                ("neff_ratio", C.c_float), ("alpha", C.c_float), ("beta", C.c_float), ("C", C.c_float),
                ("rng_seed", C.c_uint32), ("max_trace_count", C.c_int)]


SEED_DT = np.dtype([(k, "f4") for k in ("x", "y", "z", "vx", "vy", "vz", "score", "corr")])
XEST_DT = np.dtype([(k, "f4") for k in ("x", "y", "z", "vx", "vy", "vz", "sig", "corr")])
NODE_DT = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("vx", "f4"), ("vy", "f4"), ("vz", "f4"), ("corr", "f4"),
                    ("sig", "f4"), ("type", "i4")])


class PnrError(RuntimeError):
    pass


# pnr_allgather_fn / pnr_trace_fn (include/pnr_hip.h)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
TRACE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p)


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(HERE, "csrc")
    if force:
        subprocess.run(["make", "-s", "-C", src, "clean"], check=True)
    subprocess.run(["make", "-s", "-j4", "-C", src], check=True)
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PnrError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the PNR hot path)")
    # A process must not end up with two HIP runtimes (torch wheels bundle their own libamdhip64): if torch is
    # around, let it load its runtime first so that libpnr_hip.so binds to the same one.
    if not os.environ.get("PNR_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.pnr_last_error.restype = C.c_char_p
    L.pnr_default_params.argtypes = [C.POINTER(Params)]
    L.pnr_default_params.restype = None
    L.pnr_create.argtypes = [C.POINTER(Params), i32, C.POINTER(vp)]
    L.pnr_destroy.argtypes = [vp]
    L.pnr_destroy.restype = None
    L.pnr_set_stream.argtypes = [vp, vp]
    L.pnr_synchronize.argtypes = [vp]
    L.pnr_set_volume.argtypes = [vp, vp, i64, i64, i64]
    L.pnr_set_volume_device.argtypes = [vp, vp, i64, i64, i64]
    L.pnr_frangi.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pnr_get_frangi.argtypes = [vp] + [vp] * 5
    L.pnr_gaussian.argtypes = [vp, C.c_float, vp]
    L.pnr_hessian.argtypes = [vp, C.c_float] + [vp] * 6
    L.pnr_set_j8_v.argtypes = [vp] + [vp] * 4
    L.pnr_extract_seeds.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.pnr_extract_seeds_range.argtypes = [vp, i64, i64, C.POINTER(vp), C.POINTER(i64)]
    L.pnr_zncc_batch.argtypes = [vp, vp, i64, vp, vp]
    L.pnr_score_filter_sort_seeds.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.pnr_score_filter_seeds.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.pnr_sort_seeds.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.pnr_trace_batch.argtypes = [vp, vp, i64, vp, vp, vp, i32, vp, vp, vp]
    L.pnr_replay_traces.argtypes = [C.POINTER(Params), i64, i64, i64, vp, i64, vp, vp, vp, i64, C.POINTER(i64), vp, i64,
                                    C.POINTER(i64), C.POINTER(i64)]
    L.pnr_trace_replay.argtypes = [vp, vp, i64, i64, vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.pnr_reconstruct.argtypes = [vp, i64, vp, i64, C.c_float, C.c_float, i32, C.c_float, C.c_float, i32, vp, vp, i64, C.POINTER(i64)]
    L.pnr_get_table.argtypes = [vp, C.c_char_p, vp, i64, C.POINTER(i64)]
    L.pnr_frangi_slab.argtypes = [vp, i64, i64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pnr_quantise_j8.argtypes = [vp, C.c_float, C.c_float]
    L.pnr_soma.argtypes = [vp, vp, C.POINTER(C.c_int32), C.POINTER(i64)]
    L.pnr_get_soma.argtypes = [vp, vp, i64, C.POINTER(i64), vp, vp, i64, C.POINTER(i64)]
    L.pnr_replay_traces_ctx.argtypes = [vp, vp, i64, vp, vp, vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64), C.POINTER(i64)]
    L.pnr_set_profiling.argtypes = [vp, i32]
    L.pnr_set_smc_driver.argtypes = [vp, i32]
    L.pnr_get_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]
    L.pnr_reset_kernel_ms.argtypes = [vp]
    L.pnr_expf_batch.argtypes = [vp, vp, i64, vp]
    L.pnr_eigen_batch.argtypes = [vp, vp, i64, vp, vp]
    L.pnr_get_graph.argtypes = [vp, vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64)]
    L.pnr_trace_replay_sharded.argtypes = [vp, vp, i64, i32, i32, ALLGATHER_FN, vp, vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64),
                                           C.POINTER(i64), C.POINTER(i64)]
    L.pnr_sched_playback.argtypes = [C.POINTER(Params), i64, i64, i64, vp, i64, i32, i32, ALLGATHER_FN, vp, i64, TRACE_FN, vp, i32, i32, i32, i32, i32,
                                     vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.pnr_sched_playback2.argtypes = [C.POINTER(Params), i64, i64, i64, vp, i64, i32, i32, ALLGATHER_FN, vp, i64, TRACE_FN, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                                      vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.pnr_get_trace_log.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.pnr_shm_exchange_open.argtypes = [C.c_char_p, i32, i32, i64, C.POINTER(vp)]
    L.pnr_shm_allgather.argtypes = [vp, vp, vp, i64]
    L.pnr_shm_exchange_close.argtypes = [vp]
    L.pnr_shm_exchange_close.restype = None
    L.pnr_rccl_unique_id.argtypes = [vp]
    L.pnr_rccl_exchange_open.argtypes = [vp, i32, i32, i32, i64, C.POINTER(vp)]
    L.pnr_rccl_allgather.argtypes = [vp, vp, vp, i64]
    L.pnr_rccl_allreduce_minmax.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pnr_rccl_exchange_close.argtypes = [vp]
    L.pnr_rccl_exchange_close.restype = None
    L.pnr_set_option.argtypes = [vp, C.c_char_p, i64]
    L.pnr_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    for name in EXPORTS:
        if name not in ("pnr_last_error", "pnr_default_params", "pnr_destroy", "pnr_shm_exchange_close", "pnr_rccl_exchange_close"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


# the drop-in boundary (include/pnr_hip.h)
PRODUCT_EXPORTS = ["pnr_last_error", "pnr_default_params", "pnr_create", "pnr_destroy", "pnr_set_stream", "pnr_synchronize",
                   "pnr_set_volume", "pnr_set_volume_device", "pnr_frangi", "pnr_get_frangi", "pnr_extract_seeds", "pnr_extract_seeds_range",
                   "pnr_zncc_batch", "pnr_score_filter_sort_seeds", "pnr_trace_batch", "pnr_replay_traces", "pnr_replay_traces_ctx",
                   "pnr_frangi_slab", "pnr_quantise_j8", "pnr_soma", "pnr_get_soma", "pnr_trace_replay", "pnr_reconstruct", "pnr_reconstruct_stage", "pnr_set_profiling",
                   "pnr_set_smc_driver", "pnr_get_kernel_ms", "pnr_reset_kernel_ms", "pnr_get_graph", "pnr_trace_replay_sharded",
                   "pnr_set_option", "pnr_get_option", "pnr_score_filter_seeds", "pnr_sort_seeds", "pnr_get_trace_log",
                   "pnr_shm_exchange_open", "pnr_shm_allgather", "pnr_shm_exchange_close",
                   "pnr_rccl_unique_id", "pnr_rccl_exchange_open", "pnr_rccl_allgather", "pnr_rccl_allreduce_minmax", "pnr_rccl_exchange_close"]
# test taps (include/pnr_hip_test.h): single stages of the device code and the scheduler over a host engine, for tests/ only
TEST_EXPORTS = ["pnr_gaussian", "pnr_hessian", "pnr_set_j8_v", "pnr_get_table", "pnr_expf_batch", "pnr_eigen_batch",
                "pnr_sched_playback", "pnr_sched_playback2"]
EXPORTS = PRODUCT_EXPORTS + TEST_EXPORTS


def check(rc):
    if rc != 0:
        raise PnrError(f"libpnr_hip error {rc}: {load().pnr_last_error().decode()}")


def make_params(sigmas=(2, 4, 6), somaradius=0, tolerance=5, znccth=0.3, kappa=3, step=2, ni=200, np_=20, zdist=2,
                nodepervol=4, vol=1, rng_seed=42, **const):
    p = Params()
    load().pnr_default_params(C.byref(p))
    sig = sorted(float(s) for s in sigmas)  # parse_csv_string sorts (Advantra_plugin.cpp:1885-1897)
    if len(sig) > PNR_MAX_SIGMAS:
        raise PnrError("too many sigmas")
    for i, s in enumerate(sig):
        p.sig[i] = s
    p.nsig = len(sig)
    p.somaradius, p.tolerance, p.znccth, p.kappa = somaradius, tolerance, znccth, kappa
    p.step, p.ni, p.np, p.zdist, p.nodepervol, p.vol, p.rng_seed = step, ni, np_, zdist, nodepervol, vol, rng_seed
    for k, v in const.items():
        setattr(p, k, v)
    return p


# what a new Context starts with (tests run every case with both SMC drivers by patching this; the library itself reads no
# environment variable): {"smc_driver": "phased" | "persistent", "options": {key: value}}
DEFAULTS = {"smc_driver": None, "options": {}}


class Context:
    """One GPU worth of PNR hot path (pnr_ctx)."""

    def __init__(self, params, device=0):
        self.L = load()
        self.p = params
        h = C.c_void_p()
        check(self.L.pnr_create(C.byref(params), device, C.byref(h)))
        self.h = h
        self.shape = None
        self._keep = None
        if DEFAULTS.get("smc_driver"):
            self.set_smc_driver(DEFAULTS["smc_driver"])
        for k, v in (DEFAULTS.get("options") or {}).items():
            self.set_option(k, v)

    def close(self):
        if getattr(self, "h", None):
            self.L.pnr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- volume ----
    def set_volume(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        l, h, w = img.shape
        check(self.L.pnr_set_volume(self.h, img.ctypes.data, w, h, l))
        self.shape = (l, h, w)

    def set_volume_device(self, data_ptr, shape, keepalive=None):
        l, h, w = shape
        check(self.L.pnr_set_volume_device(self.h, data_ptr, w, h, l))
        self.shape = (l, h, w)
        self._keep = keepalive

    def set_stream(self, stream_ptr):
        check(self.L.pnr_set_stream(self.h, stream_ptr))

    def synchronize(self):
        check(self.L.pnr_synchronize(self.h))

    # ---- Frangi ----
    def frangi(self):
        a, b = C.c_float(), C.c_float()
        check(self.L.pnr_frangi(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def frangi_slab(self, z_keep0, z_keep1):
        """Frangi of a slab with halo: (Jmin, Jmax) over the kept planes only, J not yet quantised (pnr_frangi_slab)"""
        a, b = C.c_float(), C.c_float()
        check(self.L.pnr_frangi_slab(self.h, z_keep0, z_keep1, C.byref(a), C.byref(b)))
        return a.value, b.value

    def quantise_j8(self, jmin, jmax):
        check(self.L.pnr_quantise_j8(self.h, jmin, jmax))

    def get_frangi(self, J=True, J8=True, V=True):
        out = {}
        if J:
            out["J"] = np.empty(self.shape, np.float32)
        if J8:
            out["J8"] = np.empty(self.shape, np.uint8)
        if V:
            for k in ("Vx", "Vy", "Vz"):
                out[k] = np.empty(self.shape, np.uint8)
        ptr = lambda k: out[k].ctypes.data if k in out else None
        check(self.L.pnr_get_frangi(self.h, ptr("J"), ptr("J8"), ptr("Vx"), ptr("Vy"), ptr("Vz")))
        return out

    def gaussian(self, sig):
        F = np.empty(self.shape, np.float32)
        check(self.L.pnr_gaussian(self.h, sig, F.ctypes.data))
        return F

    def hessian(self, sig):
        H = [np.empty(self.shape, np.float32) for _ in range(6)]
        check(self.L.pnr_hessian(self.h, sig, *[a.ctypes.data for a in H]))
        return dict(zip(("Dzz", "Dyy", "Dyz", "Dxx", "Dxy", "Dxz"), H))

    def set_j8_v(self, J8, Vx, Vy, Vz):
        arrs = [np.ascontiguousarray(a, np.uint8) for a in (J8, Vx, Vy, Vz)]
        check(self.L.pnr_set_j8_v(self.h, *[a.ctypes.data for a in arrs]))

    # ---- seeds ----
    def extract_seeds(self, z0=None, z1=None):
        ptr, n = C.c_void_p(), C.c_int64()
        if z0 is None:
            check(self.L.pnr_extract_seeds(self.h, C.byref(ptr), C.byref(n)))
        else:
            check(self.L.pnr_extract_seeds_range(self.h, z0, z1, C.byref(ptr), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, SEED_DT)
        buf = (C.c_char * (n.value * SEED_DT.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, SEED_DT).copy()

    def zncc(self, pos_dir):
        pd = np.ascontiguousarray(pos_dir, np.float32).reshape(-1, 6)
        corr = np.empty(len(pd), np.float32)
        sig = np.empty(len(pd), np.float32)
        check(self.L.pnr_zncc_batch(self.h, pd.ctypes.data, len(pd), corr.ctypes.data, sig.ctypes.data))
        return corr, sig

    def score_filter_sort(self, seeds, fn="pnr_score_filter_sort_seeds"):
        s = np.ascontiguousarray(seeds, SEED_DT).copy()
        n = C.c_int64()
        check(getattr(self.L, fn)(self.h, s.ctypes.data, len(s), C.byref(n)))
        return s[:n.value].copy()

    def score_filter(self, seeds):
        """score + threshold, order kept (a rank's own z-slab of seeds)"""
        return self.score_filter_sort(seeds, "pnr_score_filter_seeds")

    def sort_seeds(self, seeds):
        """stable sort by corr of already scored seeds (the merged list of all ranks)"""
        return self.score_filter_sort(seeds, "pnr_sort_seeds")

    # ---- tracing ----
    def trace_batch(self, seeds, dbg_iters=0):
        s = np.ascontiguousarray(seeds, SEED_DT)
        n, ni, npc = len(s), self.p.ni, self.p.np
        T = np.zeros(2 * n, np.int32)
        stop = np.zeros(2 * n, np.int32)
        xc = np.zeros((2 * n, ni), XEST_DT)
        dbg = {}
        xf = idx = neff = None
        if dbg_iters > 0:
            dbg_iters = min(dbg_iters, ni)
            xf = np.zeros((2 * n, dbg_iters, npc, 9), np.float32)
            idx = np.zeros((2 * n, dbg_iters, npc), np.int32)
            neff = np.zeros((2 * n, dbg_iters), np.float32)
            dbg = dict(xfilt=xf, idxres=idx, neff=neff)
        p = lambda a: a.ctypes.data if a is not None else None
        check(self.L.pnr_trace_batch(self.h, s.ctypes.data, n, T.ctypes.data, stop.ctypes.data, xc.ctypes.data, dbg_iters,
                                     p(xf), p(idx), p(neff)))
        return T, stop, xc, dbg

    def replay(self, seeds, T, xc):
        """host replay of the trace bookkeeping with this context's parameters, dimensions and soma (pnr_replay_traces_ctx)"""
        s = np.ascontiguousarray(seeds, SEED_DT)
        T = np.ascontiguousarray(T, np.int32)
        xc = np.ascontiguousarray(xc, XEST_DT)
        cap = int(T.sum()) + 2 + 4096
        while True:
            nodes = np.zeros(cap, NODE_DT)
            links = np.zeros((2 * cap + 2, 2), np.int32)
            nn, nl, nt = C.c_int64(), C.c_int64(), C.c_int64()
            check(self.L.pnr_replay_traces_ctx(self.h, s.ctypes.data, len(s), T.ctypes.data, xc.ctypes.data, nodes.ctypes.data, cap,
                                               C.byref(nn), links.ctypes.data, len(links), C.byref(nl), C.byref(nt)))
            if nn.value <= cap and nl.value <= len(links):
                return nodes[:nn.value].copy(), links[:nl.value].copy(), nt.value
            cap = int(nn.value) + 2

    # ---- soma path (somaradius > 0) ----
    def soma(self, want_e8=False):
        """pnr_soma: threshold, soma nodes, sparse label map (voxel index, node index) [, the eroded + blurred stack]"""
        l, h, w = self.shape
        E8 = np.zeros((l, h, w), np.uint8) if want_e8 else None
        th, n = C.c_int32(), C.c_int64()
        check(self.L.pnr_soma(self.h, E8.ctypes.data if want_e8 else None, C.byref(th), C.byref(n)))
        nn, nv = C.c_int64(), C.c_int64()
        check(self.L.pnr_get_soma(self.h, None, 0, C.byref(nn), None, None, 0, C.byref(nv)))
        nodes = np.zeros(nn.value, NODE_DT)
        vox = np.zeros(nv.value, np.int64)
        lab = np.zeros(nv.value, np.int32)
        check(self.L.pnr_get_soma(self.h, nodes.ctypes.data, len(nodes), C.byref(nn), vox.ctypes.data, lab.ctypes.data, len(vox), C.byref(nv)))
        out = dict(threshold=th.value, nodes=nodes, vox=vox, lab=lab)
        if want_e8:
            out["E8"] = E8
        return out

    def get_graph(self):
        """node graph of the last trace_replay / trace_replay_sharded (pnr_get_graph)"""
        nn, nl = C.c_int64(), C.c_int64()
        check(self.L.pnr_get_graph(self.h, None, 0, C.byref(nn), None, 0, C.byref(nl)))
        nodes = np.zeros(nn.value, NODE_DT)
        links = np.zeros((nl.value, 2), np.int32)
        check(self.L.pnr_get_graph(self.h, nodes.ctypes.data, len(nodes), C.byref(nn), links.ctypes.data, len(links), C.byref(nl)))
        return nodes, links

    def trace_log(self):
        """[(seed rank, direction, ti_limit, reason, value)] of the last trace_replay with option trace_log = 1 (pnr_get_trace_log)"""
        n = C.c_int64()
        check(self.L.pnr_get_trace_log(self.h, None, 0, C.byref(n)))
        rec = np.zeros((n.value, 5), np.int32)
        check(self.L.pnr_get_trace_log(self.h, rec.ctypes.data, n.value, C.byref(n)))
        return rec

    def trace_replay(self, seeds, first_batch=0):
        """streamed trace + replay (pnr_trace_replay): nodes, links, traces used, SMC iterations run"""
        s = np.ascontiguousarray(seeds, SEED_DT)
        nn, nl, nt, it = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self.L.pnr_trace_replay(self.h, s.ctypes.data, len(s), first_batch, None, 0, C.byref(nn), None, 0, C.byref(nl), C.byref(nt), C.byref(it)))
        nodes, links = self.get_graph()
        return nodes, links, nt.value, it.value

    def trace_replay_sharded(self, seeds, rank, world, exchange):
        """this rank's part of tracing ONE sorted seed list on `world` GPUs (pnr_trace_replay_sharded); `exchange` is an ALLGATHER_FN
        (a Python callback, e.g. multigpu.make_exchange) or a ShmExchange (the library's own shared-memory all-gather: no Python in
        the loop).  Every rank returns the same graph; the iteration count is this rank's."""
        s = np.ascontiguousarray(seeds, SEED_DT)
        nn, nl, nt, it = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        fn, user = (exchange.fn, exchange.handle) if hasattr(exchange, "handle") else (exchange, None)  # ShmExchange / RcclExchange: no Python in the loop
        check(self.L.pnr_trace_replay_sharded(self.h, s.ctypes.data, len(s), rank, world, fn, user, None, 0, C.byref(nn), None, 0,
                                              C.byref(nl), C.byref(nt), C.byref(it)))
        nodes, links = self.get_graph()
        return nodes, links, nt.value, it.value

    def have_soma(self):
        nn, nv = C.c_int64(), C.c_int64()
        return self.L.pnr_get_soma(self.h, None, 0, C.byref(nn), None, None, 0, C.byref(nv)) == 0

    def set_option(self, key, value):
        check(self.L.pnr_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64()
        check(self.L.pnr_get_option(self.h, key.encode(), C.byref(v)))
        return v.value

    def set_options(self, spec):
        """'window=512,groups=2' (bench.py / tests: PNR_BENCH_OPTS); returns what was set"""
        out = {}
        for kv in filter(None, (spec or "").split(",")):
            k, v = kv.split("=")
            self.set_option(k.strip(), int(v))
            out[k.strip()] = int(v)
        return out

    def table(self, name):
        n = C.c_int64()
        check(self.L.pnr_get_table(self.h, name.encode(), None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint32 if name == "rng" else np.float32)
        check(self.L.pnr_get_table(self.h, name.encode(), out.ctypes.data, n.value, C.byref(n)))
        return out

    def set_smc_driver(self, driver):
        """'phased' (0, default) or 'persistent' (1): how the particle filter is scheduled; results are identical."""
        d = {"phased": 0, "persistent": 1}.get(driver, driver)
        check(self.L.pnr_set_smc_driver(self.h, int(d)))

    # ---- profiling ----
    def set_profiling(self, on=True):
        check(self.L.pnr_set_profiling(self.h, int(on)))

    def kernel_ms(self, group):
        ms, n = C.c_double(), C.c_int64()
        check(self.L.pnr_get_kernel_ms(self.h, group.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_kernel_ms(self):
        check(self.L.pnr_reset_kernel_ms(self.h))

    def eigen(self, A, vectors=True):
        """test tap: the device's JAMA solver on n symmetric 3 x 3 matrices -> (V or None, d)"""
        A = np.ascontiguousarray(A, np.float64).reshape(-1, 3, 3)
        d = np.empty((len(A), 3), np.float64)
        V = np.empty((len(A), 3, 3), np.float64) if vectors else None
        check(self.L.pnr_eigen_batch(self.h, A.ctypes.data, len(A), V.ctypes.data if vectors else None, d.ctypes.data))
        return V, d

    def expf(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        check(self.L.pnr_expf_batch(self.h, x.ctypes.data, x.size, y.ctypes.data))
        return y


def replay(params, shape, seeds, T, xc):
    """Host replay of trackPos/trackNeg bookkeeping (pure host; no GPU needed)."""
    L = load()
    l, h, w = shape
    s = np.ascontiguousarray(seeds, SEED_DT)
    T = np.ascontiguousarray(T, np.int32)
    xc = np.ascontiguousarray(xc, XEST_DT)
    cap = int(T.sum()) + 2
    nodes = np.zeros(cap, NODE_DT)
    links = np.zeros((2 * cap + 2, 2), np.int32)
    nn, nl, nt = C.c_int64(), C.c_int64(), C.c_int64()
    check(L.pnr_replay_traces(C.byref(params), w, h, l, s.ctypes.data, len(s), T.ctypes.data, xc.ctypes.data,
                              nodes.ctypes.data, cap, C.byref(nn), links.ctypes.data, len(links), C.byref(nl), C.byref(nt)))
    return nodes[:nn.value].copy(), links[:nl.value].copy(), nt.value


class ShmExchange:
    """pnr_shm_exchange: an all-gather between the ranks of ONE host through POSIX shared memory (include/pnr_hip.h).  `fn` / `handle`
    are what pnr_trace_replay_sharded takes as exchange / user; allgather() is the same call for other small host-side collectives."""

    def __init__(self, name, rank, world, capacity=1 << 20):
        self.L = load()
        h = C.c_void_p()
        check(self.L.pnr_shm_exchange_open(name.encode(), rank, world, capacity, C.byref(h)))
        self.handle, self.rank, self.world, self.capacity = h, rank, world, capacity
        self.fn = C.cast(self.L.pnr_shm_allgather, ALLGATHER_FN)

    def allgather(self, block):
        """bytes of equal length on every rank -> list of `world` bytes objects"""
        n = len(block)
        send = (C.c_char * max(n, 1)).from_buffer_copy(block.ljust(1, b"\0"))
        recv = (C.c_char * max(n * self.world, 1))()
        rc = self.L.pnr_shm_allgather(self.handle, send, recv, n)
        if rc != 0:
            raise PnrError(f"shared-memory all-gather failed ({rc})")
        raw = bytes(recv)
        return [raw[r * n:(r + 1) * n] for r in range(self.world)]

    def close(self):
        if self.handle:
            self.L.pnr_shm_exchange_close(self.handle)
            self.handle = None


class RcclExchange:
    """pnr_rccl_exchange: the same collectives over RCCL from a host that is not torch (include/pnr_hip.h): ncclAllGather of one block
    per rank (`fn` / `handle` = exchange / user of pnr_trace_replay_sharded) and the (min, max) all-reduce.  `uid` = the 128 bytes
    of RcclExchange.unique_id() of one rank, handed to the others by the launcher.  A collective call: one rank per GPU."""

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(load().pnr_rccl_unique_id(buf))
        return bytes(buf)

    def __init__(self, uid, rank, world, device, capacity=1 << 20):
        self.L = load()
        h = C.c_void_p()
        check(self.L.pnr_rccl_exchange_open(uid, rank, world, device, capacity, C.byref(h)))
        self.handle, self.rank, self.world, self.capacity = h, rank, world, capacity
        self.fn = C.cast(self.L.pnr_rccl_allgather, ALLGATHER_FN)

    def allgather(self, block):
        n = len(block)
        send = (C.c_char * max(n, 1)).from_buffer_copy(block.ljust(1, b"\0"))
        recv = (C.c_char * max(n * self.world, 1))()
        check(self.L.pnr_rccl_allgather(self.handle, send, recv, n))
        raw = bytes(recv)
        return [raw[r * n:(r + 1) * n] for r in range(self.world)]

    def minmax(self, mn, mx):
        a, b = C.c_float(mn), C.c_float(mx)
        check(self.L.pnr_rccl_allreduce_minmax(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.handle:
            self.L.pnr_rccl_exchange_close(self.handle)
            self.handle = None


def sched_playback(params, shape, seeds, trace_fn, rank=0, world=1, exchange=None, block_bytes=0, window=768, groups=1, poll=4, look0=0, look_pct=-1, tentative=True, target=-1, lag=-1):
    """The streaming scheduler over a host engine that plays back map-free traces (pnr_sched_playback; no GPU): `trace_fn(pos_dir6)`
    -> (T, xc[ni][8]).  Returns nodes, links, traces used, iterations on this rank."""
    L = load()
    l, h, w = shape
    s = np.ascontiguousarray(seeds, SEED_DT)
    ni = params.ni

    def _tr(user, pd, T, xc):
        try:
            Tn, rows = trace_fn(np.ctypeslib.as_array(pd, (6,)).copy())
            T[0] = int(Tn)
            out = np.ctypeslib.as_array(C.cast(xc, C.POINTER(C.c_float)), (ni, 8))
            k = min(int(Tn), ni)
            out[:k] = np.asarray(rows, np.float32).reshape(-1, 8)[:k]
            return 0
        except Exception:  # noqa: BLE001 -- must not propagate through the C frames
            import traceback
            traceback.print_exc()
            return 1

    tcb = TRACE_FN(_tr)
    xuser = None
    if isinstance(exchange, (ShmExchange, RcclExchange)):
        exchange, xuser = exchange.fn, exchange.handle
    xcb = exchange if exchange is not None else C.cast(None, ALLGATHER_FN)
    nn, nl, nt, it = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    cap = 2 * len(s) * ni + 2
    nodes = np.zeros(cap, NODE_DT)
    links = np.zeros((2 * cap + 2, 2), np.int32)
    check(L.pnr_sched_playback2(C.byref(params), w, h, l, s.ctypes.data, len(s), rank, world, xcb, xuser, block_bytes, tcb, None, window, groups, poll, look0, look_pct,
                                int(bool(tentative)), int(target), int(lag), nodes.ctypes.data, cap, C.byref(nn), links.ctypes.data, len(links), C.byref(nl), C.byref(nt), C.byref(it)))
    return nodes[:nn.value].copy(), links[:nl.value].copy(), nt.value, it.value


def reconstruct(nodes, links, trace_rsmpl=0.0, sig2radius=0.0, refine_iter=0, epsilon2=0.0, group_radius=0.0, tree_size_min=0):
    """reconstruct() chain on the host (pnr_reconstruct): tree nodes (index 0 dummy) and parent indices (-1 = root)."""
    L = load()
    nodes = np.ascontiguousarray(nodes, NODE_DT)
    links = np.ascontiguousarray(links, np.int32).reshape(-1, 2)
    cap = max(16, 4 * len(nodes))
    while True:
        out = np.zeros(cap, NODE_DT)
        par = np.zeros(cap, np.int32)
        n = C.c_int64()
        check(L.pnr_reconstruct(nodes.ctypes.data, len(nodes), links.ctypes.data, len(links), trace_rsmpl, sig2radius, refine_iter,
                                epsilon2, group_radius, tree_size_min, out.ctypes.data, par.ctypes.data, cap, C.byref(n)))
        if n.value <= cap:
            return out[:n.value].copy(), par[:n.value].copy()
        cap = int(n.value)


def reconstruct_stage(nodes, links, stage, trace_rsmpl=0.0, sig2radius=0.0, refine_iter=0, epsilon2=0.0, group_radius=0.0):
    """the node list behind a stage of reconstruct() (pnr_reconstruct_stage: 1 _n0res_, 2 _n1_, 3 _n2_, 4 _n2tree_) -> nodes, link pairs"""
    L = load()
    nodes = np.ascontiguousarray(nodes, NODE_DT)
    links = np.ascontiguousarray(links, np.int32).reshape(-1, 2)
    L.pnr_reconstruct_stage.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int,
                                        C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    nn, nl = C.c_int64(), C.c_int64()
    check(L.pnr_reconstruct_stage(nodes.ctypes.data, len(nodes), links.ctypes.data, len(links), trace_rsmpl, sig2radius, refine_iter, epsilon2,
                                  group_radius, stage, None, 0, C.byref(nn), None, 0, C.byref(nl)))
    out = np.zeros(nn.value, NODE_DT)
    lk = np.zeros((nl.value, 2), np.int32)
    check(L.pnr_reconstruct_stage(nodes.ctypes.data, len(nodes), links.ctypes.data, len(links), trace_rsmpl, sig2radius, refine_iter, epsilon2,
                                  group_radius, stage, out.ctypes.data, len(out), C.byref(nn), lk.ctypes.data, len(lk), C.byref(nl)))
    return out, lk


def kernel_source_hash(names=("smc_phased.hip", "smc_device.h", "smc.hip", "ctx.h", "stream_sched.h")):
    """sha256 (first 16 hex digits) over the sources the bytes of a trace-iteration depend on: the SMC kernels, and the defaults
    (ctx.h: max_split, sums_deep_max ...) and the scheduler (stream_sched.h) that decide which kernel form and split a launch
    takes.  The committed PMC traffic profiles carry it, and bench.py only quotes a profile whose hash equals the hash of the sources
    it runs (a changed kernel or default is never priced with the bytes of an older one)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for n in names:
        with open(os.path.join(d, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]
