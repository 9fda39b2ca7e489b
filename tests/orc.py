"""ctypes bindings onto the test-only checkers: oracle/liborc.so (our C restatement)
and oracle/_ref/libpnr_ref.so (the reference's own frangi.cpp/seed.cpp, when built)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def _build():
    subprocess.run(["make", "-s", "-C", ORC_DIR], check=True, stdout=subprocess.DEVNULL)


def load_oracle():
    path = os.path.join(ORC_DIR, "liborc.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ORC_DIR, "pnr_oracle.c")):
        _build()
    L = C.CDLL(path)
    L.orc_imgaussian3d.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, f32p]
    L.orc_hessian3d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float] + [f32p] * 6
    L.orc_eigen3.argtypes = [f64p, f64p, f64p]
    L.orc_frangi3d.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                               f32p, C.POINTER(C.c_float), C.POINTER(C.c_float), u8p, u8p, u8p]
    L.orc_j8.argtypes = [f32p, C.c_int64, C.c_float, C.c_float, u8p]
    L.orc_extract_seeds.argtypes = [C.c_double, u8p, C.c_int, C.c_int, C.c_int, u8p, u8p, u8p, f32p, C.c_int64]
    L.orc_extract_seeds.restype = C.c_int64
    L.orc_tracker_new.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int, C.c_uint32]
    L.orc_tracker_new.restype = C.c_void_p
    L.orc_tracker_free.argtypes = [C.c_void_p]
    for nm in ("sz", "ndir"):
        getattr(L, "orc_tracker_" + nm).argtypes = [C.c_void_p]
        getattr(L, "orc_tracker_" + nm).restype = C.c_int
    for nm in ("p", "u", "w0", "w0_cws", "v", "w", "w_cws"):
        getattr(L, "orc_tracker_" + nm).argtypes = [C.c_void_p]
        getattr(L, "orc_tracker_" + nm).restype = C.POINTER(C.c_float)
    L.orc_tracker_rng.argtypes = [C.c_void_p]
    L.orc_tracker_rng.restype = C.POINTER(C.c_uint32)
    L.orc_tracker_model_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_tracker_model_count.restype = C.c_int
    L.orc_tracker_model_vuw.argtypes = [C.c_void_p, C.c_int]
    L.orc_tracker_model_vuw.restype = C.POINTER(C.c_float)
    L.orc_tracker_model_wgt.argtypes = [C.c_void_p, C.c_int]
    L.orc_tracker_model_wgt.restype = C.POINTER(C.c_float)
    L.orc_tracker_model_avg.argtypes = [C.c_void_p, C.c_int]
    L.orc_tracker_model_avg.restype = C.c_float
    L.orc_glibc_rand.argtypes = [C.c_uint32, C.c_int, u32p]
    L.orc_interp.argtypes = [C.c_float] * 3 + [u8p, C.c_int, C.c_int, C.c_int]
    L.orc_interp.restype = C.c_float
    L.orc_zncc.argtypes = [C.c_void_p] + [C.c_float] * 6 + [u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.orc_zncc.restype = C.c_float
    L.orc_trace.argtypes = [C.c_void_p, f32p, u8p, C.c_int, C.c_int, C.c_int, f32p, C.POINTER(C.c_int), C.c_int,
                            C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_trace.restype = C.c_int
    L.orc_replay.argtypes = [f32p, C.c_int64, i32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_void_p, C.c_int64, i32p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.orc_replay.restype = C.c_int64
    L.orc_frangi2d.argtypes = [u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float, f32p, C.POINTER(C.c_float), C.POINTER(C.c_float), u8p, u8p, u8p]
    L.orc_hessian2d.argtypes = [u8p, C.c_int, C.c_int, C.c_float, f32p, f32p, f32p]
    L.orc_tracker_new2.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_int, C.c_uint32, C.c_int]
    L.orc_tracker_new2.restype = C.c_void_p
    L.orc_replay_soma.argtypes = [f32p, C.c_int64, i32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, f32p, C.c_int64,
                                  C.c_void_p, C.c_int64, i32p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.orc_replay_soma.restype = C.c_int64
    L.orc_imerode_xy.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, u8p]
    L.orc_imgaussian_u8_xy.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float]
    L.orc_maxentropy_th.argtypes = [u8p, C.c_int64]
    L.orc_maxentropy_th.restype = C.c_ubyte
    L.orc_maxentropy_hist.argtypes = [np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
    L.orc_maxentropy_hist.restype = C.c_ubyte
    L.orc_conn3d.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, C.c_int64]
    L.orc_conn3d.restype = C.c_int64
    L.orc_soma_extract.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.POINTER(C.c_int), i32p, f32p, C.c_int64]
    L.orc_soma_extract.restype = C.c_int64
    L.orc_reconstruct.argtypes = [C.c_void_p, C.c_int64, i32p, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int,
                                  C.c_void_p, i32p, C.c_int64]
    L.orc_reconstruct.restype = C.c_int64
    return L


def load_ref():
    """The reference's own code (frangi.cpp/seed.cpp) or None when not built."""
    path = os.path.join(ORC_DIR, "_ref", "libpnr_ref.so")
    if not os.path.exists(path):
        if os.path.isdir("/root/reference/pnr-vaa3d"):
            _build()
        if not os.path.exists(path):
            return None
    L = C.CDLL(path)
    L.ref_imgaussian3d.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, f32p]
    L.ref_hessian3d.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float] + [f32p] * 6
    L.ref_eigen3.argtypes = [f64p, f64p, f64p]
    L.ref_frangi3d.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                               f32p, C.POINTER(C.c_float), C.POINTER(C.c_float), u8p, u8p, u8p]
    L.ref_frangi2d.argtypes = [u8p, C.c_int, C.c_int, f32p, C.c_int, C.c_float, C.c_float, f32p, C.POINTER(C.c_float), C.POINTER(C.c_float), u8p, u8p, u8p]
    L.ref_hessian2d.argtypes = [u8p, C.c_int, C.c_int, C.c_float, f32p, f32p, f32p]
    L.ref_imerode_xy.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, u8p]
    L.ref_imgaussian_u8_xy.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float]
    L.ref_extract_seeds.argtypes = [C.c_double, u8p, C.c_int, C.c_int, C.c_int, u8p, u8p, u8p, f32p, C.c_int64]
    L.ref_extract_seeds.restype = C.c_int64
    return L


# ---------------- convenience wrappers ----------------
NODE_DT = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("vx", "f4"), ("vy", "f4"), ("vz", "f4"),
                    ("corr", "f4"), ("sig", "f4"), ("type", "i4")])


def frangi2d(L, img, sigs, betaone=0.5, betatwo=15.0, prefix="orc"):
    """2-D Frangi of a single-slice stack (1, h, w): J, Jmin, Jmax, Vx, Vy, Vz (frangi.cpp:392)"""
    img = np.ascontiguousarray(img, np.uint8)
    _, h, w = img.shape
    J = np.zeros(img.shape, np.float32)
    Vx, Vy, Vz = (np.zeros(img.shape, np.uint8) for _ in range(3))
    jmin, jmax = C.c_float(), C.c_float()
    getattr(L, prefix + "_frangi2d")(img, w, h, np.asarray(sigs, np.float32), len(sigs), betaone, betatwo, J, C.byref(jmin), C.byref(jmax), Vx, Vy, Vz)
    return J, jmin.value, jmax.value, Vx, Vy, Vz


def frangi3d(L, img, sigs, zdist, alpha=0.5, beta=0.5, Cc=500.0, prefix="orc"):
    l, h, w = img.shape
    sig = np.asarray(sigs, np.float32)
    J = np.zeros(img.shape, np.float32)
    Vx, Vy, Vz = (np.zeros(img.shape, np.uint8) for _ in range(3))
    jmin, jmax = C.c_float(), C.c_float()
    getattr(L, prefix + "_frangi3d")(img, w, h, l, sig, len(sig), zdist, alpha, beta, Cc, J, C.byref(jmin), C.byref(jmax), Vx, Vy, Vz)
    return J, jmin.value, jmax.value, Vx, Vy, Vz


def j8(L, J, jmin, jmax):
    out = np.zeros(J.shape, np.uint8)
    L.orc_j8(J, J.size, jmin, jmax, out)
    return out


def extract_seeds(L, tol, J8, Vx, Vy, Vz, prefix="orc"):
    l, h, w = J8.shape
    cap = 1 << 16
    while True:
        buf = np.zeros((cap, 8), np.float32)
        n = getattr(L, prefix + "_extract_seeds")(float(tol), J8, w, h, l, Vx, Vy, Vz, buf, cap)
        if n <= cap:
            return buf[:n].copy()
        cap = int(n)


class Tracker:
    def __init__(self, L, sigs, step, np_, ni, kappa, znccth, Kc=20.0, neff_ratio=0.8, zdist=2.0, nodespervol=4, rng_seed=42, is2d=False):
        self.L = L
        self.sigs = np.asarray(sigs, np.float32)
        self.np, self.ni = np_, ni
        self.h = L.orc_tracker_new2(self.sigs, len(self.sigs), step, np_, ni, kappa, znccth, Kc, neff_ratio, zdist, nodespervol, rng_seed, int(is2d))
        assert self.h
        self.sz = L.orc_tracker_sz(self.h)
        self.ndir = L.orc_tracker_ndir(self.h)

    def __del__(self):
        try:
            self.L.orc_tracker_free(self.h)
        except Exception:
            pass

    def table(self, name):
        shp = {"p": (self.sz, 3), "u": (self.sz, 3), "w0": (self.sz,), "w0_cws": (self.sz,), "v": (self.ndir, 3),
               "w": (self.ndir, self.sz), "w_cws": (self.ndir, self.sz)}[name]
        ptr = getattr(self.L, "orc_tracker_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=shp).copy()

    def rng(self):
        return np.ctypeslib.as_array(self.L.orc_tracker_rng(self.h), shape=(self.np + 1,)).copy()

    def model(self, s):
        M = self.L.orc_tracker_model_count(self.h, s)
        vuw = np.ctypeslib.as_array(self.L.orc_tracker_model_vuw(self.h, s), shape=(M, 3)).copy()
        wgt = np.ctypeslib.as_array(self.L.orc_tracker_model_wgt(self.h, s), shape=(M,)).copy()
        return vuw, wgt, self.L.orc_tracker_model_avg(self.h, s)

    def zncc(self, img, pos_dir):
        l, h, w = img.shape
        pd = np.asarray(pos_dir, np.float32).reshape(-1, 6)
        corr = np.zeros(len(pd), np.float32)
        sig = np.zeros(len(pd), np.float32)
        s = C.c_float()
        for i, q in enumerate(pd):
            corr[i] = self.L.orc_zncc(self.h, *[float(v) for v in q], img, w, h, l, C.byref(s))
            sig[i] = s.value
        return corr, sig

    def trace(self, img, seed6, max_dbg=0):
        l, h, w = img.shape
        xc = np.zeros((self.ni, 8), np.float32)
        stop = C.c_int()
        xf = np.zeros((max(max_dbg, 1), self.np, 9), np.float32)
        idx = np.zeros((max(max_dbg, 1), self.np), np.int32)
        neff = np.zeros(max(max_dbg, 1), np.float32)
        T = self.L.orc_trace(self.h, np.ascontiguousarray(seed6, np.float32), img, w, h, l, xc, C.byref(stop), max_dbg,
                             xf.ctypes.data, idx.ctypes.data, neff.ctypes.data)
        return T, stop.value, xc, xf, idx, neff


def soma_extract(L, img, somaradius, cap=4096):
    """(E8, threshold, smap, soma nodes (x, y, z, r)) of the soma path (Advantra_plugin.cpp:2426-2448)"""
    l, h, w = img.shape
    img = np.ascontiguousarray(img, np.uint8)
    E8 = np.zeros_like(img)
    smap = np.zeros(img.shape, np.int32)
    nodes4 = np.zeros((cap, 4), np.float32)
    th = C.c_int()
    n = L.orc_soma_extract(img, w, h, l, int(somaradius), E8, C.byref(th), smap.reshape(-1), nodes4.reshape(-1), cap)
    assert n <= cap
    return E8, th.value, smap, nodes4[:n].copy()


def replay(L, seeds, T, xc, ni, shape, nodespervol, vol, max_trace_count=5000, smap=None, soma4=None):
    l, h, w = shape
    nsoma = 0 if soma4 is None else len(soma4)
    cap = int(T.sum()) + 2 + nsoma
    nodes = np.zeros(cap, NODE_DT)
    links = np.zeros((2 * cap + 2, 2), np.int32)
    nl, nt = C.c_int64(), C.c_int64()
    if smap is not None:
        nn = L.orc_replay_soma(np.ascontiguousarray(seeds, np.float32), len(seeds), np.ascontiguousarray(T, np.int32),
                               np.ascontiguousarray(xc, np.float32), ni, w, h, l, nodespervol, vol, max_trace_count,
                               np.ascontiguousarray(smap, np.int32).reshape(-1), np.ascontiguousarray(soma4, np.float32).reshape(-1), nsoma,
                               nodes.ctypes.data, cap, links, len(links), C.byref(nl), C.byref(nt))
        return nodes[:nn].copy(), links[:nl.value].copy(), nt.value
    nn = L.orc_replay(np.ascontiguousarray(seeds, np.float32), len(seeds), np.ascontiguousarray(T, np.int32),
                      np.ascontiguousarray(xc, np.float32), ni, w, h, l, nodespervol, vol, max_trace_count,
                      nodes.ctypes.data, cap, links, len(links), C.byref(nl), C.byref(nt))
    return nodes[:nn].copy(), links[:nl.value].copy(), nt.value


def reconstruct(L, nodes, links, trace_rsmpl=1.0, sig2radius=1.5, refine_iter=4, epsilon2=1e-4, group_radius=2.0, tree_size_min=10):
    nodes = np.ascontiguousarray(nodes, NODE_DT)
    links = np.ascontiguousarray(links, np.int32).reshape(-1, 2)
    cap = max(16, 4 * len(nodes))
    while True:
        out = np.zeros(cap, NODE_DT)
        par = np.zeros(cap, np.int32)
        n = L.orc_reconstruct(nodes.ctypes.data, len(nodes), links, len(links), trace_rsmpl, sig2radius, refine_iter, epsilon2,
                              group_radius, tree_size_min, out.ctypes.data, par, cap)
        if n <= cap:
            return out[:n].copy(), par[:n].copy()
        cap = int(n)
