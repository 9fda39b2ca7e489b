"""Host-side mirror of the reference's interface for the accelerated path, above the C ABI:
same class / method names, argument meaning and error behaviour as the reference's Frangi,
SeedExtractor and Tracker classes and its `advantra_func` plugin entry, so parity tests read like
calls into the reference.  All compute goes through libpnr_hip.so (HIP kernels); nothing here is a
CPU implementation of the path.

Reference interfaces mirrored (file:line under /root/reference/pnr-vaa3d/):
  Frangi::Frangi / frangi3d / imgaussian / hessian3d     frangi.h:24-41, frangi.cpp:152,291,647
  SeedExtractor::extractSeeds                            seed.h, seed.cpp:556
  Tracker::Tracker / znccBBB / trackPos / trackNeg       tracker.h:155-193, tracker.cpp:79,819,825,1891
  Advantra::dofunc("advantra_func") + reconstruction_func  Advantra_plugin.cpp:274-337, 2183-2731
"""
import os
import numpy as np
from . import lib
from .lib import Context, PnrError, SEED_DT, make_params

NRINPUTPARAMS = 11  # Advantra_plugin.cpp:60
# hard-wired plugin constants echoed into the SWC header (Advantra_plugin.cpp:63-83)
CONSTS = dict(Kc=20.0, neff_ratio=0.8, frangi_alfa=0.5, frangi_beta=0.5, frangi_C=500.0, frangi_betaone=0.5,
              frangi_betatwo=15.0, MAX_TRACE_COUNT=5000, EPSILON2=0.0001, REFINE_ITER=4, SIG2RADIUS=1.5, TRACE_RSMPL=1.0,
              GROUP_RADIUS=2.0, ENFORCE_SINGLE_TREE=0, TREE_SIZE_MIN=10, TAIL_SIZE_MIN=2)


class Frangi:
    """Frangi(sigs, zdist, alpha, beta, C, beta_one, beta_two) -- frangi.cpp:34."""

    def __init__(self, sigs, zdist, alpha=0.5, beta=0.5, C=500.0, beta_one=0.5, beta_two=15.0, device=0):
        self.sig = list(sigs)
        self.zdist = zdist
        self.p = make_params(sigmas=sigs, zdist=zdist, alpha=alpha, beta=beta, C=C)
        self.ctx = Context(self.p, device)

    def frangi3d(self, I):
        """-> J, Jmin, Jmax, Vx, Vy, Vz  (frangi.cpp:152; I is uint8 [l][h][w])."""
        self.ctx.set_volume(I)
        jmin, jmax = self.ctx.frangi()
        o = self.ctx.get_frangi(J=True, J8=False, V=True)
        return o["J"], jmin, jmax, o["Vx"], o["Vy"], o["Vz"]

    def j8(self):
        """J -> J8 of Advantra_plugin.cpp:2499-2512 for the last frangi3d call."""
        return self.ctx.get_frangi(J=False, J8=True, V=False)["J8"]

    def imgaussian(self, I, sig):
        self.ctx.set_volume(I)
        return self.ctx.gaussian(sig)

    def hessian3d(self, I, sig):
        self.ctx.set_volume(I)
        return self.ctx.hessian(sig)


class SeedExtractor:
    @staticmethod
    def extractSeeds(tolerance, J8, Vx, Vy, Vz, ctx=None, device=0):
        """SeedExtractor::extractSeeds(tolerance, J8, w,h,l, Vx,Vy,Vz, seeds) -- seed.cpp:556."""
        if ctx is None:
            ctx = Context(make_params(tolerance=tolerance), device)
        elif abs(ctx.p.tolerance - tolerance) > 0:
            raise PnrError("context was created with a different tolerance")
        if ctx.shape != J8.shape:
            ctx.set_volume(np.zeros(J8.shape, np.uint8))  # dimensions only
        ctx.set_j8_v(J8, Vx, Vy, Vz)
        return ctx.extract_seeds()


class Tracker:
    """Tracker(sigs, step, npcles, niter, kappa, is2d, znccth, Kc, neff_ratio, zdist, nodespervol)
    -- tracker.cpp:79.  `rng_seed` pins what the reference draws from srand(time(NULL))."""

    def __init__(self, sigs, step, npcles, niter, kappa, is2d, znccth, Kc=20.0, neff_ratio=0.8, zdist=2.0, nodespervol=4,
                 vol=1, rng_seed=42, device=0):
        self.is2d = bool(is2d)  # P == 1: the 2-D tables are built when a single-slice stack is set (set_image)
        self.p = make_params(sigmas=sigs, step=step, np_=npcles, ni=niter, kappa=kappa, znccth=znccth, zdist=zdist,
                             nodepervol=nodespervol, vol=vol, rng_seed=rng_seed, Kc=Kc, neff_ratio=neff_ratio)
        self.ctx = Context(self.p, device)
        self.sz = len(self.ctx.table("w0"))
        self.ndir = len(self.ctx.table("v")) // 3

    def set_image(self, img):
        if (img.shape[0] == 1) != self.is2d:
            raise PnrError("Tracker(is2d=%s) needs a %s stack" % (self.is2d, "single-slice" if self.is2d else "multi-slice"))
        self.ctx.set_volume(img)
        self.sz = len(self.ctx.table("w0"))
        self.ndir = len(self.ctx.table("v")) // 3

    def znccBBB(self, pos_dir):
        """corr, sig for n poses (x,y,z,vx,vy,vz) -- tracker.cpp:1891."""
        return self.ctx.zncc(pos_dir)

    def track(self, seeds, dbg_iters=0):
        """trackPos + trackNeg filters (tracker.cpp:819-933 -> iter0New/iterINew) for every seed on
        the GPU: returns T, stop, xc[, debug taps] for traces 2*i (pos) and 2*i+1 (neg)."""
        return self.ctx.trace_batch(seeds, dbg_iters)

    def replay(self, seeds, T, xc):
        """sequential bookkeeping of trackPos over the traces, in seed order -> nodes, links."""
        return self.ctx.replay(seeds, T, xc)


# ---------------------------------------------------------------------------------------------
def _load_stack(path):
    if path.endswith(".npy"):
        a = np.load(path)
    else:
        from PIL import Image  # multi-page 8-bit TIFF (simple_loadimage_wrapper's role, Advantra_plugin.cpp:2241)
        im = Image.open(path)
        a = np.stack([np.array(im.copy()) for _ in _frames(im)])
    if a.ndim != 3:
        raise PnrError("need a 3-D stack")
    return np.ascontiguousarray(a.astype(np.uint8))


def _frames(im):
    i = 0
    while True:
        try:
            im.seek(i)
        except EOFError:
            return
        yield i
        i += 1


def print_help():
    print("**** usage of Advantra tracing **** \n"
          "vaa3d -x Advantra -f advantra_func -i <inimg_file> -p <neuritesigmas> <somaradius> <tolerance> <znccth> <kappa> "
          "<step> <ni> <np> <zdist> <nodepervol> <vol>\n")


def write_swc(path, nodes, links, sig2r=1.0, name="Advantra", comment="", type_override=-1):
    """save_nodelist (Advantra_plugin.cpp:480-523): one line per (node, neighbour) pair, ids repeat,
    parent = -1 for isolated nodes; node 0 (dummy) is skipped."""
    nbr = [[] for _ in range(len(nodes))]
    for a, b in links:  # a.nbr.push_back(b); b.nbr.push_back(a)
        nbr[a].append(int(b))
        nbr[b].append(int(a))
    with open(path, "w") as f:
        f.write(f"#name {name}\n")
        for ln in comment.split("\n"):
            if ln:
                f.write(("#comment " if not ln.startswith("#") else "") + ln + "\n")
        f.write("##n,type,x,y,z,radius,parent\n")
        for i in range(1, len(nodes)):
            nd = nodes[i]
            t = int(nd["type"]) if type_override == -1 else type_override
            for par in (nbr[i] or [-1]):
                f.write(f"{i} {t} {nd['x']:.3f} {nd['y']:.3f} {nd['z']:.3f} {sig2r * nd['sig']:.3f} {par}\n")


def write_swc_tree(path, tree, parent, sig2r=1.0, name="Advantra", comment="", type_override=-1):
    """save_nodelist for a tree list (each node carries 0 or 1 link: its parent)."""
    with open(path, "w") as f:
        f.write(f"#name {name}\n")
        for ln in comment.split("\n"):
            if ln:
                f.write(("#comment " if not ln.startswith("#") else "") + ln + "\n")
        f.write("##n,type,x,y,z,radius,parent\n")
        for i in range(1, len(tree)):
            nd = tree[i]
            t = int(nd["type"]) if type_override == -1 else type_override
            f.write(f"{i} {t} {nd['x']:.3f} {nd['y']:.3f} {nd['z']:.3f} {sig2r * nd['sig']:.3f} {int(parent[i])}\n")


def _comment(paras, channel=1):
    keys = ["neuritesigmas", "somaradius", "tolerance", "znccth", "kappa", "step", "ni", "np", "zdist", "nodepervol", "vol"]
    s = "email: miro@braincadet.com\n#params:\n#channel=%d" % channel
    for k, v in zip(keys, paras):
        s += f"\n#{k}={v}"
    s += "\n#------------------------"
    for k, v in CONSTS.items():
        s += f"\n#{k}={v:g}" if isinstance(v, float) else f"\n#{k}={v}"
    return s


def advantra_func(infiles, paras, device=0, rng_seed=42, image=None, verbose=True, out_suffix=""):
    """Advantra::dofunc(\"advantra_func\", ...) (Advantra_plugin.cpp:274-337): `infiles` = list of
    image paths (first is used), `paras` = the 11 positional parameters as strings.  Returns False
    on a usage error (missing image / wrong parameter count), 0 on a range error, True on success.
    Writes <inimg>_Advantra<suffix>.swc.  `image` may supply the stack directly (tests)."""
    import sys
    if not infiles and image is None:
        print("Need input image. ", file=sys.stderr)
        return False
    if len(paras) != NRINPUTPARAMS:
        print(f"\nNeeds {NRINPUTPARAMS} input parameters.\n", file=sys.stderr)
        print_help()
        return False
    sig = sorted(float(s) for s in str(paras[0]).split(",") if s != "")
    somaradius, tolerance, znccth, kappa = int(paras[1]), float(paras[2]), float(paras[3]), float(paras[4])
    step, ni, npc, zdist, nodepervol, vol = int(paras[5]), int(paras[6]), int(paras[7]), float(paras[8]), int(paras[9]), int(paras[10])
    checks = [(somaradius < 0, "somaradius out of range"), (tolerance < 0, "tolerance out of range"),
              (znccth < 0 or znccth > 1, "znccth out of range"), (kappa < 0 or kappa > 5, "kappa out of range"),
              (step < 1, "step out of range"), (ni <= 0, "ni out of range"), (npc <= 0, "np out of range"),
              (zdist < 1, "zdist out of range"), (nodepervol <= 2 or nodepervol > 20, "nodepervol out of range"),
              (vol not in (1, 5, 9, 11, 19, 27), "vol can be 1,5,9,11,19,27")]
    for bad, msg in checks:
        if bad:
            print(msg, file=sys.stderr)  # v3d_msg(...)
            return 0
    img = image if image is not None else _load_stack(infiles[0])
    p = make_params(sigmas=sig, somaradius=somaradius, tolerance=tolerance, znccth=znccth, kappa=kappa, step=step, ni=ni,
                    np_=npc, zdist=zdist, nodepervol=nodepervol, vol=vol, rng_seed=rng_seed)
    ctx = Context(p, device)
    res = run_pipeline(ctx, img, verbose=verbose)
    if infiles:
        out = f"{infiles[0]}_Advantra{out_suffix}.swc"
        write_swc_tree(out, res["tree"], res["parent"], comment=_comment(paras))
        res["swc"] = out
    advantra_func.last = res
    ctx.close()
    return True


def run_pipeline(ctx, img, verbose=False, max_seeds=None, one_shot=False, reconstruct=True):
    """reconstruction_func's hot path (Advantra_plugin.cpp:2488-2710): Frangi -> J8 -> seeds ->
    score/filter/sort -> trace all seeds on the GPU -> host replay."""
    import time
    t = [time.time()]
    ctx.set_volume(img) if isinstance(img, np.ndarray) else None
    soma = ctx.soma() if ctx.p.somaradius > 0 else None  # SOMA EXTR. (Advantra_plugin.cpp:2426-2486), before Frangi
    t[0] = time.time() if soma is None else t[0]
    jmin, jmax = ctx.frangi(); t.append(time.time())
    seeds_init = ctx.extract_seeds(); t.append(time.time())
    seeds = ctx.score_filter_sort(seeds_init); t.append(time.time())
    if max_seeds is not None:
        seeds = seeds[:max_seeds]
    if one_shot:  # every seed traced to its map-free end, then one replay (keeps T / xc for inspection)
        T, stop, xc, _ = ctx.trace_batch(seeds); t.append(time.time())
        nodes, links, ntr = ctx.replay(seeds, T, xc); t.append(time.time())
        iters = int((T + (T < ctx.p.ni)).sum())
    else:         # production form: seed-rank batches with early DENSITY stops (same node graph)
        T = stop = xc = None
        nodes, links, ntr, iters = ctx.trace_replay(seeds); t.append(time.time()); t.append(time.time())
    tree, parent = lib.reconstruct(nodes, links) if reconstruct else (None, None)  # reconstruct(n0, ...) :2729
    t.append(time.time())
    if verbose:
        names = ["frangi", "seed extraction", "seed selection & sorting", "tracing", "replay", "reconstruct"]
        for nm, a, b in zip(names, t[:-1], t[1:]):
            print(f"{nm}... {b - a:.3f} sec.")
        print(f"{len(seeds_init) / 1000.0}k seeds -> {len(seeds) / 1000.0}k seeds, {ntr} traces, {len(nodes) - 1} nodes")
    return dict(soma=soma, Jmin=jmin, Jmax=jmax, seeds_init=seeds_init, seeds=seeds, T=T, stop=stop, xc=xc, nodes=nodes, links=links,
                ntraces=ntr, iters=iters, tree=tree, parent=parent, times=np.diff(t))
