"""The C++ host side (pnr_amd/host): advantra_func contract of Advantra::dofunc
(Advantra_plugin.cpp:274-337) behind the head-less CLI, and -- on the GPU -- the same node graph /
SWC as the Python mirror for a multi-page TIFF input."""
import os
import subprocess
import sys
import numpy as np
import pytest
import synth
import pnr_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "pnr_amd", "host", "advantra_cli")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import swc_diff  # noqa: E402  -- the numeric SWC diff tool (ids / types / parents identical, x / y / z / r within 2e-3)


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300)


def test_cli_usage_errors():
    if not os.path.exists(CLI):
        subprocess.run(["make", "-s", "-C", os.path.dirname(CLI)], check=True)
    r = run("-f", "advantra_func", "-i", "x.tif", "-p", "2,4,6", "0", "5")
    assert r.returncode == 1 and "Needs 11 input parameters." in r.stderr and "usage of Advantra" in r.stdout
    r = run("-f", "advantra_func", "-p", *("2,4,6 0 5 0.3 3 2 200 20 2 4 1".split()))
    assert r.returncode == 1 and "Need input image" in r.stderr
    for bad, msg in (("2,4,6 0 5 0.3 9 2 200 20 2 4 1", "kappa out of range"), ("2,4,6 0 5 0.3 3 2 200 20 2 4 3", "vol can be 1,5,9,11,19,27"),
                     ("2,4,6 0 5 0.3 3 2 200 20 2 2 1", "nodepervol out of range"), ("2,4,6 0 5 1.3 3 2 200 20 2 4 1", "znccth out of range")):
        r = run("-f", "advantra_func", "-i", "x.tif", "-p", *bad.split())
        assert r.returncode == 0 and msg in r.stderr  # dofunc: v3d_msg + return 0, then "return true"
    assert run("-f", "help").returncode == 0
    assert run("-f", "nonsense").returncode == 1


def test_cli_fails_loudly_without_a_gpu(tmp_path):
    """a valid call on a machine without an MI355X: the device library's error on stderr, no SWC, exit status 1 (there is no CPU
    path; the reference's own contract -- status 0 after a parameter / image message -- only covers its own failure modes)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    if not os.path.exists(CLI):
        subprocess.run(["make", "-s", "-C", os.path.dirname(CLI)], check=True)
    raw = str(tmp_path / "s.raw")
    synth.synth(32, 24, 8, seed=1).tofile(raw)
    for extra in ((), ("--ranks", "2")):
        r = run("-d", "32,24,8", *extra, "-f", "advantra_func", "-i", raw, "-p", *"2 0 5 0.3 3 2 20 20 2 4 5".split())
        assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)
        assert not os.path.exists(raw + "_Advantra.swc")


@pytest.mark.gpu
def test_cli_matches_python_pipeline(tmp_path):
    from PIL import Image
    img = synth.synth(64, 56, 32, seed=2)
    tif = str(tmp_path / "stack.tif")
    pages = [Image.fromarray(z) for z in img]
    pages[0].save(tif, save_all=True, append_images=pages[1:], compression=None)
    paras = "2,3 0 5 0.3 3 2 40 50 2 4 5".split()
    r = run("-f", "advantra_func", "-i", tif, "-p", *paras)
    assert r.returncode == 0, r.stderr
    swc = tif + "_Advantra.swc"
    assert os.path.exists(swc)
    p = pnr_amd.make_params(sigmas=[2, 3], tolerance=5, znccth=0.3, kappa=3, step=2, ni=40, np_=50, zdist=2, nodepervol=4, vol=5)
    ctx = pnr_amd.Context(p, 0)
    res = pnr_amd.advantra.run_pipeline(ctx, img)
    ref = str(tmp_path / "ref.swc")
    pnr_amd.write_swc_tree(ref, res["tree"], res["parent"])
    ok, msg = swc_diff.diff(swc, ref, tol=2e-3)
    assert ok and len(swc_diff.read_swc(swc)[0]) > 50, msg
    # the tool as a command: equal files -> 0, a moved node -> 1 with the first difference named
    tool = [sys.executable, os.path.join(ROOT, "scripts", "swc_diff.py")]
    assert subprocess.run(tool + [swc, ref], capture_output=True, text=True).returncode == 0
    lines = open(ref).read().splitlines()
    k = next(i for i, ln in enumerate(lines) if ln and ln[0] != "#") + 3
    t = lines[k].split()
    t[3] = f"{float(t[3]) + 0.5:.3f}"
    moved = str(tmp_path / "moved.swc")
    open(moved, "w").write("\n".join(lines[:k] + [" ".join(t)] + lines[k + 1:]) + "\n")
    r2 = subprocess.run(tool + [swc, moved], capture_output=True, text=True)
    assert r2.returncode == 1 and "y differs at node #3" in r2.stdout, r2.stdout


@pytest.mark.gpu
def test_cli_config0_literal_vs_oracle(tmp_path, oracle):
    """BASELINE configs[0] as written: a 128 x 128 x 64 synthetic TIFF, scales = {2}, 50 particles, README parameters otherwise
    (ni 200, step 2, zdist 2, nodepervol 4, vol 1), end to end through advantra_func of the C++ host (advantra_cli -> C ABI -> HIP)
    into an SWC file -- against the oracle's whole pipeline on the same stack: Frangi -> J8 -> extractSeeds -> znccBBB filter + sort
    -> every trace to its map-free end -> trackPos / trackNeg replay -> reconstruct() -> SWC.  Node and trace counts equal, SWC equal
    (ids, types, parents identical; coordinates and radii within the SWC text precision)."""
    import re
    import orc
    from PIL import Image
    img = synth.synth(128, 128, 64, seed=1)
    tif = str(tmp_path / "config0.tif")
    pages = [Image.fromarray(z) for z in img]
    pages[0].save(tif, save_all=True, append_images=pages[1:], compression=None)
    r = run("-f", "advantra_func", "-i", tif, "-p", *"2 0 5 0.3 3 2 200 50 2 4 1".split())
    assert r.returncode == 0, r.stderr
    swc = tif + "_Advantra.swc"
    assert os.path.exists(swc)
    sigs, np_, ni, zdist = [2.0], 50, 200, 2.0
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, sigs, zdist)
    s = orc.extract_seeds(oracle, 5, orc.j8(oracle, J, jmin, jmax), Vx, Vy, Vz)
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    corr, _ = To.zncc(img, s[:, :6])
    s[:, 7] = corr
    s = s[corr >= 0.3]
    s = s[np.argsort(-s[:, 7], kind="stable")]
    Ts, xcs = [], []
    for sd in s:
        for sgn in (1, -1):
            q = sd[:6].copy()
            q[3:] *= sgn
            Tn, _, xc, *_ = To.trace(img, q)
            Ts.append(Tn)
            xcs.append(xc)
    nodes_o, links_o, nt = orc.replay(oracle, s, np.array(Ts, np.int32), np.stack(xcs), ni, img.shape, 4, 1)
    tree_o, par_o = orc.reconstruct(oracle, nodes_o, links_o)
    m = re.search(r"(\d+) trace nodes, (\d+) traces", r.stdout)
    assert m and int(m.group(1)) == len(nodes_o) - 1 and int(m.group(2)) == nt, (m and m.groups(), len(nodes_o) - 1, nt)
    ref = str(tmp_path / "oracle.swc")
    pnr_amd.write_swc_tree(ref, tree_o, par_o)
    ok, msg = swc_diff.diff(swc, ref, tol=2e-3)
    assert ok and len(swc_diff.read_swc(swc)[0]) > 300, msg


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["soma", "slice"])
def test_cli_soma_and_single_slice(tmp_path, case):
    """the C++ host runs the soma path for somaradius > 0 and the 2-D mode for a one-page TIFF, like the Python mirror"""
    from PIL import Image
    if case == "soma":
        img = synth.add_somas(synth.synth(64, 56, 32, seed=2), ((20, 28, 16, 6), (48, 20, 14, 5)))
        paras, kw = "2,3 3 5 0.3 3 2 25 40 2 4 1".split(), dict(somaradius=3, ni=25, np_=40, vol=1)
    else:
        img = np.ascontiguousarray(synth.synth(96, 80, 9, seed=4).max(0, keepdims=True))
        paras, kw = "2,3 0 5 0.3 3 2 25 40 2 4 1".split(), dict(somaradius=0, ni=25, np_=40, vol=1)
    tif = str(tmp_path / "stack.tif")
    pages = [Image.fromarray(z) for z in img]
    pages[0].save(tif, save_all=True, append_images=pages[1:], compression=None)
    r = run("-f", "advantra_func", "-i", tif, "-p", *paras)
    assert r.returncode == 0, r.stderr
    if case == "soma":
        assert "soma regions" in r.stdout
    ctx = pnr_amd.Context(pnr_amd.make_params(sigmas=[2, 3], tolerance=5, znccth=0.3, kappa=3, step=2, zdist=2, nodepervol=4, **kw), 0)
    res = pnr_amd.advantra.run_pipeline(ctx, img)
    ref = str(tmp_path / "ref.swc")
    pnr_amd.write_swc_tree(ref, res["tree"], res["parent"])
    ok, msg = swc_diff.diff(tif + "_Advantra.swc", ref, tol=2e-3)
    ids, types, _, _ = swc_diff.read_swc(tif + "_Advantra.swc")
    assert ok and len(ids) > 20, msg
    if case == "soma":
        assert (types == 1).sum() >= 1  # a SOMA-typed node in the SWC


def test_cli_rejects_malformed_tiff(tmp_path):
    """a TIFF whose directory chain loops, whose value count exceeds the file or whose page is larger than the file ends with a
    message, not with an endless loop or a multi-gigabyte allocation (CPU only: the loader runs before any device call)"""
    import struct
    paras = "2 0 5 0.3 3 2 10 20 2 4 1".split()

    def tiff(entries, next_ifd, extra=b""):
        body = struct.pack("<H", len(entries)) + b"".join(struct.pack("<HHII", *e) for e in entries) + struct.pack("<I", next_ifd)
        return b"II" + struct.pack("<HI", 42, 8) + body + extra

    base = [(256, 4, 1, 4), (257, 4, 1, 4), (258, 3, 1, 8), (259, 3, 1, 1), (277, 3, 1, 1)]
    strip = [(273, 4, 1, 8 + 2 + 12 * 7 + 4), (279, 4, 1, 16)]
    cases = {
        "loop.tif": (tiff(base + strip, 8, b"\x07" * 16), "loops"),                                     # IFD points at itself
        "count.tif": (tiff(base + [(273, 4, 0x40000000, 200), (279, 4, 1, 16)], 0, b"\x07" * 16), ""),   # 2^30 strip offsets in a 130-byte file
        "huge.tif": (tiff([(256, 4, 1, 60000), (257, 4, 1, 60000)] + base[2:] + strip, 0, b"\x07" * 16), "larger than the file"),
    }
    for name, (blob, msg) in cases.items():
        f = tmp_path / name
        f.write_bytes(blob)
        r = run("-f", "advantra_func", "-i", str(f), "-p", *paras)
        assert r.returncode == 0 and r.stderr.strip() and msg in r.stderr, (name, r.stderr)  # dofunc prints and returns, no crash
        assert not os.path.exists(str(f) + "_Advantra.swc")


@pytest.mark.gpu
def test_cli_verbose_prints_the_reference_trace_lines(tmp_path):
    """-v: per-trace progress and stop reasons in the reference's words (Advantra_plugin.cpp:2677; tracker.cpp:866,879,908,916);
    --save-midres: the node graph before reconstruct() (:2099); same SWC as without the flags"""
    from PIL import Image
    img = synth.synth(64, 56, 32, seed=2)
    tif = str(tmp_path / "stack.tif")
    pages = [Image.fromarray(z) for z in img]
    pages[0].save(tif, save_all=True, append_images=pages[1:], compression=None)
    paras = "2,3 0 5 0.3 3 2 40 50 2 4 5".split()
    quiet = run("-f", "advantra_func", "-i", tif, "-p", *paras)
    swc_quiet = open(tif + "_Advantra.swc").read()
    r = run("-v", "--save-midres", "-f", "advantra_func", "-i", tif, "-p", *paras)
    assert r.returncode == 0 and quiet.returncode == 0, r.stderr
    assert open(tif + "_Advantra.swc").read() == swc_quiet and os.path.exists(tif + "_n0_.swc")
    for tap in ("_n0res_", "_n1_", "_n2_", "_n2tree_"):  # the other saveMidres lists of reconstruct() (:2112-2141)
        ids = swc_diff.read_swc(tif + tap + ".swc")[0]
        assert len(ids) > 50, tap
    # --single-tree: the plugin's ENFORCE_SINGLE_TREE branch (:2142-2152) writes the largest tree to <inimg>_Advantra1.swc
    r1 = run("--single-tree", "-f", "advantra_func", "-i", tif, "-p", *paras)
    assert r1.returncode == 0 and os.path.exists(tif + "_Advantra1.swc")
    ids1, _, _, par1 = swc_diff.read_swc(tif + "_Advantra1.swc")
    ids_all, _, _, par_all = swc_diff.read_swc(tif + "_Advantra.swc")
    assert 10 < len(ids1) <= len(ids_all) and (np.asarray(par1) == -1).sum() == 1 and (np.asarray(par_all) == -1).sum() >= 1
    out = r.stdout
    ntr = out.count("\nTrace: ")
    ends = sum(out.count(k) for k in ("], DENSITY, nodespervol=4", "], success=0, corr=", "], TRACK LIMIT, niter=40", "], SOMA, idx="))
    assert ntr > 5 and ends == 2 * ntr  # trackPos + trackNeg of every trace that was used
    assert "% seeds used" in out and "seed extraction..." in out and "Trace: " not in quiet.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,soma", [(2, False), (3, False), (2, True)])
def test_cli_ranks_write_the_one_gpu_swc(tmp_path, ranks, soma):
    """advantra_cli --ranks N: N forked processes (here sharing the one GPU), z-slabs of Frangi / seeds, sorted seeds dealt round-robin,
    finished traces exchanged through the shared-memory all-gather, replay on every rank -- the SWC rank 0 writes is byte for byte the
    SWC of the single-process run; with somaradius > 0 as well"""
    from PIL import Image
    img = synth.synth(64, 56, 32, seed=2)
    if soma:
        img = synth.add_somas(img, ((20, 28, 16, 6), (48, 20, 14, 5)))
    paras = ("2,3 3 5 0.3 3 2 25 40 2 4 1" if soma else "2,3 0 5 0.3 3 2 40 50 2 4 5").split()
    outs = []
    for n in (1, ranks):
        d = tmp_path / f"r{n}"
        d.mkdir()
        tif = str(d / "stack.tif")
        pages = [Image.fromarray(z) for z in img]
        pages[0].save(tif, save_all=True, append_images=pages[1:], compression=None)
        flags = ["--ranks", str(n), "--share-gpu"] if n > 1 else []
        r = run(*flags, "-f", "advantra_func", "-i", tif, "-p", *paras)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(open(tif + "_Advantra.swc").read())
        assert "seeds used" in r.stdout
    assert outs[0] == outs[1] and outs[0].count("\n") > 30
