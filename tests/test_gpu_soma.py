"""GPU parity of the soma path (somaradius > 0) through the C ABI: eroded + blurred stack, threshold, soma
nodes and label map vs the oracle (bit-exact: bytes, integers and f32 running means in the same order), then the whole
path -- seeds in the soma dropped, traces that reach the soma stopped and linked to its node, final tree list."""
import numpy as np
import pytest
import orc
import synth
import pnr_amd
from pnr_amd import lib

pytestmark = pytest.mark.gpu


def _stack(shape=(32, 56, 64), seed=2, somas=((20, 28, 16, 6), (48, 20, 14, 5))):
    l, h, w = shape
    return synth.add_somas(synth.synth(w, h, l, seed=seed), somas)


@pytest.mark.parametrize("shape,rad,somas", [((32, 56, 64), 3, ((20, 28, 16, 6), (48, 20, 14, 5))), ((9, 20, 23), 2, ((10, 10, 4, 4),)),
                                             ((24, 40, 48), 4, ())])
def test_soma_extraction_vs_oracle(oracle, shape, rad, somas):
    img = _stack(shape, 3, somas)
    E8o, tho, smapo, n4o = orc.soma_extract(oracle, img, rad)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], somaradius=rad, np_=20, ni=5), 0)
    c.set_volume(img)
    s = c.soma(want_e8=True)
    assert np.array_equal(s["E8"], E8o)
    assert s["threshold"] == tho
    assert len(s["nodes"]) == len(n4o)
    got4 = np.stack([s["nodes"][k] for k in ("x", "y", "z", "sig")], -1) if len(n4o) else np.zeros((0, 4), np.float32)
    assert np.array_equal(got4, n4o)
    assert np.all(s["nodes"]["type"] == 1) and np.all(s["nodes"]["corr"] == -np.finfo(np.float32).max)
    fg = np.flatnonzero(smapo.reshape(-1) > 0)
    assert np.array_equal(s["vox"], fg) and np.array_equal(s["lab"], smapo.reshape(-1)[fg])
    if somas:
        assert len(n4o) >= 1 and len(fg) > 20


def test_no_soma_when_radius_zero():
    img = _stack()
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], somaradius=0, np_=20, ni=5), 0)
    c.set_volume(img)
    s = c.soma()
    assert len(s["nodes"]) == 0 and len(s["vox"]) == 0


def test_filter_needs_soma_first():
    img = _stack()
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], somaradius=3, np_=20, ni=5), 0)
    c.set_volume(img)
    c.frangi()
    with pytest.raises(lib.PnrError, match="pnr_soma"):
        c.score_filter_sort(c.extract_seeds())


@pytest.mark.parametrize("driver", ["phased", "persistent"])
def test_end_to_end_with_soma_vs_oracle(oracle, driver):
    """somaradius = 3 on a stack with two cell bodies: oracle = soma path + Frangi + seeds (those inside the soma dropped,
    Advantra_plugin.cpp:2561-2564) + tracker + replay with the soma stop (tracker.cpp:858-869) + reconstruct chain."""
    img = _stack()
    sigs, zdist, np_, ni, rad = [2.0, 3.0], 2.0, 40, 25, 3
    E8o, tho, smapo, n4o = orc.soma_extract(oracle, img, rad)
    assert len(n4o) >= 1
    J, jmin, jmax, Vx, Vy, Vz = orc.frangi3d(oracle, img, sigs, zdist)
    so = orc.extract_seeds(oracle, 5, orc.j8(oracle, J, jmin, jmax), Vx, Vy, Vz)
    l, h, w = img.shape
    vox = np.round(so[:, 2]).astype(np.int64) * w * h + np.round(so[:, 1]).astype(np.int64) * w + np.round(so[:, 0]).astype(np.int64)
    keep = smapo.reshape(-1)[vox] == 0
    assert (~keep).sum() > 0  # some seeds sit inside a soma
    so = so[keep]
    To = orc.Tracker(oracle, sigs, 2, np_, ni, 3.0, 0.3, zdist=zdist)
    corr, _ = To.zncc(img, so[:, :6])
    so[:, 7] = corr
    so = so[corr >= 0.3]
    so = so[np.argsort(-so[:, 7], kind="stable")]
    Tn, xcs = [], []
    for sd in so:
        for sgn in (1, -1):
            q = sd[:6].copy(); q[3:] *= sgn
            t, st, xco, *_ = To.trace(img, q)
            Tn.append(t); xcs.append(xco)
    nodes_o, links_o, nt_o = orc.replay(oracle, so, np.array(Tn, np.int32), np.stack(xcs), ni, img.shape, 4, 1, smap=smapo, soma4=n4o)
    tree_o, par_o = orc.reconstruct(oracle, nodes_o, links_o)

    p = pnr_amd.make_params(sigmas=sigs, somaradius=rad, np_=np_, ni=ni, zdist=zdist)
    c = pnr_amd.Context(p, 0)
    c.set_smc_driver(driver)
    res = pnr_amd.advantra.run_pipeline(c, img)
    assert len(res["seeds"]) == len(so)
    nodes, links = res["nodes"], res["links"]
    assert len(nodes) == len(nodes_o) and np.array_equal(links, links_o)
    for k in nodes.dtype.names:
        assert np.array_equal(nodes[k], nodes_o[k]), k
    nsoma = len(n4o)
    assert np.all(nodes["type"][1:1 + nsoma] == 1)
    soma_links = ((links <= nsoma) & (links >= 1)).any(1).sum()
    print("soma nodes", nsoma, "links into a soma", soma_links, "nodes", len(nodes))
    assert soma_links > 0  # a trace reached a cell body and was linked to it
    assert len(res["tree"]) == len(tree_o) and np.array_equal(res["parent"], par_o)
    for k in tree_o.dtype.names:
        assert np.array_equal(res["tree"][k], tree_o[k]), k
    # the one-shot form (trace everything, one replay) gives the same graph
    T, stop, xc, _ = c.trace_batch(res["seeds"])
    n1, l1, _ = c.replay(res["seeds"], T, xc)
    assert len(n1) == len(nodes) and np.array_equal(l1, links)


@pytest.mark.parametrize("name", ["soma_64x56x32_r3", "soma_23x20x9_r2"])
def test_soma_filters_vs_golden(name):
    """eroded + blurred stack against the reference's own Frangi::imerode / u8 Frangi::imgaussian (tests/golden)"""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz")))
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], somaradius=int(g["rad"]), np_=20, ni=5), 0)
    c.set_volume(g["img"])
    s = c.soma(want_e8=True)
    assert np.array_equal(s["E8"], g["blurred"])
