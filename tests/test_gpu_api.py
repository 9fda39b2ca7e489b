"""Small contracts of the C ABI that the pipeline tests do not touch: options, the graph kept in the context, the trace log."""
import numpy as np
import pytest
import synth
import pnr_amd
from pnr_amd import lib

pytestmark = pytest.mark.gpu


def test_options_roundtrip_and_errors(monkeypatch):
    monkeypatch.setitem(lib.DEFAULTS, "options", {})  # (this test is about the library's own defaults: no PNR_TEST_OPTIONS here)
    c = pnr_amd.Context(pnr_amd.make_params(sigmas=[2.0], np_=20, ni=5), 0)
    assert c.get_option("groups") == 0 and c.get_option("window") == 0 and c.get_option("look_pct") == -1
    c.set_option("window", 64)
    assert c.get_option("window") == 64
    assert 1 <= c.get_option("host_threads_effective") <= 32
    c.set_option("local_ranks", 4)
    c.set_option("host_threads", 3)
    assert c.get_option("host_threads_effective") == 3
    with pytest.raises(pnr_amd.PnrError, match="unknown option"):
        c.set_option("no_such_knob", 1)
    with pytest.raises(pnr_amd.PnrError, match="outside"):
        c.set_option("groups", 9)
    with pytest.raises(pnr_amd.PnrError, match="no node graph"):
        c.get_graph()
    assert c.set_options("poll=3, groups=1") == {"poll": 3, "groups": 1} and c.get_option("poll") == 3


def test_trace_log_accounts_for_every_node():
    """option trace_log: one record per replayed trace; the iterations they kept add up to the nodes of the graph, and the reasons
    are the reference's four (tracker.cpp:866,879,908,916)"""
    img = synth.synth(80, 64, 32, seed=4)
    p = pnr_amd.make_params(sigmas=[2.0], np_=32, ni=40, zdist=2.0, nodepervol=3, vol=5)
    c = pnr_amd.Context(p, 0)
    c.set_volume(img)
    c.frangi()
    seeds = c.score_filter_sort(c.extract_seeds())
    c.set_option("trace_log", 1)
    nodes, links, ntr, iters = c.trace_replay(seeds)
    log = c.trace_log()
    assert len(log) == 2 * ntr > 20
    assert set(np.unique(log[:, 3])) <= {0, 1, 2, 3} and (log[:, 3] == 2).sum() > 0  # DENSITY stops happen on this stack
    assert int(log[:, 2].sum()) == len(nodes) - 1
    assert np.all(np.diff(log[:, 0]) >= 0) and set(np.unique(log[:, 1])) == {0, 1}  # seed order, both directions
    dens = log[log[:, 3] == 2]
    assert np.all(dens[:, 4] == p.nodepervol)
    n2, l2 = c.get_graph()
    assert len(n2) == len(nodes) and np.array_equal(l2, links)
    c.set_option("trace_log", 0)
    c.trace_replay(seeds[:5])
    assert len(c.trace_log()) == 0
