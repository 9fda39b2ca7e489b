"""A fixed slice of the randomised parity campaign (tests/fuzzcase.py; scripts/fuzz_parity.py runs it open-ended) inside `-m gpu`:
random small stacks and parameters -- shape, 1..4 scales, zdist, np, ni, step, kappa, tolerance, znccth, nodepervol, vol, planted
cell bodies with somaradius > 0, single-slice stacks, scheduler knobs --, the HIP path against the oracle stage by stage.  Every
comparison inside a case is for EQUALITY OF BYTES (J, J8, V, seeds, seed scores, T / stop reason / every centroid estimate of
every trace, the streamed against the one-shot node graph, replay and reconstruct() against the oracle's)."""
import pytest
import fuzzcase

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first,count,driver", [(0, 20, None), (20, 20, None), (5000, 10, "persistent"), (7000, 6, None)])
def test_fuzz_cases_bit_exact(oracle, first, count, driver):
    stats = {}
    for case in range(first, first + count):
        desc = {}
        try:
            fuzzcase.run_case(oracle, case, stats, desc, driver=driver, big=(first == 7000))
        except AssertionError as e:
            raise AssertionError(f"case {desc}: {e}") from e
    assert stats.get("traces", 0) > count and stats.get("voxels", 0) > 0
