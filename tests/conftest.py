import os
import sys
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def default_options():
    """PNR_TEST_OPTIONS="key=value,key=value": scheduler / kernel-form options every Context of the session starts with (the results
    must not depend on them: running the suite with e.g. sums_run_max=100000 puts every launch through the form that option
    selects)"""
    spec = os.environ.get("PNR_TEST_OPTIONS", "")
    if spec:
        import pnr_amd.lib as lib
        lib.DEFAULTS["options"] = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}
    yield


@pytest.fixture(scope="session", autouse=True)
def built_tree():
    """A fresh checkout has no binaries: compile the HIP library (hipcc cross-compiles without a GPU) and the head-less CLI once
    per session, as __graft_entry__.build() does.  This only builds -- pnr_amd.lib.load() still fails loudly without the .so."""
    import subprocess
    import pnr_amd.lib as lib
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    if not os.path.exists(os.path.join(ROOT, "pnr_amd", "host", "advantra_cli")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "pnr_amd", "host")], check=True)


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.load_oracle()


@pytest.fixture(scope="session")
def ref():
    import orc
    return orc.load_ref()


def golden_cases():
    gd = os.path.join(HERE, "golden")
    return sorted(f[:-4] for f in os.listdir(gd) if f.startswith("g") and f.endswith(".npz"))


@pytest.fixture(scope="session", params=golden_cases())
def golden(request):
    import numpy as np
    return dict(np.load(os.path.join(HERE, "golden", request.param + ".npz")))
