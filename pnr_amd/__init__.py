"""pnr_amd -- MI355X (gfx950) implementation of the PNR / Advantra neurite-tracer hot path:
multi-scale Frangi vesselness, seed extraction and the batched SMC particle tracer, as
hand-written HIP kernels behind a C ABI (include/pnr_hip.h).  See DESIGN.md."""
from . import lib  # noqa: F401
from .lib import Context, Params, PnrError, make_params  # noqa: F401
from .advantra import Frangi, SeedExtractor, Tracker, advantra_func, write_swc, write_swc_tree  # noqa: F401
