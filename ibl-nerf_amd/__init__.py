"""ibl_nerf_amd — MI355X-native forward/inference renderer for IBL-NeRF's hot path.

Only what the path needs: `csrc/` (HIP kernels + the C-ABI of include/iblnerf.h)
and the host-side mirror of the reference's Python seam (`render_decomp`, `render_rays`,
`network_query_fn`, `sample_pdf`, `get_rays`, `create_IBLNeRF`).  Import as `ibl_nerf_amd`
through `_pkg.load()` at the repo root.  The HIP library is required; there is no CPU fallback.
"""
import os as _os

# HIP deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and streams that share a queue run one after the other.  Renderer._render_pair
# renders the two halves of a frame on two streams; beside an RCCL communicator's own streams, four queues leave the pair on ONE queue (measured: no overlap, -3.5 %).
# Read by the HIP runtime when it initialises — i.e. effective if this package is imported before the process's first HIP call; otherwise harmless.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import binding, checkpoint, config, dataset, dist, export, model, renderer, render_views  # noqa: F401,E402
from .model import IBLNeRF, create_IBLNeRF, network_query_fn  # noqa: F401,E402
from .export import render_decomp_path  # noqa: F401,E402
from .renderer import Renderer, get_rays, render_decomp  # noqa: F401,E402

__all__ = ["binding", "checkpoint", "config", "dataset", "dist", "export", "model", "renderer", "render_views", "render_decomp_path", "IBLNeRF", "create_IBLNeRF", "network_query_fn",
           "Renderer", "get_rays", "render_decomp"]
