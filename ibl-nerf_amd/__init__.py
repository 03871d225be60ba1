"""ibl_nerf_amd — MI355X-native forward/inference renderer for IBL-NeRF's hot path.

Only what the path needs: `csrc/` (HIP kernels + the C-ABI of include/iblnerf.h)
and the host-side mirror of the reference's Python seam (`render_decomp`, `render_rays`,
`network_query_fn`, `sample_pdf`, `get_rays`, `create_IBLNeRF`).  Import as `ibl_nerf_amd`
through `_pkg.load()` at the repo root.  The HIP library is required; there is no CPU fallback.
"""
from . import binding, checkpoint, config, dataset, dist, export, model, renderer, render_views  # noqa: F401
from .model import IBLNeRF, create_IBLNeRF, network_query_fn  # noqa: F401
from .export import render_decomp_path  # noqa: F401
from .renderer import Renderer, get_rays, render_decomp  # noqa: F401

__all__ = ["binding", "checkpoint", "config", "dataset", "dist", "export", "model", "renderer", "render_views", "render_decomp_path", "IBLNeRF", "create_IBLNeRF", "network_query_fn",
           "Renderer", "get_rays", "render_decomp"]
