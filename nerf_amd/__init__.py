"""nerf_amd -- MI355X (gfx950) implementation of NeRF's per-ray render core behind the reference's
render() / render_rays() / create_nerf() Python API.

    from nerf_amd.run_nerf import render, create_nerf, batchify_rays, render_path
    from nerf_amd.run_nerf_helpers import get_rays, ndc_rays, sample_pdf, get_embedder, init_nerf_model
    from nerf_amd.nerf_renderer import render_rays, run_network, raw2outputs

Python is host glue only: tensors are torch (ROCm) tensors and every arithmetic step runs in
nerf_amd/csrc/libnerfhip.so (C ABI in include/nerf_hip.h).  Importing this package does not load
the library; the first call does, and fails loudly if it has not been built.
"""
__version__ = "0.1.0"
