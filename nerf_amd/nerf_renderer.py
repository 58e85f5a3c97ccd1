"""Counterpart of the reference's NERFCO.nerf_renderer module (imported at run_nerf.py:12; its
source is not in the reference tree): run_network, raw2outputs, render_rays.

Semantics follow the call sites in run_nerf.py (:20, :209-214, :216-238, :775, :796-823), the
in-tree simplified twin (tiny_nerf.ipynb, render_rays cell) and sample_pdf's shape contract
(run_nerf_helpers.py:185-213).  Every stage is
one HIP kernel launch on the current stream: depths, fused encoding+MLP, compositing,
resampling+merge, fused encoding+MLP, compositing.
"""
import torch

from . import _native as nv
from . import ops
from . import training
from .run_nerf_helpers import NerfModel


SINGLE_CALL = True  # inference through nerf_render_rays (one C-ABI call per chunk) when nothing else is asked for


def _check_embedders(embed_fn, embeddirs_fn):
    m = getattr(embed_fn, 'multires', None)
    mv = getattr(embeddirs_fn, 'multires', None)
    if m != 10 or mv != 4:
        raise NotImplementedError(
            "the fused MLP kernel encodes points with multires=10 and view directions with "
            f"multires_views=4 (got {m}, {mv}); pass the callables returned by get_embedder")


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64, c2w=None):
    """run_nerf.py:209-214: inputs [N,S,3], viewdirs [N,3] -> [N,S,4] = (r,g,b,sigma) before the
    activations.  Encoding and all twelve Dense layers run inside one kernel, so `netchunk`
    (run_nerf_helpers.py:216-223, a memory bound that does not change results) is not needed."""
    if not isinstance(fn, NerfModel):
        raise TypeError("run_network: network_fn must come from init_nerf_model")
    if viewdirs is None:
        raise NotImplementedError("run_network: the built network takes view directions (use_viewdirs=True)")
    _check_embedders(embed_fn, embeddirs_fn)
    pts = nv.f32(inputs)
    lead = tuple(pts.shape[:-1])
    if pts.dim() == 2:
        pts = pts[:, None, :]
    pts = pts.reshape(pts.shape[0], -1, 3)
    vd = nv.f32(viewdirs, pts.device).reshape(-1, 3)
    raw = ops.mlp_forward_pts(fn.packed(), pts, vd)
    return raw.reshape(lead + (4,))


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0., white_bkgd=False, noise=None):
    """raw [N,S,4], z_vals [N,S], rays_d [N,3] -> rgb_map, disp_map, acc_map, weights, depth_map."""
    raw = nv.f32(raw)
    z = nv.f32(z_vals, raw.device)
    d = nv.f32(rays_d, raw.device)
    if raw_noise_std > 0. and noise is None:
        noise = torch.randn(z.shape, dtype=torch.float32, device=raw.device)
    return ops.composite_fwd(raw, z, d, 3, white_bkgd, noise, raw_noise_std)


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False,
                perturb=0., N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0.,
                verbose=False, perturb_u=None, pdf_u=None, noise0=None, noise1=None, **fork_kwargs):
    """Volumetric rendering of a ray batch [N, 11] = (o, d, near, far, viewdir).

    Returns a dict with rgb_map [N,3], disp_map, acc_map, depth_map [N]; with N_importance > 0 also
    rgb0, disp0, acc0, z_std; with retraw the last network output raw [N,S,4].
    `perturb_u` / `pdf_u` let a caller supply the uniforms otherwise drawn with torch.rand, and
    `noise0` / `noise1` ([N,Sc] / [N,Sc+Si]) the standard normals that raw_noise_std scales in the
    coarse / fine compositing (otherwise drawn with torch.randn).
    Fork-only keyword arguments (choose_keypoint_closest_depth, ...) are accepted and ignored.
    """
    rays = nv.f32(ray_batch)
    n, width = rays.shape
    if width < 11:
        raise NotImplementedError("render_rays: the built network needs view directions (ray width 11)")
    fused = getattr(network_query_fn, 'nerf_amd_fused', False)
    rec = training.recording()  # inside a GradientTape: record for the backward pass
    if rec and not fused:
        raise NotImplementedError("render_rays: gradients need the network_query_fn built by create_nerf")

    if perturb > 0.:
        if perturb_u is None:
            perturb_u = torch.rand((n, N_samples), dtype=torch.float32, device=rays.device)
        perturb_u = nv.f32(perturb_u, rays.device)
        if tuple(perturb_u.shape) != (n, N_samples):
            raise ValueError(f"render_rays: perturb_u must be [{n},{N_samples}]")
    else:
        perturb_u = None

    # Inference with nothing to hand back but the maps: the whole core is one call of the C ABI
    # (same six launches, intermediates in a workspace).  Per-stage calls remain for the tape, for
    # `retraw`, for density noise, for foreign network_query_fn's and while bench.py's per-kernel
    # timers are armed.
    if (SINGLE_CALL and fused and not rec and not retraw and not raw_noise_std > 0. and n > 0
            and ops.MLP_TIMING is None and ops.HBM_TIMING is None
            and isinstance(network_fn, NerfModel) and (network_fine is None or isinstance(network_fine, NerfModel))):
        if N_importance > 0 and perturb > 0.:
            if pdf_u is None:
                pdf_u = torch.rand((n, N_importance), dtype=torch.float32, device=rays.device)
            pdf_u = nv.f32(pdf_u, rays.device)
            if tuple(pdf_u.shape) != (n, N_importance):
                raise ValueError(f"render_rays: pdf_u must be [{n},{N_importance}]")
        else:
            pdf_u = None
        return ops.render_rays_single_call(network_fn.packed(), network_fine.packed() if network_fine is not None else None,
                                           rays, N_samples, N_importance, lindisp, white_bkgd, perturb_u, pdf_u)

    def query(z, net):
        if rec:
            return training.MLPQuery.apply(rays, z, net, *net.params)
        if fused:
            return ops.mlp_forward_rays(net.packed(), rays, z)
        pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
        return nv.f32(network_query_fn(pts, rays[:, 8:11], net), rays.device)

    def composite(raw, z, noise):
        if raw_noise_std > 0.:
            if noise is None:
                noise = torch.randn(z.shape, dtype=torch.float32, device=rays.device)
            noise = nv.f32(noise, rays.device)
            if tuple(noise.shape) != tuple(z.shape):
                raise ValueError(f"render_rays: noise must be {tuple(z.shape)}, got {tuple(noise.shape)}")
        else:
            noise = None
        if rec:
            return training.Composite.apply(raw, z, rays, bool(white_bkgd), noise, float(raw_noise_std))
        return ops.composite_fwd(raw, z, rays[:, 3:6], width, white_bkgd, noise, raw_noise_std)

    z_vals = ops.sample_z(rays, N_samples, lindisp, perturb_u)
    raw = query(z_vals, network_fn)
    rgb_map, disp_map, acc_map, weights, depth_map = composite(raw, z_vals, noise0)
    ret = {}
    if N_importance > 0:
        ret.update(rgb0=rgb_map, disp0=disp_map, acc0=acc_map)
        z_vals, _, z_std = ops.resample_merge(z_vals, weights.detach(), N_importance, det=(perturb == 0.),
                                              u=pdf_u, want_samples=False)
        raw = query(z_vals, network_fine if network_fine is not None else network_fn)
        rgb_map, disp_map, acc_map, weights, depth_map = composite(raw, z_vals, noise1)
        ret['z_std'] = z_std
    ret.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map, depth_map=depth_map)
    if retraw:
        ret['raw'] = raw
    return ret
