"""Tensor-level wrappers over the C ABI: torch tensors in, torch tensors out, on the current HIP
stream.  Shape checks live here (the kernels trust their arguments); arithmetic lives in the
library.  Nothing in this module computes on the CPU."""
import ctypes

import numpy as np
import torch

from . import _native as nv


def _dev(t=None):
    if t is not None and isinstance(t, torch.Tensor) and t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise nv.NerfHipError("no GPU visible: nerf_amd runs on MI355X only (there is no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def get_rays(H, W, focal, c2w):
    """run_nerf_helpers.py:122-129 -> rays_o, rays_d as [H,W,3] fp32 tensors."""
    dev = _dev(c2w if isinstance(c2w, torch.Tensor) else None)
    m = np.ascontiguousarray(torch.as_tensor(c2w).detach().cpu().numpy()[:3, :4], dtype=np.float32)
    rays_o = torch.empty((int(H), int(W), 3), dtype=torch.float32, device=dev)
    rays_d = torch.empty_like(rays_o)
    with torch.cuda.device(dev):
        nv.check(nv.lib().nerf_get_rays(int(H), int(W), float(focal),
                                        m.ctypes.data_as(ctypes.c_void_p), nv.ptr(rays_o),
                                        nv.ptr(rays_d), nv.stream()), "nerf_get_rays")
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """run_nerf_helpers.py:145-180 on [...,3] tensors."""
    o = nv.f32(rays_o)
    d = nv.f32(rays_d, o.device)
    if o.shape != d.shape or o.shape[-1] != 3:
        raise ValueError(f"ndc_rays: rays_o {tuple(o.shape)} and rays_d {tuple(d.shape)} must both be [...,3]")
    oo, od = torch.empty_like(o), torch.empty_like(d)
    with torch.cuda.device(o.device):
        nv.check(nv.lib().nerf_ndc_rays(o.numel() // 3, int(H), int(W), float(focal), float(near),
                                        nv.ptr(o), nv.ptr(d), nv.ptr(oo), nv.ptr(od), nv.stream()),
                 "nerf_ndc_rays")
    return oo, od


def build_ray_batch(rays_o, rays_d, view_src, ndc, H, W, focal, near, far):
    """render()'s assembly of the [N, 8|11] ray batch (run_nerf.py:79-108).
    near/far: python scalars or per-ray tensors [N] / [N,1]."""
    o = nv.f32(rays_o).reshape(-1, 3)
    d = nv.f32(rays_d, o.device).reshape(-1, 3)
    n = o.shape[0]
    if d.shape[0] != n:
        raise ValueError("build_ray_batch: rays_o and rays_d disagree on the ray count")
    v = None
    if view_src is not None:
        v = nv.f32(view_src, o.device).reshape(-1, 3)
        if v.shape[0] != n:
            raise ValueError("build_ray_batch: view directions disagree on the ray count")
    width = 11 if v is not None else 8
    near_t = far_t = None
    near_s = far_s = 0.0
    if isinstance(near, (torch.Tensor, np.ndarray)) and isinstance(far, (torch.Tensor, np.ndarray)):
        near_t = nv.f32(near, o.device).reshape(-1)
        far_t = nv.f32(far, o.device).reshape(-1)
        if near_t.numel() != n or far_t.numel() != n:
            raise ValueError("build_ray_batch: per-ray near/far must have one value per ray")
    else:
        near_s, far_s = float(near), float(far)
    out = torch.empty((n, width), dtype=torch.float32, device=o.device)
    with torch.cuda.device(o.device):
        nv.check(nv.lib().nerf_build_ray_batch(n, nv.ptr(o), nv.ptr(d), nv.ptr(v), int(bool(ndc)),
                                               int(H), int(W), float(focal), nv.ptr(near_t),
                                               nv.ptr(far_t), near_s, far_s, nv.ptr(out), width,
                                               nv.stream()), "nerf_build_ray_batch")
    return out


def gather_pixel_rays(H, W, focal, poses, images, img_index, pix_index):
    """(batch_rays [2,n,3], target [n,3] or None) for pixels pix_index (row*W+col) of images img_index.
    poses [n_img,3,4] fp32 and images [n_img,H,W,C>=3] fp32 (or None) live on the GPU."""
    poses = nv.f32(poses)
    dev = poses.device
    if poses.dim() != 3 or tuple(poses.shape[1:]) != (3, 4):
        raise ValueError(f"gather_pixel_rays: poses must be [n_img,3,4], got {tuple(poses.shape)}")
    n_img = poses.shape[0]
    channels = 0
    if images is not None:
        images = nv.f32(images, dev)
        if images.dim() != 4 or tuple(images.shape[:3]) != (n_img, H, W) or images.shape[3] < 3:
            raise ValueError(f"gather_pixel_rays: images must be [{n_img},{H},{W},>=3], got {tuple(images.shape)}")
        channels = images.shape[3]
    ii = torch.as_tensor(img_index, device=dev).to(torch.int32).contiguous().reshape(-1)
    pi = torch.as_tensor(pix_index, device=dev).to(torch.int32).contiguous().reshape(-1)
    if ii.shape != pi.shape:
        raise ValueError("gather_pixel_rays: img_index and pix_index differ in length")
    n = ii.numel()
    rays = torch.empty((2, n, 3), dtype=torch.float32, device=dev)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev) if images is not None else None
    if n == 0:
        return rays, rgb
    with torch.cuda.device(dev):
        nv.check(nv.lib().nerf_gather_pixel_rays(int(H), int(W), float(focal), n_img, nv.ptr(poses), nv.ptr(images),
                                                 channels, nv.ptr(ii), nv.ptr(pi), n, nv.ptr(rays), nv.ptr(rgb),
                                                 nv.stream()), "nerf_gather_pixel_rays")
    return rays, rgb


def embed(x, L):
    """Embedder.embed (run_nerf_helpers.py:22-56) on [...,3] -> [..., 3+6L]."""
    x = nv.f32(x)
    if x.shape[-1] != 3:
        raise ValueError("embed: last dimension must be 3")
    out = torch.empty(tuple(x.shape[:-1]) + (3 + 6 * L,), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        nv.check(nv.lib().nerf_embed(x.numel() // 3, nv.ptr(x), int(L), nv.ptr(out), nv.stream()),
                 "nerf_embed")
    return out


def sample_z(rays, S, lindisp=False, perturb_u=None):
    """Stratified depths [N,S] for a ray batch [N,width>=8]."""
    rays = nv.f32(rays)
    n, width = rays.shape
    u = None
    if perturb_u is not None:
        u = nv.f32(perturb_u, rays.device)
        if tuple(u.shape) != (n, S):
            raise ValueError(f"sample_z: perturb_u must be [{n},{S}]")
    z = torch.empty((n, S), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        nv.check(nv.lib().nerf_sample_z(n, int(S), nv.ptr(rays), width, int(bool(lindisp)),
                                        nv.ptr(u), nv.ptr(z), nv.stream()), "nerf_sample_z")
    return z


# When a list, every MLP launch appends (start_event, end_event, points) -- bench.py's roofline leg.
MLP_TIMING = None


# When a dict, the compositing and resampling launches append (start_event, end_event,
# algorithmic bytes) under their kernel's name -- bench.py's HBM-bound kernels.
HBM_TIMING = None


def _timed_hbm(name, nbytes, launch):
    if HBM_TIMING is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    HBM_TIMING.setdefault(name, []).append((e0, e1, nbytes))


def _timed_mlp(points, launch):
    if MLP_TIMING is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    MLP_TIMING.append((e0, e1, points))


def packed_bytes():
    return int(nv.lib().nerf_mlp_packed_bytes())


def mlp_pack(params, packed):
    """params: 24 contiguous fp32 CUDA tensors in get_weights() order; packed: uint8 [packed_bytes()]."""
    arr = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in params])
    with torch.cuda.device(packed.device):
        nv.check(nv.lib().nerf_mlp_pack(arr, nv.ptr(packed), nv.stream()), "nerf_mlp_pack")


def mlp_forward_rays(packed, rays, z):
    """raw [N,S,4] for points o + d*z of a ray batch [N,11] with depths z [N,S]."""
    n, width = rays.shape
    S = z.shape[1]
    if z.shape[0] != n:
        raise ValueError("mlp_forward_rays: rays and z disagree on the ray count")
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _timed_mlp(n * S, lambda: nv.check(
            nv.lib().nerf_mlp_forward_rays(nv.ptr(packed), nv.ptr(rays), width, nv.ptr(z), n, S,
                                           nv.ptr(raw), nv.stream()), "nerf_mlp_forward_rays"))
    return raw


def mlp_forward_pts(packed, pts, viewdirs):
    """raw [N,S,4] for explicit points [N,S,3] and per-ray view directions [N,3]."""
    n, S = pts.shape[0], pts.shape[1]
    if tuple(viewdirs.shape) != (n, 3) or pts.shape[2] != 3:
        raise ValueError("mlp_forward_pts: need pts [N,S,3] and viewdirs [N,3]")
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=pts.device)
    with torch.cuda.device(pts.device):
        _timed_mlp(n * S, lambda: nv.check(
            nv.lib().nerf_mlp_forward_pts(nv.ptr(packed), nv.ptr(pts), nv.ptr(viewdirs), n, S,
                                          nv.ptr(raw), nv.stream()), "nerf_mlp_forward_pts"))
    return raw


def composite_fwd(raw, z, rays_d, d_stride, white_bkgd=False, noise=None, noise_std=0.0,
                  want_weights=True):
    """raw2outputs.  rays_d is any fp32 tensor whose row i starts the direction of ray i when read
    with stride d_stride (e.g. the ray batch offset by 3 columns)."""
    n, S = z.shape
    dev = z.device
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    weights = torch.empty((n, S), dtype=torch.float32, device=dev) if want_weights else None
    with torch.cuda.device(dev):
        # algorithmic bytes: raw 16 S + z 4 S + direction 12 read, weights 4 S + maps 24 written
        _timed_hbm(f"composite_fwd S={S}", n * (24 * S + 36), lambda: nv.check(
            nv.lib().nerf_composite_fwd(nv.ptr(raw), nv.ptr(z), ctypes.c_void_p(rays_d.data_ptr()),
                                        int(d_stride), n, S, int(bool(white_bkgd)),
                                        nv.ptr(noise), float(noise_std), nv.ptr(rgb),
                                        nv.ptr(disp), nv.ptr(acc), nv.ptr(depth),
                                        nv.ptr(weights), nv.stream()), "nerf_composite_fwd"))
    return rgb, disp, acc, weights, depth


def sample_pdf(bins, weights, N_samples, det=False, u=None):
    """run_nerf_helpers.py:185-213 on [...,nb] bins and [...,nb-1] weights."""
    bins = nv.f32(bins)
    weights = nv.f32(weights, bins.device)
    nb = bins.shape[-1]
    if weights.shape[-1] != nb - 1 or weights.shape[:-1] != bins.shape[:-1]:
        raise ValueError(f"sample_pdf: bins {tuple(bins.shape)} need weights [..., {nb - 1}]")
    lead = tuple(bins.shape[:-1])
    n = bins.numel() // nb
    if not det:
        if u is None:
            u = torch.rand(lead + (N_samples,), dtype=torch.float32, device=bins.device)
        u = nv.f32(u, bins.device)
        if tuple(u.shape) != lead + (N_samples,):
            raise ValueError("sample_pdf: u must be [..., N_samples]")
    else:
        u = None
    out = torch.empty(lead + (N_samples,), dtype=torch.float32, device=bins.device)
    with torch.cuda.device(bins.device):
        nv.check(nv.lib().nerf_sample_pdf(nv.ptr(bins), nv.ptr(weights), n, nb, int(N_samples),
                                          int(bool(det)), nv.ptr(u), nv.ptr(out), nv.stream()),
                 "nerf_sample_pdf")
    return out


def resample_merge(z_vals, weights, N_importance, det=True, u=None, want_samples=True):
    """z_mid -> sample_pdf(weights[:,1:-1]) -> sort(concat) of render_rays, one kernel.
    Returns z_fine [N,Sc+Si], z_samples [N,Si] (or None), z_std [N]."""
    n, Sc = z_vals.shape
    dev = z_vals.device
    if tuple(weights.shape) != (n, Sc):
        raise ValueError("resample_merge: weights must match z_vals")
    if not det:
        if u is None:
            u = torch.rand((n, N_importance), dtype=torch.float32, device=dev)
        u = nv.f32(u, dev)
        if tuple(u.shape) != (n, N_importance):
            raise ValueError("resample_merge: u must be [N, N_importance]")
    else:
        u = None
    z_out = torch.empty((n, Sc + N_importance), dtype=torch.float32, device=dev)
    z_samples = torch.empty((n, N_importance), dtype=torch.float32, device=dev) if want_samples else None
    z_std = torch.empty((n,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        # algorithmic bytes: z and weights read (8 Sc), merged depths written (4 (Sc + Si))
        _timed_hbm(f"resample_merge {Sc}+{N_importance}", n * (8 * Sc + 4 * (Sc + N_importance)), lambda: nv.check(
            nv.lib().nerf_resample_merge(nv.ptr(z_vals), nv.ptr(weights), n, Sc,
                                         int(N_importance), int(bool(det)), nv.ptr(u),
                                         nv.ptr(z_out), nv.ptr(z_samples), nv.ptr(z_std),
                                         nv.stream()), "nerf_resample_merge"))
    return z_out, z_samples, z_std


# (device, stream) -> grow-only workspace of render_rays_single_call.  One per stream: calls on
# different streams of one GPU may overlap and must not share intermediates; a workspace that is
# outgrown is only dropped by the stream that used it, behind the work that stream has queued.
_RENDER_WS = {}


def render_rays_single_call(packed_coarse, packed_fine, rays, N_samples, N_importance, lindisp=False,
                            white_bkgd=False, perturb_u=None, pdf_u=None):
    """The whole per-ray core through ONE call of the C ABI (nerf_render_rays).  Returns the dict
    render_rays returns at raw_noise_std = 0 without `raw`."""
    n, width = rays.shape
    dev = rays.device
    need = int(nv.lib().nerf_render_rays_workspace_bytes(n, int(N_samples), int(N_importance)))
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _RENDER_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _RENDER_WS[key] = torch.empty((max(need, 1),), dtype=torch.uint8, device=dev)
        ws.record_stream(torch.cuda.current_stream(dev))
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    ret = dict(rgb_map=new(n, 3), disp_map=new(n), acc_map=new(n), depth_map=new(n))
    if N_importance > 0:
        ret.update(rgb0=new(n, 3), disp0=new(n), acc0=new(n), z_std=new(n))
    g = lambda k: nv.ptr(ret.get(k))
    with torch.cuda.device(dev):
        nv.check(nv.lib().nerf_render_rays(nv.ptr(packed_coarse), nv.ptr(packed_fine), nv.ptr(rays), width, n,
                                           int(N_samples), int(N_importance), int(bool(lindisp)), int(bool(white_bkgd)),
                                           nv.ptr(perturb_u), nv.ptr(pdf_u), g('rgb_map'), g('disp_map'), g('acc_map'),
                                           g('depth_map'), g('rgb0'), g('disp0'), g('acc0'), g('z_std'), nv.ptr(ws),
                                           nv.stream()), "nerf_render_rays")
    return ret


def composite_bwd(raw, z, rays_d, d_stride, white_bkgd, noise, noise_std, g_rgb, g_acc=None, g_depth=None):
    """d_raw [N,S,4] from the gradients of rgb_map / acc_map / depth_map (None = zero)."""
    n, S = z.shape
    d_raw = torch.empty((n, S, 4), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        nv.check(nv.lib().nerf_composite_bwd(nv.ptr(raw), nv.ptr(z), ctypes.c_void_p(rays_d.data_ptr()),
                                             int(d_stride), n, S, int(bool(white_bkgd)), nv.ptr(noise),
                                             float(noise_std), nv.ptr(g_rgb), nv.ptr(g_acc),
                                             nv.ptr(g_depth), nv.ptr(d_raw), nv.stream()),
                 "nerf_composite_bwd")
    return d_raw


def act_bytes(n_points):
    return int(nv.lib().nerf_mlp_act_bytes(int(n_points)))


def train_layout():
    """Block numbers of the saved training tensors (nerf_mlp_train_layout), as a dict."""
    out = (ctypes.c_int32 * 10)()
    nv.check(nv.lib().nerf_mlp_train_layout(out, 10), "nerf_mlp_train_layout")
    names = ("act_blocks", "pe", "h0", "v", "vpe", "mask", "grad_blocks", "d_v", "d_pre0", "d_alpha")
    return dict(zip(names, (int(x) for x in out)))


def mlp_forward_train(packed, rays, z, act):
    """mlp_forward_rays that also fills `act` (uint8 [act_bytes(N*S)]) for the backward pass."""
    n, width = rays.shape
    S = z.shape[1]
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=rays.device)
    if act.numel() < act_bytes(n * S):
        raise ValueError("mlp_forward_train: activation scratch too small")
    with torch.cuda.device(rays.device):
        _timed_mlp(n * S, lambda: nv.check(
            nv.lib().nerf_mlp_forward_train(nv.ptr(packed), nv.ptr(rays), width, nv.ptr(z), n, S,
                                            nv.ptr(raw), nv.ptr(act), nv.stream()), "nerf_mlp_forward_train"))
    return raw


def packed_bwd_bytes():
    return int(nv.lib().nerf_mlp_packed_bwd_bytes())


def mlp_pack_bwd(params, packed_bwd):
    arr = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in params])
    with torch.cuda.device(packed_bwd.device):
        nv.check(nv.lib().nerf_mlp_pack_bwd(arr, nv.ptr(packed_bwd), nv.stream()), "nerf_mlp_pack_bwd")


def grad_bytes(n_points):
    return int(nv.lib().nerf_mlp_grad_bytes(int(n_points)))


def partial_bytes():
    return int(nv.lib().nerf_mlp_partial_bytes())


GRAD_SATURATED, GRAD_UNDERFLOW = 1, 2


def mlp_backward_status(partial):
    """(flags, gscale_used, max|d_raw|) of the last mlp_backward that used `partial`; waits for it."""
    flags, gs, mx = ctypes.c_int(0), ctypes.c_float(0.), ctypes.c_float(0.)
    with torch.cuda.device(partial.device):
        nv.check(nv.lib().nerf_mlp_backward_status(nv.ptr(partial), ctypes.byref(flags), ctypes.byref(gs),
                                                   ctypes.byref(mx), nv.stream()), "nerf_mlp_backward_status")
    return flags.value, gs.value, mx.value


def mlp_backward(packed_bwd, act, d_raw, gscale, gbuf, partial, grads, accumulate=False):
    """Parameter gradients (24 tensors shaped like the parameters, written in place) from
    d_raw [N,S,4] and the activations saved by mlp_forward_train.  gscale None or 0: the fp16
    gradient scale is formed on the device from max|d_raw| (largest scaled element in [256, 512));
    gscale < 0: the same with the target [-gscale, -2 gscale), a power of two; > 0: used as given."""
    gscale = 0.0 if gscale is None else gscale
    n_points = d_raw.numel() // 4
    if gbuf.numel() < grad_bytes(n_points) or partial.numel() < partial_bytes():
        raise ValueError("mlp_backward: scratch too small")
    arr = (ctypes.c_void_p * 24)(*[g.data_ptr() for g in grads])
    with torch.cuda.device(d_raw.device):
        nv.check(nv.lib().nerf_mlp_backward(nv.ptr(packed_bwd), nv.ptr(act), nv.ptr(d_raw), n_points,
                                            float(gscale), nv.ptr(gbuf), nv.ptr(partial), arr,
                                            int(bool(accumulate)), nv.stream()), "nerf_mlp_backward")
