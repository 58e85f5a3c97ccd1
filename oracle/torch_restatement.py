"""TEST INFRASTRUCTURE ONLY -- unfused torch restatement of the reference's TF1.15 op sequence
(CPU tensors for the timed baseline; it follows the device of its inputs, so the full-size tests
can also run it in float64 next to the kernels).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED (see oracle/nerf_oracle.py for why).  Written independently of the NumPy oracle so
the two can cross-check each other, op for op as the reference issues them: a separate positional
encoding concat, twelve Dense layers as matmul + bias + relu, cumprod, searchsorted, two gathers,
sort; rays in `chunk`s (run_nerf.py:15-25) and points in `netchunk`s (run_nerf_helpers.py:216-223).
It is differentiable, so torch autograd in float64 gives the gradient ground truth for the
backward kernels, and in float32 it is the "reference CPU path" that bench.py times
(label: restatement of the TF1.15 op graph -- never "TF1.15").
"""
import math
import torch


def embed(x, L):
    """run_nerf_helpers.py:22-56"""
    out = [x]
    for k in range(L):
        f = 2.0 ** k
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def mlp(ws, e, D=8, input_ch=63, skips=(4,)):
    """run_nerf_helpers.py:89-112 (use_viewdirs=True); ws in get_weights() order
    (8 point layers, bottleneck, view, rgb, alpha)."""
    pts, views = e[..., :input_ch], e[..., input_ch:]
    h = pts
    for i in range(D):
        h = torch.relu(torch.matmul(h, ws[2 * i]) + ws[2 * i + 1])
        if i in skips:
            h = torch.cat([pts, h], -1)
    o = 2 * D
    alpha = torch.matmul(h, ws[o + 6]) + ws[o + 7]
    bott = torch.matmul(h, ws[o]) + ws[o + 1]
    v = torch.relu(torch.matmul(torch.cat([bott, views], -1), ws[o + 2]) + ws[o + 3])
    rgb = torch.matmul(v, ws[o + 4]) + ws[o + 5]
    return torch.cat([rgb, alpha], -1)


def run_network(pts, viewdirs, ws, netchunk=1024 * 64, L=10, Lv=4):
    """run_nerf.py:209-214 closure + run_nerf_helpers.py:216-223 batchify."""
    flat = pts.reshape(-1, 3)
    e = embed(flat, L)
    dirs = viewdirs[:, None, :].expand(pts.shape).reshape(-1, 3)
    e = torch.cat([e, embed(dirs, Lv)], -1)
    outs = [mlp(ws, e[i:i + netchunk]) for i in range(0, e.shape[0], netchunk)]
    return torch.cat(outs, 0).reshape(list(pts.shape[:-1]) + [4])


def raw2outputs(raw, z, rays_d, white_bkgd=False, noise=None, raw_noise_std=0.):
    """tiny_nerf.ipynb render_rays cell + upstream additions (ray-norm scaling, disparity, white bkgd)."""
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.full_like(z[..., :1], 1e10)], -1)
    dists = dists * torch.linalg.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sig = raw[..., 3]
    if raw_noise_std > 0.:
        sig = sig + noise * raw_noise_std
    alpha = 1. - torch.exp(-torch.relu(sig) * dists)
    t = 1. - alpha + 1e-10
    trans = torch.cumprod(t, -1)
    trans = torch.cat([torch.ones_like(trans[..., :1]), trans[..., :-1]], -1)
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1. / torch.clamp_min(depth_map / acc_map, 1e-10)
    if white_bkgd:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_pdf(bins, weights, n, det=True, u=None):
    """run_nerf_helpers.py:185-213"""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0., 1., n, dtype=bins.dtype, device=bins.device).expand(list(cdf.shape[:-1]) + [n])
    u = u.contiguous()
    inds = torch.searchsorted(cdf.contiguous(), u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, cdf.shape[-1] - 1)
    cdf_g0, cdf_g1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bins_g0, bins_g1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_g1 - cdf_g0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g0) / denom
    return bins_g0 + t * (bins_g1 - bins_g0)


def render_rays(rb, ws_c, ws_f, N_samples=64, N_importance=128, white_bkgd=False, lindisp=False,
                perturb_u=None, pdf_u=None, netchunk=1024 * 64):
    """Upstream render_rays (see nerf_oracle.render_rays for the evidence it follows)."""
    o, d, near, far, vd = rb[:, 0:3], rb[:, 3:6], rb[:, 6:7], rb[:, 7:8], rb[:, 8:11]
    t = torch.linspace(0., 1., N_samples, dtype=rb.dtype, device=rb.device)
    z = near * (1. - t) + far * t if not lindisp else 1. / (1. / near * (1. - t) + 1. / far * t)
    z = z.expand(rb.shape[0], N_samples)
    if perturb_u is not None:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * perturb_u
    pts = o[:, None, :] + d[:, None, :] * z[..., :, None]
    raw0 = raw = run_network(pts, vd, ws_c, netchunk)
    rgb0, disp0, acc0, w, depth0 = raw2outputs(raw, z, d, white_bkgd)
    w0 = w
    z_mid = .5 * (z[..., 1:] + z[..., :-1])
    zs = sample_pdf(z_mid, w[..., 1:-1], N_importance, det=(pdf_u is None), u=pdf_u).detach()
    z, _ = torch.sort(torch.cat([z, zs], -1), -1)
    pts = o[:, None, :] + d[:, None, :] * z[..., :, None]
    raw = run_network(pts, vd, ws_f, netchunk)
    rgb, disp, acc, w, depth = raw2outputs(raw, z, d, white_bkgd)
    return dict(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, rgb0=rgb0, disp0=disp0,
                acc0=acc0, z_std=torch.std(zs, -1, unbiased=False), raw=raw, z_vals=z, weights=w,
                raw0=raw0, weights0=w0)


def render_flat(rb, ws_c, ws_f, chunk=1024 * 32, **kw):
    """run_nerf.py:15-25 batchify_rays over an already assembled [N,11] ray batch."""
    outs = {}
    for i in range(0, rb.shape[0], chunk):
        r = render_rays(rb[i:i + chunk], ws_c, ws_f, **kw)
        for k, v in r.items():
            outs.setdefault(k, []).append(v)
    return {k: torch.cat(v, 0) for k, v in outs.items()}


def loss_fn(out, target):
    """run_nerf.py:781,:809-811: fine + coarse mean squared error."""
    return torch.mean((out['rgb_map'] - target) ** 2) + torch.mean((out['rgb0'] - target) ** 2)


def keras_adam(params, grads, ms, vs, it, lr, b1=0.9, b2=0.999, eps=1e-7):
    """tf.keras Adam dense update (eps outside the bias correction)."""
    t = it + 1
    lr_t = lr * math.sqrt(1. - b2 ** t) / (1. - b1 ** t)
    for p, g, m, v in zip(params, grads, ms, vs):
        m.mul_(b1).add_(g, alpha=1. - b1)
        v.mul_(b2).addcmul_(g, g, value=1. - b2)
        p.sub_(lr_t * m / (v.sqrt() + eps))
