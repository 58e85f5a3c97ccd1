"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy) of the reference's per-ray render core.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (nerf_amd/) never does.

PARITY UNPINNED.  The reference holds no golden vectors, known-answer tests or fixtures for this
path, and it cannot be imported here (TensorFlow 1.15 / imageio / configargparse are not installed,
and the NERFCO submodule that owns render_rays / run_network / raw2outputs is absent from the
tree).  This file therefore restates the algorithm from the reference's source text, function by
function (citations below are file:line under the reference checkout), and is itself pinned only by
analytic known-answer tests (tests/test_oracle_kat.py) and by a second, independently written torch
restatement (oracle/torch_restatement.py).  For the three functions whose source is absent
(render_rays, run_network, raw2outputs) it follows the public bmild/nerf code this fork derives
from, constrained by the in-tree call sites (run_nerf.py:20, :209-214, :216-238, :775, :796-823),
the in-tree simplified twin (tiny_nerf.ipynb render_rays cell) and sample_pdf's shape contract
(run_nerf_helpers.py:185-213).

Every function takes `dtype` (np.float64 = ground truth, np.float32 = the reference's arithmetic
type) so that fp32 rounding noise can be separated from real disagreement.
"""
import numpy as np


# ----------------------------------------------------------------------------------------------
# misc (run_nerf_helpers.py:11-17)
# ----------------------------------------------------------------------------------------------
def img2mse(x, y):
    """run_nerf_helpers.py:11 -- mean over every element of the squared difference."""
    return np.mean(np.square(x - y))


def mse2psnr(x):
    """run_nerf_helpers.py:14"""
    return -10. * np.log(x) / np.log(10.)


def to8b(x):
    """run_nerf_helpers.py:17"""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


# ----------------------------------------------------------------------------------------------
# positional encoding (run_nerf_helpers.py:22-75)
# ----------------------------------------------------------------------------------------------
def embed(x, multires, dtype=np.float64):
    """[x, sin(x f0), cos(x f0), sin(x f1), ...], f_k = 2**k, k < multires.

    run_nerf_helpers.py:41-42 (freq_bands = 2**linspace(0, L-1, L)), :34-36 (include_input),
    :46-50 (sin before cos, per frequency), :56 (concat on the last axis).
    """
    x = np.asarray(x, dtype=dtype)
    outs = [x]
    freq_bands = (2.0 ** np.linspace(0., multires - 1, multires)).astype(dtype)
    for f in freq_bands:
        for fn in (np.sin, np.cos):
            outs.append(fn(x * f))
    return np.concatenate(outs, -1)


def get_embedder(multires, i=0, dtype=np.float64):
    """run_nerf_helpers.py:59-75.  i == -1 -> identity with 3 channels."""
    if i == -1:
        return (lambda x: np.asarray(x, dtype=dtype)), 3
    return (lambda x: embed(x, multires, dtype)), 3 + 3 * 2 * multires


# ----------------------------------------------------------------------------------------------
# MLP (run_nerf_helpers.py:80-117)
# ----------------------------------------------------------------------------------------------
def layer_shapes(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,),
                 use_viewdirs=True):
    """(in, out) of every Dense layer in `get_weights()` order.

    Order: the D point layers, then (use_viewdirs) bottleneck, view layer, rgb, alpha -- the
    order Keras lists the functional model's layers in (depth-sorted; the alpha head, created at
    run_nerf_helpers.py:102, sorts last because it feeds the final concat directly).  Without
    view directions: the D point layers then the single output layer (:114).
    """
    shapes = []
    k = input_ch
    for i in range(D):
        shapes.append((k, W))
        k = W + input_ch if i in skips else W
    if use_viewdirs:
        shapes.append((k, 256))                        # bottleneck  (:103, width fixed at 256)
        shapes.append((256 + input_ch_views, W // 2))  # view layer  (:110)
        shapes.append((W // 2, 3))                     # rgb         (:111)
        shapes.append((k, 1))                          # alpha       (:102)
    else:
        shapes.append((k, output_ch))
    return shapes


def init_nerf_weights(rng, D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=(4,),
                      use_viewdirs=True, dtype=np.float32):
    """Keras Dense defaults: glorot-uniform kernel [in,out], zero bias (run_nerf_helpers.py:83)."""
    ws = []
    for (fi, fo) in layer_shapes(D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs):
        lim = np.sqrt(6. / (fi + fo))
        ws.append(rng.uniform(-lim, lim, size=(fi, fo)).astype(dtype))
        ws.append(np.zeros((fo,), dtype=dtype))
    return ws


def mlp_forward(weights, x, D=8, input_ch=63, input_ch_views=27, skips=(4,), use_viewdirs=True,
                dtype=np.float64, operand_dtype=None, alpha_operand_dtype='same'):
    """run_nerf_helpers.py:89-114.  x is [m, input_ch + input_ch_views]; returns [m, 4] = (rgb, alpha).

    `operand_dtype` (np.float16) emulates a matrix unit that takes its operands in that type and
    accumulates wider: every kernel and every Dense layer's input is rounded to it, biases and sums
    stay in `dtype`.  It is the arithmetic the fp16-MFMA kernel implements, so tests can separate
    what rounding the operands costs (any such implementation pays it) from what the kernel adds.
    `alpha_operand_dtype` overrides it for the two operands of the density head alone (the last
    hidden activation as that head reads it, and its kernel), to price that head separately."""
    def rounder(t):
        if t is None:
            return lambda a: a

        def q(a):
            with np.errstate(over='ignore'):
                r = np.asarray(a, dtype=np.float64).astype(t)
            big = np.finfo(t).max       # the kernel's conversions saturate instead of producing inf
            return np.clip(r.astype(dtype), -big, big)
        return q
    q = rounder(operand_dtype)
    qa = q if isinstance(alpha_operand_dtype, str) else rounder(alpha_operand_dtype)
    ws = [np.asarray(w, dtype=dtype) for w in weights]
    x = np.asarray(x, dtype=dtype)
    pts, views = q(x[..., :input_ch]), q(x[..., input_ch:input_ch + input_ch_views])
    h = pts
    for i in range(D):
        h_wide = np.maximum(h @ q(ws[2 * i]) + ws[2 * i + 1], 0)
        h = q(h_wide)
        if i in skips:
            h = np.concatenate([pts, h], -1)          # :98-99, order [pts, h]
            h_wide = h
    if use_viewdirs:
        o = 2 * D
        alpha = qa(h_wide) @ qa(ws[o + 6]) + ws[o + 7]  # :102 linear
        bott = q(h @ q(ws[o + 0]) + ws[o + 1])         # :103 linear (no ReLU)
        v = np.concatenate([bott, views], -1)          # :104-105
        v = q(np.maximum(v @ q(ws[o + 2]) + ws[o + 3], 0))   # :110
        rgb = v @ q(ws[o + 4]) + ws[o + 5]             # :111 linear
        return np.concatenate([rgb, alpha], -1)        # :112
    return h @ q(ws[2 * D]) + ws[2 * D + 1]            # :114


def batchify(fn, chunk):
    """run_nerf_helpers.py:216-223"""
    if chunk is None:
        return fn
    return lambda inputs: np.concatenate(
        [fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)


def run_network(inputs, viewdirs, weights, multires=10, multires_views=4, i_embed=0,
                netchunk=1024 * 64, dtype=np.float64, **mlp_kw):
    """NERFCO.nerf_renderer.run_network (source absent; closure at run_nerf.py:209-214; contract
    pinned by extract_mesh.ipynb's net_fn(flat[:,None,:], viewdirs=zeros, ...) -> [n,1,4]).

    inputs [N,S,3], viewdirs [N,3] or None -> [N,S,4] (r,g,b,sigma pre-activation).
    """
    inputs = np.asarray(inputs, dtype=dtype)
    embed_fn, input_ch = get_embedder(multires, i_embed, dtype)
    flat = inputs.reshape(-1, inputs.shape[-1])
    e = embed_fn(flat)
    input_ch_views = 0
    if viewdirs is not None:
        embeddirs_fn, input_ch_views = get_embedder(multires_views, i_embed, dtype)
        dirs = np.broadcast_to(np.asarray(viewdirs, dtype=dtype)[:, None], inputs.shape)
        e = np.concatenate([e, embeddirs_fn(dirs.reshape(-1, 3))], -1)
    fn = lambda t: mlp_forward(weights, t, input_ch=input_ch, input_ch_views=input_ch_views,
                               use_viewdirs=viewdirs is not None, dtype=dtype, **mlp_kw)
    out = batchify(fn, netchunk)(e)
    return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])


# ----------------------------------------------------------------------------------------------
# rays (run_nerf_helpers.py:122-180)
# ----------------------------------------------------------------------------------------------
def get_rays(H, W, focal, c2w, dtype=np.float32):
    """run_nerf_helpers.py:122-129 / :135-142.  No half-pixel offset, directions not normalised."""
    c2w = np.asarray(c2w, dtype=dtype)
    focal = dtype(focal)
    i, j = np.meshgrid(np.arange(W, dtype=dtype), np.arange(H, dtype=dtype), indexing='xy')
    dirs = np.stack([(i - dtype(W * .5)) / focal, -(j - dtype(H * .5)) / focal, -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], rays_d.shape).copy()
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d, dtype=np.float32):
    """run_nerf_helpers.py:145-180, operation order kept (`-1./(W/(2.*focal))`, :167,:171)."""
    rays_o = np.asarray(rays_o, dtype=dtype)
    rays_d = np.asarray(rays_d, dtype=dtype)
    near = dtype(near)
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    cw = dtype(-1. / (W / (2. * focal)))
    ch = dtype(-1. / (H / (2. * focal)))
    o0 = cw * rays_o[..., 0] / rays_o[..., 2]
    o1 = ch * rays_o[..., 1] / rays_o[..., 2]
    o2 = dtype(1.) + dtype(2.) * near / rays_o[..., 2]
    d0 = cw * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = ch * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = dtype(-2.) * near / rays_o[..., 2]
    return np.stack([o0, o1, o2], -1), np.stack([d0, d1, d2], -1)


# ----------------------------------------------------------------------------------------------
# hierarchical sampling (run_nerf_helpers.py:185-213)
# ----------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, N_samples, det=False, u=None, dtype=np.float32):
    """Inverse-CDF resampling.  `u` ([..., N_samples] in [0,1)) replaces tf.random.uniform (:198)
    when det is False so that a test can feed the same numbers to both sides."""
    bins = np.asarray(bins, dtype=dtype)
    weights = np.asarray(weights, dtype=dtype) + dtype(1e-5)      # :188 (rebinding, caller's array untouched)
    pdf = weights / np.sum(weights, -1, keepdims=True)            # :189
    cdf = np.cumsum(pdf, -1, dtype=dtype)                         # :190
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)  # :191
    if det:
        u = np.linspace(0., 1., N_samples).astype(dtype)          # :195
        u = np.broadcast_to(u, list(cdf.shape[:-1]) + [N_samples])
    else:
        assert u is not None, "oracle needs explicit uniforms when det=False"
        u = np.asarray(u, dtype=dtype)
    flat_cdf = cdf.reshape(-1, cdf.shape[-1])
    flat_u = u.reshape(-1, N_samples)
    inds = np.stack([np.searchsorted(c, uu, side='right') for c, uu in zip(flat_cdf, flat_u)], 0)
    inds = inds.reshape(u.shape)                                  # :201
    below = np.maximum(0, inds - 1)                               # :202
    above = np.minimum(cdf.shape[-1] - 1, inds)                   # :203
    cdf_g0 = np.take_along_axis(cdf, below, -1)
    cdf_g1 = np.take_along_axis(cdf, above, -1)
    bins_g0 = np.take_along_axis(bins, below, -1)
    bins_g1 = np.take_along_axis(bins, above, -1)
    denom = cdf_g1 - cdf_g0                                       # :208
    denom = np.where(denom < dtype(1e-5), np.ones_like(denom), denom)  # :209
    t = (u - cdf_g0) / denom                                      # :210
    return bins_g0 + t * (bins_g1 - bins_g0)                      # :211


# ----------------------------------------------------------------------------------------------
# compositing (tiny_nerf.ipynb render_rays cell: sigma/rgb/dists/alpha/weights/maps lines;
# full-NeRF additions follow the public upstream raw2outputs -- source absent from the tree)
# ----------------------------------------------------------------------------------------------
def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0., white_bkgd=False, noise=None,
                scale_dists_by_ray_norm=True, dtype=np.float32):
    """raw [N,S,4], z_vals [N,S], rays_d [N,3] -> rgb_map, disp_map, acc_map, weights, depth_map.

    `scale_dists_by_ray_norm`: upstream multiplies dists by ||rays_d|| (default here); the in-tree
    tiny twin does not.  `noise` ([N,S], standard normal) replaces tf.random.normal.
    """
    raw = np.asarray(raw, dtype=dtype)
    z_vals = np.asarray(z_vals, dtype=dtype)
    rays_d = np.asarray(rays_d, dtype=dtype)
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = np.concatenate([dists, np.broadcast_to(dtype(1e10), z_vals[..., :1].shape)], -1)
    if scale_dists_by_ray_norm:
        dists = dists * np.linalg.norm(rays_d[..., None, :], axis=-1)
    rgb = 1. / (1. + np.exp(-raw[..., :3]))
    sig = raw[..., 3]
    if raw_noise_std > 0.:
        assert noise is not None
        sig = sig + np.asarray(noise, dtype=dtype) * dtype(raw_noise_std)
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        alpha = dtype(1.) - np.exp(-np.maximum(sig, 0) * dists)
        t = dtype(1.) - alpha + dtype(1e-10)
        trans = np.cumprod(t, -1, dtype=dtype)
        trans = np.concatenate([np.ones_like(trans[..., :1]), trans[..., :-1]], -1)  # exclusive
        weights = alpha * trans
        rgb_map = np.sum(weights[..., None] * rgb, -2)
        depth_map = np.sum(weights * z_vals, -1)
        acc_map = np.sum(weights, -1)
        disp_map = dtype(1.) / np.maximum(dtype(1e-10), depth_map / acc_map)
    if white_bkgd:
        rgb_map = rgb_map + (dtype(1.) - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


def sample_z(near, far, N_samples, lindisp=False, perturb_u=None, dtype=np.float32):
    """Stratified depths along each ray (upstream render_rays front; in-tree twin:
    tiny_nerf.ipynb `z_vals = tf.linspace(near, far, N_samples)`).  near/far are [N,1]."""
    near = np.asarray(near, dtype=dtype)
    far = np.asarray(far, dtype=dtype)
    t = np.linspace(0., 1., N_samples).astype(dtype)
    if not lindisp:
        z = near * (dtype(1.) - t) + far * t
    else:
        with np.errstate(divide='ignore'):
            z = dtype(1.) / (dtype(1.) / near * (dtype(1.) - t) + dtype(1.) / far * t)
    z = np.broadcast_to(z, [near.shape[0], N_samples]).astype(dtype)
    if perturb_u is not None:
        mids = dtype(.5) * (z[..., 1:] + z[..., :-1])
        upper = np.concatenate([mids, z[..., -1:]], -1)
        lower = np.concatenate([z[..., :1], mids], -1)
        z = lower + (upper - lower) * np.asarray(perturb_u, dtype=dtype)
    return z


def render_rays(ray_batch, weights_coarse, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, weights_fine=None, white_bkgd=False, raw_noise_std=0.,
                perturb_u=None, pdf_u=None, noise0=None, noise1=None, dtype=np.float32,
                scale_dists_by_ray_norm=True, **net_kw):
    """NERFCO.nerf_renderer.render_rays (source absent; call site run_nerf.py:20; kwargs
    run_nerf.py:216-238,775; consumers of the returned keys run_nerf.py:116,:796-823,:975-979).

    Random draws are explicit inputs (perturb_u [N,Sc], pdf_u [N,Si], noise0/noise1) so both sides
    of a parity test can share them; with perturb=0 and raw_noise_std=0 none is needed.
    """
    rb = np.asarray(ray_batch, dtype=dtype)
    rays_o, rays_d = rb[:, 0:3], rb[:, 3:6]
    viewdirs = rb[:, -3:] if rb.shape[-1] > 8 else None
    near, far = rb[:, 6:7], rb[:, 7:8]
    z_vals = sample_z(near, far, N_samples, lindisp, perturb_u if perturb > 0. else None, dtype)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
    raw = run_network(pts, viewdirs, weights_coarse, dtype=dtype, **net_kw)
    rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(
        raw, z_vals, rays_d, raw_noise_std, white_bkgd, noise0, scale_dists_by_ray_norm, dtype)
    ret = {}
    if N_importance > 0:
        rgb0, disp0, acc0, depth0 = rgb_map, disp_map, acc_map, depth_map
        raw0, weights0, z_vals0 = raw, weights, z_vals   # kept for tests that look at the coarse pass
        z_mid = dtype(.5) * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], N_importance, det=(perturb == 0.),
                               u=pdf_u, dtype=dtype)
        z_vals = np.sort(np.concatenate([z_vals, z_samples], -1), -1)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
        wf = weights_fine if weights_fine is not None else weights_coarse
        raw = run_network(pts, viewdirs, wf, dtype=dtype, **net_kw)
        rgb_map, disp_map, acc_map, weights, depth_map = raw2outputs(
            raw, z_vals, rays_d, raw_noise_std, white_bkgd, noise1, scale_dists_by_ray_norm, dtype)
        ret.update(rgb0=rgb0, disp0=disp0, acc0=acc0, depth0=depth0,
                   z_std=np.std(z_samples, -1), z_samples=z_samples, weights0=weights0, z_vals0=z_vals0)
        if retraw:
            ret['raw0'] = raw0
    ret.update(rgb_map=rgb_map, disp_map=disp_map, acc_map=acc_map, depth_map=depth_map,
               weights=weights, z_vals=z_vals)
    if retraw:
        ret['raw'] = raw
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """run_nerf.py:15-25.  Per-ray random inputs, if any, are sliced with the rays."""
    per_ray = ('perturb_u', 'pdf_u', 'noise0', 'noise1')
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        kw = {k: (v[i:i + chunk] if (k in per_ray and v is not None) else v) for k, v in kwargs.items()}
        ret = render_rays(rays_flat[i:i + chunk], **kw)
        for k, v in ret.items():
            all_ret.setdefault(k, []).append(v)
    return {k: np.concatenate(v, 0) for k, v in all_ret.items()}


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1.,
           use_viewdirs=False, c2w_staticcam=None, dtype=np.float32, **kwargs):
    """run_nerf.py:28-119.  Returns [rgb_map, disp_map, acc_map, extras]."""
    if c2w is not None:
        rays_o, rays_d = get_rays(H, W, focal, c2w, dtype)                      # :63
    elif len(rays) == 3:                                                         # :66-69
        rays_o, rays_d = np.asarray(rays[0], dtype=dtype), np.asarray(rays[1], dtype=dtype)
        depth = np.asarray(rays[2], dtype=dtype).reshape(-1)
        far = depth[:, None] * dtype(1.1)
        near = depth[:, None] * dtype(0.9)
    else:
        rays_o, rays_d = np.asarray(rays[0], dtype=dtype), np.asarray(rays[1], dtype=dtype)  # :71
    kwargs.pop('K', None)                                                        # :75-77
    viewdirs = None
    if use_viewdirs:
        viewdirs = rays_d                                                        # :81
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, focal, c2w_staticcam, dtype)         # :84
        viewdirs = viewdirs / np.linalg.norm(viewdirs, axis=-1, keepdims=True)   # :88
        viewdirs = viewdirs.reshape(-1, 3).astype(dtype)                         # :89
    sh = rays_d.shape                                                            # :91
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d, dtype)        # :94-95
    rays_o = rays_o.reshape(-1, 3).astype(dtype)
    rays_d = rays_d.reshape(-1, 3).astype(dtype)
    near_a = np.asarray(near, dtype=dtype) * np.ones_like(rays_d[..., :1])       # :100-102
    far_a = np.asarray(far, dtype=dtype) * np.ones_like(rays_d[..., :1])
    rb = np.concatenate([rays_o, rays_d, near_a, far_a], -1)                     # :105
    if use_viewdirs:
        rb = np.concatenate([rb, viewdirs], -1)                                  # :108
    all_ret = batchify_rays(rb, chunk, dtype=dtype, **kwargs)                    # :111
    for k in all_ret:
        all_ret[k] = all_ret[k].reshape(list(sh[:-1]) + list(all_ret[k].shape[1:]))  # :112-114
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    return [all_ret[k] for k in k_extract] + [{k: v for k, v in all_ret.items() if k not in k_extract}]


# ----------------------------------------------------------------------------------------------
# optimiser (run_nerf.py:601-605, :825-826; Keras-1.15 Adam formulation)
# ----------------------------------------------------------------------------------------------
def lr_schedule(lrate, lrate_decay, it):
    """ExponentialDecay(lrate, decay_steps=lrate_decay*1000, decay_rate=0.1), no staircase
    (run_nerf.py:603-604); `it` is optimizer.iterations (0-based)."""
    if lrate_decay > 0:
        return lrate * 0.1 ** (it / (lrate_decay * 1000.))
    return lrate


def keras_adam_step(theta, g, m, v, it, lr, beta1=0.9, beta2=0.999, eps=1e-7, dtype=np.float64):
    """tf.keras.optimizers.Adam (non-amsgrad) dense update, t = it + 1:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMAs; theta -= lr_t*m/(sqrt(v)+eps)  (eps un-corrected)."""
    t = it + 1
    lr_t = dtype(lr) * np.sqrt(1. - beta2 ** t) / (1. - beta1 ** t)
    m = beta1 * m + (1. - beta1) * g
    v = beta2 * v + (1. - beta2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v
