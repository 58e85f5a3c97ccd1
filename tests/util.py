"""Shared test helpers: the flag set create_nerf reads, synthetic poses and ray batches."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def default_args(**over):
    """The hot-path flags with the reference's defaults (args_and_config.py) and the blender paper
    config's sampling (paper_configs/blender_config.txt: 64 coarse, 128 fine, use_viewdirs,
    white_bkgd)."""
    a = argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, N_importance=128, N_samples=64, netchunk=1024 * 64,
        chunk=1024 * 32, perturb=0., raw_noise_std=0., white_bkgd=True, lindisp=False,
        dataset_type='blender', no_ndc=False, basedir='/tmp/nerf_amd_logs', expname='test',
        ft_path=None, no_reload=True, keypoint_embedding_size=None,
        learnable_embeddings_filename=None, choose_keypoint_closest_depth=False,
        lrate=5e-4, lrate_decay=500, N_rand=4096)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def pose_spherical(theta, phi, radius):
    """Camera-to-world of a camera on a sphere looking at the origin (the construction of
    load_blender.py:12-39: translate along z, rotate by phi about x, by theta about y, then the
    blender axis swap)."""
    t = np.eye(4)
    t[2, 3] = radius
    p = phi / 180. * np.pi
    rp = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]])
    th = theta / 180. * np.pi
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    c2w = rt @ rp @ t
    return np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w


def synthetic_rays(n, seed=0, near=2., far=6.):
    """Ray batch [n,11] as render() assembles it: origins on a radius-4 sphere, directions toward
    uniform targets in [-1,1]^3 (not normalised), near/far, unit view directions."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = 4. * o / np.linalg.norm(o, axis=-1, keepdims=True)
    tgt = rng.uniform(-1., 1., size=(n, 3))
    d = tgt - o
    d = d * rng.uniform(0.5, 1.5, size=(n, 1))
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rb = np.concatenate([o, d, np.full((n, 1), near), np.full((n, 1), far), vd], -1)
    return rb.astype(np.float32)


def trained_like_weights(rng, scale_hidden=1.0, sigma_bias=0.5, alpha_scale=1.0, pe_decay=0.0):
    """Glorot-uniform weights (Keras default) with small non-zero biases so every bias path is
    exercised; the alpha bias is raised a little so that rays are neither empty nor opaque.

    Knobs that move the network away from its initialisation, towards what optimisation produces
    (no trained weights exist offline): `scale_hidden` multiplies every kernel (activations grow
    by that factor per layer); `alpha_scale` sharpens the density head (|sigma| in the hundreds,
    as the histogram printed in the reference's extract_mesh.ipynb shows for a trained scene);
    `pe_decay` > 0 damps the rows fed by octave k of the point encoding by 2^(-pe_decay k), the
    spectral fall-off of a fitted field (with white-noise weights on sin(512 x) the field is a
    different random number at every sample and any two arithmetics disagree after resampling)."""
    from oracle import nerf_oracle as orc
    ws = orc.init_nerf_weights(rng)
    for i in range(1, len(ws), 2):
        ws[i] = rng.uniform(-0.1, 0.1, size=ws[i].shape).astype(np.float32)
    for i in range(0, len(ws), 2):
        ws[i] = (ws[i] * scale_hidden).astype(np.float32)
    if pe_decay > 0.:
        for wi in (0, 10):           # layer 0 and the skip rows of layer 5: [x(3), (sin, cos)(6) per octave]
            for k in range(10):
                ws[wi][3 + 6 * k: 9 + 6 * k] *= np.float32(2.0 ** (-pe_decay * k))
    ws[22] = (ws[22] * alpha_scale).astype(np.float32)
    ws[23] = ws[23] + np.float32(sigma_bias)
    return ws


# ---- discontinuities of the reference's own arithmetic ---------------------------------------------
# Two steps of the path are discontinuous functions of the network output, so two arithmetics that
# differ by rounding -- the reference's fp32 against fp64 included -- can land on different sides and
# then differ by far more than the rounding.  The parity tests find the rays on which the two sides
# being compared DID land on different sides (an exact criterion, not a neighbourhood), compare every
# other ray at the stated tolerance and bound the share of rays left out.
def last_sigma_crossed(raw_ref, raw_got, noise=None, noise_std=0.):
    """[N] bool.  The last sample's interval is 1e10 long (raw2outputs), so its alpha is a step
    function of the sign of its sigma (+ noise x std when the density noise is on): rays on which
    the two network outputs [N,S,4] put that sigma on different sides of zero."""
    a = np.asarray(raw_ref, dtype=np.float64)[:, -1, 3]
    b = np.asarray(raw_got, dtype=np.float64)[:, -1, 3]
    if noise is not None and noise_std > 0.:
        a = a + np.asarray(noise, np.float64)[:, -1] * noise_std
        b = b + np.asarray(noise, np.float64)[:, -1] * noise_std
    return (a > 0) != (b > 0)


def resample_crossed(w_ref, w_got, n_importance, u=None):
    """[N] bool.  sample_pdf (run_nerf_helpers.py:185-213) replaces the mass of a bin under 1e-5 by 1
    (:209): inside such a bin every u maps to the bin's left edge, so a sample jumps by a bin width
    when its u moves across one of the bin's knots.  Rays that own a sample which the two coarse
    weight sets (w_ref, w_got: [N,Sc]) put into DIFFERENT bins, one of them such a bin.  (A sample
    that changes between two ordinary bins moves continuously and is not counted.)"""
    def cdf_of(w):
        ww = np.asarray(w, dtype=np.float64)[:, 1:-1] + 1e-5
        pdf = ww / ww.sum(-1, keepdims=True)
        cdf = np.concatenate([np.zeros_like(pdf[:, :1]), np.cumsum(pdf, -1)], -1)
        return cdf, np.diff(cdf, axis=-1) < 1e-5
    cdf_r, guard_r = cdf_of(w_ref)
    cdf_g, guard_g = cdf_of(w_got)
    n, nb = guard_r.shape
    uu = np.broadcast_to(np.linspace(0., 1., n_importance), (n, n_importance)) if u is None else np.asarray(u, np.float64)
    out = np.zeros(n, bool)
    for i in range(n):
        br = np.clip(np.searchsorted(cdf_r[i], uu[i], side='right') - 1, 0, nb - 1)
        bg = np.clip(np.searchsorted(cdf_g[i], uu[i], side='right') - 1, 0, nb - 1)
        out[i] = np.any((br != bg) & (guard_r[i, br] | guard_g[i, bg]))
    return out


def assert_rgb_parity(err, crossed, tol=2e-3, max_crossed=0.02, what=""):
    """Every ray outside `crossed` within `tol` (max abs over its channels), and at most `max_crossed`
    of the rays inside it.  err: [N] or [N,C]; crossed: [N] bool."""
    crossed = np.asarray(crossed, bool).reshape(-1)
    err = np.asarray(err, dtype=np.float64).reshape(len(crossed), -1).max(-1)
    worst = float(err[~crossed].max()) if (~crossed).any() else 0.
    print(f"{what}: {crossed.mean():.4f} of {len(crossed)} rays across a discontinuity of the reference "
          f"(max error {float(err[crossed].max()) if crossed.any() else 0.:.2e}); every other ray within {worst:.2e}")
    assert worst < tol, f"{what}: {worst:.3e} on a ray that crossed no discontinuity"
    assert crossed.mean() <= max_crossed, f"{what}: {crossed.mean():.3f} of the rays left out"


def fitted_like_weights(rng, **kw):
    """trained_like_weights with the spectral fall-off of a fitted field (octave k of the point
    encoding damped by 2^-k).  Comparisons that run THROUGH the resampling step use these: with
    white-noise weights on sin(512 x) the field is a different random number every 1e-3 along a ray,
    so the 1e-4-of-a-bin displacement that any rounding of the coarse weights gives the fine samples
    changes the colour by more than the tolerance -- a property of that field under the reference's
    own arithmetic, not of the implementation (the stage-wise comparisons, which give both sides the
    same depths, use the white-noise fields too)."""
    kw.setdefault('pe_decay', 1.0)
    return trained_like_weights(rng, **kw)


def gpu_coarse_pass(model, rays11, N_samples, lindisp=False, white_bkgd=False, perturb_u=None, noise0=None,
                    noise_std=0.):
    """The coarse half of render_rays through the per-stage entry points, for tests that need the
    HIP path's coarse network output and weights next to what render() returns: (z [N,Sc],
    raw [N,Sc,4], weights [N,Sc]) as numpy arrays."""
    from nerf_amd import ops
    z = ops.sample_z(rays11, N_samples, lindisp, perturb_u)
    raw = ops.mlp_forward_rays(model.packed(), rays11, z)
    w = ops.composite_fwd(raw, z, rays11[:, 3:6], rays11.shape[1], white_bkgd, noise0, noise_std)[3]
    return z.cpu().numpy(), raw.cpu().numpy(), w.cpu().numpy()


def crossed_rays(ref_raw0, ref_raw, ref_w0, got_raw0, got_raw, got_w0, n_importance, pdf_u=None, noise0=None,
                 noise1=None, noise_std=0.):
    """[N] bool: rays on which the reference side (coarse output raw0, fine output raw, coarse
    weights w0) and the side under test landed on different sides of one of the reference's
    discontinuities (last_sigma_crossed for either network, resample_crossed)."""
    out = last_sigma_crossed(ref_raw0, got_raw0, noise0, noise_std)
    out = out | last_sigma_crossed(ref_raw, got_raw, noise1, noise_std)
    return out | resample_crossed(ref_w0, got_w0, n_importance, pdf_u)


def check_rows_against_restatement(gpu, got_rgb, rays11, models, white, Sc, Si, lindisp=False, tol=2e-3, what="rows"):
    """A slice of a full-size render (got_rgb [n,3], the ray batch rows rays11 [n,11] it came from)
    against the float64 torch restatement of the whole path run next to it on the GPU, free-running:
    `tol` on every ray that crossed none of the reference's discontinuities, and the PSNR criterion
    on all of them."""
    import torch
    from nerf_amd import ops
    from oracle import torch_restatement as trs
    wc = [torch.as_tensor(w, device=gpu).double() for w in models['model'].get_weights()]
    wf = [torch.as_tensor(w, device=gpu).double() for w in models['model_fine'].get_weights()]
    rays11 = rays11.contiguous()
    out = trs.render_flat(rays11.double(), wc, wf, N_samples=Sc, N_importance=Si, white_bkgd=white, lindisp=lindisp)
    ref = out['rgb_map']
    got = got_rgb.reshape(-1, 3).double()
    # the same rows once more through the per-stage entry points, for the network outputs render() does not return
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=gpu)
    zc, raw_c, w_c = gpu_coarse_pass(models['model'], rays11, Sc, lindisp, white)
    zf = ops.resample_merge(t(zc), t(w_c), Si, det=True)[0]
    raw_f = ops.mlp_forward_rays(models['model_fine'].packed(), rays11, zf)
    assert torch.equal(ops.composite_fwd(raw_f, zf, rays11[:, 3:6], 11, white)[0], got_rgb.reshape(-1, 3))
    host = lambda k: out[k].cpu().numpy()
    crossed = crossed_rays(host('raw0'), host('raw'), host('weights0'), raw_c, raw_f.cpu().numpy(), w_c, Si)
    assert_rgb_parity((got - ref).abs().cpu().numpy(), crossed, tol, what=what)
    g = torch.Generator(device=gpu)
    g.manual_seed(1)
    target = (ref + torch.randn(ref.shape, generator=g, device=gpu, dtype=torch.float64) * 0.05).clamp(0, 1)
    psnr = lambda x: float(-10. * torch.log10(torch.mean((x - target) ** 2)))
    assert abs(psnr(got) - psnr(ref)) <= 0.05
