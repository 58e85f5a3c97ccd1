"""GPU parity: every HIP entry point against the CPU oracle on the same seeded inputs.

Tolerances (from the north-star statement of the path): the fp32 ray / compositing / sampling
kernels agree with the fp64 oracle within 1e-5; the fp16-MFMA MLP within a max abs rgb error of
2e-3 after the sigmoid.
"""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as orc
from tests.util import default_args, pose_spherical, synthetic_rays, trained_like_weights

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-5
RGB_TOL = 2e-3


def _t(x, gpu):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=gpu)


# ---------------------------------------------------------------------------------------------
# rays
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,W,focal", [(4, 4, 2.0), (37, 53, 41.7), (400, 400, 555.5555155968841)])
def test_get_rays_matches_fp32_oracle_bitwise(gpu, H, W, focal):
    from nerf_amd.run_nerf_helpers import get_rays
    c2w = pose_spherical(37., -30., 4.)[:3, :4].astype(np.float32)
    o, d = get_rays(H, W, focal, torch.as_tensor(c2w))
    ro, rd = orc.get_rays(H, W, focal, c2w, np.float32)
    assert o.shape == (H, W, 3) and d.shape == (H, W, 3)
    np.testing.assert_array_equal(o.cpu().numpy(), ro)
    np.testing.assert_array_equal(d.cpu().numpy(), rd)


def test_get_rays_centre_pixel_known_answer(gpu):
    from nerf_amd.run_nerf_helpers import get_rays, get_rays_np
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    o, d = get_rays(4, 4, 2.0, torch.as_tensor(c2w))
    np.testing.assert_array_equal(d[2, 2].cpu().numpy(), [0., 0., -1.])
    np.testing.assert_array_equal(d[0, 0].cpu().numpy(), [-1., 1., -1.])
    on, dn = get_rays_np(4, 4, 2.0, c2w)
    assert isinstance(on, np.ndarray)
    np.testing.assert_array_equal(dn, d.cpu().numpy())


def test_ndc_rays(gpu):
    from nerf_amd.run_nerf_helpers import ndc_rays
    H, W, focal = 378, 504, 407.5
    rng = np.random.default_rng(3)
    o = rng.normal(size=(1000, 3)).astype(np.float32) * 0.3
    d = rng.normal(size=(1000, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.5
    oo, od = ndc_rays(H, W, focal, 1., _t(o, gpu), _t(d, gpu))
    ro, rd = orc.ndc_rays(H, W, focal, 1., o, d, np.float64)
    np.testing.assert_allclose(oo.cpu().numpy(), ro, atol=FP32_TOL, rtol=1e-5)
    np.testing.assert_allclose(od.cpu().numpy(), rd, atol=FP32_TOL, rtol=1e-5)
    # known answer: the optical axis through the origin maps to o'=(0,0,-1), d'=(0,0,2)
    ko, kd = ndc_rays(H, W, focal, 1., _t([[0., 0., 0.]], gpu), _t([[0., 0., -1.]], gpu))
    np.testing.assert_array_equal(ko.cpu().numpy(), [[0., 0., -1.]])
    np.testing.assert_array_equal(kd.cpu().numpy(), [[0., 0., 2.]])


@pytest.mark.parametrize("ndc", [False, True])
def test_render_ray_batch_assembly(gpu, ndc):
    from nerf_amd import ops
    H, W, focal = 20, 30, 25.0
    c2w = pose_spherical(10., -20., 3.)[:3, :4].astype(np.float32)
    ro, rd = orc.get_rays(H, W, focal, c2w, np.float32)
    rb = ops.build_ray_batch(_t(ro, gpu), _t(rd, gpu), _t(rd, gpu), ndc, H, W, focal, 0.5, 3.5)
    vd = rd / np.linalg.norm(rd, axis=-1, keepdims=True)
    if ndc:
        eo, ed = orc.ndc_rays(H, W, focal, 1., ro, rd, np.float64)
    else:
        eo, ed = ro, rd
    exp = np.concatenate([eo.reshape(-1, 3), ed.reshape(-1, 3), np.full((H * W, 1), 0.5),
                          np.full((H * W, 1), 3.5), vd.reshape(-1, 3)], -1)
    np.testing.assert_allclose(rb.cpu().numpy(), exp, atol=FP32_TOL, rtol=1e-5)


def test_embed(gpu):
    from nerf_amd.run_nerf_helpers import get_embedder
    fn, dim = get_embedder(10, 0)
    fnv, dimv = get_embedder(4, 0)
    assert (dim, dimv) == (63, 27)
    x = np.random.default_rng(0).uniform(-4, 4, size=(257, 3)).astype(np.float32)
    e = fn(_t(x, gpu)).cpu().numpy()
    ref = orc.embed(x, 10, np.float64)
    # |d sin(512 x)| = 512 |dx|: the fp32 argument's own rounding is outside this comparison,
    # because the oracle starts from the same fp32 x and forms x*f exactly (f is a power of two)
    np.testing.assert_allclose(e, ref, atol=2e-6)
    z = fn(_t(np.zeros((1, 3)), gpu)).cpu().numpy()[0]
    assert np.all(z[:3] == 0) and np.all(z[3:].reshape(10, 2, 3)[:, 0] == 0) and np.all(z[3:].reshape(10, 2, 3)[:, 1] == 1)
    ident, d3 = get_embedder(10, -1)
    assert d3 == 3


# ---------------------------------------------------------------------------------------------
# depth sampling
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,lindisp,perturb", [(64, False, False), (64, True, False), (256, False, True),
                                               (7, True, True)])
def test_sample_z(gpu, S, lindisp, perturb):
    from nerf_amd import ops
    rb = synthetic_rays(300, seed=1, near=2., far=6.)
    rb[:, 6] = np.linspace(0.5, 2., 300)
    rb[:, 7] = np.linspace(4., 9., 300)
    u = np.random.default_rng(2).uniform(size=(300, S)).astype(np.float32) if perturb else None
    z = ops.sample_z(_t(rb, gpu), S, lindisp, None if u is None else _t(u, gpu))
    ref = orc.sample_z(rb[:, 6:7], rb[:, 7:8], S, lindisp, u, np.float64)
    np.testing.assert_allclose(z.cpu().numpy(), ref, atol=FP32_TOL, rtol=2e-6)


# ---------------------------------------------------------------------------------------------
# compositing
# ---------------------------------------------------------------------------------------------
def _composite_case(n, S, seed):
    rng = np.random.default_rng(seed)
    raw = rng.normal(size=(n, S, 4)).astype(np.float32) * 2.0
    raw[..., 3] = rng.normal(size=(n, S)) * 3.0 - 1.0
    z = np.sort(rng.uniform(2., 6., size=(n, S)), -1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    return raw, z, d


@pytest.mark.parametrize("S", [1, 2, 63, 64, 65, 128, 192, 384])
@pytest.mark.parametrize("white", [False, True])
def test_composite_fwd(gpu, S, white):
    from nerf_amd.nerf_renderer import raw2outputs
    raw, z, d = _composite_case(513, S, 10 + S)
    out = raw2outputs(_t(raw, gpu), _t(z, gpu), _t(d, gpu), 0., white)
    ref = orc.raw2outputs(raw, z, d, 0., white, dtype=np.float64)
    names = ["rgb_map", "disp_map", "acc_map", "weights", "depth_map"]
    for name, a, b in zip(names, out, ref):
        a = a.cpu().numpy()
        if name == "disp_map":
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=FP32_TOL, err_msg=name)
        else:
            np.testing.assert_allclose(a, b, atol=FP32_TOL * max(1., float(np.max(np.abs(b)))), err_msg=name)


def test_composite_noise_and_edge_cases(gpu):
    from nerf_amd import ops
    raw, z, d = _composite_case(100, 64, 5)
    noise = np.random.default_rng(6).normal(size=(100, 64)).astype(np.float32)
    out = ops.composite_fwd(_t(raw, gpu), _t(z, gpu), _t(d, gpu), 3, False, _t(noise, gpu), 0.7)
    ref = orc.raw2outputs(raw, z, d, 0.7, False, noise=noise, dtype=np.float64)
    np.testing.assert_allclose(out[0].cpu().numpy(), ref[0], atol=FP32_TOL)
    np.testing.assert_allclose(out[3].cpu().numpy(), ref[3], atol=FP32_TOL)
    # sigma == 0 everywhere: all maps 0, white background -> rgb 1, disparity is 0/0 = NaN as in
    # the reference's 1/max(1e-10, depth/acc)
    raw0 = raw.copy()
    raw0[..., 3] = -1.0
    out0 = ops.composite_fwd(_t(raw0, gpu), _t(z, gpu), _t(d, gpu), 3, True)
    assert torch.all(out0[0] == 1) and torch.all(out0[2] == 0) and torch.all(out0[4] == 0)
    assert torch.all(torch.isnan(out0[1]))
    # one opaque sample: rgb is that sample's colour, acc 1, depth its z
    raw1 = raw.copy()
    raw1[..., 3] = -1.0
    raw1[:, 17, 3] = 1e12
    out1 = ops.composite_fwd(_t(raw1, gpu), _t(z, gpu), _t(d, gpu), 3, False)
    col = 1. / (1. + np.exp(-raw1[:, 17, :3].astype(np.float64)))
    np.testing.assert_allclose(out1[0].cpu().numpy(), col, atol=FP32_TOL)
    np.testing.assert_allclose(out1[2].cpu().numpy(), 1., atol=FP32_TOL)
    np.testing.assert_allclose(out1[4].cpu().numpy(), z[:, 17], atol=FP32_TOL * 6)


def test_composite_constant_density_known_answer(gpu):
    """Constant sigma on uniform spacing: T_i = exp(-sigma*delta*i), acc = 1 (last dist 1e10)."""
    from nerf_amd import ops
    S, sigma, delta = 64, 0.8, 0.0625
    z = (2. + delta * np.arange(S))[None].astype(np.float32)
    raw = np.zeros((1, S, 4), np.float32)
    raw[..., 3] = sigma
    d = np.array([[0., 0., 1.]], np.float32)
    out = ops.composite_fwd(_t(raw, gpu), _t(z, gpu), _t(d, gpu), 3, False)
    T = np.exp(-sigma * delta * np.arange(S))
    w = T * (1 - np.exp(-sigma * delta))
    w[-1] = T[-1]
    np.testing.assert_allclose(out[3].cpu().numpy()[0], w, atol=FP32_TOL)
    np.testing.assert_allclose(out[2].cpu().numpy(), 1., atol=FP32_TOL)
    np.testing.assert_allclose(out[0].cpu().numpy(), 0.5, atol=FP32_TOL)


# ---------------------------------------------------------------------------------------------
# hierarchical sampling
# ---------------------------------------------------------------------------------------------
def _pdf_reference(bins, w, ns, det, u):
    """fp64 oracle samples plus, per sample, the error an fp32 evaluation may legitimately show.

    sample = b0 + (u - c0)/denom * (b1 - b0): a CDF rounding error e (a few fp32 ulps of 1.0)
    moves the sample by e/denom * (b1 - b0), and denom is only guarded from below at 1e-5, so
    samples that fall in bins of tiny mass are ill-conditioned in the reference itself.  Where a
    bin's mass is under the 1e-5 guard the map u -> sample jumps at the bin's edge ("knife edge"):
    a u within fp32 rounding of such a CDF knot may land on either side.  Returns ref, tol, knife.
    """
    ref = orc.sample_pdf(bins, w, ns, det=det, u=u, dtype=np.float64)
    ww = w.astype(np.float64) + 1e-5
    pdf = ww / ww.sum(-1, keepdims=True)
    cdf = np.concatenate([np.zeros_like(pdf[:, :1]), np.cumsum(pdf, -1)], -1)
    uu = np.broadcast_to(np.linspace(0., 1., ns), ref.shape) if det else u.astype(np.float64)
    dist = np.abs(uu[:, :, None] - cdf[:, None, :])
    knot = np.argmin(dist, -1)
    knot_dist = np.take_along_axis(dist, knot[..., None], -1)[..., 0]
    nbins = pdf.shape[-1]
    b64 = bins.astype(np.float64)
    cond = np.diff(b64, axis=-1) / np.where(pdf < 1e-5, 1., pdf)        # |d sample / d cdf| per bin
    small = pdf < 1.2e-5
    pick = lambda arr, idx, fill: np.where((idx >= 0) & (idx < nbins),
                                           np.take_along_axis(arr, np.clip(idx, 0, nbins - 1), -1), fill)
    inds = np.stack([np.searchsorted(c, x, side='right') for c, x in zip(cdf, uu)], 0) - 1  # bin of u
    worst = np.maximum.reduce([pick(cond, inds + k, 0.) for k in (-1, 0, 1)])
    tol = 1e-5 * np.maximum(1., np.abs(ref)) + 1e-6 * worst
    knife = (knot_dist < 2e-6) & (pick(small, knot - 1, False) | pick(small, knot, False))
    return ref, tol, knife


def _assert_samples(s, ref, tol, knife):
    bad = np.abs(s - ref) > tol
    assert not np.any(bad & ~knife), f"max err {np.max(np.abs(s - ref)[bad & ~knife])}"
    assert knife.mean() < 0.05


@pytest.mark.parametrize("nb,ns,det", [(63, 128, True), (63, 128, False), (255, 128, True), (5, 7, False),
                                       (2, 1, True), (64, 64, True), (100, 33, False), (300, 64, True)])
def test_sample_pdf(gpu, nb, ns, det):
    from nerf_amd.run_nerf_helpers import sample_pdf
    rng = np.random.default_rng(nb * 7 + ns)
    n = 301
    bins = np.sort(rng.uniform(2., 6., size=(n, nb)), -1).astype(np.float32)
    w = (rng.uniform(size=(n, nb - 1)) ** 4).astype(np.float32)
    w[::7] = 0.  # all-zero rows -> uniform pdf from the 1e-5 floor
    u = None if det else rng.uniform(size=(n, ns)).astype(np.float32)
    s = sample_pdf(_t(bins, gpu), _t(w, gpu), ns, det=det, u=None if det else _t(u, gpu))
    ref, tol, knife = _pdf_reference(bins, w, ns, det, u)
    _assert_samples(s.cpu().numpy(), ref, tol, knife)
    # on well-conditioned inputs (no tiny-mass bins) the plain 1e-5 bound holds everywhere
    w2 = (0.2 + rng.uniform(size=(n, nb - 1))).astype(np.float32)
    s2 = sample_pdf(_t(bins, gpu), _t(w2, gpu), ns, det=det, u=None if det else _t(u, gpu)).cpu().numpy()
    ref2, _, knife2 = _pdf_reference(bins, w2, ns, det, u)
    assert np.max(np.abs(s2 - ref2)[~knife2]) < 3e-5


def test_sample_pdf_known_answers(gpu):
    from nerf_amd.run_nerf_helpers import sample_pdf
    bins = np.linspace(2., 6., 63)[None].astype(np.float32)
    w = np.ones((1, 62), np.float32)
    s = sample_pdf(_t(bins, gpu), _t(w, gpu), 128, det=True).cpu().numpy()
    np.testing.assert_allclose(s[0], np.linspace(2., 6., 128), atol=2e-5)
    w = np.zeros((1, 62), np.float32)
    w[0, 20] = 1.
    s = sample_pdf(_t(bins, gpu), _t(w, gpu), 128, det=True).cpu().numpy()
    inside = (s[0] >= bins[0, 20] - 1e-6) & (s[0] <= bins[0, 21] + 1e-6)
    assert inside.sum() >= 125  # all but the few samples carried by the 1e-5 floor mass


@pytest.mark.parametrize("Sc,Si,det", [(64, 128, True), (64, 128, False), (256, 128, True), (5, 3, False),
                                       (64, 100, False), (128, 64, True), (65, 128, True), (300, 128, False)])
def test_resample_merge(gpu, Sc, Si, det):
    from nerf_amd import ops
    rng = np.random.default_rng(Sc + Si)
    n = 257
    z = np.sort(rng.uniform(2., 6., size=(n, Sc)), -1).astype(np.float32)
    w = (rng.uniform(size=(n, Sc)) ** 6).astype(np.float32)
    u = None if det else rng.uniform(size=(n, Si)).astype(np.float32)
    zo, zs, zstd = ops.resample_merge(_t(z, gpu), _t(w, gpu), Si, det=det, u=None if det else _t(u, gpu))
    zo, zs, zstd = zo.cpu().numpy(), zs.cpu().numpy(), zstd.cpu().numpy()
    z_mid = (.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    ref_s, tol, knife = _pdf_reference(z_mid, w[:, 1:-1], Si, det, u)
    _assert_samples(zs, ref_s, tol, knife)
    # the merged list is exactly sort(concat(coarse depths, the kernel's own samples))
    assert np.all(np.diff(zo, axis=-1) >= 0), "merged depths must be sorted"
    np.testing.assert_array_equal(zo, np.sort(np.concatenate([z, zs], -1), -1))
    np.testing.assert_allclose(zstd, np.std(zs.astype(np.float64), -1), atol=FP32_TOL)


@pytest.mark.parametrize("Sc,Si,det", [(64, 128, True), (100, 64, False)])
def test_resample_merge_many_rays_per_wave(gpu, Sc, Si, det):
    """More rays than resident waves: a wave walks several rays (reading the next one's row
    ahead); the result of every ray equals that of the same ray processed in a small call."""
    from nerf_amd import ops
    g = torch.Generator(device=gpu).manual_seed(Sc)
    n = 70001
    z = torch.sort(torch.rand((n, Sc), generator=g, device=gpu) * 4 + 2, dim=-1).values
    w = torch.rand((n, Sc), generator=g, device=gpu) ** 6
    u = None if det else torch.rand((n, Si), generator=g, device=gpu)
    zo, zs, zstd = ops.resample_merge(z, w, Si, det=det, u=u)
    for lo, hi in ((0, 300), (32768, 33000), (69900, 70001)):
        zo1, zs1, zstd1 = ops.resample_merge(z[lo:hi].contiguous(), w[lo:hi].contiguous(), Si, det=det,
                                            u=None if det else u[lo:hi].contiguous())
        assert torch.equal(zo[lo:hi], zo1) and torch.equal(zs[lo:hi], zs1) and torch.equal(zstd[lo:hi], zstd1)


# ---------------------------------------------------------------------------------------------
# MLP
# ---------------------------------------------------------------------------------------------
def _model_from(ws, gpu):
    from nerf_amd.run_nerf_helpers import init_nerf_model
    m = init_nerf_model(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4],
                        use_viewdirs=True, device=gpu)
    m.set_weights(ws)
    return m


def _check_raw(raw, ref):
    """rgb after the sigmoid within 2e-3; sigma (pre-relu, unbounded) relative to its scale."""
    raw = raw.astype(np.float64)
    def sig(v):   # the logistic function without an overflowing exponential
        e = np.exp(-np.abs(v))
        return np.where(v >= 0, 1. / (1. + e), e / (1. + e))
    e_rgb = np.max(np.abs(sig(raw[..., :3]) - sig(ref[..., :3])))
    e_sig = np.max(np.abs(raw[..., 3] - ref[..., 3])) / max(1., float(np.max(np.abs(ref[..., 3]))))
    assert e_rgb < RGB_TOL, f"rgb error {e_rgb}"
    assert e_sig < 4e-3, f"sigma error {e_sig}"
    return e_rgb, e_sig


@pytest.mark.parametrize("n,S", [(4, 64), (7, 37), (1, 1), (33, 192)])
def test_mlp_forward_pts(gpu, n, S):
    from nerf_amd.nerf_renderer import run_network
    from nerf_amd.run_nerf_helpers import get_embedder
    rng = np.random.default_rng(n * 1000 + S)
    ws = trained_like_weights(rng)
    model = _model_from(ws, gpu)
    pts = rng.uniform(-3., 3., size=(n, S, 3)).astype(np.float32)
    vd = rng.normal(size=(n, 3))
    vd = (vd / np.linalg.norm(vd, axis=-1, keepdims=True)).astype(np.float32)
    e, _ = get_embedder(10, 0)
    ev, _ = get_embedder(4, 0)
    raw = run_network(_t(pts, gpu), _t(vd, gpu), model, e, ev).cpu().numpy()
    assert raw.shape == (n, S, 4)
    ref = orc.run_network(pts, vd, ws, dtype=np.float64)
    _check_raw(raw, ref)


@pytest.mark.parametrize("scale_hidden,alpha_scale,sigma_bias", [(1., 300., 50.), (2., 1., .5), (2., 30., 5.), (3., 1., .5)])
def test_mlp_forward_away_from_glorot_scale(gpu, scale_hidden, alpha_scale, sigma_bias):
    """Weights scaled up (activations grow by the factor per layer: x256 / x6561 by the last layer)
    and a sharpened density head (|sigma| in the hundreds and beyond).  fp16 operands carry a
    relative error, so the bound that holds at every scale is relative to the magnitude of the
    channel: 4e-3 of max|rgb pre-activation| and of max|sigma|; the absolute 2e-3 on the colour
    after the sigmoid holds while the pre-activations are of order one."""
    from nerf_amd import ops
    rng = np.random.default_rng(int(scale_hidden * 10 + alpha_scale))
    ws = trained_like_weights(rng, scale_hidden=scale_hidden, alpha_scale=alpha_scale, sigma_bias=sigma_bias)
    model = _model_from(ws, gpu)
    n, S = 16, 64
    pts = rng.uniform(-3., 3., size=(n, S, 3)).astype(np.float32)
    vd = rng.normal(size=(n, 3))
    vd = (vd / np.linalg.norm(vd, axis=-1, keepdims=True)).astype(np.float32)
    raw = ops.mlp_forward_pts(model.packed(), _t(pts, gpu), _t(vd, gpu)).cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(raw))
    ref = orc.run_network(pts, vd, ws, dtype=np.float64)
    m_rgb, m_sig = float(np.max(np.abs(ref[..., :3]))), float(np.max(np.abs(ref[..., 3])))
    e_pre = float(np.max(np.abs(raw[..., :3] - ref[..., :3])))
    e_sig = float(np.max(np.abs(raw[..., 3] - ref[..., 3])))
    def sig(v):   # the logistic function without an overflowing exponential
        e = np.exp(-np.abs(v))
        return np.where(v >= 0, 1. / (1. + e), e / (1. + e))
    e_rgb = float(np.max(np.abs(sig(raw[..., :3]) - sig(ref[..., :3]))))
    print(f"scale {scale_hidden} alpha x{alpha_scale}: max|rgb_pre| {m_rgb:.3g} max|sigma| {m_sig:.3g} "
          f"err rgb_pre {e_pre:.3g} sigma {e_sig:.3g} sigmoid(rgb) {e_rgb:.3g}")
    assert e_pre < 4e-3 * max(1., m_rgb) and e_sig < 4e-3 * max(1., m_sig)
    if m_rgb < 4.:
        assert e_rgb < RGB_TOL
    if alpha_scale >= 30.:
        assert m_sig > 100.


def test_mlp_layer_wiring_by_zeroing_blocks(gpu):
    """Each structural choice of the network is observable: the skip concat order [pts, h], the
    linear bottleneck, the [bottleneck, views] order and the (rgb, alpha) output order."""
    from nerf_amd import ops
    rng = np.random.default_rng(42)
    base = trained_like_weights(rng)
    pts = rng.uniform(-2., 2., size=(3, 64, 3)).astype(np.float32)
    vd = rng.normal(size=(3, 3))
    vd = (vd / np.linalg.norm(vd, axis=-1, keepdims=True)).astype(np.float32)
    variants = []
    for kind in ["skip_pts_rows", "skip_h_rows", "view_rows", "bott_rows", "alpha_w", "rgb_w", "layer0_last_row"]:
        ws = [w.copy() for w in base]
        if kind == "skip_pts_rows": ws[10][:63] = 0
        if kind == "skip_h_rows": ws[10][63:] = 0
        if kind == "view_rows": ws[18][256:] = 0
        if kind == "bott_rows": ws[18][:256] = 0
        if kind == "alpha_w": ws[22][:] = 0
        if kind == "rgb_w": ws[20][:] = 0
        if kind == "layer0_last_row": ws[0][62] = 0
        variants.append((kind, ws))
    for kind, ws in variants:
        model = _model_from(ws, gpu)
        raw = ops.mlp_forward_pts(model.packed(), _t(pts, gpu), _t(vd, gpu)).cpu().numpy()
        ref = orc.run_network(pts, vd, ws, dtype=np.float64)
        _check_raw(raw, ref)
        if kind == "alpha_w":
            np.testing.assert_allclose(raw[..., 3], ws[23][0], atol=1e-6)
        if kind == "rgb_w":
            np.testing.assert_allclose(raw[..., :3], np.broadcast_to(ws[21], raw[..., :3].shape), atol=1e-6)


def test_mlp_forward_rays_equals_pts(gpu):
    """Forming the points inside the kernel (o + d*z) gives the same bits as passing them in."""
    from nerf_amd import ops
    rng = np.random.default_rng(9)
    model = _model_from(trained_like_weights(rng), gpu)
    rb = synthetic_rays(300, seed=4)
    rays = _t(rb, gpu)
    z = ops.sample_z(rays, 64)
    raw_r = ops.mlp_forward_rays(model.packed(), rays, z)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    raw_p = ops.mlp_forward_pts(model.packed(), pts.contiguous(), rays[:, 8:11].contiguous())
    assert torch.equal(raw_r, raw_p)


@pytest.mark.parametrize("S", [5, 31, 33, 47, 48, 49, 300])
def test_mlp_many_tiles_per_workgroup(gpu, S):
    """More workgroup tiles (384 points: 8 waves x 48) than workgroups, rays shorter and longer than
    a wave's 48 points (and than the 32 of the training forward) and a ragged last tile: every
    point's output equals that of the same point evaluated alone with its ray's direction (one
    point per ray), and that of a small call on a slice of the rays."""
    from nerf_amd import ops
    rng = np.random.default_rng(S)
    model = _model_from(trained_like_weights(rng), gpu)
    n = (384 * 256 * 2 + 1234) // S + 1          # > 2 tiles per workgroup on a 256-CU chip
    rb = synthetic_rays(n, seed=S)
    rays = _t(rb, gpu)
    z = ops.sample_z(rays, S)
    raw = ops.mlp_forward_rays(model.packed(), rays, z)
    assert raw.shape == (n, S, 4)
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]).reshape(-1, 1, 3).contiguous()
    vd = rays[:, None, 8:11].expand(n, S, 3).reshape(-1, 3).contiguous()
    raw_p = ops.mlp_forward_pts(model.packed(), pts, vd)
    assert torch.equal(raw.reshape(-1, 4), raw_p.reshape(-1, 4))
    for lo, hi in ((0, 7), (n // 2, n // 2 + 50), (n - 3, n)):
        raw_s = ops.mlp_forward_rays(model.packed(), rays[lo:hi].contiguous(), z[lo:hi].contiguous())
        assert torch.equal(raw[lo:hi], raw_s)


def test_mlp_large_inputs_and_repack(gpu):
    """Far-away points (|x| ~ 100, sin(512x) needs exact range reduction) and weight updates."""
    from nerf_amd import ops
    rng = np.random.default_rng(77)
    ws = trained_like_weights(rng)
    model = _model_from(ws, gpu)
    pts = rng.uniform(-100., 100., size=(5, 64, 3)).astype(np.float32)
    vd = np.tile(np.array([[0., 0., 1.]], np.float32), (5, 1))
    raw = ops.mlp_forward_pts(model.packed(), _t(pts, gpu), _t(vd, gpu)).cpu().numpy()
    _check_raw(raw, orc.run_network(pts, vd, ws, dtype=np.float64))
    ws2 = trained_like_weights(np.random.default_rng(78))
    model.set_weights(ws2)
    raw2 = ops.mlp_forward_pts(model.packed(), _t(pts, gpu), _t(vd, gpu)).cpu().numpy()
    _check_raw(raw2, orc.run_network(pts, vd, ws2, dtype=np.float64))
    got = model.get_weights()
    for a, b in zip(got, ws2):
        np.testing.assert_array_equal(a, b)


def test_mlp_hidden_activation_saturates_at_fp16_max(gpu):
    """A hidden unit whose pre-activation exceeds the fp16 range is carried as 65504, not as inf:
    the outputs equal those of a network whose unit sits exactly at 65504."""
    from nerf_amd import ops
    rng = np.random.default_rng(5)
    ws = trained_like_weights(rng)
    ws[0][:, 0] = 0
    ws[1][0] = 1e6                                              # unit 0 of layer 0: relu(1e6)
    ws[2][0, :] = rng.choice([-1., 1.], size=256) * 2. ** -13   # its fan-out, exact in fp16
    pts = rng.uniform(-2., 2., size=(3, 40, 3)).astype(np.float32)
    vd = rng.normal(size=(3, 3))
    vd = (vd / np.linalg.norm(vd, axis=-1, keepdims=True)).astype(np.float32)
    raw = ops.mlp_forward_pts(_model_from(ws, gpu).packed(), _t(pts, gpu), _t(vd, gpu)).cpu().numpy()
    assert np.all(np.isfinite(raw))
    ws_sat = [w.copy() for w in ws]
    ws_sat[1][0] = 65504.
    _check_raw(raw, orc.run_network(pts, vd, ws_sat, dtype=np.float64))
    # and the unit matters: without it the outputs differ visibly
    ws_off = [w.copy() for w in ws]
    ws_off[1][0] = 0.
    assert np.max(np.abs(raw - orc.run_network(pts, vd, ws_off, dtype=np.float64))) > 1e-2


# ---------------------------------------------------------------------------------------------
# end to end
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("white_bkgd,lindisp,field,Sc,Si", [
    (True, False, 'glorot', 64, 128), (False, True, 'glorot', 64, 128), (True, False, 'glorot', 64, 64),
    (True, False, 'fitted', 64, 128), (False, True, 'fitted', 64, 64), (True, False, 'sharp', 64, 128)])
def test_render_rays_end_to_end(gpu, white_bkgd, lindisp, field, Sc, Si):
    """Whole path at 64+128 samples (the paper configurations) and 64+64 (config_fern.txt:9-11 and most
    of the shipped configs).  The tolerance is the stated one, max abs rgb error 2e-3, on EVERY ray
    except those on which the two sides landed on different sides of one of the reference's own
    discontinuities (tests/util.py: sign of the last sigma, sub-1e-5-mass bins of sample_pdf); their
    share is bounded.

    (1) stage-wise, every field: the oracle re-runs the fine pass on the depths the GPU produced, so
        what is compared is the fp16 network and the fp32 compositing alone.
    (2) free-running (the oracle resamples for itself), 'fitted' and 'sharp' fields: the same bound.
        On the white-noise 'glorot' fields the free-running comparison is ill-conditioned under the
        reference's own arithmetic (tests/util.py:fitted_like_weights) and is held to the
        image-quality criterion instead: PSNR against a common target moves by <= 0.05 dB.

    field 'sharp': the fitted spectrum with the density head sharpened 300x (|sigma| up to several
    hundred, the magnitude extract_mesh.ipynb prints for a trained scene).  The final colour meets
    the bound.  The COARSE colour rgb0 does not: sigma carries the relative error of fp16 operands
    (about 3e-4 rms, 1.5e-3 max, of max|sigma|; tests/test_precision_budget.py), and on 64 coarse
    samples alpha = 1 - exp(-sigma * 0.25) turns that into percents where a sample sits on a
    surface.  There the kernel is compared with the oracle computing on fp16-rounded operands (what
    any fp16-input matrix unit does): 2e-3 on every ray; against the fp64 oracle rgb0 is bounded at
    2e-2 with 90 % of the rays under 2e-3."""
    from nerf_amd import ops
    from nerf_amd.run_nerf import create_nerf, render
    from tests.util import assert_rgb_parity, crossed_rays, fitted_like_weights, gpu_coarse_pass, last_sigma_crossed
    args = default_args(white_bkgd=white_bkgd, lindisp=lindisp, N_samples=Sc, N_importance=Si)
    kw_train, kw_test, start, grad_vars, models = create_nerf(args)
    rng = np.random.default_rng(5)
    make = {'glorot': trained_like_weights, 'fitted': fitted_like_weights,
            'sharp': lambda r: fitted_like_weights(r, alpha_scale=300., sigma_bias=50.)}[field]
    wc, wf = make(rng), make(rng)
    models['model'].set_weights(wc)
    models['model_fine'].set_weights(wf)
    assert start == 0 and len(grad_vars) == 48
    n = 256
    rb = synthetic_rays(n, seed=11)
    rays = (_t(rb[:, 0:3], gpu), _t(rb[:, 3:6], gpu))
    rgb, disp, acc, extras = render(1, 1, 1., chunk=100, rays=rays, near=2., far=6., retraw=True, **kw_test)
    assert rgb.shape == (n, 3) and disp.shape == (n,) and extras['raw'].shape == (n, Sc + Si, 4)
    assert set(extras) >= {'rgb0', 'disp0', 'acc0', 'z_std', 'raw', 'depth_map'}
    rgb = rgb.cpu().numpy()
    okw = dict(ndc=False, near=2., far=6., use_viewdirs=True, weights_coarse=wc, weights_fine=wf, N_samples=Sc,
               N_importance=Si, white_bkgd=white_bkgd, lindisp=lindisp, perturb=0., retraw=True, dtype=np.float64)
    ref = orc.render(1, 1, 1., chunk=100, rays=(rb[:, 0:3], rb[:, 3:6]), **okw)
    rex = ref[3]
    rays11 = ops.build_ray_batch(rays[0], rays[1], rays[1], False, 1, 1, 1., 2., 6.)
    zc, raw_c, w_c = gpu_coarse_pass(models['model'], rays11, Sc, lindisp, white_bkgd)
    raw_f = extras['raw'].cpu().numpy()
    # the coarse pass has no resampling in front of it
    cross_c = last_sigma_crossed(rex['raw0'], raw_c)
    e0 = np.abs(extras['rgb0'].cpu().numpy() - rex['rgb0'])
    a0 = np.abs(extras['acc0'].cpu().numpy() - rex['acc0'])
    if field == 'sharp':
        assert np.abs(rex['raw'][..., 3]).max() > 300.
        e0r = e0.max(-1)[~cross_c]
        print(f"sharp: coarse rgb0 against the fp64 oracle: max {e0r.max():.2e} q90 {np.quantile(e0r, 0.9):.2e}")
        assert np.quantile(e0r, 0.9) < RGB_TOL and e0r.max() < 2e-2
        ref16 = orc.render(1, 1, 1., chunk=100, rays=(rb[:, 0:3], rb[:, 3:6]), operand_dtype=np.float16, **okw)[3]
        assert_rgb_parity(np.abs(extras['rgb0'].cpu().numpy() - ref16['rgb0']), last_sigma_crossed(ref16['raw0'], raw_c),
                          RGB_TOL, what="sharp: coarse rgb0 against the oracle on fp16 operands")
    else:
        assert_rgb_parity(e0, cross_c, RGB_TOL, what=f"{field}: coarse rgb0")
        assert_rgb_parity(a0, cross_c, RGB_TOL, what=f"{field}: coarse acc0")
    # (1) stage-wise: the fine pass of the oracle on the GPU's own depths
    zf, _, _ = ops.resample_merge(_t(zc, gpu), _t(w_c, gpu), Si, det=True)
    zf = zf.cpu().numpy().astype(np.float64)
    r11 = rays11.cpu().numpy().astype(np.float64)
    pts = r11[:, None, 0:3] + r11[:, None, 3:6] * zf[:, :, None]
    raw_stage = orc.run_network(pts, r11[:, 8:11], wf, dtype=np.float64)
    rgb_stage = orc.raw2outputs(raw_stage, zf, r11[:, 3:6], 0., white_bkgd, dtype=np.float64)[0]
    assert_rgb_parity(np.abs(rgb - rgb_stage), last_sigma_crossed(raw_stage, raw_f), RGB_TOL,
                      what=f"{field} {Sc}+{Si}: final rgb, stage-wise")
    # (2) free-running
    err = np.abs(rgb - ref[0])
    crossed = crossed_rays(rex['raw0'], rex['raw'], rex['weights0'], raw_c, raw_f, w_c, Si)
    if field == 'glorot':
        print(f"glorot {Sc}+{Si}: free-running max {err.max():.2e} q99 {np.quantile(err, 0.99):.2e} (white-noise field: PSNR criterion)")
    else:
        assert_rgb_parity(err, crossed, RGB_TOL, what=f"{field} {Sc}+{Si}: final rgb, free-running")
        assert_rgb_parity(np.abs(acc.cpu().numpy() - ref[2]), crossed, RGB_TOL, what=f"{field} {Sc}+{Si}: acc, free-running")
    target = np.clip(ref[0] + np.random.default_rng(0).normal(size=ref[0].shape) * 0.05, 0, 1)
    psnr = lambda x: -10. * np.log10(np.mean((x - target) ** 2))
    print(f"{field}: PSNR delta {psnr(rgb) - psnr(ref[0]):+.4f} dB")
    assert abs(psnr(rgb) - psnr(ref[0])) <= 0.05  # every ray, crossed ones included


def test_render_full_image_ndc(gpu):
    """Camera path: get_rays -> viewdirs -> ndc_rays -> near 0 / far 1 (the LLFF route)."""
    from nerf_amd.run_nerf import create_nerf, render
    args = default_args(white_bkgd=False, dataset_type='llff', N_samples=32, N_importance=32)
    kw_train, kw_test, _, _, models = create_nerf(args)
    assert 'ndc' not in kw_test and 'lindisp' not in kw_test  # run_nerf.py:230-233
    rng = np.random.default_rng(8)
    wc, wf = trained_like_weights(rng), trained_like_weights(rng)
    models['model'].set_weights(wc)
    models['model_fine'].set_weights(wf)
    H, W, focal = 12, 16, 14.0
    c2w = np.concatenate([np.eye(3), np.array([[0.05], [0.02], [0.1]])], 1).astype(np.float32)
    rgb, disp, acc, extras = render(H, W, focal, chunk=100, c2w=torch.as_tensor(c2w), near=0., far=1., **kw_test)
    ref = orc.render(H, W, focal, chunk=100, c2w=c2w, ndc=True, near=0., far=1., use_viewdirs=True,
                     weights_coarse=wc, weights_fine=wf, N_samples=32, N_importance=32,
                     white_bkgd=False, perturb=0., dtype=np.float64)
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and extras['rgb0'].shape == (H, W, 3)
    np.testing.assert_allclose(rgb.cpu().numpy(), ref[0], atol=RGB_TOL)
    np.testing.assert_allclose(acc.cpu().numpy(), ref[2], atol=RGB_TOL)
