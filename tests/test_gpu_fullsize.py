"""Parity at the benchmark's full sizes (1,048,576 rays, 64 coarse + 128 fine samples; 256 + 128
for the long-scan stress case).  The NumPy oracle takes hours at these sizes, so every kernel is
checked here against the independent torch restatement of the reference's op sequence run in
float64 on the same device, plus properties that hold at any size: chunk invariance of render(),
sortedness and multiset equality of the merged depths, weights summing to the accumulated opacity.

Tolerances as everywhere: fp32 compositing / sampling 1e-5; fp16-MFMA MLP 2e-3 max abs rgb.
"""
import numpy as np
import pytest
import torch

from oracle import torch_restatement as trs
from tests.util import check_rows_against_restatement, default_args, fitted_like_weights, synthetic_rays, trained_like_weights

pytestmark = pytest.mark.gpu

N_FULL = 1024 * 1024
FP32_TOL = 1e-5
RGB_TOL = 2e-3


def _rays(n, gpu, seed=0):
    """The synthetic batch of tests/util.synthetic_rays, generated on the device (same construction)."""
    g = torch.Generator(device=gpu)
    g.manual_seed(seed)
    o = torch.randn((n, 3), generator=g, device=gpu)
    o = 4. * o / o.norm(dim=-1, keepdim=True)
    tgt = torch.rand((n, 3), generator=g, device=gpu) * 2 - 1
    d = (tgt - o) * (0.5 + torch.rand((n, 1), generator=g, device=gpu))
    vd = d / d.norm(dim=-1, keepdim=True)
    return torch.cat([o, d, torch.full((n, 1), 2., device=gpu), torch.full((n, 1), 6., device=gpu), vd], -1).contiguous()


@pytest.mark.parametrize("S,white", [(64, False), (192, True), (384, False)])
def test_composite_full_size(gpu, S, white):
    """raw2outputs over 1,048,576 rays against float64 torch, in 131072-ray pieces."""
    from nerf_amd import ops
    g = torch.Generator(device=gpu)
    g.manual_seed(S)
    n = N_FULL if S <= 192 else N_FULL // 2
    rays = _rays(n, gpu, seed=S)
    raw = torch.randn((n, S, 4), generator=g, device=gpu)
    raw[..., 3] = raw[..., 3] * 3. + 0.5
    z = ops.sample_z(rays, S, False, torch.rand((n, S), generator=g, device=gpu))
    rgb, disp, acc, w, depth = ops.composite_fwd(raw, z, rays[:, 3:6], 11, white)
    assert bool(torch.isfinite(rgb).all()) and float(acc.max()) <= 1. + FP32_TOL and float(w.min()) >= 0.
    assert float((w.sum(-1) - acc).abs().max()) < FP32_TOL
    worst = 0.
    for i in range(0, n, 131072):
        sl = slice(i, i + 131072)
        r = trs.raw2outputs(raw[sl].double(), z[sl].double(), rays[sl, 3:6].double(), white)
        for got, ref in ((rgb[sl], r[0]), (acc[sl], r[2]), (w[sl], r[3])):
            worst = max(worst, float((got.double() - ref).abs().max()))
        worst = max(worst, float(((depth[sl].double() - r[4]).abs() / 6.).max()))
        ok = r[2] > 1e-3  # disparity divides by the accumulated opacity
        worst = max(worst, float((disp[sl].double() - r[1])[ok].abs().max()))
    assert worst < FP32_TOL, worst


@pytest.mark.parametrize("Sc,Si", [(64, 128), (256, 128)])
def test_resample_merge_full_size(gpu, Sc, Si):
    """sample_pdf + sort(concat) over 1,048,576 rays: merged depths are sorted, are exactly the
    multiset union of coarse depths and samples, and the samples match float64 torch."""
    from nerf_amd import ops
    g = torch.Generator(device=gpu)
    g.manual_seed(Sc)
    n = N_FULL
    rays = _rays(n, gpu, seed=Sc)
    z = ops.sample_z(rays, Sc, False, torch.rand((n, Sc), generator=g, device=gpu))
    weights = torch.rand((n, Sc), generator=g, device=gpu) * 0.9 + 0.1   # well-conditioned bins
    for det in (True, False):
        u = None if det else torch.rand((n, Si), generator=g, device=gpu)
        z_out, z_samp, z_std = ops.resample_merge(z, weights, Si, det=det, u=u)
        assert bool((z_out[:, 1:] >= z_out[:, :-1]).all())
        both, _ = torch.sort(torch.cat([z, z_samp], -1), -1)
        assert torch.equal(both, z_out)
        worst = 0.
        for i in range(0, n, 262144):
            sl = slice(i, i + 262144)
            zd = z[sl].double()
            ref = trs.sample_pdf(.5 * (zd[:, 1:] + zd[:, :-1]), weights[sl, 1:-1].double(), Si, det=det,
                                 u=None if det else u[sl].double())
            worst = max(worst, float((z_samp[sl].double() - ref).abs().max()))
            worst = max(worst, float((z_std[sl].double() - torch.std(ref, -1, unbiased=False)).abs().max()))
        assert worst < 3e-5, worst  # depths span [2, 6]: 1e-5 relative


def test_mlp_full_chunk_against_float32_matmuls(gpu):
    """One bench-sized coarse launch (131072 rays x 64 = 8,388,608 points): rgb after the sigmoid
    within 2e-3 of the unfused float32 restatement at every point."""
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import init_nerf_model
    rng = np.random.default_rng(3)
    ws = trained_like_weights(rng)
    model = init_nerf_model(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4],
                            use_viewdirs=True, device=gpu)
    model.set_weights(ws)
    n, S = 131072, 64
    rays = _rays(n, gpu, seed=5)
    z = ops.sample_z(rays, S)
    raw = ops.mlp_forward_rays(model.packed(), rays, z)
    wt = [torch.as_tensor(w, device=gpu) for w in ws]
    e_rgb = e_sig = 0.
    sig_scale = 1.
    prev = torch.backends.cuda.matmul.allow_tf32
    torch.backends.cuda.matmul.allow_tf32 = False
    try:
        for i in range(0, n, 8192):
            sl = slice(i, i + 8192)
            pts = rays[sl, None, 0:3] + rays[sl, None, 3:6] * z[sl, :, None]
            ref = trs.run_network(pts, rays[sl, 8:11], wt, netchunk=1 << 20)
            e_rgb = max(e_rgb, float((torch.sigmoid(raw[sl, :, :3]) - torch.sigmoid(ref[..., :3])).abs().max()))
            e_sig = max(e_sig, float((raw[sl, :, 3] - ref[..., 3]).abs().max()))
            sig_scale = max(sig_scale, float(ref[..., 3].abs().max()))
    finally:
        torch.backends.cuda.matmul.allow_tf32 = prev
    assert e_rgb < RGB_TOL, e_rgb
    assert e_sig / sig_scale < 4e-3, (e_sig, sig_scale)


def test_render_full_size_is_chunk_invariant(gpu):
    """render() of the 1,048,576-ray benchmark batch: the `chunk` size does not change a bit of the
    result (run_nerf.py:39-40 'Does not affect final results'), maps are finite and in range."""
    from nerf_amd.run_nerf import create_nerf, render
    kw_train, kw_test, _, _, models = create_nerf(default_args(white_bkgd=True))
    rng = np.random.default_rng(12)
    models['model'].set_weights(fitted_like_weights(rng))
    models['model_fine'].set_weights(fitted_like_weights(rng))
    rays = _rays(N_FULL, gpu, seed=21)
    o, d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
    a = render(1, 1, 1., chunk=1024 * 128, rays=(o, d), near=2., far=6., **kw_test)
    b = render(1, 1, 1., chunk=1024 * 32, rays=(o, d), near=2., far=6., **kw_test)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    for k in ('rgb0', 'acc0', 'z_std', 'depth_map'):
        assert torch.equal(a[3][k], b[3][k])
    rgb, disp, acc = a[:3]
    assert rgb.shape == (N_FULL, 3) and bool(torch.isfinite(rgb).all())
    assert float(rgb.min()) >= -FP32_TOL and float(rgb.max()) <= 1. + FP32_TOL
    assert float(acc.min()) >= 0. and float(acc.max()) <= 1. + FP32_TOL
    # a 4096-ray slice against the float64 restatement of the whole path, free-running
    from nerf_amd import ops
    sl = slice(500000, 504096)
    rays11 = ops.build_ray_batch(o[sl], d[sl], d[sl], False, 1, 1, 1., 2., 6.)   # as render() assembles them
    check_rows_against_restatement(gpu, rgb[sl], rays11, models, True, 64, 128, what="rays 500000..504095 of the 1M-ray batch")


def test_llff_view_256_plus_128_samples(gpu):
    """The forward-facing stress configuration at full size: a 1008 x 756 view through get_rays ->
    viewdirs -> ndc_rays (near plane 1) -> near 0 / far 1 with 256 coarse + 128 fine samples
    (long per-ray scans: 4 and 6 chunks of 64).  Chunk invariance over the whole view and the
    float64 restatement of the whole path on two rows of pixels."""
    from nerf_amd import ops
    from nerf_amd.run_nerf import create_nerf, render
    H, W, focal = 756, 1008, 815.0
    args = default_args(white_bkgd=False, dataset_type='llff', N_samples=256, N_importance=128)
    kw_train, kw_test, _, _, models = create_nerf(args)
    rng = np.random.default_rng(31)
    models['model'].set_weights(fitted_like_weights(rng))
    models['model_fine'].set_weights(fitted_like_weights(rng))
    c2w = np.concatenate([np.eye(3), np.array([[0.05], [0.02], [0.1]])], 1).astype(np.float32)
    a = render(H, W, focal, chunk=1024 * 64, c2w=torch.as_tensor(c2w), near=0., far=1., **kw_test)
    b = render(H, W, focal, chunk=1024 * 32, c2w=torch.as_tensor(c2w), near=0., far=1., **kw_test)
    rgb, disp, acc = a[:3]
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and a[3]['rgb0'].shape == (H, W, 3)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert bool(torch.isfinite(rgb).all()) and float(acc.max()) <= 1. + FP32_TOL
    # rows 300 and 301 (2016 rays) against the restatement, fed the same ray batch
    o, d = ops.get_rays(H, W, focal, torch.as_tensor(c2w))
    rays = ops.build_ray_batch(o, d, d, True, H, W, focal, 0., 1.)
    sl = slice(300 * W, 302 * W)
    check_rows_against_restatement(gpu, rgb.reshape(-1, 3)[sl], rays[sl], models, False, 256, 128, what="rows 300-301 of the NDC view")
