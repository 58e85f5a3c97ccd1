"""CPU tests: the NumPy oracle against analytic known answers, and against the independently
written torch restatement.  (The reference holds no golden vectors for this path, so these are
what pins the oracle.)"""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as orc
from oracle import torch_restatement as trs
from tests.util import synthetic_rays, trained_like_weights


def test_embedding_dims_and_zero_pattern():
    # out dims 63 / 27 are what the reference's notebooks print ("MODEL 63 27", "(?, 90) (?, 63) (?, 27)")
    assert orc.get_embedder(10)[1] == 63 and orc.get_embedder(4)[1] == 27
    assert orc.get_embedder(10, -1)[1] == 3
    e = orc.embed(np.zeros((1, 3)), 10)[0]
    assert np.all(e[:3] == 0)
    blocks = e[3:].reshape(10, 2, 3)
    assert np.all(blocks[:, 0] == 0) and np.all(blocks[:, 1] == 1)  # sin 0, cos 0 per frequency
    x = np.array([[np.pi / 2, np.pi / 4, np.pi / 8]])
    e = orc.embed(x, 4)[0]
    np.testing.assert_allclose(e[3:6], np.sin(x[0]), atol=1e-15)       # f = 1: sin
    np.testing.assert_allclose(e[6:9], np.cos(x[0]), atol=1e-15)       # f = 1: cos
    np.testing.assert_allclose(e[9:12], np.sin(2 * x[0]), atol=1e-15)  # f = 2: sin
    assert abs(e[9] - 0.) < 1e-15 and abs(e[10] - 1.) < 1e-15


def test_get_rays_known_answer():
    c2w = np.concatenate([np.eye(3), np.array([[1.], [2.], [3.]])], 1)
    o, d = orc.get_rays(4, 4, 2.0, c2w)
    np.testing.assert_array_equal(d[2, 2], [0., 0., -1.])   # pixel (W/2, H/2) looks down -z
    np.testing.assert_array_equal(d[0, 0], [-1., 1., -1.])  # i = column, j = row, y flipped
    np.testing.assert_array_equal(d[0, 3], [0.5, 1., -1.])
    np.testing.assert_array_equal(o[1, 2], [1., 2., 3.])
    # rotation applies as R . dir
    R = np.array([[0., -1., 0.], [1., 0., 0.], [0., 0., 1.]])
    o, d = orc.get_rays(4, 4, 2.0, np.concatenate([R, np.zeros((3, 1))], 1))
    np.testing.assert_array_equal(d[0, 0], R @ np.array([-1., 1., -1.]))


def test_ndc_known_answer():
    o, d = orc.ndc_rays(100, 200, 50., 1., np.array([[0., 0., 0.]]), np.array([[0., 0., -1.]]))
    np.testing.assert_array_equal(o, [[0., 0., -1.]])
    np.testing.assert_array_equal(d, [[0., 0., 2.]])


def test_compositing_known_answers():
    S, sigma, delta = 64, 0.8, 0.0625
    z = (2. + delta * np.arange(S))[None]
    raw = np.zeros((1, S, 4))
    raw[..., 3] = sigma
    d = np.array([[0., 0., 1.]])
    rgb, disp, acc, w, depth = orc.raw2outputs(raw, z, d, dtype=np.float64)
    T = np.exp(-sigma * delta * np.arange(S))
    we = T * (1 - np.exp(-sigma * delta))
    we[-1] = T[-1]  # last dist = 1e10 -> alpha = 1
    np.testing.assert_allclose(w[0], we, atol=1e-8)
    np.testing.assert_allclose(acc, 1., atol=1e-8)
    np.testing.assert_allclose(rgb, 0.5, atol=1e-8)  # sigmoid(0)
    # ray-norm scaling: doubling ||d|| doubles the optical depth
    _, _, _, w2, _ = orc.raw2outputs(raw, z, 2 * d, dtype=np.float64)
    np.testing.assert_allclose(w2[0, 1], np.exp(-2 * sigma * delta) * (1 - np.exp(-2 * sigma * delta)), atol=1e-8)
    _, _, _, w3, _ = orc.raw2outputs(raw, z, 2 * d, scale_dists_by_ray_norm=False, dtype=np.float64)
    np.testing.assert_allclose(w3, w, atol=1e-12)
    # empty space: everything 0, white background gives 1, disparity 0/0
    raw[..., 3] = -5.
    rgb, disp, acc, w, depth = orc.raw2outputs(raw, z, d, white_bkgd=True, dtype=np.float64)
    assert np.all(rgb == 1.) and np.all(acc == 0.) and np.all(np.isnan(disp))
    # one opaque sample
    raw[0, 10] = [0.3, -0.2, 1.5, 1e4]
    rgb, disp, acc, w, depth = orc.raw2outputs(raw, z, d, dtype=np.float64)
    np.testing.assert_allclose(rgb[0], 1 / (1 + np.exp(-np.array([0.3, -0.2, 1.5]))), atol=1e-9)
    np.testing.assert_allclose(depth[0], z[0, 10], atol=1e-8)
    np.testing.assert_allclose(disp[0], 1 / z[0, 10], atol=1e-8)


def test_sample_pdf_known_answers():
    bins = np.linspace(2., 6., 63)[None]
    s = orc.sample_pdf(bins, np.ones((1, 62)), 128, det=True, dtype=np.float64)
    np.testing.assert_allclose(s[0], np.linspace(2., 6., 128), atol=1e-9)
    w = np.zeros((1, 62))
    w[0, 20] = 1.
    s = orc.sample_pdf(bins, w, 128, det=True, dtype=np.float64)[0]
    assert np.sum((s >= bins[0, 20]) & (s <= bins[0, 21])) >= 126
    # caller's weights are not modified (the reference's `weights += 1e-5` rebinds a tensor)
    assert w[0, 0] == 0.
    # searchsorted side='right', u = 1 -> inds = len(cdf) -> above clamps to the last edge
    assert abs(s[-1] - bins[0, -1]) < 1e-6


def test_sample_z_and_perturb_bounds():
    near, far = np.full((3, 1), 2.), np.full((3, 1), 6.)
    z = orc.sample_z(near, far, 64, dtype=np.float64)
    np.testing.assert_allclose(z[0], np.linspace(2., 6., 64))
    zd = orc.sample_z(near, far, 64, lindisp=True, dtype=np.float64)
    np.testing.assert_allclose(1. / zd[0], np.linspace(1 / 2., 1 / 6., 64))
    u = np.random.default_rng(0).uniform(size=(3, 64))
    zp = orc.sample_z(near, far, 64, perturb_u=u, dtype=np.float64)
    mids = .5 * (z[:, 1:] + z[:, :-1])
    assert np.all(zp[:, 1:-1] >= mids[:, :-1]) and np.all(zp[:, 1:-1] <= mids[:, 1:])
    assert np.all(np.diff(zp, axis=-1) >= 0)


def test_mlp_shapes_and_param_count():
    shapes = orc.layer_shapes()
    assert shapes[0] == (63, 256) and shapes[5] == (319, 256) and shapes[8] == (319 - 63, 256)
    assert shapes[9] == (283, 128) and shapes[10] == (128, 3) and shapes[11] == (256, 1)
    n = sum(a * b + b for a, b in shapes)
    assert n == 595844  # "weights are ~5MB" for two networks (README)
    macs = sum(a * b for a, b in shapes)
    assert macs == 593408


def test_mlp_wiring_against_handwritten_forward():
    rng = np.random.default_rng(0)
    ws = trained_like_weights(rng)
    x = rng.normal(size=(5, 90))
    out = orc.mlp_forward(ws, x)
    pts, views = x[:, :63], x[:, 63:]
    h = pts
    for i in range(8):
        if i == 5:
            h = np.concatenate([pts, h], -1)
        h = np.maximum(h @ ws[2 * i] + ws[2 * i + 1], 0)
    alpha = h @ ws[22] + ws[23]
    bott = h @ ws[16] + ws[17]
    v = np.maximum(np.concatenate([bott, views], -1) @ ws[18] + ws[19], 0)
    rgb = v @ ws[20] + ws[21]
    np.testing.assert_allclose(out, np.concatenate([rgb, alpha], -1), atol=1e-12)


def test_keras_adam_three_steps_by_hand():
    theta = np.array([1., -2., 0.5, 3.])
    m = np.zeros(4)
    v = np.zeros(4)
    g = np.array([0.1, -0.2, 0.3, 0.])
    th = theta.copy()
    for it in range(3):
        lr = orc.lr_schedule(5e-4, 250, it)
        th, m, v = orc.keras_adam_step(th, g, m, v, it, lr)
    # constant gradient: m_hat = g, v_hat = g^2 -> step = lr_t-free form lr * g/(|g| + eps*...)
    exp = theta.copy()
    mm = np.zeros(4)
    vv = np.zeros(4)
    for it in range(3):
        t = it + 1
        lr = 5e-4 * 0.1 ** (it / 250000.)
        mm = 0.9 * mm + 0.1 * g
        vv = 0.999 * vv + 0.001 * g * g
        exp = exp - lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * mm / (np.sqrt(vv) + 1e-7)
    np.testing.assert_allclose(th, exp, rtol=1e-14)
    assert th[3] == theta[3]  # zero gradient, zero step
    np.testing.assert_allclose(theta[0] - th[0], 3 * 5e-4, rtol=1e-3)  # |step| ~ lr per iteration


def test_numpy_oracle_agrees_with_torch_restatement():
    rng = np.random.default_rng(1)
    wc, wf = trained_like_weights(rng), trained_like_weights(rng)
    rb = synthetic_rays(24, seed=2)
    ref = orc.render_rays(rb, wc, 64, retraw=True, N_importance=128, weights_fine=wf, white_bkgd=True,
                          dtype=np.float64)
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    out = trs.render_rays(t(rb), [t(w) for w in wc], [t(w) for w in wf], white_bkgd=True)
    for k in ['rgb_map', 'disp_map', 'acc_map', 'depth_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'raw']:
        np.testing.assert_allclose(out[k].numpy(), ref[k], atol=1e-9, err_msg=k)
    # with stratified jitter and random inverse-CDF uniforms shared by both sides
    pu = rng.uniform(size=(24, 64))
    uu = rng.uniform(size=(24, 128))
    ref = orc.render_rays(rb, wc, 64, N_importance=128, weights_fine=wf, perturb=1., perturb_u=pu,
                          pdf_u=uu, dtype=np.float64)
    out = trs.render_rays(t(rb), [t(w) for w in wc], [t(w) for w in wf], perturb_u=t(pu), pdf_u=t(uu))
    np.testing.assert_allclose(out['rgb_map'].numpy(), ref['rgb_map'], atol=1e-9)


def test_fp32_oracle_noise_floor():
    """fp32 evaluation of the same graph stays within the 1e-5 budget of the compositing path."""
    rng = np.random.default_rng(3)
    raw = rng.normal(size=(50, 192, 4)) * 2
    z = np.sort(rng.uniform(2, 6, size=(50, 192)), -1)
    d = rng.normal(size=(50, 3))
    a = orc.raw2outputs(raw, z, d, dtype=np.float64)
    b = orc.raw2outputs(raw, z, d, dtype=np.float32)
    for x, y in zip(a[:1] + a[2:], b[:1] + b[2:]):
        assert np.max(np.abs(x - y)) < 1e-5 * max(1., np.max(np.abs(x)))


def test_render_shapes_and_keys():
    rng = np.random.default_rng(4)
    wc = trained_like_weights(rng)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    out = orc.render(3, 5, 4., chunk=7, c2w=c2w, ndc=False, near=2., far=6., use_viewdirs=True,
                     weights_coarse=wc, weights_fine=wc, N_samples=8, N_importance=4, dtype=np.float64)
    assert out[0].shape == (3, 5, 3) and out[1].shape == (3, 5) and out[2].shape == (3, 5)
    assert out[3]['rgb0'].shape == (3, 5, 3) and out[3]['z_std'].shape == (3, 5)
