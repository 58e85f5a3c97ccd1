"""GPU parity of the training-side kernels against torch-CPU autograd (float64) of the restatement."""
import numpy as np
import pytest
import torch

from oracle import torch_restatement as trs
from tests.util import synthetic_rays, trained_like_weights

pytestmark = pytest.mark.gpu


def _t(x, gpu):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=gpu)


@pytest.mark.parametrize("S,white", [(64, False), (64, True), (128, True), (192, True), (100, False), (1, False), (384, True)])
def test_composite_bwd(gpu, S, white):
    from nerf_amd import ops
    rng = np.random.default_rng(S)
    n = 97
    raw = (rng.normal(size=(n, S, 4)) * 1.5).astype(np.float32)
    raw[..., 3] = rng.normal(size=(n, S)) * 2.0
    raw[::5, S // 2, 3] = 50.0  # an opaque sample in the middle of some rays
    z = np.sort(rng.uniform(2., 6., size=(n, S)), -1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    g_rgb = rng.normal(size=(n, 3)).astype(np.float32)
    g_acc = rng.normal(size=(n,)).astype(np.float32)
    g_depth = rng.normal(size=(n,)).astype(np.float32)
    # keep the last sample away from sigma = 0 (its alpha is a step function there)
    raw[:, -1, 3] = np.where(np.abs(raw[:, -1, 3]) < 0.05, 0.5, raw[:, -1, 3])
    rt = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
    rgb, disp, acc, w, depth = trs.raw2outputs(rt, torch.tensor(z, dtype=torch.float64),
                                               torch.tensor(d, dtype=torch.float64), white)
    loss = (rgb * torch.tensor(g_rgb, dtype=torch.float64)).sum() + (acc * torch.tensor(g_acc, dtype=torch.float64)).sum() \
        + (depth * torch.tensor(g_depth, dtype=torch.float64)).sum()
    loss.backward()
    ref = rt.grad.numpy()
    got = ops.composite_bwd(_t(raw, gpu), _t(z, gpu), _t(d, gpu), 3, white, None, 0.,
                            _t(g_rgb, gpu), _t(g_acc, gpu), _t(g_depth, gpu)).cpu().numpy()
    scale = max(1., float(np.max(np.abs(ref))))
    np.testing.assert_allclose(got, ref, atol=2e-5 * scale, rtol=2e-4)
    # rgb gradient only (the training loss)
    got2 = ops.composite_bwd(_t(raw, gpu), _t(z, gpu), _t(d, gpu), 3, white, None, 0., _t(g_rgb, gpu)).cpu().numpy()
    rt2 = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
    rgb2 = trs.raw2outputs(rt2, torch.tensor(z, dtype=torch.float64), torch.tensor(d, dtype=torch.float64), white)[0]
    (rgb2 * torch.tensor(g_rgb, dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(got2, rt2.grad.numpy(), atol=2e-5 * scale, rtol=2e-4)


def test_forward_train_matches_inference_and_saves_activations(gpu):
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import init_nerf_model
    from oracle import nerf_oracle as orc
    rng = np.random.default_rng(3)
    ws = trained_like_weights(rng)
    m = init_nerf_model(8, 256, 63, 27, 4, [4], True, device=gpu)
    m.set_weights(ws)
    rb = synthetic_rays(9, seed=5)   # 9 rays x 64 = 576 points: 3 workgroups, last one partial
    rays = _t(rb, gpu)
    z = ops.sample_z(rays, 64)
    act = torch.zeros((ops.act_bytes(9 * 64),), dtype=torch.uint8, device=gpu)
    raw_t = ops.mlp_forward_train(m.packed(), rays, z, act)
    raw_i = ops.mlp_forward_rays(m.packed(), rays, z)
    # the saving forward is the inference kernel plus stores: same bits
    assert torch.equal(raw_t, raw_i)
    # decode the saved h7 of the first wave tile (points 0..31) and compare with the oracle
    lay = ops.train_layout()
    blk = act.view(-1, lay['act_blocks'], 1024)[0]
    h7 = blk[lay['h0'] + 16 * 7: lay['h0'] + 16 * 8].contiguous().view(torch.float16).view(16, 64, 8).float().cpu().numpy()
    zz = z.cpu().numpy().astype(np.float64)
    r = rb.astype(np.float64)
    pts = (r[:, None, 0:3] + r[:, None, 3:6] * zz[:, :, None]).reshape(-1, 3)[:32]
    e = orc.embed(pts, 10)
    hcur = e
    W = [w.astype(np.float64) for w in ws]
    for i in range(8):
        if i == 5:
            hcur = np.concatenate([e, hcur], -1)
        hcur = np.maximum(hcur @ W[2 * i] + W[2 * i + 1], 0)
    # block b = k-step (b >> 1) of point tile (b & 1); lane = point + 16 g; element j = feature
    # 32 s + 16 (j >> 2) + 4 g + (j & 3)
    for b in range(16):
        s_, pt = b >> 1, b & 1
        for lane in range(64):
            c, g = 16 * pt + (lane & 15), lane >> 4
            for j in range(8):
                f = 32 * s_ + 16 * (j >> 2) + 4 * g + (j & 3)
                assert abs(h7[b, lane, j] - hcur[c, f]) < 2e-3 * max(1., abs(hcur[c, f])), (b, lane, j)
    # the mask block of h7: with i = 2 t + (r >> 1), bit (i & 15) + 16 (r & 1) of dword [pt][i >> 4] of
    # lane (point, g) = unit 16 t + 4 g + r
    mk = blk[lay['mask'] + 7].contiguous().view(torch.int32).view(64, 4).cpu().numpy().astype(np.uint32)
    for lane in range(64):
        g = lane >> 4
        for pt in range(2):
            c = 16 * pt + (lane & 15)
            for t in range(16):
                for r_ in range(4):
                    i_ = 2 * t + (r_ >> 1)
                    bit = (int(mk[lane, 2 * pt + (i_ >> 4)]) >> ((i_ & 15) + 16 * (r_ & 1))) & 1
                    f = 16 * t + 4 * g + r_
                    val = h7[2 * (f >> 5) + pt, (lane & 15) + 16 * ((f >> 2) & 3), 4 * ((f & 31) >> 4) + (f & 3)]
                    assert bit == int(val != 0), (lane, pt, t, r_)


def _decode_blocks(buf, blocks_per_tile, blk0, nblocks, n_points):
    """Saved fp16 activation blocks [wave tile][block][lane][8] -> [points, 16*nblocks] in feature
    order (block b = k-step b >> 1 of point tile b & 1, lane = point + 16 g, element j = feature
    32 s + 16 (j >> 2) + 4 g + (j & 3))."""
    b = buf.view(-1, blocks_per_tile, 1024)[:, blk0:blk0 + nblocks].contiguous().view(torch.float16)
    b = b.view(-1, nblocks, 64, 8).double().cpu().numpy()
    blk = np.arange(nblocks)[:, None, None]
    lane = np.arange(64)[None, :, None]
    j = np.arange(8)[None, None, :]
    f = 32 * (blk >> 1) + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3)
    c = np.broadcast_to(16 * (blk & 1) + (lane & 15), f.shape)
    out = np.zeros((b.shape[0] * 32, nblocks * 16))
    for t in range(b.shape[0]):
        out[t * 32 + c, f] = b[t]
    return out[:n_points]


def _mlp_grads_reference(ws, rb, z, d_raw, act):
    """fp64 torch autograd of sum(raw * d_raw) w.r.t. the 24 parameter arrays.

    A ReLU unit whose pre-activation is within the fp16 forward's rounding of zero may be on in
    one arithmetic and off in the other; its gradient is then all or nothing.  The backward pass
    must follow the forward pass it belongs to, so the reference takes the on/off pattern from
    the activations the GPU forward saved and differentiates pre * mask instead of relu(pre)."""
    from nerf_amd import ops
    n_points = z.size
    lay = ops.train_layout()
    masks = [torch.tensor(_decode_blocks(act, lay['act_blocks'], lay['h0'] + 16 * l, 16, n_points) != 0) for l in range(8)]
    mask_v = torch.tensor(_decode_blocks(act, lay['act_blocks'], lay['v'], 8, n_points) != 0)
    W = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws]
    r = torch.tensor(rb, dtype=torch.float64)
    zz = torch.tensor(z, dtype=torch.float64)
    pts = (r[:, None, 0:3] + r[:, None, 3:6] * zz[:, :, None]).reshape(-1, 3)
    dirs = r[:, None, 8:11].expand(r.shape[0], zz.shape[1], 3).reshape(-1, 3)
    pe, ve = trs.embed(pts, 10), trs.embed(dirs, 4)
    h = pe
    flips = 0
    for i in range(8):
        if i == 5:
            h = torch.cat([pe, h], -1)
        pre = h @ W[2 * i] + W[2 * i + 1]
        flips += int(((pre > 0) != masks[i]).sum())
        h = pre * masks[i]
    alpha = h @ W[22] + W[23]
    bott = h @ W[16] + W[17]
    pre = torch.cat([bott, ve], -1) @ W[18] + W[19]
    flips += int(((pre > 0) != mask_v).sum())
    v = pre * mask_v
    raw = torch.cat([v @ W[20] + W[21], alpha], -1)
    (raw * torch.tensor(d_raw, dtype=torch.float64).reshape(-1, 4)).sum().backward()
    return [w.grad.numpy() for w in W], flips / (n_points * (8 * 256 + 128))


@pytest.mark.parametrize("n,S,scale_hidden", [(8, 64, 1.), (5, 192, 1.), (3, 37, 1.), (8, 64, 2.), (8, 64, 3.)])
def test_mlp_backward(gpu, n, S, scale_hidden):
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import init_nerf_model
    rng = np.random.default_rng(100 + n)
    ws = trained_like_weights(rng, scale_hidden=scale_hidden)
    m = init_nerf_model(8, 256, 63, 27, 4, [4], True, device=gpu)
    m.set_weights(ws)
    rb = synthetic_rays(n, seed=n)
    rays = _t(rb, gpu)
    z = ops.sample_z(rays, S)
    d_raw = (rng.normal(size=(n, S, 4)) * 1e-4).astype(np.float32)
    act = torch.zeros((ops.act_bytes(n * S),), dtype=torch.uint8, device=gpu)
    ops.mlp_forward_train(m.packed(), rays, z, act)
    pb = torch.empty((ops.packed_bwd_bytes(),), dtype=torch.uint8, device=gpu)
    ops.mlp_pack_bwd([p.detach() for p in m.params], pb)
    gbuf = torch.empty((ops.grad_bytes(n * S),), dtype=torch.uint8, device=gpu)
    part = torch.empty((ops.partial_bytes(),), dtype=torch.uint8, device=gpu)
    grads = [torch.full_like(p, 7.0) for p in m.params]
    ops.mlp_backward(pb, act, _t(d_raw, gpu), 4096.0, gbuf, part, grads)
    ref, flip_rate = _mlp_grads_reference(ws, rb, z.cpu().numpy(), d_raw, act)
    assert flip_rate < 2e-3  # units whose on/off state differs between fp16 and fp64 forward
    names = [f"{'W' if i % 2 == 0 else 'b'}{i // 2}" for i in range(24)]
    for name, g, r in zip(names, grads, ref):
        g = g.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(g)), name
        # fp16 operands, fp32 accumulation: compare in the norm, plus a loose elementwise bound
        rel = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        assert rel < 5e-3, (name, rel)
        assert np.max(np.abs(g - r)) < 2e-2 * np.max(np.abs(r)) + 1e-12, name
    # the automatic scale gives the same answer to rounding; kernels scaled 3x make the gradient
    # grow by more than the 128x of headroom it leaves on the way down (x3 per layer), and that
    # is reported, with NaN gradients, not returned as numbers
    auto = [torch.zeros_like(p) for p in m.params]
    ops.mlp_backward(pb, act, _t(d_raw, gpu), None, gbuf, part, auto)
    flags = ops.mlp_backward_status(part)[0]
    assert flags == (ops.GRAD_SATURATED if scale_hidden >= 3. else 0)
    for name, g, r in zip(names, auto, ref):
        if flags:
            assert bool(torch.isnan(g).all()), name
            continue
        rel = np.linalg.norm(g.cpu().numpy() - r) / max(np.linalg.norm(r), 1e-30)
        assert rel < 5e-3, (name, rel)
    # accumulate mode adds on top
    grads2 = [g.clone() for g in grads]
    ops.mlp_backward(pb, act, _t(d_raw, gpu), 4096.0, gbuf, part, grads2, accumulate=True)
    for a, b in zip(grads, grads2):
        assert torch.allclose(b, 2 * a, rtol=1e-3, atol=1e-12)


def test_mlp_backward_many_tiles_is_the_sum_of_its_slices(gpu):
    """A batch of more 256-point tiles than workgroups (the saving forward's workgroups each walk
    several tiles): outputs equal inference bit for bit, and the weight gradients of the whole
    batch equal the sum of the gradients of its two halves, computed by separate calls."""
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import init_nerf_model
    rng = np.random.default_rng(21)
    m = init_nerf_model(8, 256, 63, 27, 4, [4], True, device=gpu)
    m.set_weights(trained_like_weights(rng))
    n, S = 1100, 192                       # 211,200 points = 825 tiles
    rays = _t(synthetic_rays(n, seed=8), gpu)
    z = ops.sample_z(rays, S)
    g = torch.Generator(device=gpu).manual_seed(1)
    d_raw = torch.randn((n, S, 4), generator=g, device=gpu) * 1e-3
    pb = torch.empty((ops.packed_bwd_bytes(),), dtype=torch.uint8, device=gpu)
    ops.mlp_pack_bwd([p.detach() for p in m.params], pb)
    part = torch.empty((ops.partial_bytes(),), dtype=torch.uint8, device=gpu)

    def fwd_bwd(lo, hi):
        r, zz, dr = rays[lo:hi].contiguous(), z[lo:hi].contiguous(), d_raw[lo:hi].contiguous()
        pts = (hi - lo) * S
        act = torch.zeros((ops.act_bytes(pts),), dtype=torch.uint8, device=gpu)
        raw = ops.mlp_forward_train(m.packed(), r, zz, act)
        gbuf = torch.empty((ops.grad_bytes(pts),), dtype=torch.uint8, device=gpu)
        grads = [torch.zeros_like(p) for p in m.params]
        ops.mlp_backward(pb, act, dr, 4096.0, gbuf, part, grads)
        return raw, grads

    raw, full = fwd_bwd(0, n)
    assert torch.equal(raw, ops.mlp_forward_rays(m.packed(), rays, z))
    h = 601                                # the halves do not end on a tile boundary
    raw_a, ga = fwd_bwd(0, h)
    raw_b, gb = fwd_bwd(h, n)
    assert torch.equal(raw_a, raw[:h]) and torch.equal(raw_b, raw[h:])
    for i, (gf, x, y) in enumerate(zip(full, ga, gb)):
        gs = x + y
        err = float((gf - gs).norm() / gs.norm().clamp_min(1e-20))
        assert err < 1e-4, f"tensor {i}: {err}"


def test_mse_and_keras_adam(gpu):
    from nerf_amd import training
    from nerf_amd.run_nerf_helpers import img2mse, mse2psnr
    from oracle import nerf_oracle as orc
    rng = np.random.default_rng(0)
    x = rng.uniform(size=(1000, 3)).astype(np.float32)
    y = rng.uniform(size=(1000, 3)).astype(np.float32)
    xt = _t(x, gpu).requires_grad_(True)
    loss = img2mse(xt, _t(y, gpu))
    np.testing.assert_allclose(float(loss.detach()), orc.img2mse(x.astype(np.float64), y.astype(np.float64)), rtol=1e-5)
    (3.0 * loss).backward()
    np.testing.assert_allclose(xt.grad.cpu().numpy(), 3.0 * 2 * (x - y) / x.size, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(float(mse2psnr(loss.detach())), orc.mse2psnr(float(loss.detach())), rtol=1e-5)
    # Keras Adam with the reference's decay schedule, 5 steps, against the oracle in float64
    theta0 = rng.normal(size=(5000,)).astype(np.float32)
    flat = _t(theta0, gpu)
    views = [flat[:3000].view(30, 100), flat[3000:]]
    opt = training.Adam(training.ExponentialDecay(5e-4, 250 * 1000, 0.1))
    th, m, v = theta0.astype(np.float64), np.zeros(5000), np.zeros(5000)
    for it in range(5):
        g = rng.normal(size=(5000,)).astype(np.float32) * 10 ** rng.uniform(-6, 0)
        gt = _t(g, gpu)
        opt.apply_gradients(zip([gt[:3000].view(30, 100), gt[3000:]], views))
        th, m, v = orc.keras_adam_step(th, g.astype(np.float64), m, v, it, orc.lr_schedule(5e-4, 250, it))
    assert opt.iterations == 5
    np.testing.assert_allclose(flat.cpu().numpy(), th, rtol=2e-6, atol=2e-7)
    # the optimiser state in Keras' get_weights() order, so that an optimiser kept in the models
    # dictionary goes through the same checkpoint writer as a network (run_nerf.py:838-842)
    ws = opt.get_weights()
    assert int(ws[0]) == 5 and len(ws) == 3
    np.testing.assert_allclose(ws[1], m, rtol=2e-5, atol=1e-6 * np.abs(m).max())   # fp32 sums of mixed signs
    np.testing.assert_allclose(ws[2], v, rtol=2e-5, atol=1e-6 * np.abs(v).max())
    import tempfile
    from nerf_amd import checkpoint
    with tempfile.TemporaryDirectory() as d:
        back = np.load(checkpoint.save_weights(opt, 'optimizer', 5, d, 'exp'), allow_pickle=True)
        assert len(back) == 3 and int(back[0]) == 5


@pytest.mark.parametrize("n,coarse", [(1, True), (37, False), (4096, True), (100003, True)])
def test_photometric_loss_in_one_launch(gpu, n, coarse):
    """training.photometric_loss (what train_step uses) against the oracle's img2mse / mse2psnr and
    against img2mse + img2mse through the tape: loss, both terms, both PSNRs, both gradients; plain,
    with a weighted loss, and with grad_weight."""
    from nerf_amd import training
    from nerf_amd.run_nerf_helpers import img2mse
    from oracle import nerf_oracle as orc
    rng = np.random.default_rng(n)
    x, x0, y = (rng.uniform(size=(n, 3)).astype(np.float32) for _ in range(3))
    f64 = lambda a: a.astype(np.float64)
    want, want0 = orc.img2mse(f64(x), f64(y)), orc.img2mse(f64(x0), f64(y))
    for weight, outer in [(1.0, None), (1.0, 3.0), (5.0, None)]:
        xt, x0t = _t(x, gpu).requires_grad_(True), _t(x0, gpu).requires_grad_(True)
        with training.GradientTape() as tape:
            loss, img_loss, img_loss0, psnr, psnr0 = training.photometric_loss(xt, x0t if coarse else None, _t(y, gpu),
                                                                             grad_weight=weight)
            top = loss if outer is None else loss * outer
        gx, gx0 = tape.gradient(top, [xt, x0t])
        np.testing.assert_allclose(float(img_loss), want, rtol=1e-5)
        np.testing.assert_allclose(float(psnr), orc.mse2psnr(want), rtol=1e-5, atol=1e-5)
        scale = weight * (outer or 1.0)
        np.testing.assert_allclose(gx.cpu().numpy(), scale * 2 * (f64(x) - y) / x.size, rtol=2e-6, atol=1e-12)
        if coarse:
            np.testing.assert_allclose(float(loss.detach()), want + want0, rtol=1e-5)
            np.testing.assert_allclose(float(img_loss0), want0, rtol=1e-5)
            np.testing.assert_allclose(float(psnr0), orc.mse2psnr(want0), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(gx0.cpu().numpy(), scale * 2 * (f64(x0) - y) / x.size, rtol=2e-6, atol=1e-12)
        else:
            np.testing.assert_allclose(float(loss.detach()), want, rtol=1e-5)
            assert img_loss0 is None and psnr0 is None and not bool(gx0.any())
    # the two-node form a caller's own loop would write gives the same numbers
    xt, x0t = _t(x, gpu).requires_grad_(True), _t(x0, gpu).requires_grad_(True)
    with training.GradientTape() as tape:
        loss2 = img2mse(xt, _t(y, gpu)) + img2mse(x0t, _t(y, gpu))
    hx, hx0 = tape.gradient(loss2, [xt, x0t])
    np.testing.assert_allclose(float(loss2.detach()), want + want0, rtol=1e-5)
    np.testing.assert_allclose(hx.cpu().numpy(), 2 * (f64(x) - y) / x.size, rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(hx0.cpu().numpy(), 2 * (f64(x0) - y) / x.size, rtol=2e-6, atol=1e-12)
    # same inputs, same bits (one workgroup, no atomics)
    a = training.photometric_loss(_t(x, gpu), _t(x0, gpu), _t(y, gpu))[0]
    b = training.photometric_loss(_t(x, gpu), _t(x0, gpu), _t(y, gpu))[0]
    assert float(a) == float(b)


@pytest.mark.parametrize("n", [96, 4096])
def test_training_step_gradients_end_to_end(gpu, n):
    """tape.gradient over render() (64+128 samples, perturb and random resampling uniforms shared
    with the reference) against torch autograd of the restatement in float64, at a small batch and
    at the reference's default N_rand.  The fp16 gradient scale is left to the library."""
    from nerf_amd import ops, training
    from nerf_amd.run_nerf import create_nerf, render
    from nerf_amd.run_nerf_helpers import img2mse
    from tests.util import default_args
    args = default_args(white_bkgd=True, perturb=1.)
    kw_train, kw_test, _, grad_vars, models = create_nerf(args)
    rng = np.random.default_rng(21)
    wc, wf = trained_like_weights(rng), trained_like_weights(rng)
    models['model'].set_weights(wc)
    models['model_fine'].set_weights(wf)
    assert models['model'].grad_scale is None and models['model_fine'].grad_scale is None
    rb = synthetic_rays(n, seed=8)
    target = rng.uniform(size=(n, 3)).astype(np.float32)
    pu = rng.uniform(size=(n, 64)).astype(np.float32)
    uu = rng.uniform(size=(n, 128)).astype(np.float32)
    with training.GradientTape() as tape:
        rgb, disp, acc, extras = render(1, 1, 1., chunk=1024 * 32, rays=(_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)),
                                        near=2., far=6., retraw=True, perturb_u=_t(pu, gpu), pdf_u=_t(uu, gpu),
                                        **kw_train)
        loss = img2mse(rgb, _t(target, gpu)) + img2mse(extras['rgb0'], _t(target, gpu))
    grads = tape.gradient(loss, grad_vars)
    assert len(grads) == 48
    for m in models.values():
        flags, gs, mx = m.grad_status()
        assert flags == 0 and mx > 0
        assert 256. <= gs * mx < 512. and np.log2(gs) == round(np.log2(gs))   # a power of two, on target
    # float64 autograd of the restatement, on the GPU, a slice of rays at a time
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=gpu)
    Wc = [t64(w).requires_grad_(True) for w in wc]
    Wf = [t64(w).requires_grad_(True) for w in wf]
    ref_loss = 0.
    for i in range(0, n, 512):
        sl = slice(i, i + 512)
        out = trs.render_rays(t64(rb[sl]), Wc, Wf, white_bkgd=True, perturb_u=t64(pu[sl]), pdf_u=t64(uu[sl]))
        tt = t64(target[sl])
        part = (((out['rgb_map'] - tt) ** 2).sum() + ((out['rgb0'] - tt) ** 2).sum()) / (n * 3)
        part.backward()
        ref_loss += float(part)
    np.testing.assert_allclose(float(loss), ref_loss, rtol=2e-3)
    ref = [w.grad.cpu().numpy() for w in Wc + Wf]
    got = [g.cpu().numpy().astype(np.float64) for g in grads]
    # whole-gradient agreement per network (ReLU units and resampling bins that flip between the
    # fp16 and the fp64 forward move individual entries, not the direction)
    for lo, hi, name in [(0, 24, 'coarse'), (24, 48, 'fine')]:
        a = np.concatenate([g.ravel() for g in got[lo:hi]])
        b = np.concatenate([g.ravel() for g in ref[lo:hi]])
        assert np.all(np.isfinite(a)), name
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
        assert cos > 0.999 and rel < 5e-2, (name, cos, rel)


def _backward_setup(gpu, n, S, seed, d_raw_scale=1e-4):
    from nerf_amd import ops
    from nerf_amd.run_nerf_helpers import init_nerf_model
    rng = np.random.default_rng(seed)
    ws = trained_like_weights(rng)
    m = init_nerf_model(8, 256, 63, 27, 4, [4], True, device=gpu)
    m.set_weights(ws)
    rb = synthetic_rays(n, seed=seed)
    rays = _t(rb, gpu)
    z = ops.sample_z(rays, S)
    d_raw = _t((rng.normal(size=(n, S, 4)) * d_raw_scale).astype(np.float32), gpu)
    act = torch.zeros((ops.act_bytes(n * S),), dtype=torch.uint8, device=gpu)
    ops.mlp_forward_train(m.packed(), rays, z, act)
    gbuf = torch.empty((ops.grad_bytes(n * S),), dtype=torch.uint8, device=gpu)
    part = torch.empty((ops.partial_bytes(),), dtype=torch.uint8, device=gpu)
    return m, ws, rb, z, d_raw, act, gbuf, part


@pytest.mark.parametrize("n,S", [(40, 64), (700, 192)])
def test_mlp_backward_same_inputs_same_bits(gpu, n, S):
    """The weight gradients are summed in a fixed order: two calls on the same inputs agree bit
    for bit (few wave tiles: fewer workgroups than CUs; many: one per CU, several tiles each),
    whatever the scratch buffers held before."""
    from nerf_amd import ops
    m, ws, rb, z, d_raw, act, gbuf, part = _backward_setup(gpu, n, S, 5)
    runs = []
    for fill in (0, 255):
        gbuf.fill_(fill)
        part.fill_(fill)
        grads = [torch.full_like(p, float('nan')) for p in m.params]
        ops.mlp_backward(m.packed_bwd(), act, d_raw, None, gbuf, part, grads)
        assert ops.mlp_backward_status(part)[0] == 0
        runs.append(grads)
    for a, b in zip(*runs):
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_gradient_scale_is_automatic_and_misuse_is_reported(gpu):
    """gscale <= 0 forms the factor on the device; an explicit factor is used as given and a
    grossly wrong one is reported (status bits, NaN gradients) instead of being silently used."""
    from nerf_amd import ops
    n, S = 8, 64
    m, ws, rb, z, d_raw, act, gbuf, part = _backward_setup(gpu, n, S, 108)
    pb = m.packed_bwd()
    ref, flip_rate = _mlp_grads_reference(ws, rb, z.cpu().numpy(), d_raw.cpu().numpy(), act)

    def run(gscale):
        grads = [torch.zeros_like(p) for p in m.params]
        ops.mlp_backward(pb, act, d_raw, gscale, gbuf, part, grads)
        return grads, ops.mlp_backward_status(part)

    def worst_rel(grads):
        return max(np.linalg.norm(g.cpu().numpy() - r) / max(np.linalg.norm(r), 1e-30) for g, r in zip(grads, ref))

    grads, (flags, gs, mx) = run(None)
    assert flags == 0 and abs(mx - float(d_raw.abs().max())) < 1e-12 and 256. <= gs * mx < 512.
    assert worst_rel(grads) < 5e-3
    # an explicit factor equal to the automatic one gives the same bits
    again, (flags2, gs2, _) = run(gs)
    assert flags2 == 0 and gs2 == gs and all(torch.equal(a, b) for a, b in zip(grads, again))
    # 2^10 too high: the first conversions already saturate -> reported, gradients are NaN
    high, (fh, _, _) = run(gs * 1024.)
    assert fh & ops.GRAD_SATURATED
    assert all(bool(torch.isnan(g).all()) for g in high)
    # 2^10 too low is still inside the fp16 range (scaled d_raw around 1): sound, and not flagged
    low, (fl, _, _) = run(gs / 1024.)
    assert fl == 0 and worst_rel(low) < 5e-3
    # 2^30 too low: nothing but denormals and zeros reach the weight gradients -> reported
    lost, (fu, _, _) = run(gs / 2. ** 30)
    assert fu & ops.GRAD_UNDERFLOW
    assert all(bool(torch.isnan(g).all()) for g in lost)
    # a NaN in d_raw is reported, not averaged away
    bad = d_raw.clone()
    bad[3, 7, 1] = float('nan')
    grads_nan = [torch.zeros_like(p) for p in m.params]
    ops.mlp_backward(pb, act, bad, None, gbuf, part, grads_nan)
    assert ops.mlp_backward_status(part)[0] & ops.GRAD_SATURATED
    # all-zero d_raw: zero gradients, nothing to report
    zero = [torch.ones_like(p) for p in m.params]
    ops.mlp_backward(pb, act, torch.zeros_like(d_raw), None, gbuf, part, zero)
    assert ops.mlp_backward_status(part)[0] == 0 and all(float(g.abs().max()) == 0. for g in zero)


def test_training_reduces_loss(gpu):
    """A few Adam steps on a fixed batch lower the loss, re-packing the weights every step."""
    from nerf_amd import training
    from nerf_amd.run_nerf import create_nerf, render
    from tests.util import default_args
    torch.manual_seed(0)
    args = default_args(white_bkgd=False, perturb=1.)
    kw_train, kw_test, _, grad_vars, models = create_nerf(args)
    n = 512
    rb = synthetic_rays(n, seed=3)
    rays = (_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu))
    target = _t(np.random.default_rng(1).uniform(0.2, 0.8, size=(n, 3)), gpu)
    opt = training.Adam(training.ExponentialDecay(5e-4, 250 * 1000, 0.1))
    fn = lambda r: render(1, 1, 1., chunk=1024 * 32, rays=r, near=2., far=6., retraw=True, **kw_train)
    losses = []
    before = [p.detach().clone() for p in grad_vars]
    for it in range(12):
        loss, psnr, psnr0 = training.train_step(fn, rays, target, grad_vars, opt)
        losses.append(float(loss))
        assert np.isfinite(losses[-1])
    assert opt.iterations == 12
    assert losses[-1] < 0.8 * losses[0], losses
    assert all(not torch.equal(a, b.detach()) for a, b in zip(before, grad_vars))
    # the inference path sees the updated weights (the packed streams were refreshed)
    with torch.no_grad():
        rgb, _, _, _ = render(1, 1, 1., chunk=1024 * 32, rays=rays, near=2., far=6., **kw_test)
    assert float(torch.mean((rgb - target) ** 2)) < losses[0]


def _dp_worker(rank, world_size, port, tmp):
    """One of two data-parallel ranks sharing the one GPU of the box (gloo moves the device
    tensors; on a multi-GPU node the same code runs over RCCL)."""
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nerf_amd import parallel, training
    from nerf_amd.run_nerf import create_nerf, render
    from tests.util import default_args
    torch.cuda.set_device(0)
    gpu = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    try:
        torch.manual_seed(1234 + rank)          # different initial weights per rank: create_nerf must broadcast
        args = default_args(white_bkgd=True, perturb=1.)
        kw_train, kw_test, _, grad_vars, models = create_nerf(args)
        nets = [models['model'], models['model_fine']]
        start = [m.flat.detach().clone() for m in nets]
        both = [torch.empty_like(start[0]) for _ in range(world_size)]
        dist.all_gather(both, start[0])
        assert torch.equal(both[0], both[1]), "replicas differ after create_nerf"
        n = 97                                  # shards of 49 and 48 rays
        rb = synthetic_rays(n, seed=3)
        rng = np.random.default_rng(5)
        target = rng.uniform(size=(n, 3)).astype(np.float32)
        pu = rng.uniform(size=(n, 64)).astype(np.float32)
        uu = rng.uniform(size=(n, 128)).astype(np.float32)
        lo, hi = parallel.shard_bounds(n, rank, world_size)

        def fn_for(sl):
            return lambda r: render(1, 1, 1., chunk=1024, rays=r, near=2., far=6., retraw=True,
                                    perturb_u=_t(pu[sl], gpu), pdf_u=_t(uu[sl], gpu), **kw_train)
        # reference: the whole batch on this rank alone, no exchange
        from nerf_amd.run_nerf_helpers import img2mse
        # (a tape on its own exchanges nothing: only train_step starts collectives)
        with training.GradientTape() as tape:
            rgb, _, _, ex = fn_for(slice(0, n))((_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)))
            full_loss = img2mse(rgb, _t(target, gpu)) + img2mse(ex['rgb0'], _t(target, gpu))
        full = [g.clone() for g in tape.gradient(full_loss, grad_vars)]
        # the data-parallel step on this rank's shard
        seen = {}

        class Probe(training.Adam):
            def apply_gradients(self, gv):
                gv = list(gv)
                seen['grads'] = [g.clone() for g, _ in gv]
                super().apply_gradients(gv)
        opt = Probe(training.ExponentialDecay(5e-4, 250 * 1000, 0.1))
        sl = slice(lo, hi)
        loss, psnr, psnr0 = training.train_step(fn_for(sl), (_t(rb[sl, :3], gpu), _t(rb[sl, 3:6], gpu)),
                                                _t(target[sl], gpu), grad_vars, opt)
        np.testing.assert_allclose(float(loss), float(full_loss), rtol=1e-4)      # the global mean loss
        for lo_i, hi_i, name in [(0, 24, 'coarse'), (24, 48, 'fine')]:
            a = torch.cat([g.reshape(-1) for g in seen['grads'][lo_i:hi_i]]).double()
            b = torch.cat([g.reshape(-1) for g in full[lo_i:hi_i]]).double()
            rel = float((a - b).norm() / b.norm())
            assert rel < 2e-3, (name, rel)      # same sum, split differently over fp16 chains
        # both replicas took the same step
        for m, s0 in zip(nets, start):
            assert not torch.equal(m.flat, s0)
            dist.all_gather(both, m.flat.detach())
            assert torch.equal(both[0], both[1]), "replicas diverged"
        # a network queried twice under one tape (network_fine=None with N_importance > 0: the
        # coarse network also does the fine pass, run_nerf.py:216-227): its two gradient
        # contributions are summed by autograd and exchanged ONCE
        kw_one = dict(kw_train, network_fine=None)

        def fn_one(sl):
            return lambda r: render(1, 1, 1., chunk=1024, rays=r, near=2., far=6., retraw=True,
                                    perturb_u=_t(pu[sl], gpu), pdf_u=_t(uu[sl], gpu), **kw_one)
        cvars = grad_vars[:24]
        with training.GradientTape() as tape:
            rgb, _, _, ex = fn_one(slice(0, n))((_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)))
            full_loss = img2mse(rgb, _t(target, gpu)) + img2mse(ex['rgb0'], _t(target, gpu))
        full = [g.clone() for g in tape.gradient(full_loss, cvars)]
        seen.clear()
        loss, _, _ = training.train_step(fn_one(sl), (_t(rb[sl, :3], gpu), _t(rb[sl, 3:6], gpu)),
                                         _t(target[sl], gpu), cvars, Probe(training.ExponentialDecay(5e-4, 250 * 1000, 0.1)))
        np.testing.assert_allclose(float(loss), float(full_loss), rtol=1e-4)
        a = torch.cat([g.reshape(-1) for g in seen['grads']]).double()
        b = torch.cat([g.reshape(-1) for g in full]).double()
        assert float((a - b).norm() / b.norm()) < 2e-3
        dist.all_gather(both, nets[0].flat.detach())
        assert torch.equal(both[0], both[1]), "replicas diverged (network queried twice)"
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_saturated_step_is_skipped_and_the_scale_backs_off(gpu):
    """Kernels scaled 3x amplify the gradient by more than the default automatic scale leaves
    room for (test_mlp_backward): the saturated passes yield NaN gradients, which the Adam kernel
    skips (weights and moments untouched); the target of the automatic scale is lowered from the
    status of earlier passes and the following steps use sound gradients.  No step ever writes a
    non-finite weight.  An explicit target (gscale < 0) does the same in one call."""
    from nerf_amd import ops, training
    from nerf_amd.run_nerf_helpers import init_nerf_model
    rng = np.random.default_rng(108)
    ws = trained_like_weights(rng, scale_hidden=3.)
    m = init_nerf_model(8, 256, 63, 27, 4, [4], True, device=gpu)
    m.set_weights(ws)
    n, S = 8, 64
    rays = _t(synthetic_rays(n, seed=8), gpu)
    z = ops.sample_z(rays, S)
    d_raw = _t((rng.normal(size=(n, S, 4)) * 1e-4).astype(np.float32), gpu)
    opt = training.Adam(1e-5)
    start = m.flat.detach().clone()
    flags, targets = [], []
    for it in range(10):
        with training.GradientTape() as tape:
            raw = training.MLPQuery.apply(rays, z, m, *m.params)
            loss = (raw * d_raw).sum()
        grads = tape.gradient(loss, m.params)
        before = m.flat.detach().clone()
        opt.apply_gradients(zip(grads, m.params))
        f = m.grad_status()[0]
        flags.append(f)
        targets.append(m.auto_target)
        assert bool(torch.isfinite(m.flat).all()), it
        if f & ops.GRAD_SATURATED:   # NaN gradients: the step was skipped as a whole
            assert all(bool(torch.isnan(g).all()) for g in grads) and torch.equal(m.flat, before), it
    assert flags[0] & ops.GRAD_SATURATED and flags[1] & ops.GRAD_SATURATED, flags   # the default target is too high here ...
    assert flags[-1] == 0 and targets[-1] < 256., (flags, targets)                  # ... and has come down by itself
    assert not torch.equal(m.flat, start)
    # the same in one call
    act = torch.zeros((ops.act_bytes(n * S),), dtype=torch.uint8, device=gpu)
    ops.mlp_forward_train(m.packed(), rays, z, act)
    gbuf = torch.empty((ops.grad_bytes(n * S),), dtype=torch.uint8, device=gpu)
    part = torch.empty((ops.partial_bytes(),), dtype=torch.uint8, device=gpu)
    g = [torch.zeros_like(p) for p in m.params]
    ops.mlp_backward(m.packed_bwd(), act, d_raw, -4.0, gbuf, part, g)
    fl, gs, mx = ops.mlp_backward_status(part)
    assert fl == 0 and 4. <= gs * mx < 8. and all(bool(torch.isfinite(x).all()) for x in g)
    with pytest.raises(Exception):
        ops.mlp_backward(m.packed_bwd(), act, d_raw, -3.0, gbuf, part, g)   # not a power of two


def test_data_parallel_train_step_two_ranks(gpu, tmp_path):
    """training.train_step under torch.distributed with world_size 2 and unequal shards: the
    summed gradient equals the single-rank gradient of the whole batch, the reported loss is the
    global mean, and the replicas stay bit-identical."""
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_graphed_train_step_equals_train_step(gpu):
    """GraphedTrainStep (forward + backward of both networks as one HIP graph, replayed) against
    train_step on the same batches and uniforms: the same weights bit for bit after four steps;
    a moved gradient-scale target makes it capture anew; with its own random streams it trains."""
    from nerf_amd import training
    from nerf_amd.run_nerf import create_nerf, render
    from tests.util import default_args
    n = 300
    rng = np.random.default_rng(23)
    batches = [(synthetic_rays(n, seed=50 + i), rng.uniform(size=(n, 3)).astype(np.float32)) for i in range(4)]
    pu, uu = _t(rng.uniform(size=(n, 64)), gpu), _t(rng.uniform(size=(n, 128)), gpu)

    def setup():
        torch.manual_seed(7)
        kw_train, _, _, grad_vars, models = create_nerf(default_args(white_bkgd=True, perturb=1.))
        opt = training.Adam(training.ExponentialDecay(5e-4, 250 * 1000, 0.1))
        fn = lambda r: render(1, 1, 1., chunk=1024 * 32, rays=r, near=2., far=6., retraw=True,
                              perturb_u=pu, pdf_u=uu, **kw_train)
        return kw_train, grad_vars, models, opt, fn
    kw_train, grad_vars, models, opt, fn = setup()
    eager = [training.train_step(fn, (_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)), _t(tg, gpu), grad_vars, opt) for rb, tg in batches]
    eager_losses = [float(l) for l, _, _ in eager]
    want = [models[k].flat.detach().clone() for k in ('model', 'model_fine')]
    kw_train, grad_vars, models, opt, fn = setup()
    step = training.GraphedTrainStep(fn, n, grad_vars, opt, [models['model'], models['model_fine']])
    got_losses = []
    for rb, tg in batches:
        loss, psnr, psnr0 = step((_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)), _t(tg, gpu))
        got_losses.append(float(loss))
        np.testing.assert_allclose(float(psnr), -10. * np.log10(float(loss) - 10 ** (-float(psnr0) / 10.)), rtol=1e-3)
    assert step.captures == 1 and opt.iterations == 4
    assert got_losses == eager_losses, (got_losses, eager_losses)
    for k, w in zip(('model', 'model_fine'), want):
        assert torch.equal(models[k].flat, w), k
    # a moved target (what review_grad_status does after a saturated pass) is a kernel argument: captured anew
    models['model_fine'].auto_target = 16.0
    rb, tg = batches[0]
    loss, _, _ = step((_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)), _t(tg, gpu))
    assert step.captures == 2 and np.isfinite(float(loss))
    flags, gs, mx = models['model_fine'].grad_status()
    assert flags == 0 and 16. <= gs * mx < 32.
    # free-running (torch.rand inside the graph: new uniforms at every replay) on one batch: the loss falls
    kw_train, grad_vars, models, opt, _ = setup()
    fn_rand = lambda r: render(1, 1, 1., chunk=1024 * 32, rays=r, near=2., far=6., retraw=True, **kw_train)
    step = training.GraphedTrainStep(fn_rand, n, grad_vars, opt, [models['model'], models['model_fine']])
    losses = [float(step((_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)), _t(tg, gpu))[0]) for _ in range(40)]
    assert losses[-1] < 0.9 * losses[0] and len(set(losses)) == 40, losses[::8]


def _rccl_alone_worker(rank, port, tmp):
    """A process group of ONE rank over RCCL: with parallel.EXCHANGE_WHEN_ALONE the data-parallel
    paths run through the real backend (broadcast of the replicas, the all-reduce started inside
    the backward pass on RCCL's stream and waited for before the optimiser, the all-gather of a
    sharded render) and, the exchanges being identities, must reproduce the plain paths bit for bit
    (a batch of 1024 rays: the ray-count weight is a power of two)."""
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nerf_amd import parallel, training
    from nerf_amd.run_nerf import create_nerf, render
    from tests.util import default_args
    torch.cuda.set_device(0)
    gpu = torch.device("cuda", 0)
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=gpu)
        probe = torch.ones(4, device=gpu)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
    except Exception as e:  # no RCCL on this machine: nothing to test
        open(os.path.join(tmp, "unavailable"), "w").write(repr(e))
        return
    try:
        n = 1024
        rb = synthetic_rays(n, seed=13)
        rng = np.random.default_rng(17)
        o, d = _t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)
        target = _t(rng.uniform(size=(n, 3)).astype(np.float32), gpu)
        pu = _t(rng.uniform(size=(n, 64)).astype(np.float32), gpu)
        uu = _t(rng.uniform(size=(n, 128)).astype(np.float32), gpu)

        def run(exchange):
            parallel.EXCHANGE_WHEN_ALONE = exchange
            assert parallel.exchanging() == exchange
            torch.manual_seed(99)
            kw_train, kw_test, _, grad_vars, models = create_nerf(default_args(white_bkgd=True, perturb=1.))
            opt = training.Adam(training.ExponentialDecay(5e-4, 250 * 1000, 0.1))
            fn = lambda r: render(1, 1, 1., chunk=1024 * 32, rays=r, near=2., far=6., retraw=True,
                                  perturb_u=pu, pdf_u=uu, **kw_train)
            out = [training.train_step(fn, (o, d), target, grad_vars, opt) for _ in range(3)]
            losses = [float(l) for l, _, _ in out]
            weights = [models[k].flat.detach().clone() for k in ('model', 'model_fine')]
            with torch.no_grad():
                rgb, disp, acc, extras = render(1, 1, 1., chunk=300, rays=(o, d), near=2., far=6., **kw_test)
                if exchange:
                    rgb_s, disp_s, acc_s, extras_s = parallel.render(1, 1, 1., chunk=300, rays=(o, d), near=2., far=6., **kw_test)
                    assert torch.equal(rgb_s, rgb) and torch.equal(disp_s, disp) and torch.equal(acc_s, acc)
                    assert sorted(extras_s) == sorted(extras) and all(torch.equal(extras_s[k], extras[k]) for k in extras)
            return losses, weights, rgb
        plain = run(False)
        over_rccl = run(True)
        assert plain[0] == over_rccl[0], (plain[0], over_rccl[0])
        assert plain[0][-1] < plain[0][0]
        for a, b in zip(plain[1], over_rccl[1]):
            assert torch.equal(a, b), float((a - b).abs().max())
        assert torch.equal(plain[2], over_rccl[2])
        open(os.path.join(tmp, "ok"), "w").write("ok")
    finally:
        parallel.EXCHANGE_WHEN_ALONE = False
        dist.destroy_process_group()


def test_data_parallel_paths_over_rccl_with_one_rank(gpu, tmp_path):
    """The collectives of the product (replica broadcast, in-backward all-reduce, sharded-render
    all-gather) on the RCCL backend itself, which needs one GPU per rank: a one-rank group."""
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_alone_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    if os.path.exists(tmp_path / "unavailable"):
        pytest.skip("RCCL process group could not be created here: " + open(tmp_path / "unavailable").read())
    assert os.path.exists(tmp_path / "ok")
