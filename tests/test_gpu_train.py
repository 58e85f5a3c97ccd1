"""GPU parity of the training-side kernels against torch-CPU autograd (float64) of the restatement."""
import numpy as np
import pytest
import torch

from oracle import torch_restatement as trs
from tests.util import synthetic_rays, trained_like_weights

pytestmark = pytest.mark.gpu


def _t(x, gpu):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=gpu)


@pytest.mark.parametrize("S,white", [(64, False), (64, True), (192, True), (100, False), (1, False), (384, True)])
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
    assert torch.equal(raw_t, raw_i)
    # decode the saved h7 of the first wave tile (points 0..31) and compare with the oracle
    blk = act.view(-1, 158, 1024)[0]
    h7 = blk[4 + 16 * 7: 4 + 16 * 8].contiguous().view(torch.float16).view(16, 64, 8).float().cpu().numpy()
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
    for k in range(16):
        for lane in range(64):
            c, hh = lane & 31, lane >> 5
            for j in range(8):
                f = 32 * (k >> 1) + 16 * (k & 1) + 8 * (j >> 2) + 4 * hh + (j & 3)
                assert abs(h7[k, lane, j] - hcur[c, f]) < 2e-3 * max(1., abs(hcur[c, f])), (k, lane, j)
