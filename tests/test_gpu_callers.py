"""GPU tests of the callers and data formats either side of the per-ray path: render_path
(run_nerf.py:122-161), checkpoint reload through create_nerf (:244-261, :838-846), the data-set
readers and option reader feeding the training step (load_blender.py:71-178, load_llff.py:269-342,
args_and_config.py), the 800 x 800 view and the no_ndc + lindisp configuration at 256 coarse
samples, and the remaining branches of render() (:66-69 depth-bounded rays, :82-84 static camera,
raw_noise_std > 0 through render_rays)."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as orc
from oracle import torch_restatement as trs
from tests.util import (assert_rgb_parity, crossed_rays, default_args, fitted_like_weights, gpu_coarse_pass,
                        pose_spherical, synthetic_rays, trained_like_weights)

pytestmark = pytest.mark.gpu

RGB_TOL = 2e-3   # fp16-MFMA MLP: max abs rgb error
FP32_TOL = 1e-5  # fp32 compositing / sampling


def _t(x, gpu):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=gpu)


def _assert_free_running(what, got_rgb, got_raw, ref, models, rays11, Sc, Si, white, lindisp=False, tol=RGB_TOL, **rnd):
    """Final colour of a free-running render against the oracle's (ref = orc.render(..., retraw=True)):
    max abs error `tol` on every ray that crossed none of the reference's discontinuities."""
    rex = ref[3]
    zc, raw_c, w_c = gpu_coarse_pass(models['model'], rays11, Sc, lindisp, white, rnd.get('perturb_u'), rnd.get('noise0'),
                                     rnd.get('noise_std', 0.))
    host = lambda k: None if rnd.get(k) is None else rnd[k].cpu().numpy()
    crossed = crossed_rays(rex['raw0'].reshape(-1, Sc, 4), rex['raw'].reshape(-1, Sc + Si, 4), rex['weights0'].reshape(-1, Sc),
                           raw_c, got_raw.reshape(-1, Sc + Si, 4).cpu().numpy(), w_c, Si, host('pdf_u'), host('noise0'),
                           host('noise1'), rnd.get('noise_std', 0.))
    assert_rgb_parity(np.abs(got_rgb.reshape(-1, 3).cpu().numpy() - np.asarray(ref[0]).reshape(-1, 3)), crossed, tol, what=what)


def _nets(args, seed):
    """Both networks with the spectrum of a fitted field: these tests compare THROUGH the resampling
    step (tests/util.py:fitted_like_weights)."""
    from nerf_amd.run_nerf import create_nerf
    kw_train, kw_test, start, grad_vars, models = create_nerf(args)
    rng = np.random.default_rng(seed)
    wc, wf = fitted_like_weights(rng), fitted_like_weights(rng)
    models['model'].set_weights(wc)
    models['model_fine'].set_weights(wf)
    return kw_train, kw_test, grad_vars, models, wc, wf


# ---- render_path (run_nerf.py:122-161) ------------------------------------------------------------
@pytest.mark.parametrize("render_factor", [0, 2])
def test_render_path(gpu, tmp_path, capsys, render_factor):
    from nerf_amd.imageio_lite import imread
    from nerf_amd.run_nerf import render, render_path
    from nerf_amd.run_nerf_helpers import to8b
    kw_train, kw_test, _, models, wc, wf = _nets(default_args(white_bkgd=True, N_samples=32, N_importance=32), 3)
    H, W = 12, 10
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)
    poses = np.stack([pose_spherical(a, -30., 4.) for a in (-180., -60., 75.)], 0).astype(np.float32)
    rng = np.random.default_rng(0)
    gt = rng.uniform(size=(3, H, W, 3)).astype(np.float32)
    kw = dict(kw_test, near=2., far=6.)
    savedir = str(tmp_path)
    rgbs, extras = render_path(poses, [H, W, focal], 40, kw, gt_imgs=gt, savedir=savedir, render_factor=render_factor)
    h, w, f = (H, W, focal) if render_factor == 0 else (H // render_factor, W // render_factor, focal / render_factor)
    assert isinstance(rgbs, np.ndarray) and rgbs.shape == (3, h, w, 3)
    # extras are stacked per pose, with the disparity map added under 'disp' (:142)
    assert set(extras) >= {'disp', 'rgb0', 'disp0', 'acc0', 'z_std', 'depth_map'}
    assert extras['disp'].shape == (3, h, w) and extras['rgb0'].shape == (3, h, w, 3)
    printed = []
    for s in capsys.readouterr().out.split('\n'):
        try:
            printed.append(float(s))
        except ValueError:
            pass
    for i in range(3):
        rgb, disp, acc, ex = render(h, w, f, chunk=40, c2w=torch.as_tensor(poses[i, :3, :4]), **kw)
        np.testing.assert_array_equal(rgbs[i], rgb.cpu().numpy())
        np.testing.assert_array_equal(extras['disp'][i], disp.cpu().numpy())
        np.testing.assert_array_equal(extras['depth_map'][i], ex['depth_map'].cpu().numpy())
        back = imread(os.path.join(savedir, '{:03d}.png'.format(i)))
        np.testing.assert_array_equal(back, to8b(rgbs[i]))
        if render_factor == 0:  # per-view PSNR against the ground truth is printed (:148-150)
            p = -10. * np.log10(np.mean(np.square(rgbs[i] - gt[i])))
            assert any(abs(s - p) < 1e-3 for s in printed), (p, printed)
    if render_factor != 0:
        assert not printed
    # the first view against the float64 oracle
    from nerf_amd import ops
    ref = orc.render(h, w, f, chunk=40, c2w=poses[0, :3, :4], ndc=False, near=2., far=6., use_viewdirs=True,
                     weights_coarse=wc, weights_fine=wf, N_samples=32, N_importance=32, white_bkgd=True,
                     perturb=0., retraw=True, dtype=np.float64)
    c2w0 = torch.as_tensor(poses[0, :3, :4])
    again = render(h, w, f, chunk=40, c2w=c2w0, retraw=True, **kw)
    np.testing.assert_array_equal(again[0].cpu().numpy(), rgbs[0])
    o, d = ops.get_rays(h, w, f, c2w0)
    rays11 = ops.build_ray_batch(o, d, d, False, h, w, f, 2., 6.)
    _assert_free_running("render_path view 0", again[0], again[3]['raw'], ref, models, rays11, 32, 32, True)


# ---- checkpoints through create_nerf (run_nerf.py:244-261, :838-846) ---------------------------------
def test_checkpoint_reload_through_create_nerf(gpu, tmp_path):
    from nerf_amd import checkpoint
    from nerf_amd.run_nerf import create_nerf, render
    base = str(tmp_path)
    args = default_args(white_bkgd=True, basedir=base, expname='exp', no_reload=False)
    kw_train, kw_test, start, grad_vars, models = create_nerf(args)
    assert start == 0                       # nothing to reload yet
    rng = np.random.default_rng(17)
    wc, wf = trained_like_weights(rng), trained_like_weights(rng)
    models['model'].set_weights(wc)
    models['model_fine'].set_weights(wf)
    rb = synthetic_rays(200, seed=2)
    rays = (_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu))
    want = render(1, 1, 1., chunk=64, rays=rays, near=2., far=6., **kw_test)
    # an older and a newer checkpoint: the newest coarse file wins and names the fine file (:247-261)
    old = {k: m for k, m in models.items()}
    for k in ('model', 'model_fine'):
        checkpoint.save_weights(old[k], k, 199999, base, 'exp')
    junk = trained_like_weights(np.random.default_rng(99))
    models['model'].set_weights(junk)
    models['model_fine'].set_weights(junk)
    for k in ('model', 'model_fine'):
        checkpoint.save_weights(models[k], k, 1000, base, 'exp')
    assert sorted(os.listdir(os.path.join(base, 'exp'))) == ['model_001000.npy', 'model_199999.npy',
                                                             'model_fine_001000.npy', 'model_fine_199999.npy']
    kw_train2, kw_test2, start2, grad_vars2, models2 = create_nerf(args)
    assert start2 == 199999 + 1                                  # :254
    for a, b in zip(models2['model'].get_weights(), wc):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(models2['model_fine'].get_weights(), wf):
        np.testing.assert_array_equal(a, b)
    got = render(1, 1, 1., chunk=64, rays=rays, near=2., far=6., **kw_test2)
    same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=-1.), torch.nan_to_num(y, nan=-1.))  # disp is NaN where acc == 0
    for x, y in zip(got[:3], want[:3]):
        assert same(x, y)
    assert same(got[3]['rgb0'], want[3]['rgb0'])
    # ft_path names one file explicitly (:244-246); no_reload ignores everything (:250)
    args_ft = default_args(white_bkgd=True, basedir=base, expname='exp', no_reload=False,
                           ft_path=os.path.join(base, 'exp', 'model_001000.npy'))
    _, _, start3, _, models3 = create_nerf(args_ft)
    assert start3 == 1001
    np.testing.assert_array_equal(models3['model_fine'].get_weights()[0], junk[0])
    _, _, start4, _, _ = create_nerf(default_args(basedir=base, expname='exp', no_reload=True))
    assert start4 == 0


# ---- data sets and options into the path -------------------------------------------------------------
def test_blender_config_to_training_steps(gpu, tmp_path):
    """config file -> config_parser -> load_blender_data(half_res) -> white-background blend
    (run_nerf.py:415-416) -> RayBatcher (no_batching, pre-crop) -> three optimisation steps, then a
    render_path view of the loaded render poses."""
    from nerf_amd import training
    from nerf_amd.args_and_config import config_parser
    from nerf_amd.load_blender import load_blender_data
    from nerf_amd.ray_batches import RayBatcher
    from nerf_amd.run_nerf import create_nerf, render, render_path
    from tests.test_loaders import _make_blender
    data = str(tmp_path / "scene")
    os.makedirs(data)
    truth = _make_blender(data, H=16, W=16, counts=(4, 2, 2))
    cfg = tmp_path / "config.txt"
    cfg.write_text(f"expname = synth\nbasedir = {tmp_path}/logs\ndatadir = {data}\ndataset_type = blender\n"
                   "half_res = True\nno_batching = True\nN_samples = 64\nN_importance = 128\nuse_viewdirs = True\n"
                   "white_bkgd = True\nlrate_decay = 500\nN_rand = 48\nprecrop_iters = 2\nprecrop_frac = 0.5\n"
                   "testskip = 1\nno_reload = True\n")
    args = config_parser().parse_args(['--config', str(cfg)])
    assert args.N_rand == 48 and args.half_res and args.no_batching and args.N_importance == 128
    images, poses, render_poses, hwf, i_split, extras = load_blender_data(
        args.datadir, args.half_res, args.testskip, args.image_extn, trainskip=args.trainskip,
        train_frames_field=args.frames_field)
    i_train, i_val, i_test = i_split
    H, W, focal = hwf
    assert (H, W) == (8, 8) and images.shape == (8, 8, 8, 4)
    near, far = 2., 6.
    images = images[..., :3] * images[..., -1:] + (1. - images[..., -1:])             # :416
    torch.manual_seed(0)
    kw_train, kw_test, start, grad_vars, models = create_nerf(args)
    assert kw_train['ndc'] is False and kw_train['perturb'] == 1. and kw_train['N_samples'] == 64
    batcher = RayBatcher(H, W, focal, poses, images, i_train, args.N_rand, use_batching=not args.no_batching,
                         precrop_iters=args.precrop_iters, precrop_frac=args.precrop_frac, device=gpu, seed=1)
    opt = training.Adam(training.ExponentialDecay(args.lrate, args.lrate_decay * 1000, 0.1))
    fn = lambda r: render(H, W, focal, chunk=args.chunk, rays=r, near=near, far=far, retraw=True, **kw_train)
    before = [p.detach().clone() for p in grad_vars]
    losses = []
    for i in range(3):
        batch_rays, target_s = batcher.sample(i)
        n_i = batch_rays.shape[1]
        assert batch_rays.shape == (2, n_i, 3) and target_s.shape == (n_i, 3)
        assert n_i == (16 if i < args.precrop_iters else 48)      # the 4x4 central crop holds 16 pixels
        loss, psnr, psnr0 = training.train_step(fn, batch_rays, target_s, grad_vars, opt)
        losses.append(float(loss))
    assert np.all(np.isfinite(losses)) and opt.iterations == 3
    assert all(not torch.equal(a, b.detach()) for a, b in zip(before, grad_vars))
    # the targets really are pixels of the loaded images
    flat = torch.as_tensor(images[i_train]).reshape(-1, 3).to(gpu)
    assert bool((target_s[:, None, :] == flat[None]).all(-1).any(-1).all())
    rgbs, ex = render_path(render_poses[:1], hwf, args.chunk, dict(kw_test, near=near, far=far))
    assert rgbs.shape == (1, 8, 8, 3) and np.all(np.isfinite(rgbs))


def test_llff_data_to_ndc_render(gpu, tmp_path):
    """poses_bounds.npy + images_4/ -> load_llff_data -> hwf / poses split as train() does
    (run_nerf.py:337-362) -> render(ndc) of the held-out view and of a spiral pose, against the
    float64 oracle; the RayBatcher's pooled batches over the training views feed one step."""
    from nerf_amd import training
    from nerf_amd.load_llff import load_llff_data
    from nerf_amd.ray_batches import RayBatcher
    from nerf_amd.run_nerf import render
    from tests.test_loaders import _make_llff
    _make_llff(str(tmp_path), n=5, H=6, W=10, factor=4)
    images, poses, bds, render_poses, i_test = load_llff_data(str(tmp_path), 4, recenter=True, bd_factor=.75)
    hwf = poses[0, :3, -1]
    poses = poses[:, :3, :4]
    H, W, focal = int(hwf[0]), int(hwf[1]), float(hwf[2])
    assert (H, W, focal) == (6, 10, 100.)
    args = default_args(white_bkgd=False, dataset_type='llff', N_samples=32, N_importance=32)
    kw_train, kw_test, grad_vars, models, wc, wf = _nets(args, 8)
    assert 'ndc' not in kw_test                                  # run_nerf.py:230-233: render()'s default, True
    for c2w in (poses[i_test], render_poses[7, :3, :4]):
        rgb, disp, acc, extras = render(H, W, focal, chunk=32, c2w=torch.as_tensor(c2w), near=0., far=1., **kw_test)
        ref = orc.render(H, W, focal, chunk=32, c2w=np.asarray(c2w, np.float32), ndc=True, near=0., far=1.,
                         use_viewdirs=True, weights_coarse=wc, weights_fine=wf, N_samples=32, N_importance=32,
                         white_bkgd=False, perturb=0., dtype=np.float64)
        assert rgb.shape == (H, W, 3)
        np.testing.assert_allclose(rgb.cpu().numpy(), ref[0], atol=RGB_TOL)
        np.testing.assert_allclose(acc.cpu().numpy(), ref[2], atol=RGB_TOL)
    i_train = np.array([i for i in range(images.shape[0]) if i != i_test])
    batcher = RayBatcher(H, W, focal, poses, images, i_train, 40, use_batching=True, device=gpu, seed=0)
    opt = training.Adam(training.ExponentialDecay(5e-4, 250 * 1000, 0.1))
    fn = lambda r: render(H, W, focal, chunk=1024, rays=r, near=0., far=1., retraw=True, **dict(kw_train, perturb=1.))
    batch_rays, target_s = batcher.sample(0)
    loss, psnr, psnr0 = training.train_step(fn, batch_rays, target_s, grad_vars, opt)
    assert np.isfinite(float(loss)) and np.isfinite(float(psnr0))


# ---- full-size views -------------------------------------------------------------------------------
def _check_rows_against_restatement(gpu, rgb, rays, rows, W, models, white, Sc, Si, lindisp=False):
    from tests.util import check_rows_against_restatement
    sl = slice(rows[0] * W, rows[1] * W)
    check_rows_against_restatement(gpu, rgb.reshape(-1, 3)[sl], rays[sl], models, white, Sc, Si, lindisp,
                                   what=f"rows {rows} of the full-size view")


def test_blender_view_800x800(gpu):
    """The full-resolution novel view of the blender paper configuration: 640,000 rays from one
    pose, 64 + 128 samples, white background, chunk 32768 (20 render_rays calls).  Chunk
    invariance bit for bit, and two pixel rows against the float64 restatement of the whole path."""
    from nerf_amd import ops
    from nerf_amd.run_nerf import render
    H = W = 800
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)
    kw_train, kw_test, _, models, wc, wf = _nets(default_args(white_bkgd=True), 41)
    c2w = torch.as_tensor(pose_spherical(-120., -30., 4.)[:3, :4].astype(np.float32))
    a = render(H, W, focal, chunk=1024 * 32, c2w=c2w, near=2., far=6., **kw_test)
    b = render(H, W, focal, chunk=1024 * 128, c2w=c2w, near=2., far=6., **kw_test)
    rgb, disp, acc = a[:3]
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and acc.shape == (H, W)
    assert a[3]['rgb0'].shape == (H, W, 3) and a[3]['z_std'].shape == (H, W)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert bool(torch.isfinite(rgb).all())
    assert float(acc.min()) >= 0. and float(acc.max()) <= 1. + FP32_TOL
    o, d = ops.get_rays(H, W, focal, c2w)
    rays = ops.build_ray_batch(o, d, d, False, H, W, focal, 2., 6.)
    _check_rows_against_restatement(gpu, rgb, rays, (400, 402), W, models, True, 64, 128)


def test_blender_view_400x400_half_res(gpu):
    """The half-resolution blender view at its own size (config_lego.txt: half_res, white_bkgd,
    use_viewdirs; 64 + 128 samples as paper_configs/blender_config.txt:16-17): 160,000 rays from one
    pose, chunk 32768, i.e. five render_rays calls of which the last is ragged (28,928 rays).
    Chunk invariance bit for bit against one 160,000-ray call, and two pixel rows through the
    middle of the view against the float64 restatement of the whole path."""
    from nerf_amd import ops
    from nerf_amd.run_nerf import render
    H = W = 400
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)      # load_blender.py:158-167 at half_res
    kw_train, kw_test, _, models, wc, wf = _nets(default_args(white_bkgd=True), 47)
    c2w = torch.as_tensor(pose_spherical(-180., -30., 4.)[:3, :4].astype(np.float32))
    a = render(H, W, focal, chunk=1024 * 32, c2w=c2w, near=2., far=6., **kw_test)
    b = render(H, W, focal, chunk=H * W, c2w=c2w, near=2., far=6., **kw_test)
    rgb, disp, acc = a[:3]
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and acc.shape == (H, W)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(torch.nan_to_num(x, nan=-1.), torch.nan_to_num(y, nan=-1.))
    for k in ('rgb0', 'acc0', 'z_std', 'depth_map'):
        assert torch.equal(a[3][k], b[3][k])
    assert bool(torch.isfinite(rgb).all())
    assert float(acc.min()) >= 0. and float(acc.max()) <= 1. + FP32_TOL
    o, d = ops.get_rays(H, W, focal, c2w)
    rays = ops.build_ray_batch(o, d, d, False, H, W, focal, 2., 6.)
    _check_rows_against_restatement(gpu, rgb, rays, (200, 202), W, models, True, 64, 128)


def test_llff_view_no_ndc_lindisp_256_samples(gpu):
    """The forward-facing view without NDC: sampling linear in disparity between
    near = 0.9 min(bds) and far = max(bds) (run_nerf.py:357-359) with 256 coarse + 128 fine samples."""
    from nerf_amd import ops
    from nerf_amd.run_nerf import render
    H, W, focal = 756, 1008, 815.0
    args = default_args(white_bkgd=False, dataset_type='llff', no_ndc=True, lindisp=True, N_samples=256,
                        N_importance=128)
    kw_train, kw_test, _, models, wc, wf = _nets(args, 43)
    assert kw_test['ndc'] is False and kw_test['lindisp'] is True           # run_nerf.py:230-233
    near, far = 0.9 * 1.3333, 12.5
    c2w = torch.as_tensor(np.concatenate([np.eye(3), np.array([[0.05], [0.02], [0.1]])], 1).astype(np.float32))
    a = render(H, W, focal, chunk=1024 * 64, c2w=c2w, near=near, far=far, **kw_test)
    b = render(H, W, focal, chunk=1024 * 32, c2w=c2w, near=near, far=far, **kw_test)
    rgb, disp, acc = a[:3]
    assert rgb.shape == (H, W, 3)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert bool(torch.isfinite(rgb).all()) and float(acc.max()) <= 1. + FP32_TOL
    o, d = ops.get_rays(H, W, focal, c2w)
    rays = ops.build_ray_batch(o, d, d, False, H, W, focal, near, far)
    # the depths really are uniform in 1/z
    z = ops.sample_z(rays[:4], 256, True)
    inv = 1. / z.double()
    steps = inv[:, 1:] - inv[:, :-1]
    assert float((steps - steps.mean(-1, keepdim=True)).abs().max()) < 1e-6
    _check_rows_against_restatement(gpu, rgb, rays, (300, 301), W, models, False, 256, 128, lindisp=True)


# ---- the remaining branches of render() ------------------------------------------------------------
def test_render_depth_bounded_rays(gpu):
    """rays = (o, d, depth): near / far = 0.9 / 1.1 x depth per ray (run_nerf.py:66-69)."""
    from nerf_amd.run_nerf import render
    kw_train, kw_test, _, models, wc, wf = _nets(default_args(white_bkgd=False, N_samples=32, N_importance=32), 5)
    n = 150
    rb = synthetic_rays(n, seed=3)
    depth = np.random.default_rng(4).uniform(3., 5., size=(n,)).astype(np.float32)
    rgb, disp, acc, extras = render(1, 1, 1., chunk=64, rays=(_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu), _t(depth, gpu)),
                                    near=123., far=456., retraw=True, **kw_test)
    ref = orc.render(1, 1, 1., chunk=64, rays=(rb[:, :3], rb[:, 3:6], depth), ndc=False, use_viewdirs=True,
                     weights_coarse=wc, weights_fine=wf, N_samples=32, N_importance=32, white_bkgd=False,
                     perturb=0., retraw=True, dtype=np.float64)
    assert rgb.shape == (n, 3)
    from nerf_amd import ops
    o, d = _t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu)
    rays11 = ops.build_ray_batch(o, d, d, False, 1, 1, 1., _t(0.9 * depth, gpu), _t(1.1 * depth, gpu))
    _assert_free_running("depth-bounded rays", rgb, extras['raw'], ref, models, rays11, 32, 32, False)
    # the expected depth lies inside each ray's own bounds
    dm = extras['depth_map'].cpu().numpy() / np.maximum(acc.cpu().numpy(), 1e-6)
    ok = acc.cpu().numpy() > 0.5
    assert np.all(dm[ok] > 0.9 * depth[ok] - 1e-3) and np.all(dm[ok] < 1.1 * depth[ok] + 1e-3)


def test_render_static_camera(gpu):
    """c2w_staticcam: rays from the static camera, view directions from c2w (run_nerf.py:82-84)."""
    from nerf_amd.run_nerf import render
    kw_train, kw_test, _, models, wc, wf = _nets(default_args(white_bkgd=True, N_samples=32, N_importance=32), 6)
    H, W = 9, 11
    focal = 12.0
    c2w = pose_spherical(40., -30., 4.)[:3, :4].astype(np.float32)
    static = pose_spherical(-10., -50., 4.)[:3, :4].astype(np.float32)
    rgb, disp, acc, _ = render(H, W, focal, chunk=50, c2w=torch.as_tensor(c2w), c2w_staticcam=torch.as_tensor(static),
                               near=2., far=6., **kw_test)
    ref = orc.render(H, W, focal, chunk=50, c2w=c2w, c2w_staticcam=static, ndc=False, near=2., far=6.,
                     use_viewdirs=True, weights_coarse=wc, weights_fine=wf, N_samples=32, N_importance=32,
                     white_bkgd=True, perturb=0., dtype=np.float64)
    np.testing.assert_allclose(rgb.cpu().numpy(), ref[0], atol=RGB_TOL)
    # and it differs from both plain renders (the branch is observable)
    plain = render(H, W, focal, chunk=50, c2w=torch.as_tensor(static), near=2., far=6., **kw_test)[0]
    assert float((plain - rgb).abs().max()) > 1e-3


def test_render_rays_with_density_noise(gpu):
    """raw_noise_std > 0 through render_rays: sigma + noise * std before the relu, in the coarse
    and the fine compositing, with the normals supplied to both sides."""
    from nerf_amd.run_nerf import render
    args = default_args(white_bkgd=False, raw_noise_std=1.0, perturb=1.)
    kw_train, kw_test, _, models, wc, wf = _nets(args, 7)
    assert kw_train['raw_noise_std'] == 1.0 and kw_test['raw_noise_std'] == 0.       # run_nerf.py:235-238
    n = 128
    rb = synthetic_rays(n, seed=9)
    rng = np.random.default_rng(2)
    pu = rng.uniform(size=(n, 64)).astype(np.float32)
    uu = rng.uniform(size=(n, 128)).astype(np.float32)
    n0 = rng.normal(size=(n, 64)).astype(np.float32)
    n1 = rng.normal(size=(n, 192)).astype(np.float32)
    rays = (_t(rb[:, :3], gpu), _t(rb[:, 3:6], gpu))
    with torch.no_grad():
        rgb, disp, acc, extras = render(1, 1, 1., chunk=50, rays=rays, near=2., far=6., retraw=True,
                                        perturb_u=_t(pu, gpu), pdf_u=_t(uu, gpu), noise0=_t(n0, gpu),
                                        noise1=_t(n1, gpu), **kw_train)
        quiet = render(1, 1, 1., chunk=50, rays=rays, near=2., far=6., perturb_u=_t(pu, gpu), pdf_u=_t(uu, gpu),
                       **dict(kw_train, raw_noise_std=0.))[3]['rgb0']
    ref = orc.render(1, 1, 1., chunk=50, rays=(rb[:, :3], rb[:, 3:6]), ndc=False, near=2., far=6., use_viewdirs=True,
                     weights_coarse=wc, weights_fine=wf, N_samples=64, N_importance=128, white_bkgd=False,
                     perturb=1., raw_noise_std=1.0, perturb_u=pu, pdf_u=uu, noise0=n0, noise1=n1, retraw=True,
                     dtype=np.float64)
    # the noise is visible
    assert float((quiet - extras['rgb0']).abs().max()) > 1e-2
    from nerf_amd import ops
    rays11 = ops.build_ray_batch(rays[0], rays[1], rays[1], False, 1, 1, 1., 2., 6.)
    _assert_free_running("density noise", rgb, extras['raw'], ref, models, rays11, 64, 128, False, perturb_u=_t(pu, gpu),
                         pdf_u=_t(uu, gpu), noise0=_t(n0, gpu), noise1=_t(n1, gpu), noise_std=1.0)
    # a wrong-shaped noise array is rejected
    with pytest.raises(ValueError):
        render(1, 1, 1., chunk=50, rays=rays, near=2., far=6., noise0=_t(n0[:, :5], gpu), **kw_train)


# ---- the whole core in one C-ABI call ----------------------------------------------------------------
@pytest.mark.parametrize("Sc,Si,lindisp,white,random", [(64, 128, False, True, False), (64, 128, False, False, True),
                                                        (40, 0, True, True, False), (256, 128, True, False, True)])
def test_render_rays_single_call_equals_stagewise(gpu, Sc, Si, lindisp, white, random):
    """nerf_render_rays (what render_rays uses for plain inference) gives the bits of the six
    per-stage calls, with and without supplied uniforms, with and without a fine pass."""
    from nerf_amd import nerf_renderer
    args = default_args(white_bkgd=white, lindisp=lindisp, N_samples=Sc, N_importance=Si, perturb=1. if random else 0.)
    kw_train, kw_test, _, models, wc, wf = _nets(args, 12) if Si > 0 else (None,) * 6
    if Si == 0:
        from nerf_amd.run_nerf import create_nerf
        kw_train, kw_test, _, _, models = create_nerf(args)
        models['model'].set_weights(trained_like_weights(np.random.default_rng(12)))
    kw = dict(kw_train if random else kw_test)
    n = 777
    rays = _t(synthetic_rays(n, seed=13), gpu)
    rng = np.random.default_rng(3)
    if random:
        kw.update(perturb_u=_t(rng.uniform(size=(n, Sc)), gpu), pdf_u=_t(rng.uniform(size=(n, max(Si, 1))), gpu) if Si else None)
    for k in ('ndc', 'use_viewdirs'):
        kw.pop(k, None)
    try:
        nerf_renderer.SINGLE_CALL = True
        one = nerf_renderer.render_rays(rays, **kw)
        nerf_renderer.SINGLE_CALL = False
        six = nerf_renderer.render_rays(rays, **kw)
    finally:
        nerf_renderer.SINGLE_CALL = True
    assert set(one) == set(six) and ('rgb0' in one) == (Si > 0)
    for k in six:
        assert one[k].shape == six[k].shape, k
        assert torch.equal(torch.nan_to_num(one[k], nan=-1.), torch.nan_to_num(six[k], nan=-1.)), k


def test_renders_on_two_streams_do_not_share_intermediates(gpu):
    """Two ray batches rendered on two torch streams, enqueued back to back so that they are in
    flight together, give the bits of the same renders made one after the other: the one-call
    path keeps its intermediates in a workspace per (device, stream)."""
    from nerf_amd import nerf_renderer, ops
    args = default_args(white_bkgd=True)
    kw_train, kw_test, _, models, wc, wf = _nets(args, 31)
    kw = {k: v for k, v in kw_test.items() if k not in ('ndc', 'use_viewdirs')}
    batches = [_t(synthetic_rays(n, seed=s), gpu) for n, s in ((4001, 41), (2503, 43))]
    want = [nerf_renderer.render_rays(b, **kw) for b in batches]   # also packs the weights, on the current stream
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu) for _ in batches]
    got = []
    for b, st in zip(batches, streams):
        with torch.cuda.stream(st):
            got.append(nerf_renderer.render_rays(b, **kw))
    torch.cuda.synchronize()
    for b, st in zip(batches, streams):          # a workspace of its own per stream
        assert (b.device, st.cuda_stream) in ops._RENDER_WS
    assert len({ops._RENDER_WS[(b.device, st.cuda_stream)].data_ptr() for b, st in zip(batches, streams)}) == 2
    for w, g in zip(want, got):
        for k in w:
            assert torch.equal(torch.nan_to_num(w[k], nan=-1.), torch.nan_to_num(g[k], nan=-1.)), k


# ---- the whole loop on a scene with a known answer ---------------------------------------------------
def test_trains_an_analytic_scene_end_to_end(gpu):
    """Twelve 32x32 views of a unit sphere coloured by its normal on a white background (rendered
    analytically from get_rays_np's rays), nine for training: RayBatcher -> train_step (both
    networks, 64+128 samples, perturb, Keras Adam with the reference's decay) for 300 steps, then
    novel views of the three held-out poses.  Untrained networks give about 8 dB; the test asks for
    22 dB on the held-out views (a run reaches 26)."""
    from nerf_amd import training
    from nerf_amd.load_blender import pose_spherical
    from nerf_amd.ray_batches import RayBatcher
    from nerf_amd.run_nerf import create_nerf, render
    from nerf_amd.run_nerf_helpers import get_rays_np
    H = W = 32
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)

    def sphere_image(c2w):
        o, d = get_rays_np(H, W, focal, c2w)
        o, d = o.reshape(-1, 3).astype(np.float64), d.reshape(-1, 3).astype(np.float64)
        dn = d / np.linalg.norm(d, axis=-1, keepdims=True)
        b = (o * dn).sum(-1)
        disc = b * b - ((o * o).sum(-1) - 1.0)
        hit = disc > 0
        p = o + dn * (-b - np.sqrt(np.where(hit, disc, 0.)))[:, None]
        return np.where(hit[:, None], 0.5 + 0.5 * p, 1.0).reshape(H, W, 3).astype(np.float32)

    thetas = np.linspace(-180, 180, 13)[:-1]
    poses = np.stack([pose_spherical(th, -30. + 20 * (i % 3), 4.)[:3, :4] for i, th in enumerate(thetas)]).astype(np.float32)
    images = np.stack([sphere_image(p) for p in poses])
    held_out = np.arange(12) % 4 == 3
    i_train, i_test = np.arange(12)[~held_out], np.arange(12)[held_out]
    torch.manual_seed(0)
    kw_train, kw_test, _, grad_vars, _ = create_nerf(default_args(white_bkgd=True, perturb=1.))
    batcher = RayBatcher(H, W, focal, poses, images, i_train, 1024, use_batching=True, device=gpu, seed=0)
    opt = training.Adam(training.ExponentialDecay(5e-4, 250 * 1000, 0.1))
    fn = lambda r: render(H, W, focal, chunk=32768, rays=r, near=2., far=6., retraw=True, **kw_train)

    def held_out_psnr():
        out = []
        for i in i_test:
            rgb = render(H, W, focal, chunk=32768, c2w=torch.as_tensor(poses[i]), near=2., far=6., **kw_test)[0]
            out.append(-10. * np.log10(float(((rgb.cpu().numpy() - images[i]) ** 2).mean())))
        return float(np.mean(out))

    start = held_out_psnr()
    for i in range(300):
        rays, target = batcher.sample(i)
        loss, psnr, psnr0 = training.train_step(fn, rays, target, grad_vars, opt)
    end = held_out_psnr()
    assert start < 12. and end > 22. and float(psnr) > 25., (start, end, float(psnr))


def _sharded_render_worker(rank, world_size, port, tmp):
    """One of two ranks sharing the one GPU of the box (gloo; over RCCL on a multi-GPU node)."""
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nerf_amd import parallel
    from nerf_amd.run_nerf import create_nerf, render, render_path
    torch.cuda.set_device(0)
    gpu = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    try:
        torch.manual_seed(50 + rank)               # create_nerf broadcasts rank 0's weights
        kw_train, kw_test, start, _, models = create_nerf(default_args())
        assert start == 0
        rb = synthetic_rays(1001, seed=9)          # odd: shards of 501 and 500 rays, several chunks each
        rays = (torch.as_tensor(rb[:, :3]).to(gpu), torch.as_tensor(rb[:, 3:6]).to(gpu))
        one = render(1, 1, 1., chunk=128, rays=rays, near=2., far=6., **kw_test)
        both = parallel.render(1, 1, 1., chunk=128, rays=rays, near=2., far=6., **kw_test)
        for a, b in zip(one[:3], both[:3]):
            assert a.shape == b.shape and torch.equal(a, b)
        assert set(one[3]) == set(both[3])
        for k in one[3]:
            assert torch.equal(one[3][k], both[3][k]), k
        # a camera view through render_path, sharded by a key in its render_kwargs
        H = W = 9
        poses = [pose_spherical(30., -30., 4.)]
        r1, e1 = render_path(poses, (H, W, 12.), 64, dict(kw_test, near=2., far=6.))
        r2, e2 = render_path(poses, (H, W, 12.), 64, dict(kw_test, near=2., far=6., shard_over_ranks=True))
        assert r1.shape == (1, H, W, 3) and np.array_equal(r1, r2) and np.array_equal(e1['disp'], e2['disp'])
        # the step a run resumes from is rank 0's on every rank
        assert parallel.broadcast_int(7 + rank) == 7
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_render_equals_single_rank_render_two_ranks(gpu, tmp_path):
    """parallel.render / render(shard_over_ranks=True) on two ranks: every rank gets, bit for bit,
    what one process renders alone (run_nerf.py:28-161 with the ray batch split over the ranks)."""
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_render_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
