"""Training ray batches (run_nerf.py:615-656, :695-766): the pixel-id logic on the CPU, the
gather kernel against the table the reference would have built (fp32 NumPy restatement)."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as orc
from tests.util import pose_spherical


def _reference_table(H, W, focal, poses, images):
    """rays_rgb of run_nerf.py:634-656 for the given (training) views: [N*H*W, 3, 3]."""
    rays = np.stack([np.stack(orc.get_rays(H, W, focal, p[:3, :4], np.float32), 0) for p in poses], 0)  # [N,2,H,W,3]
    rays_rgb = np.concatenate([rays, images[:, None, ..., :3]], 1)        # [N,3,H,W,3]
    rays_rgb = np.transpose(rays_rgb, [0, 2, 3, 1, 4])                    # [N,H,W,3,3]
    return rays_rgb.reshape(-1, 3, 3).astype(np.float32)


def test_precrop_window_and_pixel_order():
    from nerf_amd.ray_batches import precrop_window, window_pixels
    H, W = 20, 30
    r0, r1, c0, c1 = precrop_window(H, W, 0.5)
    assert (r0, r1, c0, c1) == (10 - 5, 10 + 5, 15 - 7, 15 + 7)
    # the reference flattens meshgrid(rows, cols, indexing='ij') (run_nerf.py:750-759)
    rr, cc = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing='ij')
    np.testing.assert_array_equal(window_pixels(H, W, (r0, r1, c0, c1)), (rr * W + cc).reshape(-1))
    np.testing.assert_array_equal(window_pixels(H, W), np.arange(H * W))
    assert precrop_window(7, 9, 1.0) == (0, 6, 0, 8)   # odd sizes: int(H//2 * frac) on both sides


def _scene(n_views=5, H=12, W=16, seed=0):
    rng = np.random.default_rng(seed)
    poses = np.stack([pose_spherical(40. * k - 90., -30. + 5 * k, 4.) for k in range(n_views)], 0).astype(np.float32)
    images = rng.random((n_views, H, W, 4)).astype(np.float32)   # RGBA as load_blender returns
    return H, W, 20.5, poses, images


@pytest.mark.gpu
def test_gather_pixel_rays_equals_table_lookup(gpu):
    from nerf_amd import ops
    H, W, focal, poses, images = _scene()
    table = _reference_table(H, W, focal, poses, images)
    rng = np.random.default_rng(1)
    ids = rng.integers(0, table.shape[0], size=777)
    rays, rgb = ops.gather_pixel_rays(H, W, focal, torch.as_tensor(poses[:, :3, :4]).to(gpu),
                                      torch.as_tensor(images).to(gpu), ids // (H * W), ids % (H * W))
    assert rays.shape == (2, 777, 3) and rgb.shape == (777, 3)
    np.testing.assert_array_equal(rays[0].cpu().numpy(), table[ids, 0])
    np.testing.assert_array_equal(rays[1].cpu().numpy(), table[ids, 1])
    np.testing.assert_array_equal(rgb.cpu().numpy(), table[ids, 2])
    # colours are optional
    rays2, none = ops.gather_pixel_rays(H, W, focal, torch.as_tensor(poses[:, :3, :4]).to(gpu), None,
                                        ids // (H * W), ids % (H * W))
    assert none is None and torch.equal(rays2, rays)
    # empty batch
    r0, c0 = ops.gather_pixel_rays(H, W, focal, torch.as_tensor(poses[:, :3, :4]).to(gpu),
                                   torch.as_tensor(images).to(gpu), [], [])
    assert r0.shape == (2, 0, 3) and c0.shape == (0, 3)


@pytest.mark.gpu
def test_pooled_batches_cover_every_training_ray_once_per_epoch(gpu):
    from nerf_amd.ray_batches import RayBatcher
    H, W, focal, poses, images = _scene()
    i_train = [0, 2, 3]
    table = _reference_table(H, W, focal, poses[i_train], images[i_train])
    n = table.shape[0]
    N_rand = 100                                   # 576 rays: 5 full batches + one of 76
    b = RayBatcher(H, W, focal, poses, images, i_train, N_rand, use_batching=True, device=gpu, seed=7)
    perm = np.random.default_rng(5).permutation(n)
    b.shuffle(perm)
    got = []
    for step in range(6):
        rays, tgt = b.sample(step)
        sl = perm[step * N_rand:(step + 1) * N_rand]
        assert rays.shape[1] == len(sl)            # the last slice of an epoch is short (run_nerf.py:711)
        np.testing.assert_array_equal(rays[0].cpu().numpy(), table[sl, 0])
        np.testing.assert_array_equal(rays[1].cpu().numpy(), table[sl, 1])
        np.testing.assert_array_equal(tgt.cpu().numpy(), table[sl, 2])
        got.append(torch.cat([rays[0], rays[1], tgt], -1).cpu().numpy())
    got = np.concatenate(got, 0)
    assert got.shape[0] == n
    ref = table.reshape(n, 9)
    assert sorted(map(bytes, got)) == sorted(map(bytes, ref))   # a permutation of the table
    # the pool is exhausted: the next call reshuffles on its own and starts a new epoch
    rays, _ = b.sample(6)
    assert rays.shape[1] == N_rand and b.i_batch == N_rand


@pytest.mark.gpu
def test_single_image_batches_respect_precrop(gpu):
    from nerf_amd.ray_batches import RayBatcher, precrop_window, window_pixels
    H, W, focal, poses, images = _scene(H=20, W=30)
    i_train = [1, 4]
    b = RayBatcher(H, W, focal, poses, images, i_train, 40, use_batching=False, precrop_iters=3,
                   precrop_frac=0.5, device=gpu, seed=3)
    table = _reference_table(H, W, focal, poses[i_train], images[i_train]).reshape(2, H * W, 3, 3)
    win = window_pixels(H, W, precrop_window(H, W, 0.5))
    sel = np.random.default_rng(2).choice(len(win), size=40, replace=False)
    rays, tgt = b.sample(0, image=1, select=sel)            # i < precrop_iters: central crop
    np.testing.assert_array_equal(rays[1].cpu().numpy(), table[1, win[sel], 1])
    np.testing.assert_array_equal(tgt.cpu().numpy(), table[1, win[sel], 2])
    sel_full = np.random.default_rng(4).choice(H * W, size=40, replace=False)
    rays, tgt = b.sample(3, image=0, select=sel_full)       # afterwards: the whole image
    np.testing.assert_array_equal(rays[0].cpu().numpy(), table[0, sel_full, 0])
    np.testing.assert_array_equal(rays[1].cpu().numpy(), table[0, sel_full, 1])
    np.testing.assert_array_equal(tgt.cpu().numpy(), table[0, sel_full, 2])
    # unpinned draws: N_rand distinct pixels of one training image, inside the crop while it applies
    rays, tgt = b.sample(1)
    assert rays.shape == (2, 40, 3)
    d = rays[1].cpu().numpy()
    hits = [(np.abs(table[k, :, 1][None] - d[:, None]).max(-1) == 0) for k in range(2)]
    k = int(np.argmax([h.any(1).all() for h in hits]))
    pix = hits[k].argmax(1)
    assert len(set(pix.tolist())) == 40 and set(pix.tolist()) <= set(win.tolist())


@pytest.mark.gpu
def test_data_parallel_batchers_split_one_global_batch(gpu):
    """RayBatcher(rank, world_size): the ranks draw the same global batch from a common seed and
    keep disjoint contiguous slices of it, in both modes (run_nerf.py:695-766 run by R processes)."""
    from nerf_amd.ray_batches import RayBatcher
    H, W, focal, poses, images = _scene()
    i_train = [0, 1, 3]
    N_rand, R = 101, 3                             # slices of 34, 34 and 33 rays
    for batching in (True, False):
        whole = RayBatcher(H, W, focal, poses, images, i_train, N_rand, use_batching=batching, device=gpu, seed=11)
        parts = [RayBatcher(H, W, focal, poses, images, i_train, N_rand, use_batching=batching, device=gpu, seed=11,
                            rank=r, world_size=R) for r in range(R)]
        for step in range(3):
            rays, tgt = whole.sample(step)
            got = [p.sample(step) for p in parts]
            assert [g[0].shape[1] for g in got] == [34, 34, 33]
            assert torch.equal(torch.cat([g[0] for g in got], 1), rays)
            assert torch.equal(torch.cat([g[1] for g in got], 0), tgt)
    with pytest.raises(ValueError):
        RayBatcher(H, W, focal, poses, images, i_train, N_rand, device=gpu, rank=0, world_size=2)   # no common seed
