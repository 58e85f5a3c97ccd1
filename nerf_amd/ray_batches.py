"""Training ray batches: the step of the reference's train() that runs right before render()
(run_nerf.py:615-656 table construction, :695-731 shuffled slices, :736-766 single-image sampling
with the central pre-crop).

MI355X-first shape of the same thing: the reference materialises a [N*H*W, 3, 3] fp32 table of
(origin, direction, colour) on the host and permutes the whole table every epoch.  Here the posed
images stay resident in HBM (12 B/pixel), an epoch is a permutation of pixel ids drawn on the GPU,
and a batch is one kernel launch that forms the N_rand rays from the poses with get_rays'
arithmetic (bit-identical to a table lookup) and gathers their colours -- 36 B/ray of traffic per
step instead of a 36 B x N*H*W shuffle per epoch.

Two quirks of the fork are not reproduced: `rays_rgb` is indexed by i_train twice
(run_nerf.py:642, :647), and depth-loss batches pass the origins as directions (:729).
"""
import numpy as np
import torch

from . import _native as nv
from . import ops


def precrop_window(H, W, precrop_frac):
    """run_nerf.py:748-753: rows [r0, r1) x columns [c0, c1) of the central crop."""
    dH = int(H // 2 * precrop_frac)
    dW = int(W // 2 * precrop_frac)
    return H // 2 - dH, H // 2 + dH, W // 2 - dW, W // 2 + dW


def window_pixels(H, W, window=None):
    """Flat pixel ids (row*W + column) of a window in the 'ij' meshgrid order the reference
    flattens its coords in (run_nerf.py:750-759)."""
    r0, r1, c0, c1 = (0, H, 0, W) if window is None else window
    rows = np.arange(r0, r1, dtype=np.int64)[:, None]
    cols = np.arange(c0, c1, dtype=np.int64)[None, :]
    return (rows * W + cols).reshape(-1)


class RayBatcher:
    """batcher.sample(i) -> (batch_rays [2,B,3], target_s [B,3]) for optimisation step i.

    use_batching=True  (run_nerf.py:695-731): every ray of every training image in one pool,
        reshuffled when the pool is exhausted; the last batch of an epoch may be short, as the
        reference's slice is.
    use_batching=False (run_nerf.py:736-766): one random training image per step, N_rand pixels
        without replacement, restricted to the central crop while i < precrop_iters.
    """

    def __init__(self, H, W, focal, poses, images, i_train, N_rand, use_batching=True,
                 precrop_iters=0, precrop_frac=0.5, device=None, seed=None, rank=0, world_size=1):
        self.H, self.W, self.focal = int(H), int(W), float(focal)
        dev = ops._dev() if device is None else torch.device(device)
        self.device = dev
        i_train = np.asarray(i_train, dtype=np.int64).reshape(-1)
        poses = np.asarray(poses, dtype=np.float32) if not isinstance(poses, torch.Tensor) else poses
        images = np.asarray(images, dtype=np.float32) if not isinstance(images, torch.Tensor) else images
        # only the training views are kept on the GPU
        self.poses = nv.f32(torch.as_tensor(poses)[torch.as_tensor(i_train)][:, :3, :4], dev)
        self.images = nv.f32(torch.as_tensor(images)[torch.as_tensor(i_train)], dev)
        if tuple(self.images.shape[1:3]) != (self.H, self.W):
            raise ValueError(f"RayBatcher: images are {tuple(self.images.shape[1:3])}, expected {(self.H, self.W)}")
        self.n_img = int(self.poses.shape[0])
        self.N_rand = int(N_rand)
        self.use_batching = bool(use_batching)
        self.precrop_iters, self.precrop_frac = int(precrop_iters), float(precrop_frac)
        # Data parallel: N_rand is the GLOBAL batch.  Every rank draws the same permutation / image
        # / pixel selection from the same seed and keeps its contiguous slice of the N_rand picks
        # (parallel.shard_bounds), so the ranks' shards are disjoint and together are exactly the
        # batch one process would have drawn.
        self.rank, self.world_size = int(rank), int(world_size)
        if not 0 <= self.rank < self.world_size:
            raise ValueError(f"RayBatcher: rank {rank} outside world of {world_size}")
        if self.world_size > 1 and seed is None:
            raise ValueError("RayBatcher: data-parallel ranks need a common seed to draw the same global batch")
        self.gen = torch.Generator(device=dev)
        self.host_gen = torch.Generator()  # the image of a step is drawn on the host: no device round trip
        if seed is not None:
            self.gen.manual_seed(int(seed))
            self.host_gen.manual_seed(int(seed) + 1)
        self._window_ids = {}  # window -> its flat pixel ids, resident on the device
        self.n_rays = self.n_img * self.H * self.W
        self.perm = None
        self.i_batch = 0

    # -- pooled batching ---------------------------------------------------------------------
    def shuffle(self, perm=None):
        """Starts an epoch (run_nerf.py:697-708).  `perm` lets a caller supply the permutation."""
        if perm is None:
            perm = torch.randperm(self.n_rays, device=self.device, generator=self.gen)
        self.perm = torch.as_tensor(perm, device=self.device).to(torch.int64)
        if self.perm.numel() != self.n_rays:
            raise ValueError("RayBatcher.shuffle: permutation has the wrong length")
        self.i_batch = 0

    def _pooled(self):
        if self.perm is None or self.i_batch >= self.n_rays:
            self.shuffle()
        ids = self._mine(self.perm[self.i_batch:self.i_batch + self.N_rand])
        self.i_batch += self.N_rand
        hw = self.H * self.W
        return ops.gather_pixel_rays(self.H, self.W, self.focal, self.poses, self.images,
                                     torch.div(ids, hw, rounding_mode='floor'), ids % hw)

    def _mine(self, picks):
        """This rank's contiguous slice of a global batch of picks."""
        if self.world_size == 1:
            return picks
        from .parallel import shard_bounds
        lo, hi = shard_bounds(int(picks.shape[0]), self.rank, self.world_size)
        return picks[lo:hi]

    # -- one image per step ------------------------------------------------------------------
    def _single(self, i, image=None, select=None):
        if image is None:
            image = int(torch.randint(self.n_img, (1,), generator=self.host_gen))
        window = precrop_window(self.H, self.W, self.precrop_frac) if i < self.precrop_iters else None
        pix = self._window_ids.get(window)
        if pix is None:  # two windows at most: the central crop and the whole image
            pix = self._window_ids[window] = torch.as_tensor(window_pixels(self.H, self.W, window), device=self.device)
        if select is None:
            select = torch.randperm(pix.numel(), device=self.device, generator=self.gen)[:self.N_rand]
        pix = pix[self._mine(torch.as_tensor(select, device=self.device).to(torch.int64))]
        return ops.gather_pixel_rays(self.H, self.W, self.focal, self.poses, self.images,
                                     torch.full_like(pix, image), pix)

    def sample(self, i, image=None, select=None):
        """Batch of step i.  `image` (slot in i_train) and `select` (indices into the window's
        pixel list) pin the random choices of the single-image mode."""
        if self.use_batching:
            return self._pooled()
        return self._single(int(i), image, select)
