"""Drop-in counterparts of the reference's run_nerf_helpers.py on torch (ROCm) tensors.

Same names, argument meaning and return structure as the reference (file:line cited per
function), so `from nerf_amd.run_nerf_helpers import *` can replace the reference's star import
(run_nerf.py:7).  All arithmetic runs in libnerfhip.so on the GPU.
"""
import weakref

import numpy as np
import torch

from . import _native as nv
from . import ops

__all__ = ["img2mse", "mse2psnr", "to8b", "Embedder", "get_embedder", "NerfModel", "init_nerf_model",
           "get_rays", "get_rays_np", "ndc_rays", "sample_pdf", "batchify"]


# ---- misc (run_nerf_helpers.py:11-17) ----------------------------------------------------------
def img2mse(x, y):
    """run_nerf_helpers.py:11 -- mean squared error over every element (differentiable in x)."""
    from .training import mse
    return mse(x, y)


def mse2psnr(x):
    """run_nerf_helpers.py:14"""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x, dtype=torch.float32)
    return torch.log(x) * float(-10. / np.log(10.))


def to8b(x):
    """run_nerf_helpers.py:17 (NumPy in, uint8 out, as in the reference)."""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


# ---- positional encoding (run_nerf_helpers.py:22-75) -------------------------------------------
class Embedder:
    """Same constructor keywords as the reference (:24-53).  Only the configuration the reference
    itself ever builds is accepted: 3 input dims, include_input, log-sampled 2^k frequencies and
    (sin, cos) -- that is what the kernels encode."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs.get('input_dims', 3)
        n = kwargs['num_freqs']
        if d != 3 or not kwargs.get('include_input', True) or not kwargs.get('log_sampling', True) \
                or kwargs.get('max_freq_log2', n - 1) != n - 1:
            raise NotImplementedError("Embedder: only include_input=True, input_dims=3, "
                                      "log_sampling=True, max_freq_log2=num_freqs-1 is built")
        self.num_freqs = n
        self.out_dim = d + 2 * d * n

    def embed(self, inputs):
        return ops.embed(inputs, self.num_freqs)


def get_embedder(multires, i=0):
    """run_nerf_helpers.py:59-75 -> (embed_fn, out_dim)."""
    if i == -1:
        return (lambda x: x), 3
    eo = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=['sin', 'cos'])

    def embed(x, eo=eo):
        return eo.embed(x)
    embed.multires = multires
    return embed, eo.out_dim


# ---- model (run_nerf_helpers.py:80-117) --------------------------------------------------------
# flat-buffer address -> model, so that an optimiser that updates the buffer with a raw kernel can
# tell the owner to re-pack its weights
_MODELS_BY_STORAGE = weakref.WeakValueDictionary()


class NerfModel:
    """What init_nerf_model returns: the parameters of one network with the slice of the Keras
    Model interface create_nerf and train() use (get_weights / set_weights / trainable_variables,
    run_nerf.py:181,:253,:841), plus the packed fp16 stream the MFMA kernels read.

    Parameter order (get_weights()): kernel [in,out], bias [out] of the D point layers, then
    bottleneck, view layer, rgb, alpha."""

    def __init__(self, D, W, input_ch, input_ch_views, output_ch, skips, use_viewdirs, device=None,
                 generator=None):
        if not (D == 8 and W == 256 and input_ch == 63 and input_ch_views == 27
                and tuple(skips) == (4,) and use_viewdirs and output_ch == 4):
            raise NotImplementedError(
                "the MI355X MLP kernels are built for the reference's configured network only: "
                "D=8, W=256, input_ch=63 (multires=10), input_ch_views=27 (multires_views=4), "
                f"skips=[4], use_viewdirs=True; got D={D} W={W} input_ch={input_ch} "
                f"input_ch_views={input_ch_views} skips={list(skips)} use_viewdirs={use_viewdirs}")
        self.D, self.W, self.input_ch, self.input_ch_views = D, W, input_ch, input_ch_views
        self.skips, self.use_viewdirs = tuple(skips), use_viewdirs
        dev = ops._dev() if device is None else torch.device(device)
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        self.device = dev
        self.shapes = []
        k = input_ch
        for i in range(D):
            self.shapes.append((k, W))
            k = W + input_ch if i in skips else W
        self.shapes += [(k, 256), (256 + input_ch_views, W // 2), (W // 2, 3), (k, 1)]
        # one flat fp32 buffer holds every parameter (the fused Adam kernel updates it in one
        # launch); the 24 tensors handed out are views into it
        sizes = []
        for (fi, fo) in self.shapes:
            sizes += [fi * fo, fo]
        self.flat = torch.zeros((sum(sizes),), dtype=torch.float32, device=dev)
        self.params = []
        at = 0
        for k, (fi, fo) in enumerate(self.shapes):  # Keras Dense defaults: glorot-uniform kernel, zero bias (:83)
            lim = float(np.sqrt(6. / (fi + fo)))
            w = self.flat[at:at + fi * fo].view(fi, fo)
            w.copy_((torch.rand((fi, fo), dtype=torch.float32, device=dev, generator=generator) * 2 - 1) * lim)
            at += fi * fo
            bvec = self.flat[at:at + fo]
            at += fo
            self.params.append(w.requires_grad_(True))
            self.params.append(bvec.requires_grad_(True))
        self._packed = torch.empty((ops.packed_bytes(),), dtype=torch.uint8, device=dev) if dev.type == 'cuda' else None
        self._packed_bwd = None
        self._packed_version = None
        self._packed_bwd_version = None
        self._epoch = 0          # bumped by updates that bypass torch (the fused Adam kernel)
        # factor applied to d_raw before the fp16 gradient chain; None = formed on the device from
        # max|d_raw| at every backward pass (what callers should leave it at) so that the largest
        # scaled element lies in [auto_target, 2 auto_target).  auto_target is lowered by itself
        # when a backward pass reports saturation (review_grad_status)
        self.grad_scale = None
        self.auto_target = 256.0
        self._status_ring = None   # two pinned host copies of the status block + their events
        self._status_n = 0
        self._clean_reviews = 0
        self._grad_partial = None  # scratch of the last backward pass; its head is the status block
        self._grad_flat = None     # flat gradient buffer of the last backward pass (24 views tile it)
        self._grad_work = None     # pending all-reduce of that buffer under data parallelism
        self._grad_epoch = -1
        _MODELS_BY_STORAGE[self.flat.data_ptr()] = self

    @staticmethod
    def live():
        """The networks that currently exist in this process."""
        return list(_MODELS_BY_STORAGE.values())

    # Keras-like surface
    @property
    def trainable_variables(self):
        return list(self.params)

    def get_weights(self):
        return [p.detach().cpu().numpy() for p in self.params]

    def set_weights(self, weights):
        weights = list(weights)
        if len(weights) != len(self.params):
            raise ValueError(f"set_weights: expected {len(self.params)} arrays, got {len(weights)}")
        with torch.no_grad():
            for p, w in zip(self.params, weights):
                w = torch.as_tensor(np.asarray(w), dtype=torch.float32)
                if tuple(w.shape) != tuple(p.shape):
                    raise ValueError(f"set_weights: shape {tuple(w.shape)} does not match {tuple(p.shape)}")
                p.copy_(w)
        self.mark_dirty()

    def count_params(self):
        return sum(p.numel() for p in self.params)

    def _version(self):
        return (self._epoch,) + tuple(p._version for p in self.params)

    def packed(self):
        """The fp16 fragment stream for the current parameter values (re-packed when they change)."""
        ver = self._version()
        if ver != self._packed_version:
            with torch.no_grad():
                ops.mlp_pack([p.detach() for p in self.params], self._packed)
            self._packed_version = ver
        return self._packed

    def packed_bwd(self):
        """The transposed stream the gradient chain reads."""
        ver = self._version()
        if self._packed_bwd is None:
            self._packed_bwd = torch.empty((ops.packed_bwd_bytes(),), dtype=torch.uint8, device=self.device)
        if ver != self._packed_bwd_version:
            with torch.no_grad():
                ops.mlp_pack_bwd([p.detach() for p in self.params], self._packed_bwd)
            self._packed_bwd_version = ver
        return self._packed_bwd

    def grad_status(self):
        """(flags, gscale_used, max|d_raw|) of this network's last backward pass; flags is a
        combination of ops.GRAD_SATURATED / ops.GRAD_UNDERFLOW (0 = the gradients are sound).
        Waits for the GPU, so call it where a synchronisation is acceptable (e.g. when printing)."""
        if self._grad_partial is None:
            return 0, 0.0, 0.0
        return ops.mlp_backward_status(self._grad_partial)

    # The automatic gradient scale looks after itself, the way dynamic loss scaling does: every
    # backward pass leaves its status block in a pinned host buffer (asynchronous copy, no wait),
    # and the pass after the next one reads it.  A saturated pass produced NaN gradients, which
    # the Adam kernel skipped; the target is then lowered 16-fold for the following passes.  An
    # underflow raises it 16-fold; 1000 sound passes in a row raise it two-fold, up to 256.  The
    # two-pass delay keeps the host from ever waiting for the pass the GPU is still working on.
    def post_grad_status(self):
        if self._grad_partial is None or not self._grad_partial.is_cuda:
            return
        if self._status_ring is None:
            self._status_ring = [(torch.zeros(8, dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
        host, event = self._status_ring[self._status_n % 2]
        host.copy_(self._grad_partial[:32].view(torch.int32), non_blocking=True)
        event.record()
        self._status_n += 1

    def review_grad_status(self):
        """Adjusts auto_target from the status of the backward pass before the last one."""
        if self._status_ring is None or self._status_n < 2 or self.grad_scale is not None:
            return
        host, event = self._status_ring[self._status_n % 2]   # the older of the two
        event.synchronize()
        flags = int(host[4])
        if flags & ops.GRAD_SATURATED:
            self.auto_target = max(self.auto_target / 16.0, 2.0 ** -8)
            self._clean_reviews = 0
        elif flags & ops.GRAD_UNDERFLOW:
            self.auto_target = min(self.auto_target * 16.0, 2.0 ** 12)
            self._clean_reviews = 0
        else:
            self._clean_reviews += 1
            if self._clean_reviews >= 1000 and self.auto_target < 256.0:
                self.auto_target *= 2.0
                self._clean_reviews = 0

    def mark_dirty(self):
        """Call after writing the parameters behind torch's back (raw kernels on self.flat)."""
        self._epoch += 1


def init_nerf_model(D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4],
                    use_viewdirs=False, device=None, generator=None):
    """run_nerf_helpers.py:80-117."""
    return NerfModel(int(D), int(W), int(input_ch), int(input_ch_views), int(output_ch), list(skips),
                     bool(use_viewdirs), device=device, generator=generator)


# ---- rays (run_nerf_helpers.py:122-180) --------------------------------------------------------
def get_rays(H, W, focal, c2w):
    """run_nerf_helpers.py:122-129 -> rays_o, rays_d [H,W,3] on the GPU."""
    return ops.get_rays(H, W, focal, c2w)


def get_rays_np(H, W, focal, c2w):
    """run_nerf_helpers.py:135-142 -> NumPy arrays (computed on the GPU, copied back)."""
    o, d = ops.get_rays(H, W, focal, c2w)
    return o.cpu().numpy(), d.cpu().numpy()


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """run_nerf_helpers.py:145-180."""
    return ops.ndc_rays(H, W, focal, near, rays_o, rays_d)


# ---- hierarchical sampling (run_nerf_helpers.py:185-213) ---------------------------------------
def sample_pdf(bins, weights, N_samples, det=False, u=None):
    """run_nerf_helpers.py:185-213.  `u` lets a caller supply the uniforms the reference draws
    with tf.random.uniform (:198)."""
    return ops.sample_pdf(bins, weights, N_samples, det=det, u=u)


def batchify(fn, chunk):
    """run_nerf_helpers.py:216-223."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret
