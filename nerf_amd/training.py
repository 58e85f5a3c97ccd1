"""The training step of the reference (run_nerf.py:601-605 optimiser, :770-826 step) on the HIP
kernels: a tf.GradientTape-shaped recorder, the Keras Adam optimiser with its exponential decay
schedule, the photometric loss, and the ray-sharded data-parallel gradient exchange.

    optimizer = Adam(ExponentialDecay(lrate, lrate_decay * 1000, 0.1))       # run_nerf.py:601-605
    with GradientTape() as tape:                                             # :770
        rgb, disp, acc, extras = render(H, W, focal, chunk=..., rays=batch_rays,
                                        retraw=True, **render_kwargs_train)  # :773-775
        img_loss = img2mse(rgb, target_s)                                    # :781
        loss = img_loss + img2mse(extras['rgb0'], target_s)                  # :809-811
    gradients = tape.gradient(loss, grad_vars)                               # :825
    optimizer.apply_gradients(zip(gradients, grad_vars))                     # :826

Inside the tape render_rays runs the training variants of the kernels (the MLP forward also
saves its fp16 activations) as torch.autograd.Function nodes, so tape.gradient is one
torch.autograd.grad call; outside it everything takes the lean inference path.
"""
import math

import numpy as np

import torch

from . import _native as nv
from . import ops
from . import parallel

_TAPE_DEPTH = 0
_TAPE_EPOCH = 0   # bumped when a tape opens: a network's reusable gradient buffer is handed out once per tape


def recording():
    """True while a GradientTape is open (render_rays then records for the backward pass)."""
    return _TAPE_DEPTH > 0 and torch.is_grad_enabled()


class GradientTape:
    """Context manager with tf.GradientTape's two-call surface (run_nerf.py:770, :825)."""

    def __enter__(self):
        global _TAPE_DEPTH, _TAPE_EPOCH
        _TAPE_DEPTH += 1
        _TAPE_EPOCH += 1
        self._grad_mode = torch.enable_grad()
        self._grad_mode.__enter__()
        return self

    def __exit__(self, *exc):
        global _TAPE_DEPTH
        self._grad_mode.__exit__(*exc)
        _TAPE_DEPTH -= 1
        return False

    def gradient(self, loss, variables):
        variables = list(variables)
        grads = torch.autograd.grad(loss, variables, allow_unused=True)
        return [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, variables)]


# ---- autograd nodes over the kernels --------------------------------------------------------
class MLPQuery(torch.autograd.Function):
    """raw = network(rays, z); backward yields the 24 parameter gradients (none for rays / z:
    the reference stops the gradient at the resampled depths and never differentiates rays)."""

    @staticmethod
    def forward(ctx, rays, z, model, *params):
        ctx.set_materialize_grads(False)
        n, S = z.shape
        act = torch.empty((ops.act_bytes(n * S),), dtype=torch.uint8, device=rays.device)
        raw = ops.mlp_forward_train(model.packed(), rays, z, act)
        ctx.model, ctx.act = model, act
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        model = ctx.model
        if d_raw is None:  # the network output did not reach the loss
            ctx.act = None
            return (None, None, None) + (None,) * len(model.params)
        d_raw = d_raw.contiguous()
        n_points = d_raw.numel() // 4
        dev = d_raw.device
        gbuf = torch.empty((ops.grad_bytes(n_points),), dtype=torch.uint8, device=dev)
        partial = model._grad_partial
        if partial is None or partial.device != dev or partial.numel() < ops.partial_bytes():
            partial = model._grad_partial = torch.empty((ops.partial_bytes(),), dtype=torch.uint8, device=dev)
        # The network's whole gradient is ONE flat buffer (the 24 tensors are views): Adam reads it
        # with one launch and data-parallel ranks exchange it with one collective.  The buffer is
        # kept and reused from step to step; a network queried twice under one tape (coarse
        # network doing the fine pass too) gets a fresh one the second time, autograd adds them.
        first_use = getattr(model, '_grad_epoch', -1) != _TAPE_EPOCH
        if first_use and getattr(model, '_grad_flat', None) is not None and model._grad_flat.device == dev:
            flat = model._grad_flat
        else:
            flat = torch.empty_like(model.flat)
            if first_use:
                model._grad_flat = flat
        grads, at = [], 0
        for p in model.params:
            grads.append(flat[at:at + p.numel()].view(p.shape))
            at += p.numel()
        ops.mlp_backward(model.packed_bwd(), ctx.act, d_raw, model.grad_scale, gbuf, partial, grads)
        ctx.act = None
        if first_use:
            model._grad_epoch = _TAPE_EPOCH
            # data parallel: start this network's all-reduce now, so that the fine network's
            # exchange runs beside the coarse network's backward pass (autograd reaches the fine
            # network first); train_step waits for the handles before the optimiser step
            model._grad_work = parallel.allreduce_sum_(flat, async_op=True) if DP_OVERLAP else None
        return (None, None, None) + tuple(grads)


class Composite(torch.autograd.Function):
    """raw2outputs with its backward kernel; weights are returned for the (non-differentiated)
    resampling step only."""

    @staticmethod
    def forward(ctx, raw, z, rays, white_bkgd, noise, noise_std):
        ctx.set_materialize_grads(False)  # maps that do not reach the loss cost no kernels
        width = rays.shape[1]
        rgb, disp, acc, weights, depth = ops.composite_fwd(raw, z, rays[:, 3:6], width, white_bkgd,
                                                           noise, noise_std)
        ctx.save_for_backward(raw, z, rays, depth, acc)
        ctx.white_bkgd, ctx.noise, ctx.noise_std = white_bkgd, noise, noise_std
        ctx.mark_non_differentiable(weights)
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_weights, g_depth):
        raw, z, rays, depth, acc = ctx.saved_tensors
        if g_rgb is None and g_disp is None and g_acc is None and g_depth is None:
            return None, None, None, None, None, None
        if g_disp is not None:
            # disp = 1/max(1e-10, q), q = depth/acc: chain into depth and acc
            q = depth / acc
            dq = torch.where(q > 1e-10, -1. / (q * q), torch.zeros_like(q)) * g_disp
            dq = torch.nan_to_num(dq, nan=0.0, posinf=0.0, neginf=0.0)
            g_depth = dq / acc if g_depth is None else g_depth + dq / acc
            g_acc_d = torch.nan_to_num(-dq * depth / (acc * acc), nan=0.0, posinf=0.0, neginf=0.0)
            g_acc = g_acc_d if g_acc is None else g_acc + g_acc_d
        c = lambda t: None if t is None else t.contiguous()
        d_raw = ops.composite_bwd(raw, z, rays[:, 3:6], rays.shape[1], ctx.white_bkgd, ctx.noise,
                                  ctx.noise_std, c(g_rgb), c(g_acc), c(g_depth))
        return d_raw, None, None, None, None, None


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        y = nv.f32(y, x.device)
        if x.shape != y.shape:
            raise ValueError(f"img2mse: shapes {tuple(x.shape)} and {tuple(y.shape)} differ")
        loss = torch.zeros((), dtype=torch.float32, device=x.device)
        d_x = torch.empty_like(x)
        with torch.cuda.device(x.device):
            nv.check(nv.lib().nerf_mse_loss_fwd_bwd(nv.ptr(x), nv.ptr(y), x.numel(), 1.0 / x.numel(), 1.0,
                                                    nv.ptr(loss), nv.ptr(d_x), nv.stream()),
                     "nerf_mse_loss_fwd_bwd")
        ctx.save_for_backward(d_x)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_x,) = ctx.saved_tensors
        return (None if g is None else d_x * g), None


def mse(x, y):
    """img2mse (run_nerf_helpers.py:11) with the gradient produced in the same kernel pass."""
    return _MSE.apply(nv.f32(x), y)


# ---- optimiser (run_nerf.py:601-605, :826) ----------------------------------------------------
class ExponentialDecay:
    """tf.keras.optimizers.schedules.ExponentialDecay without staircase:
    lr(it) = initial * decay_rate ** (it / decay_steps)  (run_nerf.py:603-604)."""

    def __init__(self, initial_learning_rate, decay_steps, decay_rate):
        self.initial_learning_rate = float(initial_learning_rate)
        self.decay_steps = float(decay_steps)
        self.decay_rate = float(decay_rate)

    def __call__(self, step):
        return self.initial_learning_rate * self.decay_rate ** (step / self.decay_steps)


class Adam:
    """tf.keras.optimizers.Adam as TF 1.15 configures it by default: beta_1 0.9, beta_2 0.999,
    epsilon 1e-7 (outside the bias correction), no amsgrad.  `iterations` starts at 0 and is what
    the learning-rate schedule sees, exactly as in Keras (and, as in the reference, it restarts
    from 0 on resume because the optimiser state is never reloaded, run_nerf.py:244-261).

    Variables that are views tiling one flat buffer (the 24 tensors of a NerfModel) are updated
    by one fused kernel launch per buffer."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.iterations = 0
        self._groups = None
        self.owners = []  # NerfModels to notify after an update (set by bind_models)

    def bind_models(self, models):
        """Lets the optimiser tell the models that their parameters changed (re-pack on next use)."""
        self.owners = [m for m in models if hasattr(m, 'mark_dirty')]
        return self

    def get_weights(self):
        """Keras' optimizer.get_weights(): [iterations, m of every variable..., v of every variable...]
        (one flat array per parameter buffer here), so that an optimiser stored in the models
        dictionary can be checkpointed like a network."""
        out = [np.asarray(self.iterations, dtype=np.int64)]
        for key in ('m', 'v'):
            out += [g[key].detach().cpu().numpy() for g in (self._groups or [])]
        return out

    def _lr(self):
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def _build(self, variables):
        groups = {}
        for i, v in enumerate(variables):
            base = v._base if v._base is not None else v
            groups.setdefault(base.data_ptr(), [base, []])[1].append(i)
        from .run_nerf_helpers import _MODELS_BY_STORAGE
        self._groups = []
        for base, idx in groups.values():
            owner = _MODELS_BY_STORAGE.get(base.data_ptr())
            if owner is not None and owner not in self.owners:
                self.owners.append(owner)
            n = sum(variables[i].numel() for i in idx)
            at, tiled = base.data_ptr(), n == base.numel()
            for i in idx:
                tiled = tiled and variables[i].data_ptr() == at and variables[i].is_contiguous()
                at += variables[i].numel() * 4
            theta = base.view(-1) if tiled else None
            self._groups.append(dict(idx=idx, theta=theta, n=n,
                                     m=torch.zeros((n,), dtype=torch.float32, device=base.device),
                                     v=torch.zeros((n,), dtype=torch.float32, device=base.device)))

    def apply_gradients(self, grads_and_vars):
        grads_and_vars = list(grads_and_vars)
        grads = [g for g, _ in grads_and_vars]
        variables = [v for _, v in grads_and_vars]
        if self._groups is None:
            self._build(variables)
        t = self.iterations + 1
        lr_t = self._lr() * math.sqrt(1. - self.beta_2 ** t) / (1. - self.beta_1 ** t)
        with torch.no_grad():
            for grp in self._groups:
                gs = [grads[i] for i in grp['idx']]
                # gradients that already tile one flat buffer are used in place
                g0 = gs[0]
                gbase = g0._base if g0._base is not None else g0
                at, tiled = gbase.data_ptr(), gbase.numel() == grp['n']
                for g in gs:
                    tiled = tiled and g.data_ptr() == at and g.is_contiguous()
                    at += g.numel() * 4
                gflat = gbase.view(-1) if tiled else torch.cat([g.reshape(-1) for g in gs])
                if grp['theta'] is not None:
                    theta = grp['theta']
                else:
                    theta = torch.cat([variables[i].detach().reshape(-1) for i in grp['idx']])
                with torch.cuda.device(theta.device):
                    nv.check(nv.lib().nerf_adam_step(nv.ptr(theta), nv.ptr(gflat.contiguous()), nv.ptr(grp['m']),
                                                     nv.ptr(grp['v']), grp['n'], lr_t, self.beta_1,
                                                     self.beta_2, self.epsilon, nv.stream()), "nerf_adam_step")
                if grp['theta'] is None:  # scattered variables: copy the result back
                    at = 0
                    for i in grp['idx']:
                        k = variables[i].numel()
                        variables[i].detach().copy_(theta[at:at + k].view(variables[i].shape))
                        at += k
        for m in self.owners:
            m.mark_dirty()
        self.iterations += 1


# ---- data parallel ---------------------------------------------------------------------------
# Ray-sharded data parallelism.  Every rank divides its loss by the GLOBAL ray count
# (dp_loss_weight), so the SUM of the ranks' gradients is the gradient of the global mean loss
# whatever the shard sizes; nothing is divided afterwards.
DP_OVERLAP = True   # start a network's all-reduce inside its backward node (see MLPQuery.backward)


def dp_loss_weight(n_local, device=None):
    """n_local / n_global: what a rank multiplies its local MEAN loss by.  A python float on one
    rank; a 0-d fp32 tensor on `device` otherwise (formed from an all-reduced count, never read
    back to the host)."""
    _, world_size = parallel.world()
    if world_size == 1:
        return 1.0
    n_global = parallel.global_count(n_local, device)
    return (float(n_local) / n_global.clamp_min(1.)).to(torch.float32)


def allreduce_gradients(grads):
    """SUM over the ranks of a list of gradient tensors, with ONE collective per flat buffer they
    tile (a network's 24 gradients are views of one 2.38 MB buffer: no copy, no per-tensor call);
    tensors that are not views of a common buffer are packed into one bucket first.  Exchanges
    already started by the backward nodes (DP_OVERLAP) are only waited for.  Returns the list."""
    rank, world_size = parallel.world()
    if world_size == 1:
        return grads
    from .run_nerf_helpers import NerfModel
    pending = [m for m in NerfModel.live() if getattr(m, '_grad_work', None) is not None]
    done = set()
    for m in pending:
        m._grad_work.wait()
        m._grad_work = None
        done.add(m._grad_flat.data_ptr())
    groups, loose = {}, []
    for i, g in enumerate(grads):
        base = g._base if g._base is not None else g
        if base.dim() == 1 and base.is_contiguous() and base.dtype == torch.float32:
            groups.setdefault(base.data_ptr(), [base, []])[1].append(i)
        else:
            loose.append(i)
    out = list(grads)
    for ptr, (base, idx) in groups.items():
        covered = sum(grads[i].numel() for i in idx) == base.numel()
        if not covered:
            loose += idx
        elif ptr not in done:
            parallel.allreduce_sum_(base)
    if loose:
        flat = torch.cat([grads[i].reshape(-1) for i in loose])
        parallel.allreduce_sum_(flat)
        at = 0
        for i in loose:
            out[i] = flat[at:at + grads[i].numel()].view(grads[i].shape)
            at += grads[i].numel()
    return out


def train_step(render_fn, batch_rays, target_s, grad_vars, optimizer, models=()):
    """One optimisation step as the reference performs it (run_nerf.py:770-826, the plain
    photometric configuration): returns (loss, psnr, psnr0) as 0-d tensors.
    render_fn(rays) must call render(..., rays=rays, **render_kwargs_train).

    Under torch.distributed every rank passes ITS shard of the batch; the losses are weighted by
    n_local / n_global, the gradients summed over the ranks, and the returned loss is the global
    mean (psnr / psnr0 are those of the local shard)."""
    from .run_nerf_helpers import img2mse, mse2psnr
    _, world_size = parallel.world()
    weight = 1.0
    if world_size > 1:
        n_local = int(target_s.shape[0]) if hasattr(target_s, 'shape') else len(target_s)
        weight = dp_loss_weight(n_local, batch_rays[0].device if isinstance(batch_rays[0], torch.Tensor) else None)
    with GradientTape() as tape:
        rgb, disp, acc, extras = render_fn(batch_rays)
        img_loss = img2mse(rgb, target_s)
        loss = img_loss
        psnr = mse2psnr(img_loss.detach())
        psnr0 = None
        if 'rgb0' in extras:
            img_loss0 = img2mse(extras['rgb0'], target_s)
            loss = loss + img_loss0
            psnr0 = mse2psnr(img_loss0.detach())
        if world_size > 1:
            loss = loss * weight
    gradients = tape.gradient(loss, grad_vars)
    gradients = allreduce_gradients(gradients)
    optimizer.apply_gradients(zip(gradients, grad_vars))
    loss = loss.detach()
    if world_size > 1:
        loss = loss.clone()
        parallel.allreduce_sum_(loss)
    return loss, psnr, psnr0
