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
_DP_STEP = None   # set by train_step around tape.gradient: the data-parallel exchange the backward nodes start
_TAIL = 2         # floats behind a network's flat gradient: [n_local * local mean loss, n_local]


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
        # how many times this network is queried under the tape (the coarse network also does the
        # fine pass when network_fine is None)
        if getattr(model, '_node_epoch', -1) != _TAPE_EPOCH:
            model._node_epoch, model._nodes = _TAPE_EPOCH, 0
        model._nodes += 1
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
        # The network's whole gradient is ONE flat buffer (the 24 tensors are views, two more
        # floats behind them carry the step's loss and ray count under data parallelism): Adam
        # reads it with one launch and data-parallel ranks exchange it with one collective.  The
        # buffer is kept and reused from step to step, so gradients returned by a tape are valid
        # until the next tape's backward pass; a network queried twice under one tape (coarse
        # network doing the fine pass too) gets fresh buffers, which autograd adds.
        once = getattr(model, '_nodes', 1) == 1
        n_par = model.flat.numel()
        if once and getattr(model, '_grad_flat', None) is not None and model._grad_flat.device == dev:
            flat = model._grad_flat
        else:
            flat = torch.empty((n_par + _TAIL,), dtype=torch.float32, device=dev)
            if once:
                model._grad_flat = flat
        grads, at = [], 0
        for p in model.params:
            grads.append(flat[at:at + p.numel()].view(p.shape))
            at += p.numel()
        # (under graph capture the host cannot look at a status block: GraphedTrainStep does both
        # calls around its replays instead)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            model.review_grad_status()
        gscale = model.grad_scale if model.grad_scale is not None else -float(model.auto_target)
        ops.mlp_backward(model.packed_bwd(), ctx.act, d_raw, gscale, gbuf, partial, grads)
        if not capturing:
            model.post_grad_status()
        ctx.act = None
        model._grad_work = None
        if once and _DP_STEP is not None:
            # data parallel: start this network's exchange now, so that the fine network's
            # all-reduce runs beside the coarse network's backward pass (autograd reaches the fine
            # network first); train_step waits for the handles before the optimiser step.  Only a
            # train_step asks for this (a tape used on its own exchanges nothing), and only for a
            # network with a single node under the tape: the gradients of a network queried twice
            # are summed by autograd afterwards and exchanged once, by allreduce_gradients.
            flat[n_par:].copy_(_DP_STEP.tail)
            model._grad_work = parallel.allreduce_sum_(flat, async_op=True)
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


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, rgb0, target, grad_weight):
        ctx.set_materialize_grads(False)
        rgb = rgb.contiguous()
        target = nv.f32(target, rgb.device)
        if rgb.shape != target.shape or (rgb0 is not None and rgb0.shape != target.shape):
            raise ValueError(f"photometric_loss: shapes {tuple(rgb.shape)}, "
                             f"{None if rgb0 is None else tuple(rgb0.shape)} and {tuple(target.shape)} differ")
        if rgb.numel() == 0:
            raise ValueError("photometric_loss: empty batch")
        rgb0 = None if rgb0 is None else rgb0.contiguous()
        out = torch.empty((8,), dtype=torch.float32, device=rgb.device)
        d_rgb = torch.empty_like(rgb)
        d_rgb0 = None if rgb0 is None else torch.empty_like(rgb0)
        with torch.cuda.device(rgb.device):
            nv.check(nv.lib().nerf_photometric_loss(nv.ptr(rgb), nv.ptr(rgb0), nv.ptr(target), rgb.numel(),
                                                    float(grad_weight), nv.ptr(out), nv.ptr(d_rgb), nv.ptr(d_rgb0),
                                                    nv.stream()), "nerf_photometric_loss")
        ctx.save_for_backward(d_rgb, d_rgb0)
        loss, stats = out[0], out[1:5]
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g, _g_stats):
        d_rgb, d_rgb0 = ctx.saved_tensors
        if g is None:
            return None, None, None, None
        return d_rgb * g, (None if d_rgb0 is None else d_rgb0 * g), None, None


def photometric_loss(rgb, rgb0, target, grad_weight=1.0):
    """The loss of the plain photometric configuration with everything the step reports, one kernel
    launch (run_nerf.py:781, :797, :809-811): returns (loss, img_loss, img_loss0, psnr, psnr0) as
    0-d tensors, loss = img2mse(rgb, target) + img2mse(rgb0, target) being the differentiable one
    (rgb0 None: no coarse term; img_loss0 and psnr0 are then None).  The gradient that flows back
    from `loss` is grad_weight times the true one (data-parallel ranks pass their ray count, see
    train_step)."""
    rgb = nv.f32(rgb)
    rgb0 = None if rgb0 is None else nv.f32(rgb0, rgb.device)
    loss, stats = _Photometric.apply(rgb, rgb0, target, grad_weight)
    if rgb0 is None:
        return loss, stats[0], None, stats[2], None
    return loss, stats[0], stats[1], stats[2], stats[3]


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
                gflat = _tiled_flat(gs)
                if gflat is None or gflat.numel() != grp['n']:
                    gflat = torch.cat([g.reshape(-1) for g in gs])
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


def _tiled_flat(tensors):
    """The 1-d fp32 tensor that `tensors` tile back to back (views of one contiguous buffer, in
    order, no gaps), or None.  The buffer may be longer than what they cover."""
    t0 = tensors[0]
    base = t0._base if t0._base is not None else t0
    if base.dim() != 1 or not base.is_contiguous() or base.dtype != torch.float32:
        return None
    at = t0.data_ptr()
    for t in tensors:
        tb = t._base if t._base is not None else t
        if tb.data_ptr() != base.data_ptr() or tb.numel() != base.numel() or t.data_ptr() != at or not t.is_contiguous():
            return None
        at += t.numel() * 4
    off = (t0.data_ptr() - base.data_ptr()) // 4
    n = (at - t0.data_ptr()) // 4
    if off < 0 or off + n > base.numel():
        return None
    return base[off:off + n]


# ---- data parallel ---------------------------------------------------------------------------
# Ray-sharded data parallelism, two collectives per step (one per network) and nothing else: every
# rank back-propagates n_local x its local MEAN loss, the flat gradient buffers are summed over the
# ranks together with two trailing floats, [n_local x loss, n_local], and the sums are divided by
# the summed ray count, which gives the gradient and the value of the GLOBAL mean loss whatever the
# shard sizes.  (dp_loss_weight + allreduce_gradients(grads) remain for callers that weight the
# loss themselves: one more small collective for the count.)
class _DPStep:
    def __init__(self, n_local, loss_local, device):
        self.n_local = float(n_local)
        # [n_local * local mean loss, n_local] on the device, written without a host round trip
        self.tail = torch.stack([loss_local.detach().to(torch.float32) * self.n_local,
                                 torch.full((), self.n_local, dtype=torch.float32, device=device)])


def dp_loss_weight(n_local, device=None):
    """n_local / n_global: what a rank multiplies its local MEAN loss by.  A python float on one
    rank; a 0-d fp32 tensor on `device` otherwise (formed from an all-reduced count, never read
    back to the host)."""
    if not parallel.exchanging():
        return 1.0
    n_global = parallel.global_count(n_local, device)
    return (float(n_local) / n_global.clamp_min(1.)).to(torch.float32)


def allreduce_gradients(grads, step=None):
    """SUM over the ranks of a list of gradient tensors, with ONE collective per flat buffer they
    tile (a network's 24 gradients are views of one 2.38 MB buffer: no copy, no per-tensor call);
    tensors that do not tile a buffer of their own are packed into one bucket first.

    step=None: a plain sum (the caller weighted its loss by dp_loss_weight); returns the list.
    step=_DPStep (what train_step passes): the buffers were or are exchanged together with the
    step's [n_local x loss, n_local] tail, exchanges already started by the backward nodes are only
    waited for, every sum is divided by the summed ray count; returns (list, global mean loss)."""
    if not parallel.exchanging():
        return grads if step is None else (grads, None)
    from .run_nerf_helpers import NerfModel
    done = {}
    for m in NerfModel.live():
        if getattr(m, '_grad_work', None) is not None:
            m._grad_work.wait()
            m._grad_work = None
            done[m._grad_flat.data_ptr()] = m._grad_flat
    groups, loose = {}, []
    for i, g in enumerate(grads):
        base = g._base if g._base is not None else g
        if base.dim() == 1 and base.is_contiguous() and base.dtype == torch.float32:
            groups.setdefault(base.data_ptr(), [base, []])[1].append(i)
        else:
            loose.append(i)
    out = list(grads)
    tails = []   # summed [n x loss, n] of this step, one per exchange made with a tail
    for ptr, (base, idx) in groups.items():
        head = _tiled_flat([grads[i] for i in idx])
        if ptr in done:
            tails.append((base[-_TAIL:], base[:-_TAIL]))      # exchanged inside the backward node, tail included
        elif step is None and head is not None:
            parallel.allreduce_sum_(head)                     # in place, whatever lies behind the gradients
        else:
            loose += idx
    if loose:
        parts = [grads[i].reshape(-1) for i in loose] + ([step.tail] if step is not None else [])
        flat = torch.cat(parts)
        parallel.allreduce_sum_(flat)
        at = 0
        for i in loose:
            out[i] = flat[at:at + grads[i].numel()].view(grads[i].shape)
            at += grads[i].numel()
        if step is not None:
            tails.append((flat[at:], flat[:at]))
    if step is None:
        return out
    with torch.no_grad():
        for tail, head in tails:
            head.div_(tail[1])
    tail = tails[0][0]
    return out, (tail[0] / tail[1])


def train_step(render_fn, batch_rays, target_s, grad_vars, optimizer, models=()):
    """One optimisation step as the reference performs it (run_nerf.py:770-826, the plain
    photometric configuration): returns (loss, psnr, psnr0) as 0-d tensors.
    render_fn(rays) must call render(..., rays=rays, **render_kwargs_train).

    Under torch.distributed every rank passes ITS (non-empty) shard of the batch: the rank
    back-propagates n_local x its local mean loss, each network's gradient buffer is summed over
    the ranks with [n_local x loss, n_local] behind it and divided by the summed ray count, and
    the returned loss is the global mean (psnr / psnr0 are those of the local shard)."""
    global _DP_STEP
    alone = not parallel.exchanging()
    n_local = int(target_s.shape[0]) if hasattr(target_s, 'shape') else len(target_s)
    with GradientTape() as tape:
        rgb, disp, acc, extras = render_fn(batch_rays)
        # img_loss, psnr, img_loss0, psnr0 and loss = img_loss + img_loss0 in one launch; a
        # data-parallel rank's gradient carries its ray count (the sums over the ranks are divided
        # by the summed count)
        loss, _, _, psnr, psnr0 = photometric_loss(rgb, extras.get('rgb0'), target_s,
                                                   grad_weight=1.0 if alone else float(n_local))
    if alone:
        gradients = tape.gradient(loss, grad_vars)
    else:
        step = _DPStep(n_local, loss, loss.device)
        _DP_STEP = step
        try:
            gradients = tape.gradient(loss, grad_vars)
        finally:
            _DP_STEP = None
        gradients, loss = allreduce_gradients(gradients, step)
    optimizer.apply_gradients(zip(gradients, grad_vars))
    return loss.detach(), psnr, psnr0


class GraphedTrainStep:
    """train_step for one batch size with the forward and backward passes of both networks captured
    into ONE HIP graph and replayed.  A step is some fifty kernel launches issued through Python,
    about 1.2 ms of host time whatever the batch: batches of a thousand rays or fewer finish on the
    GPU before the host has enqueued them, and the captured step takes the host out of it.  (At the
    reference's default of 4096 rays the GPU is the longer side either way.)

        step = GraphedTrainStep(render_fn, N_rand, grad_vars, optimizer, [model, model_fine])
        loss, psnr, psnr0 = step(batch_rays, target_s)       # as train_step returns them

    render_fn(rays) must call render(..., rays=rays, **render_kwargs_train); batch_rays is the
    (rays_o, rays_d) pair of [N_rand, 3] tensors.  The batch is copied into the graph's own input
    buffers; the returned tensors are the graph's outputs and hold their values until the next call.
    The optimiser step runs outside the graph (its learning rate changes every step and is a
    kernel argument).  The automatic fp16 gradient scale keeps looking after itself: the status
    blocks are read between replays, and the step is captured anew when a network's target moved.
    Single process only (no gradient exchange inside the graph)."""

    def __init__(self, render_fn, n_rays, grad_vars, optimizer, models, device=None):
        if parallel.exchanging():
            raise NotImplementedError("GraphedTrainStep: single process only; use train_step under torch.distributed")
        self.render_fn, self.grad_vars, self.optimizer = render_fn, list(grad_vars), optimizer
        self.models = list(models)
        if not self.models:
            raise ValueError("GraphedTrainStep: pass the networks the step trains")
        dev = torch.device(device) if device is not None else self.models[0].device
        self.device = dev
        new = lambda: torch.zeros((int(n_rays), 3), dtype=torch.float32, device=dev)
        self.rays, self.target = (new(), new()), new()
        self.captures = 0
        self.graph = self.out = self._targets = None

    def _fwd_bwd(self):
        with GradientTape() as tape:
            rgb, disp, acc, extras = self.render_fn(self.rays)
            loss, _, _, psnr, psnr0 = photometric_loss(rgb, extras.get('rgb0'), self.target)
        return loss, psnr, psnr0, tape.gradient(loss, self.grad_vars)

    def _capture(self):
        # (the parameters' gradient-accumulation nodes were made on whatever stream first used the
        # parameters; tape.gradient never accumulates into them, so torch's warning about their
        # stream does not apply here)
        warn = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
        if warn is not None:
            warn(False)
        try:
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):   # un-captured passes first: the nodes' reusable buffers, lazy set-up
                for _ in range(2):
                    self._fwd_bwd()
            cur.wait_stream(side)
            for m in self.models:           # the graph packs the weights itself, at every replay
                m.mark_dirty()
            self._targets = [(m.grad_scale, m.auto_target) for m in self.models]
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._fwd_bwd()
        finally:
            if warn is not None:
                warn(True)
        self.captures += 1

    def __call__(self, batch_rays, target_s):
        rays_o, rays_d = batch_rays
        n = self.target.shape[0]
        if nv.f32(target_s, self.device).reshape(-1, 3).shape[0] != n:
            raise ValueError(f"GraphedTrainStep: captured for batches of {n} rays")
        with torch.no_grad():
            self.rays[0].copy_(nv.f32(rays_o, self.device).reshape(-1, 3))
            self.rays[1].copy_(nv.f32(rays_d, self.device).reshape(-1, 3))
            self.target.copy_(nv.f32(target_s, self.device).reshape(-1, 3))
        for m in self.models:
            m.review_grad_status()
        if self.graph is None or self._targets != [(m.grad_scale, m.auto_target) for m in self.models]:
            self._capture()
        self.graph.replay()
        for m in self.models:
            m.post_grad_status()
        loss, psnr, psnr0, grads = self.out
        self.optimizer.apply_gradients(zip(grads, self.grad_vars))
        return loss.detach(), psnr, psnr0
