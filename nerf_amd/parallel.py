"""Ray-sharded multi-GPU execution: one process per GPU, torch.distributed (backend "nccl" is
RCCL over xGMI on MI355X; "gloo" in the CPU tests).

Rays are independent, so the only exchanges are (a) an all-gather of the rendered maps at the end
of a sharded render (20-24 B per ray) and (b) one all-reduce(SUM) per network and training step of
the network's flat gradient buffer (595,844 floats = 2.38 MB, with the step's loss and ray count in
two more floats behind it: latency-bound over xGMI, so it travels as ONE buffer rather than 24 small
tensors, and the fine network's exchange runs beside the coarse network's backward pass).  Nothing
here touches the kernels, so the same code runs on CPU tensors under gloo.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# A process group of ONE rank has nothing to exchange, and every function here returns early for
# it.  With this set the collectives run all the same (and training.train_step takes its
# data-parallel branch): the sums and gathers are identities, but the backend, its streams, work
# handles and buffers are the real ones, which is how a machine with one GPU takes the RCCL path.
EXCHANGE_WHEN_ALONE = False


def exchanging():
    """True when the collectives have to run: a process group is up and has more than one rank
    (or EXCHANGE_WHEN_ALONE is set)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or EXCHANGE_WHEN_ALONE


def shard_bounds(n, rank, world_size):
    """Contiguous slice [lo, hi) of n rays owned by `rank`; shards differ by at most one ray."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(rays_flat, rank=None, world_size=None):
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    lo, hi = shard_bounds(rays_flat.shape[0], rank, world_size)
    return rays_flat[lo:hi]


def _staged(t):
    """The tensor a collective should be given: gloo moves device tensors through host copies of
    its own and has been seen to stall doing so when two ranks share one GPU, so under gloo device
    tensors are staged through the host explicitly (RCCL takes them as they are)."""
    return t.cpu() if (t.is_cuda and dist.get_backend() == "gloo") else t


def gather_rows(local, n_total, group=None):
    """All-gathers per-ray rows [n_local, ...] from every rank into [n_total, ...] in ray order.
    Shards may differ in length by one, so each is padded to the longest before the collective."""
    _, w = world()
    if not exchanging():
        return local
    longest = (n_total + w - 1) // w
    pad = longest - local.shape[0]
    if pad:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], 0)
    device, local = local.device, _staged(local.contiguous())
    bufs = [torch.empty_like(local) for _ in range(w)]
    dist.all_gather(bufs, local, group=group)
    bufs = [b.to(device) for b in bufs]
    parts = []
    for r, b in enumerate(bufs):
        lo, hi = shard_bounds(n_total, r, w)
        parts.append(b[:hi - lo])
    return torch.cat(parts, 0)


def render_sharded(render_flat_fn, rays_flat, keys=('rgb_map', 'disp_map', 'acc_map', 'depth_map')):
    """Each rank renders its contiguous slice of `rays_flat` with render_flat_fn(rays) -> dict and
    the maps named in `keys` are gathered so that every rank returns the full-length result.
    One collective: the maps are packed side by side into a single [n, C] tensor."""
    n = rays_flat.shape[0]
    out = render_flat_fn(shard_rays(rays_flat))
    cols = [out[k].reshape(out[k].shape[0], -1) for k in keys]
    widths = [c.shape[1] for c in cols]
    full = gather_rows(torch.cat(cols, 1), n)
    res, at = {}, 0
    for k, wd in zip(keys, widths):
        v = full[:, at:at + wd]
        res[k] = v if out[k].dim() > 1 else v[:, 0]
        at += wd
    return res


def gather_dict(local, n_total):
    """All-gathers a dict of per-ray tensors [n_local, ...] into full-length ones with ONE
    collective: the entries are packed side by side into a single [n_local, C] fp32 tensor."""
    keys = sorted(local)
    cols = [local[k].reshape(local[k].shape[0], -1).to(torch.float32) for k in keys]
    widths = [c.shape[1] for c in cols]
    full = gather_rows(torch.cat(cols, 1), n_total)
    res, at = {}, 0
    for k, wd in zip(keys, widths):
        res[k] = full[:, at:at + wd].reshape((n_total,) + tuple(local[k].shape[1:]))
        at += wd
    return res


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, **kwargs):
    """run_nerf.render (run_nerf.py:28-119) with the ray batch sharded over the ranks: every rank
    must make the same call (same camera or same ray batch); each renders its contiguous slice and
    all of them return the complete [rgb_map, disp_map, acc_map, extras].  render_path takes it
    through its render_kwargs: render_path(poses, hwf, chunk, dict(render_kwargs_test,
    shard_over_ranks=True))."""
    from .run_nerf import render as render_one
    return render_one(H, W, focal, chunk=chunk, rays=rays, c2w=c2w, shard_over_ranks=True, **kwargs)


def seed_sampling(seed):
    """Seeds this rank's random streams (stratified depths, resampling uniforms, density noise,
    drawn with torch.rand / torch.randn on the GPU) with seed + rank, so that data-parallel ranks
    jitter their shards independently.  Network weights are not affected: create_nerf broadcasts
    rank 0's."""
    r, _ = world()
    torch.manual_seed(int(seed) + r)


def broadcast_int(value, src=0):
    """The integer `value` of rank `src` on every rank (e.g. the step a run resumes from)."""
    if not exchanging():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=src)
    return int(t.item())


def global_count(n_local, device=None):
    """Sum of `n_local` over the ranks (the global ray count of a sharded batch) as a 0-d float64
    tensor on `device`: one tiny collective and no host synchronisation (the optimisation loop must
    not wait for the GPU every step); 1-rank jobs need no collective."""
    t = torch.tensor(float(n_local), dtype=torch.float64, device=device)
    if exchanging():
        allreduce_sum_(t)
    return t


def allreduce_sum_(flat, async_op=False, group=None):
    """SUM over the ranks, in place, of one flat fp32 buffer (a network's whole gradient: 2.38 MB,
    one collective -- over xGMI the cost is latency, so never one call per tensor).  Each rank's
    loss must already be divided by the GLOBAL ray count so that the sum is the gradient of the
    global mean (run_nerf.py:781,:809-811).  Returns the work handle when async_op."""
    if not exchanging():
        return None
    if flat.is_cuda and dist.get_backend() == "gloo":  # see _staged: through the host, synchronously
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return _Done() if async_op else None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class _Done:
    """Work handle of an exchange that has already finished."""
    def wait(self):
        return True


def broadcast_parameters(models, src=0):
    """Makes every rank's networks bit-identical replicas of rank `src`'s (one broadcast of the
    flat parameter buffer per network), e.g. after create_nerf drew the initial weights from each
    rank's own generator or only one rank found a checkpoint.  create_nerf broadcasts the step to
    resume from as well (broadcast_int); the optimiser is created after it with zero moments and
    iteration 0 on every rank -- the reference never reloads its state (run_nerf.py:244-261) --
    so nothing of it needs to travel."""
    if not exchanging():
        return
    for m in models:
        with torch.no_grad():
            buf = _staged(m.flat)
            dist.broadcast(buf, src=src)
            if buf is not m.flat:
                m.flat.copy_(buf)
        m.mark_dirty()
