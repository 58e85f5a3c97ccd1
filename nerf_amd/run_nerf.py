"""Drop-in counterparts of the render-side functions of the reference's run_nerf.py:
batchify_rays (:15-25), render (:28-119), render_path (:122-161), create_nerf (:164-263).
Same signatures, keyword names and return structures; tensors are torch tensors on the GPU."""
import os
import time
from collections import defaultdict

import numpy as np
import torch

from . import _native as nv
from . import nerf_renderer
from . import ops
from .checkpoint import load_weights
from .run_nerf_helpers import *  # noqa: F401,F403  (the reference star-imports its helpers, :7)
from .run_nerf_helpers import get_embedder, get_rays, init_nerf_model, to8b


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """run_nerf.py:15-25: render rays in `chunk`-sized pieces and concatenate every returned key."""
    all_ret = defaultdict(list)
    per_ray = ('perturb_u', 'pdf_u', 'noise0', 'noise1')
    for i in range(0, rays_flat.shape[0], chunk):
        kw = {k: (v[i:i + chunk] if (k in per_ray and v is not None) else v) for k, v in kwargs.items()}
        ret = nerf_renderer.render_rays(rays_flat[i:i + chunk], **kw)
        for k in ret.keys():
            all_ret[k].append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items()}


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1.,
           use_viewdirs=False, c2w_staticcam=None, shard_over_ranks=False, **kwargs):
    """run_nerf.py:28-119.  Returns [rgb_map, disp_map, acc_map, extras] with the leading shape of
    the rays ([H,W] for a camera, [B] for a ray batch).

    shard_over_ranks (not in the reference): under torch.distributed every rank must then make
    the same call; each renders a contiguous slice of the flat ray batch and one all-gather hands
    every rank the complete outputs (nerf_amd.parallel.render)."""
    if c2w is not None:
        if 'K' in kwargs:
            raise NotImplementedError("render(K=...): intrinsics-matrix rays (NERFCO.get_rays_tf_K, "
                                      "run_nerf.py:59-60) are not part of this build")
        rays_o, rays_d = get_rays(H, W, focal, c2w)                                   # :63
    else:
        if len(rays) == 3:                                                            # :66-69
            rays_o, rays_d, depth = rays
            depth = nv.f32(depth).reshape(-1)
            far = depth * 1.1
            near = depth * 0.9
        else:
            rays_o, rays_d = rays[0], rays[1]                                         # :71
        rays_o, rays_d = nv.f32(rays_o), nv.f32(rays_d)
    kwargs_copy = {k: v for k, v in kwargs.items() if k != 'K'}                       # :75-77

    view_src = None
    if use_viewdirs:
        view_src = rays_d                                                             # :81
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, focal, c2w_staticcam)                     # :84
    sh = rays_d.shape                                                                 # :91
    rays_flat = ops.build_ray_batch(rays_o, rays_d, view_src, ndc, H, W, focal, near, far)  # :92-108

    from . import parallel
    rank, world_size = parallel.world()
    if shard_over_ranks and parallel.exchanging():
        n_all = rays_flat.shape[0]
        lo, hi = parallel.shard_bounds(n_all, rank, world_size)
        per_ray = ('perturb_u', 'pdf_u', 'noise0', 'noise1')
        kw = {k: (v[lo:hi] if (k in per_ray and v is not None) else v) for k, v in kwargs_copy.items()}
        all_ret = parallel.gather_dict(batchify_rays(rays_flat[lo:hi], chunk, **kw), n_all)
    else:
        all_ret = batchify_rays(rays_flat, chunk, **kwargs_copy)                      # :111
    for k in all_ret:
        k_sh = list(sh[:-1]) + list(all_ret[k].shape[1:])                             # :113
        all_ret[k] = torch.reshape(all_ret[k], k_sh)
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


def render_path(render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """run_nerf.py:122-161: one render() per pose; PSNR printed when ground truth is given; PNGs
    written when savedir is given (through nerf_amd.imageio_lite, the reference uses imageio)."""
    H, W, focal = hwf
    if render_factor != 0:
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor
    rgbs = []
    extrass = defaultdict(list)
    t = time.time()
    for i, c2w in enumerate(render_poses):
        t = time.time()
        c2w = torch.as_tensor(np.asarray(c2w) if not isinstance(c2w, torch.Tensor) else c2w)
        rgb, disp, acc, extras = render(H, W, focal, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        extras['disp'] = disp
        rgbs.append(rgb.cpu().numpy())
        if i == 0:
            print(rgb.shape, disp.shape)
        if gt_imgs is not None and render_factor == 0:
            p = -10. * np.log10(np.mean(np.square(rgbs[-1] - np.asarray(gt_imgs[i]))))
            print(p)
        if savedir is not None:
            from .imageio_lite import imwrite
            imwrite(os.path.join(savedir, '{:03d}.png'.format(i)), to8b(rgbs[-1]))
        for k in extras:
            extrass[k].append(extras[k].cpu().numpy())
    rgbs = np.stack(rgbs, 0)
    extrass = {k: np.stack(v, 0) for k, v in extrass.items()}
    return rgbs, extrass


def create_nerf(args):
    """run_nerf.py:164-263: builds the embedders, the coarse and fine networks, the query
    closure and the two kwargs dicts; reloads the newest checkpoint unless args.no_reload.
    Returns (render_kwargs_train, render_kwargs_test, start, grad_vars, models)."""
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)                    # :167
    input_ch_views = 0
    embeddirs_fn = None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)  # :172
    output_ch = 4
    skips = [4]
    if getattr(args, 'keypoint_embedding_size', None):
        raise NotImplementedError("keypoint networks (NERFCO.nerf_keypoint_network, run_nerf.py:186-195) "
                                  "are not part of this build")
    model = init_nerf_model(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch,
                            skips=skips, input_ch_views=input_ch_views,
                            use_viewdirs=args.use_viewdirs)                            # :176-179
    grad_vars = model.trainable_variables
    models = {'model': model}
    model_fine = None
    if args.N_importance > 0:
        model_fine = init_nerf_model(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch,
                                     output_ch=output_ch, skips=skips,
                                     input_ch_views=input_ch_views,
                                     use_viewdirs=args.use_viewdirs)                   # :197-201
        grad_vars = grad_vars + model_fine.trainable_variables                         # :206
        models['model_fine'] = model_fine

    netchunk = getattr(args, 'netchunk', 1024 * 64)

    def network_query_fn(inputs, viewdirs, network_fn):                                # :209-214
        return nerf_renderer.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                         embeddirs_fn=embeddirs_fn, netchunk=netchunk,
                                         c2w=getattr(args, 'c2w', None))
    # render_rays forms the points on the fly (o + d*z) inside the MLP kernel when it sees this mark
    network_query_fn.nerf_amd_fused = True

    render_kwargs_train = {                                                            # :216-227
        'network_query_fn': network_query_fn,
        'perturb': args.perturb,
        'N_importance': args.N_importance,
        'network_fine': model_fine,
        'N_samples': args.N_samples,
        'network_fn': model,
        'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std,
        'choose_keypoint_closest_depth': getattr(args, 'choose_keypoint_closest_depth', False),
    }
    if args.dataset_type != 'llff' or args.no_ndc:                                     # :230-233
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}      # :235-238
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.

    start = 0
    basedir, expname = args.basedir, args.expname
    ft_path = getattr(args, 'ft_path', None)
    if ft_path is not None and ft_path != 'None':                                      # :244-248
        ckpts = [ft_path]
    else:
        d = os.path.join(basedir, expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d))
                 if ('model_' in f and 'fine' not in f and 'optimizer' not in f)] if os.path.isdir(d) else []
    if len(ckpts) > 0 and not args.no_reload:                                          # :250-261
        ft_weights = ckpts[-1]
        print('Reloading from', ft_weights)
        model.set_weights(load_weights(ft_weights))
        start = int(ft_weights[-10:-4]) + 1
        if model_fine is not None:
            ft_weights_fine = '{}_fine_{}'.format(ft_weights[:-11], ft_weights[-10:])
            print('Reloading fine from', ft_weights_fine)
            model_fine.set_weights(load_weights(ft_weights_fine))
    # one process per GPU: every rank must hold the same replicas (each drew its own initial
    # weights, and perhaps only some found a checkpoint) and resume from the same step
    from . import parallel
    parallel.broadcast_parameters(models.values())
    start = parallel.broadcast_int(start)
    return render_kwargs_train, render_kwargs_test, start, grad_vars, models
