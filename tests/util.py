"""Shared test helpers: the flag set create_nerf reads, synthetic poses and ray batches."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def default_args(**over):
    """The hot-path flags with the reference's defaults (args_and_config.py) and the blender paper
    config's sampling (paper_configs/blender_config.txt: 64 coarse, 128 fine, use_viewdirs,
    white_bkgd)."""
    a = argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, N_importance=128, N_samples=64, netchunk=1024 * 64,
        chunk=1024 * 32, perturb=0., raw_noise_std=0., white_bkgd=True, lindisp=False,
        dataset_type='blender', no_ndc=False, basedir='/tmp/nerf_amd_logs', expname='test',
        ft_path=None, no_reload=True, keypoint_embedding_size=None,
        learnable_embeddings_filename=None, choose_keypoint_closest_depth=False,
        lrate=5e-4, lrate_decay=500, N_rand=4096)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def pose_spherical(theta, phi, radius):
    """Camera-to-world of a camera on a sphere looking at the origin (the construction of
    load_blender.py:12-39: translate along z, rotate by phi about x, by theta about y, then the
    blender axis swap)."""
    t = np.eye(4)
    t[2, 3] = radius
    p = phi / 180. * np.pi
    rp = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1]])
    th = theta / 180. * np.pi
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    c2w = rt @ rp @ t
    return np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ c2w


def synthetic_rays(n, seed=0, near=2., far=6.):
    """Ray batch [n,11] as render() assembles it: origins on a radius-4 sphere, directions toward
    uniform targets in [-1,1]^3 (not normalised), near/far, unit view directions."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = 4. * o / np.linalg.norm(o, axis=-1, keepdims=True)
    tgt = rng.uniform(-1., 1., size=(n, 3))
    d = tgt - o
    d = d * rng.uniform(0.5, 1.5, size=(n, 1))
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rb = np.concatenate([o, d, np.full((n, 1), near), np.full((n, 1), far), vd], -1)
    return rb.astype(np.float32)


def trained_like_weights(rng, scale_hidden=1.0, sigma_bias=0.5, alpha_scale=1.0, pe_decay=0.0):
    """Glorot-uniform weights (Keras default) with small non-zero biases so every bias path is
    exercised; the alpha bias is raised a little so that rays are neither empty nor opaque.

    Knobs that move the network away from its initialisation, towards what optimisation produces
    (no trained weights exist offline): `scale_hidden` multiplies every kernel (activations grow
    by that factor per layer); `alpha_scale` sharpens the density head (|sigma| in the hundreds,
    as the histogram printed in the reference's extract_mesh.ipynb shows for a trained scene);
    `pe_decay` > 0 damps the rows fed by octave k of the point encoding by 2^(-pe_decay k), the
    spectral fall-off of a fitted field (with white-noise weights on sin(512 x) the field is a
    different random number at every sample and any two arithmetics disagree after resampling)."""
    from oracle import nerf_oracle as orc
    ws = orc.init_nerf_weights(rng)
    for i in range(1, len(ws), 2):
        ws[i] = rng.uniform(-0.1, 0.1, size=ws[i].shape).astype(np.float32)
    for i in range(0, len(ws), 2):
        ws[i] = (ws[i] * scale_hidden).astype(np.float32)
    if pe_decay > 0.:
        for wi in (0, 10):           # layer 0 and the skip rows of layer 5: [x(3), (sin, cos)(6) per octave]
            for k in range(10):
                ws[wi][3 + 6 * k: 9 + 6 * k] *= np.float32(2.0 ** (-pe_decay * k))
    ws[22] = (ws[22] * alpha_scale).astype(np.float32)
    ws[23] = ws[23] + np.float32(sigma_bias)
    return ws
