"""Generates tests/golden/render_core_v1.npz: seeded inputs and the fp64 oracle's outputs for every
stage of the per-ray render core, and tests/golden/render_core_v2.npz: the whole path once more on
networks with the spectrum of a fitted field (tests/util.py:fitted_like_weights), with what a test
needs to tell which rays crossed one of the reference's discontinuities.

The reference itself cannot produce these (TensorFlow 1.15 is not installable here and its
NERFCO submodule is absent), so the vectors come from oracle/nerf_oracle.py -- they freeze the
oracle's behaviour so that a later change to it, or to the kernels, is seen as a diff against a
committed file.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nerf_oracle as orc  # noqa: E402
from tests.util import fitted_like_weights, pose_spherical, synthetic_rays, trained_like_weights  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_core_v1.npz")
OUT2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_core_v2.npz")


def main():
    rng = np.random.default_rng(20260101)
    g = {}
    # weights are regenerated from the seed (2 x 2.4 MB would not be a "small fixture")
    g['weights_seed'] = np.array([123, 456])
    wc = trained_like_weights(np.random.default_rng(123))
    wf = trained_like_weights(np.random.default_rng(456))
    g['weights_checksum'] = np.array([float(sum(np.sum(w.astype(np.float64)) for w in wc)),
                                      float(sum(np.sum(w.astype(np.float64)) for w in wf))])
    # rays
    H, W, focal = 6, 8, 7.5
    c2w = pose_spherical(40., -25., 4.)[:3, :4].astype(np.float32)
    g['c2w'] = c2w
    g['hwf'] = np.array([H, W, focal])
    ro, rd = orc.get_rays(H, W, focal, c2w, np.float32)
    g['get_rays_o'], g['get_rays_d'] = ro, rd
    no, nd = orc.ndc_rays(H, W, focal, 1., ro, rd, np.float64)
    g['ndc_o'], g['ndc_d'] = no, nd
    # encoding
    x = rng.uniform(-3, 3, size=(16, 3)).astype(np.float32)
    g['embed_x'] = x
    g['embed_10'] = orc.embed(x, 10, np.float64)
    g['embed_4'] = orc.embed(x, 4, np.float64)
    # MLP
    pts = rng.uniform(-2.5, 2.5, size=(6, 8, 3)).astype(np.float32)
    vd = rng.normal(size=(6, 3))
    vd = (vd / np.linalg.norm(vd, axis=-1, keepdims=True)).astype(np.float32)
    g['mlp_pts'], g['mlp_viewdirs'] = pts, vd
    g['mlp_raw'] = orc.run_network(pts, vd, wc, dtype=np.float64)
    # compositing
    raw = (rng.normal(size=(12, 64, 4)) * 2).astype(np.float32)
    z = np.sort(rng.uniform(2., 6., size=(12, 64)), -1).astype(np.float32)
    d = rng.normal(size=(12, 3)).astype(np.float32)
    g['comp_raw'], g['comp_z'], g['comp_d'] = raw, z, d
    for white in (0, 1):
        out = orc.raw2outputs(raw, z, d, 0., bool(white), dtype=np.float64)
        for name, v in zip(['rgb', 'disp', 'acc', 'weights', 'depth'], out):
            g[f'comp_{name}_white{white}'] = v
    # sampling
    w = (0.2 + rng.uniform(size=(12, 62))).astype(np.float32)
    bins = (.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    u = rng.uniform(size=(12, 128)).astype(np.float32)
    g['pdf_bins'], g['pdf_weights'], g['pdf_u'] = bins, w, u
    g['pdf_samples_det'] = orc.sample_pdf(bins, w, 128, det=True, dtype=np.float64)
    g['pdf_samples_rand'] = orc.sample_pdf(bins, w, 128, det=False, u=u, dtype=np.float64)
    # depth sampling
    rb = synthetic_rays(32, seed=77)
    g['rays'] = rb
    pu = rng.uniform(size=(32, 64)).astype(np.float32)
    g['perturb_u'] = pu
    g['z_lin'] = orc.sample_z(rb[:, 6:7], rb[:, 7:8], 64, False, None, np.float64)
    g['z_lindisp'] = orc.sample_z(rb[:, 6:7], rb[:, 7:8], 64, True, None, np.float64)
    g['z_perturb'] = orc.sample_z(rb[:, 6:7], rb[:, 7:8], 64, False, pu, np.float64)
    # end to end, 64 + 128 samples
    for white in (0, 1):
        r = orc.render_rays(rb, wc, 64, retraw=True, N_importance=128, weights_fine=wf,
                            white_bkgd=bool(white), dtype=np.float64)
        for k in ['rgb_map', 'disp_map', 'acc_map', 'depth_map', 'rgb0', 'disp0', 'acc0', 'z_std', 'z_vals']:
            g[f'e2e_white{white}_{k}'] = r[k]
        if white:
            g['e2e_white1_raw_last_sigma'] = r['raw'][:, -1, 3]
    # Keras Adam, three steps on a 4-vector
    th, m, v = np.array([1., -2., 0.5, 3.]), np.zeros(4), np.zeros(4)
    gr = np.array([0.1, -0.2, 0.3, 0.])
    traj = []
    for it in range(3):
        th, m, v = orc.keras_adam_step(th, gr, m, v, it, orc.lr_schedule(5e-4, 250, it))
        traj.append(th.copy())
    g['adam_traj'] = np.stack(traj)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")

    # v2: the whole path, 64 + 128 and 64 + 64 samples, fitted-like networks
    g2 = {'weights_seed': np.array([789, 1011])}
    wc = fitted_like_weights(np.random.default_rng(789))
    wf = fitted_like_weights(np.random.default_rng(1011))
    g2['weights_checksum'] = np.array([float(sum(np.sum(w.astype(np.float64)) for w in wc)),
                                       float(sum(np.sum(w.astype(np.float64)) for w in wf))])
    rb = synthetic_rays(96, seed=78)
    g2['rays'] = rb
    for Si in (128, 64):
        for white in (0, 1):
            r = orc.render_rays(rb, wc, 64, retraw=True, N_importance=Si, weights_fine=wf, white_bkgd=bool(white),
                                dtype=np.float64)
            tag = f'e2e_{Si}_white{white}_'
            for k in ['rgb_map', 'disp_map', 'acc_map', 'depth_map', 'rgb0', 'acc0', 'z_std', 'weights0']:
                g2[tag + k] = r[k]
            g2[tag + 'raw0_last'] = r['raw0'][:, -1:, :]   # [N,1,4]: the samples whose sigma's sign is a step
            g2[tag + 'raw_last'] = r['raw'][:, -1:, :]
    np.savez_compressed(OUT2, **g2)
    print("wrote", OUT2, os.path.getsize(OUT2), "bytes")


if __name__ == "__main__":
    main()
