"""Blender ("nerf_synthetic") data set reader and its pose helper -- the counterpart of the
reference's load_blender.py (pose_spherical :12-39, load_blender_data :71-178) on NumPy and the
in-tree PNG reader (no TensorFlow, no imageio).

Returned structure is the reference's: (imgs [N,H,W,4] float32 in [0,1], poses [N,4,4] float32,
render_poses [40,4,4] float32, [H, W, focal], [i_train, i_val, i_test], extras).
"""
import json
import os

import numpy as np

from .imageio_lite import imread
from .load_llff import load_masks


def _translate_z(t):
    m = np.eye(4, dtype=np.float32)
    m[2, 3] = t
    return m


def _rotate_about_x(phi):
    c, s = np.float32(np.cos(phi)), np.float32(np.sin(phi))
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float32)


def _rotate_about_y(theta):
    c, s = np.float32(np.cos(theta)), np.float32(np.sin(theta))
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], dtype=np.float32)


_BLENDER_AXES = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)


def pose_spherical(theta, phi, radius):
    """load_blender.py:34-39: camera on a sphere of `radius` looking at the origin; angles in
    degrees.  fp32 throughout, as the reference's tf.float32 matrices are."""
    c2w = _rotate_about_x(np.float32(phi / 180. * np.pi)) @ _translate_z(np.float32(radius))
    c2w = _rotate_about_y(np.float32(theta / 180. * np.pi)) @ c2w
    return _BLENDER_AXES @ c2w


def area_halve(imgs):
    """Area resampling to half resolution: the mean of each 2x2 block (what
    tf.image.resize_area computes for an exact factor of two, load_blender.py:164)."""
    n, h, w, c = imgs.shape
    if h % 2 or w % 2:
        raise ValueError(f"half_res needs even image sizes, got {h}x{w}")
    return imgs.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4), dtype=np.float32).astype(np.float32)


def load_blender_data(basedir, half_res=False, testskip=1, image_extn='.png', get_depths=False,
                      mask_directory=None, image_field='file_path', image_dir_override=None, trainskip=1,
                      test_finish=None, train_frames_field='frames'):
    """load_blender.py:71-178.  The reference resizes to a hard-coded 400x400 when half_res is
    set (:164) while halving H and W (:165-166); here the images are halved, which is the same
    thing for the 800x800 data the flag exists for."""
    if get_depths:
        raise NotImplementedError("load_blender_data(get_depths=True): depth maps feed the fork's LINEMOD depth loss "
                                  "(run_nerf.py:715-729, :816-823; load_blender.py:42-68, :112-121), which is not part "
                                  "of this build; the depth file names are still returned in extras['depth_filenames']")
    splits = ['train', 'val', 'test']
    metas = {}
    for s in splits:
        with open(os.path.join(basedir, f'transforms_{s}.json'), 'r') as fp:
            metas[s] = json.load(fp)

    all_imgs, all_poses, all_files, all_depth_files = [], [], [], []
    counts = [0]
    for s in splits:
        meta = metas[s]
        if s == 'train' or testskip == 0:
            picked, field = slice(None, None, trainskip), train_frames_field
            if field not in meta:
                raise KeyError(f"frames_field '{field}' not in transforms_{s}.json (has {list(meta.keys())})")
        else:
            picked, field = slice(None, test_finish, testskip), 'frames'
        imgs, poses, files, depth_files = [], [], [], []
        for frame in meta[field][picked]:
            name = frame[image_field]
            path = os.path.join(basedir, name if any(e in name for e in ('png', 'jpg')) else name + image_extn)
            if image_dir_override is not None:
                path = os.path.join(image_dir_override, os.path.basename(path))
            if not os.path.isfile(path):
                raise FileNotFoundError(f"image not found: {path}")
            files.append(path)
            imgs.append(imread(path))
            poses.append(np.array(frame['transform_matrix']))
            if 'depth_path' in frame:
                dpath = frame['depth_path']
                depth_files.append(dpath)
        all_imgs.append((np.array(imgs) / 255.).astype(np.float32))   # all channels kept (RGBA)
        all_poses.append(np.array(poses).astype(np.float32))
        counts.append(counts[-1] + len(imgs))
        all_files.append(files)
        all_depth_files.append(depth_files)

    i_split = [np.arange(counts[i], counts[i + 1]) for i in range(3)]
    extras = {}
    if mask_directory is not None:
        extras['masks'] = load_masks(basedir, mask_directory, [f for fs in all_files for f in fs])

    imgs = np.concatenate(all_imgs, 0)
    poses = np.concatenate(all_poses, 0)
    H, W = imgs[0].shape[:2]
    meta = metas['test']                                            # the reference reads the last split's meta (:158)
    focal = .5 * W / np.tan(.5 * float(meta['camera_angle_x']))
    render_poses = np.stack([pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 40 + 1)[:-1]], 0)

    if half_res:
        imgs = area_halve(imgs)
        H, W, focal = H // 2, W // 2, focal / 2.
    if 'K' in meta:
        extras['K'] = np.array(meta['K'], dtype=np.float32)
    extras['filenames'] = all_files
    extras['depth_filenames'] = all_depth_files
    return imgs, poses, render_poses, [H, W, focal], i_split, extras
