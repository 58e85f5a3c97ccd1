"""LLFF ("forward-facing" / 360 real scenes) data set reader and pose utilities -- the
counterpart of the reference's load_llff.py (:61-342) on NumPy and the in-tree PNG reader.

poses_bounds.npy rows are 3x5 camera matrices [R | t | (H, W, focal)] flattened, followed by the
near and far depth bounds of that view.
"""
import os
import re

import numpy as np

from .imageio_lite import imread

_IMAGE_ENDINGS = ('JPG', 'jpg', 'png')


def get_list_of_image_filenames(base_directory):
    """load_llff.py:61-64."""
    d = os.path.join(base_directory, 'images')
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(_IMAGE_ENDINGS)]


def make_rgb(single_channel_image):
    """load_llff.py:66-67."""
    return np.repeat(single_channel_image[:, :, np.newaxis], 3, axis=-1)


def load_masks(base_directory, mask_directory, list_of_image_filenames=None):
    """load_llff.py:70-85: <mask_directory>/<frame number, 4 digits>.png, non-zero = inside."""
    if list_of_image_filenames is None:
        list_of_image_filenames = get_list_of_image_filenames(base_directory)
    masks = []
    for name in list_of_image_filenames:
        frame = int(re.findall(r'\d+', name)[-1])
        path = os.path.join(mask_directory, str(frame).zfill(4) + '.png')
        if not os.path.isfile(path):
            raise FileNotFoundError(f"mask not found: {path}")
        m = imread(path) > 0
        masks.append(m[:, :, 0] if m.ndim == 3 else m)
    return np.stack(masks, axis=0)


def _read_image(path):
    if not path.endswith('png'):
        raise NotImplementedError(f"{path}: only PNG images can be read here (convert JPEGs once, e.g. to images_<factor>/*.png)")
    return imread(path)


def _load_data(base_directory, factor=None, width=None, height=None, load_imgs=True):
    """load_llff.py:88-144.  The reference shells out to ImageMagick to create a missing
    images_<factor> directory (:9-58); here a missing directory is an error."""
    arr = np.load(os.path.join(base_directory, 'poses_bounds.npy'))
    poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0])     # [3,5,N]
    bounds = arr[:, -2:].transpose([1, 0])                           # [2,N]
    sfx = ''
    if factor is not None:
        sfx = f'_{factor}'
    elif height is not None or width is not None:
        shape = _read_image(get_list_of_image_filenames(base_directory)[0]).shape
        if height is not None:
            factor = shape[0] / float(height)
            width = int(shape[1] / factor)
        else:
            factor = shape[1] / float(width)
            height = int(shape[0] / factor)
        sfx = f'_{width}x{height}'
    else:
        factor = 1
    image_directory = os.path.join(base_directory, 'images' + sfx)
    if not os.path.isdir(image_directory):
        raise FileNotFoundError(f"{image_directory} does not exist (pre-shrunk images are expected on disk)")
    files = [os.path.join(image_directory, f) for f in sorted(os.listdir(image_directory)) if f.endswith(_IMAGE_ENDINGS)]
    if poses.shape[-1] != len(files):
        raise ValueError(f"{len(files)} images but {poses.shape[-1]} poses in {base_directory}")
    shape = _read_image(files[0]).shape
    poses[:2, 4, :] = np.array(shape[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    if not load_imgs:
        return poses, bounds
    imgs = np.stack([_read_image(f)[..., :3] / 255. for f in files], -1)   # [H,W,3,N]
    return poses, bounds, imgs


def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """load_llff.py:154-160: camera-to-world [3,4] looking along z with the given up hint."""
    back = normalize(z)
    right = normalize(np.cross(up, back))
    true_up = normalize(np.cross(back, right))
    return np.stack([right, true_up, back, pos], 1)


def ptstocam(pts, c2w):
    """load_llff.py:162-164."""
    return np.matmul(c2w[:3, :3].T, (pts - c2w[:3, 3])[..., np.newaxis])[..., 0]


def poses_avg(poses):
    """load_llff.py:166-175: mean centre, summed viewing direction, summed up; hwf of view 0."""
    hwf = poses[0, :3, -1:]
    centre = poses[:, :3, 3].mean(0)
    back = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(back, up, centre), hwf], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """load_llff.py:179-188."""
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    out = []
    for theta in np.linspace(0., 2. * np.pi * rots, int(N) + 1)[:-1]:
        eye = np.dot(c2w[:3, :4], np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads)
        look = normalize(eye - np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.])))
        out.append(np.concatenate([viewmatrix(look, up, eye), hwf], 1))
    return out


def _to_44(p):
    bottom = np.broadcast_to(np.array([0, 0, 0, 1.], dtype=p.dtype), p.shape[:-2] + (1, 4))
    return np.concatenate([p[..., :3, :4], bottom], -2)


def recenter_poses(poses):
    """load_llff.py:192-204: express every pose in the frame of the average pose."""
    out = poses + 0
    avg = _to_44(poses_avg(poses))
    out[:, :3, :4] = (np.linalg.inv(avg) @ _to_44(poses))[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """load_llff.py:210-266: recentre on the point closest to all optical axes, scale the mean
    camera distance to 1, and build a 120-view circular path at the cameras' mean height."""
    dirs = poses[:, :3, 2:3]
    origins = poses[:, :3, 3:4]
    proj = np.eye(3) - dirs * np.transpose(dirs, [0, 2, 1])            # projector off each axis
    rhs = -proj @ origins
    centre = np.squeeze(-np.linalg.inv((np.transpose(proj, [0, 2, 1]) @ proj).mean(0)) @ rhs.mean(0))
    up = normalize((poses[:, :3, 3] - centre).mean(0))
    ax1 = normalize(np.cross([.1, .2, .3], up))
    ax2 = normalize(np.cross(up, ax1))
    frame = np.stack([ax1, ax2, up, centre], 1)
    reset = np.linalg.inv(_to_44(frame[None])) @ _to_44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc                                                           # in place, as the reference does (:240)
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    radcircle = np.sqrt(rad ** 2 - zh ** 2)
    ring = []
    for th in np.linspace(0., 2. * np.pi, 120):
        eye = np.array([radcircle * np.cos(th), radcircle * np.sin(th), zh])
        back = normalize(eye)
        right = normalize(np.cross(back, np.array([0, 0, -1.])))
        ring.append(np.stack([right, normalize(np.cross(back, right)), back, eye], 1))
    ring = np.stack(ring, 0)
    hwf = poses[0, :3, -1:]
    ring = np.concatenate([ring, np.broadcast_to(hwf, ring[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, ring, bds


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """load_llff.py:269-342 -> images [N,H,W,3], poses [N,3,5], bds [N,2], render_poses, i_test."""
    poses, bds, imgs = _load_data(basedir, factor=factor)
    # [down, right, back] -> [right, up, back] column order (:276), view index first
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    images = np.moveaxis(imgs, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)

    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)

    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        c2w = poses_avg(poses)
        up = normalize(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / ((1. - dt) / close_depth + dt / inf_depth)       # the "focus depth" of the spiral
        zdelta = close_depth * .2
        rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        n_views, n_rots = 120, 2
        if path_zflat:
            zloc = -close_depth * .1
            c2w[:3, 3] = c2w[:3, 3] + zloc * c2w[:3, 2]
            rads[2] = 0.
            n_rots, n_views = 1, n_views / 2
        render_poses = render_path_spiral(c2w, up, rads, focal, zdelta, zrate=.5, rots=n_rots, N=n_views)
    render_poses = np.array(render_poses).astype(np.float32)

    c2w = poses_avg(poses)
    i_test = int(np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)))   # the view nearest the average
    return images.astype(np.float32), poses.astype(np.float32), bds, render_poses, i_test
