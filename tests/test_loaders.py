"""Data-set readers, pose utilities and the option reader (load_blender.py, load_llff.py,
args_and_config.py of the reference) on synthetic data sets written to a temp directory."""
import json
import os
import struct
import zlib

import numpy as np
import pytest

from nerf_amd import imageio_lite
from nerf_amd.args_and_config import config_parser, read_config_file

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- PNG reader ---------------------------------------------------------------------------------
def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _write_png_with_filters(path, img, filters):
    """Encoder used only here: row y is stored with PNG filter type filters[y % len]."""
    h, w, c = img.shape
    flat = img.reshape(h, w * c).astype(np.int64)
    raw = bytearray()
    for y in range(h):
        f = filters[y % len(filters)]
        raw.append(f)
        for x in range(w * c):
            a = flat[y, x - c] if x >= c else 0
            b = flat[y - 1, x] if y else 0
            cc = flat[y - 1, x - c] if (y and x >= c) else 0
            pred = [0, a, b, (a + b) // 2, _paeth(a, b, cc)][f]
            raw.append(int(flat[y, x] - pred) & 0xff)
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    color = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    data = bytes(raw)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0))
    z = zlib.compress(data, 9)
    png += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"tEXt", b"k\x00v") + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    with open(path, "wb") as fp:
        fp.write(png)


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_png_reader_undoes_every_scanline_filter(tmp_path, channels):
    rng = np.random.default_rng(channels)
    img = rng.integers(0, 256, size=(13, 17, channels), dtype=np.uint8)
    p = str(tmp_path / "f.png")
    _write_png_with_filters(p, img, [0, 1, 2, 3, 4, 4, 3])
    got = imageio_lite.imread(p)
    np.testing.assert_array_equal(got if channels > 1 else got[..., None], img)
    # and the writer / reader pair used by render_path
    imageio_lite.imwrite(p, img if channels != 2 else img[..., :1])
    back = imageio_lite.imread(p)
    np.testing.assert_array_equal(back if back.ndim == 3 else back[..., None], img if channels != 2 else img[..., :1])


def test_png_reader_rejects_garbage(tmp_path):
    p = tmp_path / "x.png"
    p.write_bytes(b"not a png at all")
    with pytest.raises(ValueError):
        imageio_lite.imread(str(p))
    p.write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 32)   # a JPEG header (image_extn = .jpg configs)
    with pytest.raises(ValueError, match="JPEG"):
        imageio_lite.imread(str(p))


def _write_png16(path, img, filters):
    """16-bit samples, big-endian, filtered byte-wise with a step of 2 bytes per channel."""
    h, w = img.shape
    be = img.astype(">u2").view(np.uint8).reshape(h, w * 2).astype(np.int64)
    bpp, raw = 2, bytearray()
    for y in range(h):
        f = filters[y % len(filters)]
        raw.append(f)
        for x in range(w * 2):
            a = be[y, x - bpp] if x >= bpp else 0
            b = be[y - 1, x] if y else 0
            cc = be[y - 1, x - bpp] if (y and x >= bpp) else 0
            raw.append(int(be[y, x] - [0, a, b, (a + b) // 2, _paeth(a, b, cc)][f]) & 0xff)
    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as fp:
        fp.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0))
                 + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


@pytest.mark.parametrize("filters", [[0], [0, 1, 2, 3, 4]])
def test_png_reader_16_bit_grey(tmp_path, filters):
    """16-bit grey PNGs come back as uint16 arrays, whatever the row filters."""
    rng = np.random.default_rng(5)
    depth = rng.integers(400, 3000, size=(9, 11)).astype(np.uint16)
    depth[2, 3], depth[7, 1] = 0, 65535
    p = str(tmp_path / "d.png")
    _write_png16(p, depth, filters)
    got = imageio_lite.imread(p)
    assert got.dtype == np.uint16
    np.testing.assert_array_equal(got, depth)


# ---- option reader ------------------------------------------------------------------------------
def test_config_defaults_match_the_reference_option_table():
    a = config_parser().parse_args([])
    assert (a.netdepth, a.netwidth, a.N_rand, a.lrate, a.lrate_decay) == (8, 256, 4096, 5e-4, 250)
    assert (a.chunk, a.netchunk, a.N_samples, a.N_importance, a.perturb) == (32768, 65536, 64, 0, 1.)
    assert (a.multires, a.multires_views, a.i_embed, a.raw_noise_std) == (10, 4, 0, 0.)
    assert (a.dataset_type, a.testskip, a.llffhold, a.factor) == ('llff', 8, 8, None)
    assert not (a.use_viewdirs or a.white_bkgd or a.half_res or a.no_ndc or a.lindisp or a.no_batching)
    assert (a.precrop_iters, a.precrop_frac, a.i_weights, a.basedir) == (0, .5, 10000, './logs/')


def test_config_file_then_command_line(tmp_path):
    cfg = os.path.join(HERE, "golden", "config_blender_example.txt")
    a = config_parser().parse_args(['--config', cfg])
    assert (a.expname, a.dataset_type, a.N_importance, a.N_rand, a.lrate_decay) == ('lego_example', 'blender', 128, 1024, 500)
    assert a.half_res and a.no_batching and a.use_viewdirs and a.white_bkgd
    assert a.precrop_iters == 500 and a.raw_noise_std == 0.0 and a.multires == 10
    b = config_parser().parse_args(['--config', cfg, '--N_rand', '2048', '--lindisp'])
    assert b.N_rand == 2048 and b.lindisp and b.N_importance == 128
    bad = tmp_path / "bad.txt"
    bad.write_text("expname = x\nnot_an_option = 3\n")
    with pytest.raises(ValueError, match="unknown option"):
        read_config_file(str(bad))
    bad.write_text("N_rand = many\n")
    with pytest.raises(ValueError, match="not int"):
        read_config_file(str(bad))
    ok = tmp_path / "ok.txt"
    ok.write_text("# comment\nno_batching = False\nfactor = 4\ndatadir = './data/x'\n")
    assert read_config_file(str(ok)) == {'no_batching': False, 'factor': 4, 'datadir': './data/x'}


# ---- blender ------------------------------------------------------------------------------------
def _make_blender(tmp, H=8, W=8, counts=(5, 3, 4), seed=0):
    from nerf_amd.load_blender import pose_spherical
    rng = np.random.default_rng(seed)
    truth = {}
    for split, n in zip(('train', 'val', 'test'), counts):
        os.makedirs(os.path.join(tmp, split), exist_ok=True)
        frames, imgs = [], []
        for k in range(n):
            img = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
            _write_png_with_filters(os.path.join(tmp, split, f"r_{k}.png"), img, [k % 5])
            pose = pose_spherical(30. * k, -20., 4.)
            frames.append({"file_path": f"./{split}/r_{k}", "transform_matrix": pose.tolist()})
            imgs.append(img)
        with open(os.path.join(tmp, f"transforms_{split}.json"), "w") as fp:
            json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, fp)
        truth[split] = np.stack(imgs, 0)
    return truth


def test_load_blender_data(tmp_path):
    from nerf_amd.load_blender import load_blender_data, pose_spherical
    truth = _make_blender(str(tmp_path))
    imgs, poses, render_poses, hwf, i_split, extras = load_blender_data(str(tmp_path), half_res=False, testskip=2)
    assert imgs.shape == (5 + 2 + 2, 8, 8, 4) and imgs.dtype == np.float32
    assert [list(s) for s in i_split] == [[0, 1, 2, 3, 4], [5, 6], [7, 8]]
    np.testing.assert_array_equal(imgs[:5], (truth['train'] / 255.).astype(np.float32))
    np.testing.assert_array_equal(imgs[5:7], (truth['val'][::2] / 255.).astype(np.float32))
    np.testing.assert_array_equal(imgs[7:], (truth['test'][::2] / 255.).astype(np.float32))
    np.testing.assert_allclose(poses[6], pose_spherical(60., -20., 4.), atol=1e-7)
    H, W, focal = hwf
    assert (H, W) == (8, 8)
    np.testing.assert_allclose(focal, .5 * 8 / np.tan(.5 * 0.6911112070083618))
    assert render_poses.shape == (40, 4, 4) and render_poses.dtype == np.float32
    np.testing.assert_allclose(render_poses[0], pose_spherical(-180., -30., 4.), atol=1e-7)
    assert len(extras['filenames'][0]) == 5
    # half_res: 2x2 area means, H, W, focal halved
    imgs2, _, _, hwf2, _, _ = load_blender_data(str(tmp_path), half_res=True, testskip=2)
    assert imgs2.shape == (9, 4, 4, 4) and hwf2[:2] == [4, 4] and hwf2[2] == focal / 2.
    np.testing.assert_allclose(imgs2[0, 1, 2], imgs[0, 2:4, 4:6].reshape(4, 4).mean(0), rtol=1e-6)
    with pytest.raises(FileNotFoundError):
        load_blender_data(str(tmp_path), image_dir_override=str(tmp_path / "nowhere"))


def test_pose_spherical_geometry():
    from nerf_amd.load_blender import pose_spherical
    for th, ph, r in [(0., -30., 4.), (123., -75., 2.5), (-180., 10., 1.)]:
        c2w = pose_spherical(th, ph, r)
        R, t = c2w[:3, :3], c2w[:3, 3]
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
        np.testing.assert_allclose(np.linalg.norm(t), r, rtol=1e-6)
        np.testing.assert_allclose(R[:, 2], t / r, atol=1e-6)      # the camera looks down -z at the origin


# ---- LLFF ---------------------------------------------------------------------------------------
def _make_llff(tmp, n=7, H=6, W=10, factor=4, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(tmp, "images"), exist_ok=True)
    os.makedirs(os.path.join(tmp, f"images_{factor}"), exist_ok=True)
    rows = []
    imgs = []
    for k in range(n):
        big = rng.integers(0, 256, size=(H * factor, W * factor, 3), dtype=np.uint8)
        small = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        imageio_lite.imwrite(os.path.join(tmp, "images", f"img_{k:03d}.png"), big)
        _write_png_with_filters(os.path.join(tmp, f"images_{factor}", f"img_{k:03d}.png"), small, [1, 4, 2])
        imgs.append(small)
        # LLFF convention: columns are [down, right, back]; cameras roughly facing -z from z ~ 0
        ang = 0.1 * (k - n / 2)
        right = np.array([np.cos(ang), 0., -np.sin(ang)])
        up = np.array([0., 1., 0.])
        back = np.cross(right, up)
        pos = np.array([0.3 * k - 1., 0.1 * rng.normal(), 0.05 * rng.normal()])
        m = np.stack([-up, right, back, pos, np.array([H * factor, W * factor, 400.])], 1)   # [3,5]
        rows.append(np.concatenate([m.reshape(-1), [2.0 + 0.1 * k, 30.0 + k]]))
    np.save(os.path.join(tmp, "poses_bounds.npy"), np.stack(rows, 0))
    return np.stack(imgs, 0)


def test_load_llff_data_forward_facing(tmp_path):
    from nerf_amd.load_llff import load_llff_data, poses_avg
    small = _make_llff(str(tmp_path))
    images, poses, bds, render_poses, i_test = load_llff_data(str(tmp_path), factor=4, recenter=True, bd_factor=.75)
    assert images.shape == (7, 6, 10, 3) and poses.shape == (7, 3, 5) and bds.shape == (7, 2)
    np.testing.assert_array_equal(images, (small / 255.).astype(np.float32))
    # hwf column: image size of the shrunk images, focal divided by the factor
    np.testing.assert_allclose(poses[:, :, 4], np.tile([6., 10., 100.], (7, 1)))
    # bounds rescaled so that the nearest bound is 1/bd_factor
    np.testing.assert_allclose(bds.min(), 1. / .75, rtol=1e-6)
    # after recentring the average pose is the identity frame
    avg = poses_avg(poses)
    np.testing.assert_allclose(avg[:3, :3], np.eye(3), atol=1e-5)
    np.testing.assert_allclose(avg[:3, 3], 0., atol=1e-5)
    # rotations stay orthonormal and right-handed
    for p in list(poses) + list(render_poses):
        np.testing.assert_allclose(p[:3, :3] @ p[:3, :3].T, np.eye(3), atol=1e-5)
        assert np.linalg.det(p[:3, :3]) > 0.99
    assert render_poses.shape == (120, 3, 5)
    assert i_test == int(np.argmin(np.sum(np.square(poses[:, :3, 3]), -1)))
    rp_flat = load_llff_data(str(tmp_path), factor=4, path_zflat=True)[3]
    assert rp_flat.shape == (60, 3, 5)
    with pytest.raises(FileNotFoundError):
        load_llff_data(str(tmp_path), factor=8)


def test_load_llff_data_spherify(tmp_path):
    from nerf_amd.load_llff import load_llff_data
    _make_llff(str(tmp_path))
    images, poses, bds, render_poses, i_test = load_llff_data(str(tmp_path), factor=4, spherify=True)
    assert render_poses.shape == (120, 3, 5)
    # mean squared camera distance from the new origin is 1; the path is a circle at the mean height
    np.testing.assert_allclose(np.mean(np.sum(poses[:, :3, 3] ** 2, -1)), 1., rtol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(render_poses[:, :3, 3], axis=-1), 1., rtol=1e-5)
    np.testing.assert_allclose(render_poses[:, 2, 3], np.mean(poses[:, 2, 3]), atol=1e-5)
    for p in render_poses:
        np.testing.assert_allclose(p[:3, :3] @ p[:3, :3].T, np.eye(3), atol=1e-5)
        np.testing.assert_allclose(p[:3, 2], p[:3, 3] / np.linalg.norm(p[:3, 3]), atol=1e-5)   # looks at the origin


def test_viewmatrix_and_recenter_properties():
    from nerf_amd.load_llff import viewmatrix, recenter_poses, poses_avg, ptstocam
    m = viewmatrix(np.array([0., 0., 2.]), np.array([0., 1., 0.]), np.array([1., 2., 3.]))
    np.testing.assert_allclose(m, np.array([[1, 0, 0, 1], [0, 1, 0, 2], [0, 0, 1, 3.]]), atol=1e-12)
    rng = np.random.default_rng(0)
    poses = []
    for _ in range(6):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)) * 0.1 + np.eye(3))
        q = q * np.sign(np.diag(q))
        poses.append(np.concatenate([q, rng.normal(size=(3, 1)), np.array([[4.], [6.], [50.]])], 1))
    poses = np.stack(poses, 0)
    rc = recenter_poses(poses)
    avg = poses_avg(rc)
    np.testing.assert_allclose(avg[:3, :3], np.eye(3), atol=1e-9)
    np.testing.assert_allclose(avg[:3, 3], 0., atol=1e-9)
    np.testing.assert_array_equal(rc[:, :, 4], poses[:, :, 4])
    np.testing.assert_allclose(ptstocam(poses[0, :3, 3][None], poses[0])[0], 0., atol=1e-12)
