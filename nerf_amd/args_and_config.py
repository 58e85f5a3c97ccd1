"""The reference's option set (args_and_config.py:3-207) without configargparse: every option
name, type and default of config_parser(), a `--config` file of `key = value` lines
(config_lego.txt, paper_configs/*.txt), command-line values overriding the file.

    parser = config_parser()
    args = parser.parse_args()            # or parse_args(['--config', 'config_lego.txt', '--N_rand', '2048'])
"""
import argparse

_FLAG = object()

# (name, type or _FLAG, default)  -- order and defaults follow args_and_config.py
OPTIONS = [
    ("expname", str, None), ("basedir", str, './logs/'), ("datadir", str, './data/llff/fern'),
    # training options (:14-39)
    ("netdepth", int, 8), ("netwidth", int, 256), ("netdepth_fine", int, 8), ("netwidth_fine", int, 256),
    ("N_rand", int, 32 * 32 * 4), ("lrate", float, 5e-4), ("lrate_decay", int, 250),
    ("chunk", int, 1024 * 32), ("netchunk", int, 1024 * 64), ("no_batching", _FLAG, False),
    ("no_reload", _FLAG, False), ("ft_path", str, None), ("random_seed", int, None),
    # pre-crop (:42-45)
    ("precrop_iters", int, 0), ("precrop_frac", float, .5),
    # rendering (:48-70)
    ("N_samples", int, 64), ("N_importance", int, 0), ("perturb", float, 1.), ("use_viewdirs", _FLAG, False),
    ("i_embed", int, 0), ("multires", int, 10), ("multires_views", int, 4), ("raw_noise_std", float, 0.),
    ("render_only", _FLAG, False), ("render_test", _FLAG, False), ("render_factor", int, 0),
    # dataset (:73-98)
    ("dataset_type", str, 'llff'), ("testskip", int, 8), ("shape", str, 'greek'), ("white_bkgd", _FLAG, False),
    ("half_res", _FLAG, False), ("factor", int, None), ("no_ndc", _FLAG, False), ("lindisp", _FLAG, False),
    ("spherify", _FLAG, False), ("llffhold", int, 8),
    # logging (:101-110)
    ("i_print", int, 100), ("i_img", int, 500), ("i_weights", int, 10000), ("i_testset", int, 50000),
    ("i_video", int, 50000),
    # fork additions (:112-205)
    ("image_extn", str, '.png'), ("mask_directory", str, None), ("mask_images", _FLAG, False),
    ("ray_masking", _FLAG, False), ("sigma_masking", _FLAG, False), ("sigma_threshold", float, 0.),
    ("get_depth_maps", _FLAG, False), ("Z_limits_from_pose", _FLAG, False),
    ("force_black_background", _FLAG, False), ("visualize_optimization", _FLAG, False),
    ("visualize_results", _FLAG, False), ("use_K", _FLAG, False), ("image_fieldname", str, 'file_path'),
    ("approximate_poses_filename", str, ''), ("centring_transforms", str, None),
    ("img_loss_threshold", float, None), ("depth_loss_threshold", float, None), ("epochs", int, 10),
    ("output_directory", str, './'), ("use_huber_loss", _FLAG, False), ("image_dir_override", str, None),
    ("resample", float, 1.0), ("object_radius", float, 0.259425),
    ("colmap_db_filename", str, './data/linemod_driller_all_llff/database.db'),
    ("colmap_keypoints_filename", str, None), ("number_of_keypoints", int, None),
    ("keypoint_embedding_size", int, None), ("category_activation", str, None),
    ("keypoint_oversample", _FLAG, False), ("keypoint_iterations_start", int, 0),
    ("keypoint_loss_coeff", float, 1.0), ("choose_keypoint_closest_depth", _FLAG, False),
    ("zero_embedding_origin", _FLAG, False), ("keypoints_filename", str, None),
    ("keypoint_detector", str, None), ("keypoint_dropout", _FLAG, False), ("keypoint_regularize", _FLAG, False),
    ("autoencoded_keypoints_filename", str, None), ("learnable_embeddings_filename", str, None),
    ("trainskip", int, 1), ("depth_from_camera", _FLAG, False), ("test_finish", int, None),
    ("near", float, None), ("far", float, None), ("frames_field", str, 'frames'), ("depth_loss", _FLAG, False),
]
_BY_NAME = {name: (typ, default) for name, typ, default in OPTIONS}


def _to_bool(text, key):
    t = text.strip().lower()
    if t in ('true', 'yes', 'on', '1'):
        return True
    if t in ('false', 'no', 'off', '0', ''):
        return False
    raise ValueError(f"config: '{key} = {text}' is not a boolean")


def read_config_file(path):
    """`key = value` lines ('#' or ';' comments, optional quotes) -> dict of typed values.
    A key that is not an option is an error, as it is for configargparse."""
    out = {}
    with open(path, 'r') as fp:
        for ln, raw in enumerate(fp, 1):
            line = raw.strip()
            if not line or line[0] in '#;':
                continue
            if '=' in line:
                key, val = line.split('=', 1)
            elif ':' in line:
                key, val = line.split(':', 1)
            else:
                key, val = line, 'true'     # a bare flag name
            key, val = key.strip().lstrip('-'), val.strip()
            if len(val) >= 2 and val[0] == val[-1] and val[0] in '"\'':
                val = val[1:-1]
            if key not in _BY_NAME:
                raise ValueError(f"{path}:{ln}: unknown option '{key}'")
            typ, _ = _BY_NAME[key]
            if typ is _FLAG:
                out[key] = _to_bool(val, key)
            elif val.lower() == 'none':
                out[key] = None
            else:
                try:
                    out[key] = typ(val)
                except ValueError:
                    raise ValueError(f"{path}:{ln}: '{key} = {val}' is not {typ.__name__}") from None
    return out


class ConfigParser(argparse.ArgumentParser):
    """argparse parser whose defaults are overridden by the `--config` file, then by the command
    line (the precedence configargparse applies)."""

    def parse_known_args(self, args=None, namespace=None):
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument('--config', type=str, default=None)
        known, _ = pre.parse_known_args(args)
        if known.config:
            self.set_defaults(**read_config_file(known.config))
        return super().parse_known_args(args, namespace)


def config_parser():
    """args_and_config.py:3-207."""
    parser = ConfigParser()
    parser.add_argument('--config', type=str, default=None, help='config file path')
    for name, typ, default in OPTIONS:
        if typ is _FLAG:
            parser.add_argument('--' + name, action='store_true', default=default)
        else:
            parser.add_argument('--' + name, type=typ, default=default)
    return parser
