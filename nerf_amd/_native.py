"""ctypes binding of libnerfhip.so (include/nerf_hip.h).  No fallback: a missing library or a
missing GPU is an error, never a silent CPU path."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnerfhip.so")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_L = _c.c_int64
_F = _c.c_float
_D = _c.c_double

# name -> (restype, argtypes); mirrors include/nerf_hip.h one to one
SIGNATURES = {
    "nerf_abi_version": (_I, []),
    "nerf_last_error": (_c.c_char_p, []),
    "nerf_get_rays": (_I, [_I, _I, _D, _P, _P, _P, _P]),
    "nerf_ndc_rays": (_I, [_L, _I, _I, _D, _D, _P, _P, _P, _P, _P]),
    "nerf_build_ray_batch": (_I, [_L, _P, _P, _P, _I, _I, _I, _D, _P, _P, _D, _D, _P, _I, _P]),
    "nerf_png_unfilter": (_I, [_P, _L, _L, _I, _P]),
    "nerf_gather_pixel_rays": (_I, [_I, _I, _D, _I, _P, _P, _I, _P, _P, _L, _P, _P, _P]),
    "nerf_embed": (_I, [_L, _P, _I, _P, _P]),
    "nerf_sample_z": (_I, [_L, _I, _P, _I, _I, _P, _P, _P]),
    "nerf_mlp_packed_bytes": (_c.c_size_t, []),
    "nerf_mlp_pack": (_I, [_P, _P, _P]),
    "nerf_mlp_forward_rays": (_I, [_P, _P, _I, _P, _L, _I, _P, _P]),
    "nerf_mlp_forward_pts": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "nerf_composite_fwd": (_I, [_P, _P, _P, _I, _L, _I, _I, _P, _F, _P, _P, _P, _P, _P, _P]),
    "nerf_sample_pdf": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _P]),
    "nerf_resample_merge": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P]),
    "nerf_render_rays_workspace_bytes": (_c.c_size_t, [_L, _I, _I]),
    "nerf_render_rays": (_I, [_P, _P, _P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nerf_render_rays_static_bytes": (_c.c_size_t, []),
    "nerf_render_rays_static": (_I, [_P, _P, _P, _I, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nerf_composite_bwd": (_I, [_P, _P, _P, _I, _L, _I, _I, _P, _F, _P, _P, _P, _P, _P]),
    "nerf_mlp_act_bytes": (_c.c_size_t, [_L]),
    "nerf_mlp_forward_train": (_I, [_P, _P, _I, _P, _L, _I, _P, _P, _P]),
    "nerf_mlp_train_layout": (_I, [_P, _I]),
    "nerf_mlp_packed_bwd_bytes": (_c.c_size_t, []),
    "nerf_mlp_pack_bwd": (_I, [_P, _P, _P]),
    "nerf_mlp_grad_bytes": (_c.c_size_t, [_L]),
    "nerf_mlp_partial_bytes": (_c.c_size_t, []),
    "nerf_mlp_backward": (_I, [_P, _P, _P, _L, _F, _P, _P, _P, _I, _P]),
    "nerf_mlp_backward_status": (_I, [_P, _P, _P, _P, _P]),
    "nerf_mse_loss_fwd_bwd": (_I, [_P, _P, _L, _F, _F, _P, _P, _P]),
    "nerf_photometric_loss": (_I, [_P, _P, _P, _L, _F, _P, _P, _P, _P]),
    "nerf_adam_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P]),
}

_lib = None


class NerfHipError(RuntimeError):
    pass


def lib():
    """Loads the shared library once.  Raises if it has not been built (python -m nerf_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfHipError(
                f"{LIB_PATH} is missing: build it with `python -m nerf_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().nerf_last_error().decode(errors="replace")
        raise NerfHipError(f"{what} failed with code {code}: {msg}")


def stream():
    """The current torch HIP stream as the void* the C ABI takes."""
    return _c.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL); insists on contiguous fp32/byte CUDA storage."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NerfHipError("expected a tensor on the GPU (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise NerfHipError("expected a contiguous tensor")
    return _c.c_void_p(t.data_ptr())


def f32(t, device=None):
    """Contiguous fp32 CUDA tensor view/copy of `t` (tensor, ndarray or nested list)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if device is None:
        if t.is_cuda:
            device = t.device
        elif not torch.cuda.is_available():
            raise NerfHipError("no GPU visible: nerf_amd runs on MI355X only (there is no CPU path)")
        else:
            device = torch.device("cuda", torch.cuda.current_device())
    return t.to(device=device, dtype=torch.float32).contiguous()
