"""nerf_render_rays_static: nerf_render_rays with its intermediates in a fixed array of the code
object instead of a caller workspace, for callers that cannot allocate device memory."""
import ctypes

import numpy as np
import pytest
import torch

from tests.util import default_args, synthetic_rays, trained_like_weights

pytestmark = pytest.mark.gpu


def _t(x, gpu):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=gpu)


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-1.), torch.nan_to_num(b, nan=-1.))   # disp is NaN where acc == 0


def test_render_rays_static_matches_workspace_form(gpu):
    from nerf_amd import _native as nv
    from nerf_amd.run_nerf import create_nerf
    lib = nv.lib()
    kw_train, kw_test, _, _, models = create_nerf(default_args(white_bkgd=True))
    rng = np.random.default_rng(4)
    models['model'].set_weights(trained_like_weights(rng))
    models['model_fine'].set_weights(trained_like_weights(rng))
    pc, pf = models['model'].packed(), models['model_fine'].packed()
    Sc, Si = 64, 128
    cap = int(lib.nerf_render_rays_static_bytes())
    assert cap == 64 << 20

    def outputs(n):
        new = lambda *s: torch.full(s, float('inf'), dtype=torch.float32, device=gpu)
        return [new(n, 3), new(n), new(n), new(n), new(n, 3), new(n), new(n), new(n)]

    def prepare(n, seed, with_u):
        rays = _t(synthetic_rays(n, seed=seed), gpu)
        r = np.random.default_rng(seed)
        pu = _t(r.uniform(size=(n, Sc)), gpu) if with_u else None
        uu = _t(r.uniform(size=(n, Si)), gpu) if with_u else None
        out = outputs(n)
        keep = (rays, pu, uu)   # alive until the call has finished
        return out, keep, [nv.ptr(pc), nv.ptr(pf), nv.ptr(rays), 11, n, Sc, Si, 0, 1, nv.ptr(pu), nv.ptr(uu)] + [nv.ptr(o) for o in out]

    def run(n, seed, static, with_u):
        out, keep, args = prepare(n, seed, with_u)
        if static:
            # inputs and outputs exist, the stream is idle: whatever the device's free memory
            # does between here and the end of the call is the entry point's own doing
            stream = nv.stream()
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            nv.check(lib.nerf_render_rays_static(*args, stream), "nerf_render_rays_static")
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == free_before, "nerf_render_rays_static allocated device memory"
        else:
            need = int(lib.nerf_render_rays_workspace_bytes(n, Sc, Si))
            ws = torch.empty((need,), dtype=torch.uint8, device=gpu)
            nv.check(lib.nerf_render_rays(*args, nv.ptr(ws), nv.stream()), "nerf_render_rays")
            torch.cuda.synchronize()
        return out

    # two calls in a row through the built-in array (each bracketed by the device's free-memory
    # figure, the first one included), the second on a smaller batch (the array still holds the
    # first call's intermediates), each against the workspace form
    n_big = 9000
    assert int(lib.nerf_render_rays_workspace_bytes(n_big, Sc, Si)) <= cap
    # set-up (n_rays = 0): the runtime loads the code object that holds the array on this GPU
    out0, keep0, args0 = prepare(1, 0, False)
    args0[4] = 0
    nv.check(lib.nerf_render_rays_static(*args0, nv.stream()), "nerf_render_rays_static(set-up)")
    assert bool(torch.isinf(out0[0]).all())            # nothing was rendered
    first = run(n_big, 1, True, False)
    second = run(2500, 2, True, True)
    for got, want in ((first, run(n_big, 1, False, False)), (second, run(2500, 2, False, True))):
        for g, w in zip(got, want):
            assert bool(torch.isfinite(torch.nan_to_num(g, nan=0.)).all())   # every output was written
            assert _same(g, w)
    # beyond the array: a clear error, nothing launched
    n_over = cap // 5000 + 1000
    assert int(lib.nerf_render_rays_workspace_bytes(n_over, Sc, Si)) > cap
    rays = _t(synthetic_rays(16, seed=0), gpu)       # never read: the size check comes first
    out = outputs(16)
    rc = lib.nerf_render_rays_static(nv.ptr(pc), nv.ptr(pf), nv.ptr(rays), 11, n_over, Sc, Si, 0, 1, None, None,
                                     *[nv.ptr(o) for o in out], nv.stream())
    assert rc == -2 and b"built-in array" in lib.nerf_last_error()
    assert bool(torch.isinf(out[0]).all())


def test_render_rays_static_under_graph_capture(gpu):
    """What the entry point is for: captured into a HIP graph (where nothing may allocate) after the
    one set-up call, then replayed twice on fresh inputs written into the captured buffers; each
    replay gives the bits of the workspace form."""
    from nerf_amd import _native as nv
    from nerf_amd.run_nerf import create_nerf
    lib = nv.lib()
    kw_train, kw_test, _, _, models = create_nerf(default_args(white_bkgd=True))
    rng = np.random.default_rng(6)
    models['model'].set_weights(trained_like_weights(rng))
    models['model_fine'].set_weights(trained_like_weights(rng))
    pc, pf = models['model'].packed(), models['model_fine'].packed()
    Sc, Si, n = 64, 128, 3000
    rays = _t(synthetic_rays(n, seed=21), gpu)
    new = lambda *s: torch.full(s, float('inf'), dtype=torch.float32, device=gpu)
    out = [new(n, 3), new(n), new(n), new(n), new(n, 3), new(n), new(n), new(n)]
    args = [nv.ptr(pc), nv.ptr(pf), nv.ptr(rays), 11, n, Sc, Si, 0, 1, None, None] + [nv.ptr(o) for o in out]
    # set-up outside the capture (n_rays = 0 loads the array's code object on this GPU)
    nv.check(lib.nerf_render_rays_static(*(args[:4] + [0] + args[5:]), nv.stream()), "nerf_render_rays_static(set-up)")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        nv.check(lib.nerf_render_rays_static(*args, nv.stream()), "nerf_render_rays_static(capture)")
    assert bool(torch.isinf(out[0]).all())            # captured, not run
    for seed in (21, 22):
        rays.copy_(_t(synthetic_rays(n, seed=seed), gpu))
        for o in out:
            o.fill_(float('inf'))
        graph.replay()
        torch.cuda.synchronize()
        want = [torch.empty_like(o) for o in out]
        ws = torch.empty((int(lib.nerf_render_rays_workspace_bytes(n, Sc, Si)),), dtype=torch.uint8, device=gpu)
        nv.check(lib.nerf_render_rays(*(args[:11] + [nv.ptr(w) for w in want]), nv.ptr(ws), nv.stream()), "nerf_render_rays")
        torch.cuda.synchronize()
        for g, w in zip(out, want):
            assert _same(g, w)
