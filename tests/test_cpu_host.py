"""CPU tests: the C-ABI library loads and exports every symbol the header declares (no compute
call), host-side logic (checkpoint format, PNG writer, error behaviour) and the multi-process
ray-sharding path under gloo with world_size 2."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "nerf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from nerf_amd import _native
    lib = _native.lib()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/nerf_hip.h but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes signature in nerf_amd/_native.py"
    assert sorted(_native.SIGNATURES) == declared
    assert lib.nerf_abi_version() == 1
    assert lib.nerf_mlp_packed_bytes() == 74 * 16384 + 10 * 1024


def test_argument_errors_are_reported_without_a_gpu():
    from nerf_amd import _native
    lib = _native.lib()
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.nerf_sample_z(4, 0, None, 11, 0, None, None, None) == -1
    assert b"nerf_sample_z" in lib.nerf_last_error()
    assert lib.nerf_mlp_forward_rays(None, None, 11, None, 1, 64, None, None) == -1
    assert lib.nerf_build_ray_batch(1, None, None, None, 0, 1, 1, 1.0, None, None, 0.0, 1.0, None, 9, None) == -1
    assert lib.nerf_photometric_loss(None, None, None, 12, 1.0, None, None, None, None) == -1
    assert b"nerf_photometric_loss" in lib.nerf_last_error()
    with pytest.raises(_native.NerfHipError):
        _native.check(-1, "probe")


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, not compute on the host."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nerf_amd import _native, ops
    with pytest.raises(_native.NerfHipError):
        ops.get_rays(4, 4, 2.0, np.eye(4)[:3])
    with pytest.raises(_native.NerfHipError):
        ops.sample_z(torch.zeros((4, 11)), 8)
    from nerf_amd import training
    from nerf_amd.run_nerf_helpers import img2mse
    with pytest.raises(_native.NerfHipError):
        training.photometric_loss(torch.zeros((4, 3)), torch.zeros((4, 3)), torch.ones((4, 3)))
    with pytest.raises(_native.NerfHipError):
        img2mse(torch.zeros((4, 3)), torch.ones((4, 3)))


def test_product_code_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nerf_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} mentions the oracle"


def test_unsupported_network_shapes_raise():
    from nerf_amd.run_nerf_helpers import NerfModel
    with pytest.raises(NotImplementedError):
        NerfModel(8, 128, 63, 27, 4, [4], True, device="cpu")
    with pytest.raises(NotImplementedError):
        NerfModel(8, 256, 63, 0, 4, [4], False, device="cpu")


def test_checkpoint_roundtrip_and_safe_loader(tmp_path):
    from nerf_amd import checkpoint

    class Net:
        def __init__(self, ws):
            self.ws = ws

        def get_weights(self):
            return self.ws

    rng = np.random.default_rng(0)
    ws = [rng.normal(size=s).astype(np.float32) for s in [(63, 256), (256,), (319, 256), (256, 1), (1,)]]
    path = checkpoint.save_weights(Net(ws), 'model', 200000, str(tmp_path), 'exp')
    assert path.endswith('model_200000.npy')
    assert int(path[-10:-4]) + 1 == 200001                        # run_nerf.py:254
    assert '{}_fine_{}'.format(path[:-11], path[-10:]).endswith('model_fine_200000.npy')  # :258
    back = checkpoint.load_weights(path)
    assert len(back) == len(ws)
    for a, b in zip(back, ws):
        np.testing.assert_array_equal(a, b)
    # the same file as the reference reads it
    ref = np.load(path, allow_pickle=True)
    assert len(ref) == len(ws) and ref[2].shape == (319, 256)
    # a pickle that names anything but NumPy array constructors is refused
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    bad = np.empty((1,), dtype=object)
    bad[0] = Evil()
    p2 = str(tmp_path / "evil.npy")
    np.save(p2, bad, allow_pickle=True)
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load_weights(p2)


def test_png_writer_roundtrip(tmp_path):
    from nerf_amd.imageio_lite import imread, imwrite
    from nerf_amd.run_nerf_helpers import to8b
    img = to8b(np.random.default_rng(0).uniform(-0.2, 1.2, size=(7, 5, 3)))
    assert img.dtype == np.uint8 and img.min() == 0 and img.max() == 255
    p = str(tmp_path / "x.png")
    imwrite(p, img)
    np.testing.assert_array_equal(imread(p), img)
    assert open(p, "rb").read(8) == b"\x89PNG\r\n\x1a\n"


def test_shard_bounds_cover_all_rays():
    from nerf_amd.parallel import shard_bounds
    for n in [0, 1, 7, 8, 640000, 762048]:
        for w in [1, 2, 3, 8]:
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            lens = [hi - lo for lo, hi in spans]
            assert max(lens) - min(lens) <= 1


def _worker(rank, world_size, port, tmp):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from nerf_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    try:
        n = 1001  # odd: shards of 501 and 500 rays
        g = torch.Generator().manual_seed(0)
        rays = torch.rand((n, 11), generator=g)

        def fake_render(r):  # deterministic per-ray function standing in for render_rays
            return {'rgb_map': r[:, 0:3] * 2 + 1, 'disp_map': r[:, 6] * 3, 'acc_map': r[:, 7], 'depth_map': r[:, 8]}
        out = parallel.render_sharded(fake_render, rays)
        full = fake_render(rays)
        for k in full:
            assert out[k].shape == full[k].shape, k
            assert torch.equal(out[k], full[k]), k
        # every per-ray output of a shard in ONE collective, whatever its trailing shape; the resume step of rank 0
        lo_r, hi_r = parallel.shard_bounds(n, rank, world_size)
        part = {'rgb_map': rays[lo_r:hi_r, 0:3], 'raw': rays[lo_r:hi_r, :8].reshape(-1, 2, 4), 'z_std': rays[lo_r:hi_r, 9]}
        whole = parallel.gather_dict(part, n)
        assert torch.equal(whole['rgb_map'], rays[:, 0:3]) and torch.equal(whole['z_std'], rays[:, 9])
        assert whole['raw'].shape == (n, 2, 4) and torch.equal(whole['raw'], rays[:, :8].reshape(-1, 2, 4))
        assert parallel.broadcast_int(40 + rank) == 40
        # data-parallel gradient, the product's own functions on CPU tensors: unequal shards (501 /
        # 500 rays), each rank's MEAN loss weighted by n_local / n_global, SUM all-reduce of the flat
        # gradient buffer in place -> the gradient of the global mean
        from nerf_amd import training
        w = torch.nn.Parameter(torch.ones(11, 3))
        b = torch.nn.Parameter(torch.zeros(3))
        target = torch.rand((n, 3), generator=g)
        mine = parallel.shard_rays(rays)
        lo, hi = parallel.shard_bounds(n, rank, world_size)
        assert float(parallel.global_count(hi - lo)) == n
        weight = training.dp_loss_weight(hi - lo)
        assert weight.dtype == torch.float32 and abs(float(weight) - (hi - lo) / n) < 1e-7
        loss = ((mine @ w + b - target[lo:hi]) ** 2).mean() * weight
        gw, gb = torch.autograd.grad(loss, [w, b])
        flat = torch.zeros(36)
        views = [flat[:33].view(11, 3), flat[33:]]
        views[0].copy_(gw)
        views[1].copy_(gb)
        out = training.allreduce_gradients(views)
        assert out[0].data_ptr() == flat.data_ptr() and out[1].data_ptr() == views[1].data_ptr()   # in place, one bucket
        w2 = torch.nn.Parameter(torch.ones(11, 3))
        b2 = torch.nn.Parameter(torch.zeros(3))
        loss_full = ((rays @ w2 + b2 - target) ** 2).mean()
        fw, fb = torch.autograd.grad(loss_full, [w2, b2])
        assert torch.allclose(out[0], fw, atol=1e-6) and torch.allclose(out[1], fb, atol=1e-6)
        # tensors that do not tile one buffer are packed into one bucket
        out2 = training.allreduce_gradients([gw.clone(), gb.clone()])
        assert torch.allclose(out2[0], fw, atol=1e-6) and torch.allclose(out2[1], fb, atol=1e-6)
        # what train_step does: every rank back-propagates n_local x its local mean loss, the
        # gradients travel with [n_local x loss, n_local] behind them in ONE bucket, and the sums are
        # divided by the summed count -> gradient and value of the global mean loss, no other collective
        w3 = torch.nn.Parameter(torch.ones(11, 3))
        b3 = torch.nn.Parameter(torch.zeros(3))
        local = ((mine @ w3 + b3 - target[lo:hi]) ** 2).mean()
        step = training._DPStep(hi - lo, local, torch.device('cpu'))
        g3 = torch.autograd.grad(local * step.n_local, [w3, b3])
        out3, loss3 = training.allreduce_gradients(list(g3), step)
        assert torch.allclose(out3[0], fw, atol=1e-6) and torch.allclose(out3[1], fb, atol=1e-6)
        assert abs(float(loss3) - float(loss_full)) < 1e-6 * float(loss_full), (float(loss3), float(loss_full))
        # the averaged loss a step reports, and identical replicas after a broadcast

        class Net:
            def __init__(self, seed):
                self.flat = torch.rand(7, generator=torch.Generator().manual_seed(seed))
                self.dirty = 0

            def mark_dirty(self):
                self.dirty += 1
        net = Net(100 + rank)
        parallel.broadcast_parameters([net])
        assert torch.equal(net.flat, Net(100).flat) and net.dirty == 1
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_ray_sharding_two_ranks_gloo(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_bench_quotes_counter_traffic_only_for_matching_kernel_sources(tmp_path):
    """bench.py's roofline.traffic comes from a committed rocprofv3 counter summary and is quoted
    only when that summary was taken on this tree's kernel sources (hash in its first line)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    body = ("== pass sq: cmd\nvoid nerf::mlp_fwd_wide_kernel<0, false, 8, 3>: n=8 total_ms=109.0 vgpr=124/0\n"
            "    SQ_WAVE_CYCLES   9.63e+10  per_ns=883\n"
            "== pass fetch: cmd\nvoid nerf::mlp_fwd_wide_kernel<0, false, 8, 3>: n=8 total_ms=109.0 vgpr=124/0\n"
            "    FETCH_SIZE       3.7e+05  per_ns=3.4\n"
            "== pass write: cmd\nvoid nerf::mlp_fwd_wide_kernel<0, false, 8, 3>: n=8 total_ms=109.0 vgpr=124/0\n"
            "    WRITE_SIZE       2.048e+06  per_ns=18.8\n")
    good = tmp_path / "r99_pmc_summary.txt"
    good.write_text(f"# source_hash {bench.kernel_source_hash()}\n" + body)
    t, note = bench.pmc_traffic_per_launch(summary=str(good))
    assert abs(t - (2 * 3.7e5 + 2.048e6) * 1024 / 8) < 1 and "same kernel sources" in note
    stale = tmp_path / "r98_pmc_summary.txt"
    stale.write_text("# source_hash 0123abcd\n" + body)
    t, note = bench.pmc_traffic_per_launch(summary=str(stale))
    assert t is None and "other kernel sources" in note
    unkeyed = tmp_path / "r97_pmc_summary.txt"
    unkeyed.write_text(body)
    assert bench.pmc_traffic_per_launch(summary=str(unkeyed))[0] is None
    assert bench.pmc_traffic_per_launch("no_such_kernel", summary=str(good))[0] is None
    # whatever is committed under profiles/ either matches this tree or is declined with a reason
    t, note = bench.pmc_traffic_per_launch()
    assert (t is None and note) or 0.75 < t / (16777216 * 20) < 1.25
