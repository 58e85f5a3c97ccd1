"""Headline benchmark: rays/s of the per-ray render core at 64 coarse + 128 fine samples per ray.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (ray-batch assembly, depths, fused encoding+MLP coarse,
compositing, resampling+merge, MLP fine, compositing) over a synthetic batch of 1,048,576 rays per
GPU that is resident in HBM before the clock starts.  With N > 1 every rank renders its own batch
(rays are independent: weak scaling, no collective on the data path) and the ranks all-gather the
rendered maps (rgb, disp, acc: 20 B/ray) at the end of the step, as a sharded image render would.
Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (the fused MLP) from HIP events
recorded around every one of its launches inside the timed steps; `hbm_kernels` does the same for
the bandwidth-bound compositing and resampling launches (algorithmic bytes / event time against
the 8 TB/s HBM peak); `cpu_baseline` times the
unfused torch-CPU restatement of the reference's TF op graph (oracle/torch_restatement.py) on a
bounded sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT = 2 * 593408          # one network, use_viewdirs (run_nerf_helpers.py:95-112)
MFMA_PEAK_TFLOPS = 2500.0            # dense fp16/bf16 MFMA, MI355X
# what a bare v_mfma_f32_16x16x32_f16 loop on random register operands sustains on this pool's
# MI355X under its power limit (tools/mfma_ceiling.hip; its output is kept under profiles/): context for
# roofline.frac, which stays priced against the spec peak
BARE_MFMA_LOOP_TFLOPS = 1880.0
HBM_PEAK_GBS = 8000.0                # HBM3E, MI355X
# bytes one optimisation step moves per query point by construction: saved fp16 activations and
# ReLU masks written (151 KiB per 32 points), fp16 layer gradients written (138 KiB), masks read
# by the chain (9 KiB), every saved block read once for the weight gradients (284 KiB)
TRAIN_BYTES_PER_POINT = (151 + 138 + 9 + 284) * 1024 / 32
N_RAYS = 1024 * 1024
N_SAMPLES, N_IMPORTANCE = 64, 128


KERNEL_SOURCES = ("mlp_forward_wide.hip", "mlp_common.h", "mlp_layout_wide.h", "mlp_layout.h", "common.h")


def kernel_source_hash():
    """sha256 over the sources the dominant kernel (mlp_fwd_wide_kernel) is built from."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "nerf_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic_per_launch(kernel="mlp_fwd_wide_kernel<0, false,", summary=None):
    """(HBM bytes per launch of the dominant kernel, note) from the newest committed rocprofv3
    counter summary (profiles/rNN_pmc_summary.txt, written by tools/profile_round.sh: FETCH_SIZE and
    WRITE_SIZE in KiB summed over n dispatches of the same launch shapes as the default bench --
    131072-ray chunks, coarse and fine launches in equal number).  FETCH_SIZE is doubled (gfx950
    tallies 128-byte read requests at 64 bytes).  bench.py cannot collect counters itself, so the
    figure is only quoted when the summary was taken on the kernel sources of this tree (its first
    line holds their hash); otherwise (None, reason)."""
    import glob
    if summary is None:
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.txt")))
        if not found:
            return None, "no counter summary under profiles/"
        summary = found[-1]
    try:
        text = open(summary).read()
    except OSError as e:
        return None, f"unreadable counter summary: {e}"
    name = os.path.basename(summary)
    first = text.splitlines()[0] if text else ""
    if not first.startswith("# source_hash"):
        return None, f"{name} does not record the kernel sources it was taken on"
    if first.split()[-1] != kernel_source_hash():
        return None, f"{name} was taken on other kernel sources than this tree's (re-run tools/profile_round.sh)"
    got = {}
    for section in text.split("== pass")[1:]:
        lines = section.splitlines()
        for i, line in enumerate(lines):
            if kernel in line and "n=" in line:
                n = int(line.split("n=")[1].split()[0])
                for follow in lines[i + 1:]:
                    if not follow.startswith("    "):
                        break
                    parts = follow.split()
                    if parts[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                        got[parts[0]] = float(parts[1]) * 1024.0 / n
                break
    if "FETCH_SIZE" not in got or "WRITE_SIZE" not in got:
        return None, f"{name} holds no FETCH_SIZE / WRITE_SIZE for {kernel}"
    return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], f"{name}, same kernel sources"


GROUPED = False  # a process group is up (more than one rank, or the one-rank RCCL rehearsal)


def backend_is_rccl():
    import torch.distributed as dist
    return dist.get_backend() == "nccl"


def _max_over_ranks_on_host(t):
    """Rehearsal backend (gloo): the reduction runs on a host copy."""
    import torch.distributed as dist
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.MAX)
    return h


def synthetic_batch(n, seed, device):
    """Origins on a radius-4 sphere, directions toward uniform targets in [-1,1]^3, not normalised."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    o = torch.randn((n, 3), generator=g)
    o = 4. * o / o.norm(dim=-1, keepdim=True)
    tgt = torch.rand((n, 3), generator=g) * 2 - 1
    d = (tgt - o) * (0.5 + torch.rand((n, 1), generator=g))
    return o.to(device), d.to(device)


def make_args():
    return argparse.Namespace(
        multires=10, multires_views=4, i_embed=0, use_viewdirs=True, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, N_importance=N_IMPORTANCE, N_samples=N_SAMPLES,
        netchunk=1024 * 64, perturb=0., raw_noise_std=0., white_bkgd=True, lindisp=False,
        dataset_type='blender', no_ndc=False, basedir='/tmp/nerf_amd_bench', expname='bench',
        ft_path=None, no_reload=True)


def cpu_baseline(n_rays, seed=0):
    """rays/s of the op-for-op torch-CPU restatement (fp32, chunk=32768, netchunk=65536) on rays of
    the half-resolution blender view (400 x 400, focal 555.6, pose_spherical(-180, -30, 4), near 2,
    far 6, white background: config_lego.txt with the paper's 64 + 128 samples): n_rays consecutive
    rays from the middle of the view -- 32768 by default, one `chunk` of its five."""
    from oracle import torch_restatement as trs
    from oracle import nerf_oracle as orc
    from nerf_amd.load_blender import pose_spherical
    rng = np.random.default_rng(seed)
    ws_c = [torch.as_tensor(w) for w in orc.init_nerf_weights(rng)]
    ws_f = [torch.as_tensor(w) for w in orc.init_nerf_weights(rng)]
    H = W = 400
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)
    ro, rd = orc.get_rays(H, W, focal, np.asarray(pose_spherical(-180., -30., 4.))[:3, :4].astype(np.float32), np.float32)
    n_rays = min(n_rays, H * W)
    first = (H * W - n_rays) // 2
    o = torch.as_tensor(ro.reshape(-1, 3)[first:first + n_rays].copy())
    d = torch.as_tensor(rd.reshape(-1, 3)[first:first + n_rays].copy())
    vd = d / d.norm(dim=-1, keepdim=True)
    rb = torch.cat([o, d, torch.full((n_rays, 1), 2.), torch.full((n_rays, 1), 6.), vd], -1)
    with torch.no_grad():
        trs.render_flat(rb[:256], ws_c, ws_f, white_bkgd=True)  # warm the thread pool
        t0 = time.perf_counter()
        trs.render_flat(rb, ws_c, ws_f, chunk=1024 * 32, white_bkgd=True, netchunk=1024 * 64)
        dt = time.perf_counter() - t0
    return n_rays / dt, dt


def train_leg(a, kw_train, models, dev, rank, world, fence):
    """Optimisation steps as run_nerf.py:770-826 performs them (N_rand rays per GPU, 64+128 samples,
    perturb=1, fwd + bwd + Keras Adam; gradients all-reduced over the ranks).  Weak scaling: the
    global batch is N_rand x n_gpus."""
    from nerf_amd import training
    from nerf_amd.run_nerf import render
    kw = dict(kw_train)
    kw['perturb'] = 1.
    grad_vars = models['model'].trainable_variables + models['model_fine'].trainable_variables
    o, d = synthetic_batch(a.n_rand, 5000 + rank, dev)
    g = torch.Generator(device="cpu")
    g.manual_seed(77 + rank)
    target = torch.rand((a.n_rand, 3), generator=g).to(dev)
    opt = training.Adam(training.ExponentialDecay(5e-4, 500 * 1000, 0.1))
    fn = lambda r: render(1, 1, 1., chunk=1024 * 32, rays=r, near=2., far=6., retraw=True, **kw)
    for _ in range(3):
        loss, _, _ = training.train_step(fn, (o, d), target, grad_vars, opt)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.train_steps):
        loss, _, _ = training.train_step(fn, (o, d), target, grad_vars, opt)
    fence()
    dt = time.perf_counter() - t0
    if GROUPED:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX) if backend_is_rccl() else t.copy_(_max_over_ranks_on_host(t))
        dt = float(t.item())
    assert np.isfinite(float(loss)), "training produced a non-finite loss"
    rays_per_s = world * a.n_rand * a.train_steps / dt
    points_per_ray = N_SAMPLES + N_SAMPLES + N_IMPORTANCE
    flop_per_ray = 3 * points_per_ray * FLOP_PER_POINT  # fwd + dX + dW
    gbs = rays_per_s * points_per_ray * TRAIN_BYTES_PER_POINT / 1e9
    return {"rays_per_s": round(rays_per_s, 1), "ms_per_step": round(dt / a.train_steps * 1e3, 3),
            "steps": a.train_steps, "n_rand_per_gpu": a.n_rand, "scaling": "weak",
            "algorithmic_tflops": round(rays_per_s * flop_per_ray / 1e12, 2),
            "frac_of_mfma_peak": round(rays_per_s * flop_per_ray / 1e12 / MFMA_PEAK_TFLOPS / world, 4),
            "bound": "hbm", "hbm_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS / world, 4),
            "final_loss": round(float(loss), 6)}


def strong_leg(a, kw_test, dev, rank, world, fence):
    """The full-resolution novel-view configuration: ONE 800 x 800 view (640,000 rays, 64+128 samples, chunk 32768)
    with the ray batch sharded over the ranks (contiguous slices, nerf_amd.parallel.render_sharded)
    and one all-gather of the packed maps (rgb, disp, acc, depth: 24 B/ray) at the end.  Strong
    scaling: the work is fixed, value = 640,000 x steps / time."""
    from nerf_amd import ops, parallel
    from nerf_amd.load_blender import pose_spherical
    from nerf_amd.run_nerf import batchify_rays
    H = W = 800
    focal = .5 * W / np.tan(.5 * 0.6911112070083618)
    c2w = torch.as_tensor(pose_spherical(-120., -30., 4.)[:3, :4])
    o, d = ops.get_rays(H, W, focal, c2w)
    rays_flat = ops.build_ray_batch(o, d, d, False, H, W, focal, 2., 6.)
    kw = {k: v for k, v in kw_test.items() if k not in ('ndc', 'use_viewdirs')}
    flat_fn = lambda r: batchify_rays(r, 1024 * 32, **kw)

    def step():
        return parallel.render_sharded(flat_fn, rays_flat)
    out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    # the exchange alone: the same all-gather on already rendered maps
    n = rays_flat.shape[0]
    lo, hi = parallel.shard_bounds(n, rank, world)
    packed = torch.cat([out['rgb_map'][lo:hi], out['disp_map'][lo:hi, None], out['acc_map'][lo:hi, None],
                        out['depth_map'][lo:hi, None]], 1).contiguous()
    parallel.gather_rows(packed, n)
    fence()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        parallel.gather_rows(packed, n)
    fence()
    dg = time.perf_counter() - t1
    if GROUPED:
        import torch.distributed as dist
        t = torch.tensor([dt, dg], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX) if backend_is_rccl() else t.copy_(_max_over_ranks_on_host(t))
        dt, dg = float(t[0].item()), float(t[1].item())
    assert out['rgb_map'].shape == (n, 3) and bool(torch.isfinite(out['rgb_map']).all())
    return {"workload": "one 800x800 view, 640000 rays, 64+128 samples, chunk 32768, ray batch sharded over the ranks",
            "rays_per_s": round(n * a.steps / dt, 1), "ms_per_view": round(dt / a.steps * 1e3, 3),
            "all_gather_ms": round(dg / a.steps * 1e3, 3), "gather_bytes_per_ray": 24, "scaling": "strong",
            "rays_per_gpu": hi - lo, "steps": a.steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=N_RAYS, help="rays per GPU per step")
    ap.add_argument("--chunk", type=int, default=1024 * 128, help="rays per render_rays call")
    ap.add_argument("--cpu-rays", type=int, default=32768, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=20, help="timed optimisation steps of the training leg (0 = skip)")
    ap.add_argument("--n-rand", type=int, default=4096, help="rays per GPU per optimisation step")
    ap.add_argument("--strong-leg", type=int, default=1, help="0 = skip the 800x800 sharded view (only allowed with --mode weak)")
    ap.add_argument("--mode", choices=("weak", "strong"), default="weak",
                    help="headline: weak = every rank renders its own --rays batch; strong = one 800x800 view sharded over the ranks")
    a = ap.parse_args()

    if os.environ.get("NERF_BENCH_WATCHDOG"):  # rehearsals: a hung rank prints where it hangs and exits
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["NERF_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # Rehearsal on a one-GPU box: NERF_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and
    # NERF_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); the numbers of
    # such a run mean nothing, it only walks the multi-rank code paths.
    share_gpu = os.environ.get("NERF_BENCH_SHARE_GPU", "0") == "1"
    backend = os.environ.get("NERF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(0 if share_gpu else local_rank)
    dev = torch.device("cuda", 0 if share_gpu else local_rank)
    # NERF_BENCH_RCCL_ALONE=1 (one process, one GPU): a ONE-rank RCCL group whose collectives are made
    # to run, so that a one-GPU box walks the multi-rank code paths over the real backend.
    global GROUPED
    rccl_alone = world == 1 and os.environ.get("NERF_BENCH_RCCL_ALONE", "0") == "1"
    GROUPED = world > 1 or rccl_alone
    dist = None
    if GROUPED:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from nerf_amd import ops, parallel
    from nerf_amd.run_nerf import create_nerf, render
    parallel.EXCHANGE_WHEN_ALONE = rccl_alone

    torch.manual_seed(rank)  # every rank draws its own initial weights; create_nerf broadcasts rank 0's
    kw_train, kw_test, _, _, models = create_nerf(make_args())
    parallel.seed_sampling(0)  # per-rank streams for the stratified depths and resampling uniforms
    o, d = synthetic_batch(a.rays, 1000 + rank, dev)
    def step():
        rgb, disp, acc, _ = render(1, 1, 1., chunk=a.chunk, rays=(o, d), near=2., far=6., **kw_test)
        if GROUPED:  # the packed maps of every rank, in rank order (one all-gather, 20 B/ray)
            parallel.gather_rows(torch.cat([rgb, disp[:, None], acc[:, None]], -1), world * a.rays)
        return rgb

    def fence():
        if GROUPED:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    ops.MLP_TIMING = []
    ops.HBM_TIMING = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rgb = step()
    fence()
    dt = time.perf_counter() - t0
    timing, ops.MLP_TIMING = ops.MLP_TIMING, None
    hbm_timing, ops.HBM_TIMING = ops.HBM_TIMING, None
    if GROUPED:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX) if backend_is_rccl() else t.copy_(_max_over_ranks_on_host(t))
        dt = float(t.item())
    assert bool(torch.isfinite(rgb).all()), "render produced non-finite colours"

    train = None
    if a.train_steps > 0:
        train = train_leg(a, kw_train, models, dev, rank, world, fence)
    strong = strong_leg(a, kw_test, dev, rank, world, fence) if (a.strong_leg or a.mode == "strong") else None

    mlp_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in timing)
    mlp_points = sum(p for _, _, p in timing)
    achieved = mlp_points * FLOP_PER_POINT / (mlp_ms * 1e-3) / 1e12
    rays_per_s = world * a.rays * a.steps / dt

    hbm_kernels = []
    for name, ev in sorted(hbm_timing.items()):
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in ev)
        gbs = sum(b for _, _, b in ev) / (ms * 1e-3) / 1e9
        hbm_kernels.append({"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 4), "launches": len(ev),
                            "avg_launch_ms": round(ms / len(ev), 4), "share_of_step": round(ms * 1e-3 / dt, 5)})

    if rank == 0:
        traffic, traffic_note = pmc_traffic_per_launch()
        if not (a.chunk == 131072 and a.rays % a.chunk == 0):
            traffic, traffic_note = None, "counter summary is for 131072-ray chunks"
        cpu = None
        if a.cpu_rays > 0 and world == 1:   # the CPU leg belongs to the one-GPU line only
            v, secs = cpu_baseline(a.cpu_rays)
            cpu = {"value": round(v, 1), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"{min(a.cpu_rays, 160000)} consecutive rays (one chunk of five) from the middle of the 400x400 "
                             f"half-res blender view, 64+128 samples, chunk=32768, netchunk=65536, fp32 torch-CPU "
                             f"restatement of the TF1.15 op graph, {secs:.1f} s; the whole 160000-ray view extrapolates "
                             f"to {160000 / v:.0f} s"}
        out = {
            "metric": "render rays/sec (64+128 samples/ray)",
            "value": round(rays_per_s, 1), "unit": "rays/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 (MFMA inputs; f32 accumulate, f32 compositing/sampling)",
            "data": "synthetic",
            "config": {"workload": f"render of a synthetic {a.rays}-ray batch per GPU, {N_SAMPLES} coarse + "
                                   f"{N_IMPORTANCE} fine samples, fused MLP fwd + raw2outputs + sample_pdf",
                       "rays_per_gpu": a.rays, "chunk": a.chunk, "networks": "2 x (D=8, W=256, skip 4, view head), random init",
                       "parallelism": f"ray-sharded x{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "bare_mfma_loop": BARE_MFMA_LOOP_TFLOPS,
                         "frac_of_bare_mfma_loop": round(achieved / BARE_MFMA_LOOP_TFLOPS, 4),
                         # bytes per launch from the committed counter passes (same launch shapes and
                         # kernel sources), not live; null with the reason when they do not match
                         "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": (mlp_points // max(1, len(timing))) * 20,
                         "kernel": "mlp_fwd_wide_kernel", "launches": len(timing),
                         "avg_launch_ms": round(mlp_ms / max(1, len(timing)), 4),
                         "flop_per_point": FLOP_PER_POINT, "points_per_launch": mlp_points // max(1, len(timing)),
                         "mlp_share_of_step": round(mlp_ms * 1e-3 / dt, 4)},
            "hbm_kernels": hbm_kernels,
            "cpu_baseline": cpu,
            "train": train,
            "strong_800x800": strong,
        }
        if a.mode == "strong":
            out.update(value=strong["rays_per_s"], ms_per_step=strong["ms_per_view"], scaling="strong")
            out["config"] = {"workload": strong["workload"], "rays_per_gpu": strong["rays_per_gpu"], "chunk": 32768,
                             "networks": out["config"]["networks"], "parallelism": f"ray-sharded x{world}"}
            out["weak_1M"] = {"rays_per_s": round(rays_per_s, 1), "ms_per_step": round(dt / a.steps * 1e3, 3)}
        print(json.dumps(out), flush=True)
    if GROUPED:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
