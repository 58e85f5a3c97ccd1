"""Same-process A/B of one optimisation step (4096 rays, 64+128 samples, perturb=1, as bench.py's
training leg): training.train_step against the same step written the long way (img2mse twice,
mse2psnr twice, a sum, the tape seeded by autograd), in alternating blocks so that the box, its
clock and its temperature are the same for both.  Prints the median block time of each.  Names
given after the two counts are environment switches of experimental library builds: train_step
is timed once more with each of them set.
    python tools/ab_train_step.py [blocks] [steps per block] [ENV_SWITCH ...]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    from nerf_amd import training
    from nerf_amd.run_nerf import create_nerf, render
    from nerf_amd.run_nerf_helpers import img2mse, mse2psnr
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    kw_train, _, _, _, models = create_nerf(bench.make_args())
    kw = dict(kw_train)
    kw['perturb'] = 1.
    grad_vars = models['model'].trainable_variables + models['model_fine'].trainable_variables
    o, d = bench.synthetic_batch(4096, 5000, dev)
    target = torch.rand((4096, 3), device=dev)
    opt = training.Adam(training.ExponentialDecay(5e-4, 500 * 1000, 0.1))
    fn = lambda r: render(1, 1, 1., chunk=1024 * 32, rays=r, near=2., far=6., retraw=True, **kw)

    def fused():
        return training.train_step(fn, (o, d), target, grad_vars, opt)

    def long_way():
        with training.GradientTape():
            rgb, disp, acc, extras = fn((o, d))
            img_loss = img2mse(rgb, target)
            psnr = mse2psnr(img_loss.detach())
            img_loss0 = img2mse(extras['rgb0'], target)
            psnr0 = mse2psnr(img_loss0.detach())
            loss = img_loss + img_loss0
        grads = torch.autograd.grad(loss, grad_vars, allow_unused=True)
        opt.apply_gradients(zip(grads, grad_vars))
        return loss.detach(), psnr, psnr0

    variants = [("train_step", fused, None), ("long way", long_way, None)]
    variants += [("train_step " + env, fused, env) for env in sys.argv[3:]]
    times = {name: [] for name, _, _ in variants}

    def run(f, env, count):
        if env:
            os.environ[env] = "1"
        try:
            for _ in range(count):
                out = f()
            torch.cuda.synchronize()
        finally:
            if env:
                del os.environ[env]
        return out

    for name, f, env in variants:
        run(f, env, 5)
    for _ in range(blocks):
        for name, f, env in variants:
            t0 = time.perf_counter()
            out = run(f, env, steps)
            times[name].append((time.perf_counter() - t0) / steps * 1e3)
    for name, ts in times.items():
        print(f"{name:12s} median {statistics.median(ts):.3f} ms/step   blocks: " + " ".join(f"{t:.3f}" for t in ts))
    print("loss", float(out[0]))


if __name__ == "__main__":
    main()
