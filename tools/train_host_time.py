"""How much of an optimisation step is the host's: for a few batch sizes, the time the Python loop
needs to ENQUEUE a step (no synchronisation inside the loop) next to the time the GPU needs to
finish it, for training.train_step and for training.GraphedTrainStep (the step as one HIP graph).
A step whose enqueue time reaches its GPU time is bound by the host.
    python tools/train_host_time.py [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    from nerf_amd import training
    from nerf_amd.run_nerf import create_nerf, render
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    kw_train, _, _, _, models = create_nerf(bench.make_args())
    kw = dict(kw_train)
    kw['perturb'] = 1.
    grad_vars = models['model'].trainable_variables + models['model_fine'].trainable_variables
    opt = training.Adam(training.ExponentialDecay(5e-4, 500 * 1000, 0.1))
    fn = lambda r: render(1, 1, 1., chunk=1024 * 32, rays=r, near=2., far=6., retraw=True, **kw)
    for n in (256, 1024, 2048, 4096):
        o, d = bench.synthetic_batch(n, 5000, dev)
        target = torch.rand((n, 3), device=dev)
        for _ in range(5):
            training.train_step(fn, (o, d), target, grad_vars, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            training.train_step(fn, (o, d), target, grad_vars, opt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        graphed = training.GraphedTrainStep(fn, n, grad_vars, opt, [models['model'], models['model_fine']])
        for _ in range(5):
            graphed((o, d), target)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(steps):
            graphed((o, d), target)
        t4 = time.perf_counter()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        print(f"N_rand {n:5d}: train_step enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step, finished {1e3 * (t2 - t0) / steps:.3f} ms/step"
              f" | GraphedTrainStep enqueue {1e3 * (t4 - t3) / steps:.3f}, finished {1e3 * (t5 - t3) / steps:.3f} ms/step")
        del graphed


if __name__ == "__main__":
    main()
