"""Iteration rate of mnist.ApproximatorTrainer (MNIST/approximate_rpqmnxy.py:119-136) on one MI355X; run from the repo root."""
import importlib, os, sys, time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
for dtype in ("f32", "bf16"):
    torch.manual_seed(0)
    M = eg.mnist.Affine_classifier().to("cuda")
    tr = eg.mnist.ApproximatorTrainer(M, 128, dtype=dtype)
    tr.code.copy_(torch.rand(128, 7, device="cuda") * 2 - 1)
    tr.capture(warmup=True)
    for _ in range(50):
        tr.step_resident()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step_resident()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"approximator fit {dtype} B=128 graph={'yes' if tr.graph is not None else 'NO'}: {dt * 1e6:.1f} us/iter = {1 / dt:.0f} iter/s", flush=True)
