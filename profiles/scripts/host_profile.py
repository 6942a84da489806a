"""host-side cost of one eager CelebA iteration (cProfile over 40 iterations): which Python calls the launch loop spends its time in.
usage: python profiles/scripts/host_profile.py [dp]   (dp: with the data-parallel schedule over no-op collectives)"""
import cProfile
import importlib
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
dev = torch.device("cuda:0")
B = 128
torch.manual_seed(0)
G, D = eg.celeba.Generator(dtype="bf16").to(dev), eg.celeba.Discriminator(dtype="bf16").to(dev)
ar = eg.dp.GradAllReduce(1) if len(sys.argv) > 1 and sys.argv[1] == "dp" else None
tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16", allreduce=ar)
g = torch.Generator(device=dev).manual_seed(1)
tr.inputs = eg.celeba.DeviceInputs(torch.randint(0, 256, (4096, 3, 64, 64), device=dev, dtype=torch.uint8, generator=g), seed=1)
for _ in range(5):
    tr.step_resident()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    tr.step_resident()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_enq / 40 * 1e3:.3f} ms/iter, with the GPU drained {t_all / 40 * 1e3:.3f} ms/iter")
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    tr.step_resident()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
