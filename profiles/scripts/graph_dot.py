"""Dump the captured CelebA iteration as a DOT file (hipGraphDebugDotPrint through torch.cuda.CUDAGraph.debug_dump) and list, for chosen kernels,
the kernels they depend on.  usage: python profiles/scripts/graph_dot.py <out.dot> [kernel-name-substring ...]"""
import importlib
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "128"))
torch.manual_seed(0)
G, D = eg.celeba.Generator(dtype="bf16").to(dev), eg.celeba.Discriminator(dtype="bf16").to(dev)
tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16")
g = torch.Generator(device=dev).manual_seed(1)
tr.load_inputs(torch.rand((B, 3, 64, 64), device=dev, generator=g) * 2 - 1, torch.randn((B, 200), device=dev, generator=g),
               torch.rand((B, 8), device=dev, generator=g) * 2 - 1, torch.randint(0, 10, (B,), device=dev, generator=g))
tr.step_resident()
torch.cuda.synchronize()
import ctypes
graph = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(graph):
    tr._step_with_inputs()
raw = graph.raw_cuda_graph()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(raw), os.path.abspath(sys.argv[1]).encode(), 1)      # 1 = verbose
print("hipGraphDebugDotPrint rc", rc)
txt = open(sys.argv[1]).read()
print(len(txt), "bytes of DOT")
