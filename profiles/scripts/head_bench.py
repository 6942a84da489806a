"""eg_dense_small_fwd_slices + eg_head_fused against the launches they replace (CelebA head: K = 16384, N = 19, B = 128), back to back on one stream.
usage: python profiles/scripts/head_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops
DEV, N, K, B, dtype = "cuda", 19, 16384, 128, 1


def timeit(fn, n=20, reps=10):
    """device time per call: n calls captured into one hipGraph (the host's ~10 us per eager launch would hide the kernels), replayed"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


for T in (1, 2, 3):
    x = torch.randn(T * B, K, device=DEV).bfloat16()
    wp = (torch.randn(N, K, device=DEV) * 0.02).bfloat16()
    bias = torch.zeros(N, device=DEV)
    sigma = torch.ones(T, device=DEV)
    code = torch.rand(B, 8, device=DEV)
    labels = torch.randint(0, 10, (B,), device=DEV)
    y = torch.empty(T * B, N, device=DEV)
    dout = torch.empty(T * B, N, device=DEV)
    dx = torch.empty_like(x)
    loss = torch.zeros(1, device=DEV)
    terms = torch.empty(3 * B, device=DEV)
    counter = torch.zeros(1, device=DEV, dtype=torch.int32)
    ws = torch.empty(16 * T * B * N, device=DEV)
    kw = dict(info=(1, 8, 10, code, labels, 1.0, 0.1, 0.5)) if T == 3 else dict(targets=(1.0, 0.0)[:T], scales=(0.5, 0.5)[:T])

    def fused():
        ns = ops.dense_small_fwd_slices(dtype, x, wp, T * B, K, K, N, ws)
        ops.head_fused(dtype, x, wp, bias, ws, ns, y, dout, dx, sigma, B, T, K, K, N, loss, terms, counter, ops.ACT_LRELU, 0.2, **kw)

    def separate():
        ops.dense_small_fwd(dtype, x, wp, bias, y, T * B, K, K, N, ws)
        if T == 3:
            ops.loss_info_rpqxy(y[:B], y[B:2 * B], y[2 * B:], N, 1, 8, 10, B, code, 8, labels, 1.0, 0.1, 0.5, loss, dout[:B], dout[B:2 * B], dout[2 * B:])
        else:
            for t in range(T):
                ops.loss_bce_sigmoid(y[t * B:(t + 1) * B], N, 0, B, 1.0, 0.5, loss, dout[t * B:(t + 1) * B])
        ops.dense_small_bwd(dtype, dout, wp, x, dx, T * B, K, K, N, ops.ACT_LRELU, 0.2, sigma, B)

    print(f"T={T}: fused {timeit(fused):7.1f} us   separate launches {timeit(separate):7.1f} us")
