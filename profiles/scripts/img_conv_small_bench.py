"""first trunk layer of the dSprites networks (Conv2d(C -> 32, 4, 2, 1) + LeakyReLU): eg_im2col_img + the K = 64 GEMM over patch rows against
eg_conv_img_mfma_n, and whether a TrunkEngine dispatches to it.  usage: python profiles/scripts/img_conv_small_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops
dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for dt, B, C, N, T in ((ops.EG_F16, 512, 3, 32, 1), (ops.EG_F16, 512, 3, 32, 2), (ops.EG_F16, 512, 3, 32, 3), (ops.EG_BF16, 128, 1, 32, 1), (ops.EG_BF16, 128, 1, 32, 3)):
    S = 64
    tdt = ops.torch_dtype(dt)
    w = torch.randn(N, C, 4, 4, device=dev) * 0.1
    wp = torch.empty(N * 64, device=dev, dtype=tdt)
    ops.pack_strided(dt, w, wp, N, C * 16, 64, 1, C * 16, 0, 1)
    bias = torch.zeros(N, device=dev)
    imgs = [torch.rand(B, C, S, S, device=dev) for _ in range(T)]
    npix = B * 1024
    kp = ops.round_up(C * 16, 8)
    patches = torch.zeros(T * npix, kp, device=dev, dtype=tdt)
    out = torch.empty(T * B, 32, 32, N, device=dev, dtype=tdt)
    sigma = torch.ones(T, device=dev)
    c = ops.make_conv(T * B, 32, 32, kp, N, 1, 1, 0)
    ep = ops.epilogue(bias=bias, sigma=sigma, sigma_rows=npix, act=ops.ACT_LRELU, slope=0.2)

    def im2col():
        for t in range(T):
            ops.im2col_img(dt, imgs[t], patches[t * npix:(t + 1) * npix], B, C, S, S, 4, 2, 1, kp)

    t_i = timeit(im2col)
    t_g = timeit(lambda: ops.conv_fwd(c, dt, patches, wp, out, ep))
    t_d = timeit(lambda: ops.conv_img_mfma(dt, imgs, wp, out, B, C, S, S, ep, N=N))
    mb = T * B * (C * S * S * 4 + 1024 * N * 2) / 1e6
    print(f"dtype {dt} B {B} C {C} N {N} T {T}: im2col {t_i:.1f} us + GEMM {t_g:.1f} us | direct {t_d:.1f} us ({mb / t_d:.2f} TB/s algorithmic)", flush=True)
