"""first Discriminator layer, B = 128: eg_im2col_img + the K = 64 GEMM over patch rows against eg_conv_img_mfma, 1 / 2 / 3 tapes"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops
dev = torch.device("cuda:0")
dt = ops.EG_BF16
B, C, S = 128, 3, 64


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


w = torch.randn(128, C, 4, 4, device=dev) * 0.1
wp = torch.empty(128 * 64, device=dev, dtype=torch.bfloat16)
ops.pack_strided(dt, w, wp, 128, 48, 64, 1, 48, 0, 1)
bias = torch.zeros(128, device=dev)
big = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
for T in (1, 2, 3):
    imgs = [torch.rand(B, C, S, S, device=dev) for _ in range(T)]
    npix = B * 1024
    patches = torch.empty(T * npix, 64, device=dev, dtype=torch.bfloat16)
    out = torch.empty(T * B, 32, 32, 128, device=dev, dtype=torch.bfloat16)
    sigma = torch.ones(T, device=dev)
    c = ops.make_conv(T * B, 32, 32, 64, 128, 1, 1, 0)
    ep = ops.epilogue(bias=bias, sigma=sigma, sigma_rows=npix, act=ops.ACT_LRELU, slope=0.1)

    def old():
        for t in range(T):
            ops.im2col_img(dt, imgs[t], patches[t * npix:(t + 1) * npix], B, C, S, S, 4, 2, 1, 64)
        ops.conv_fwd(c, dt, patches, wp, out, ep)

    t_old = timeit(old)
    t_new = timeit(lambda: ops.conv_img_mfma(dt, imgs, wp, out, B, C, S, S, ep))
    print(f"T={T}: im2col + GEMM {t_old:.1f} us | direct {t_new:.1f} us ({(T * B * (3 * 64 * 64 * 4 + 32 * 32 * 128 * 2)) / t_new / 1e6:.2f} TB/s algorithmic)")

# transposed side: ConvTranspose2d(128 -> 3) as GEMM + col2im against the one-launch kernel
a = torch.randn(B, 32, 32, 128, device=dev).to(torch.bfloat16)
w3 = torch.randn(128, C, 4, 4, device=dev) * 0.05
c2 = ops.make_conv(B, 32, 32, 128, 48, 1, 1, 0)
wp2 = torch.empty(ops.pack_fwd_elems(c2, dt), device=dev, dtype=torch.bfloat16)
ops.pack_strided(dt, w3, wp2, 48, 128, 128, C, 1, 16, 48)
cols = torch.empty(B * 1024, 48, device=dev, dtype=torch.bfloat16)
img = torch.empty(B, C, 64, 64, device=dev)
b3 = torch.zeros(C, device=dev)


def old_t():
    ops.conv_fwd(c2, dt, a, wp2, cols, None)
    ops.col2im_img(dt, cols, B, C, 32, 32, 4, 2, 1, b3, ops.ACT_TANH, 0.0, img)


print(f"ConvT 128->3: GEMM + col2im {timeit(old_t):.1f} us | one launch {timeit(lambda: ops.convt_img_mfma(dt, a, wp2, b3, img, B, C, 32, 32, ops.ACT_TANH, 0.0)):.1f} us")
