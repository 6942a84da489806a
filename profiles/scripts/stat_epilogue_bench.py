"""cost of the fused column statistics in the 8-wave kernel's epilogue: the CelebA launches that carry them, with and without, back to back"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops
dev = torch.device("cuda:0")
dt = ops.EG_BF16
ws = torch.zeros(eg.engine.SPLITK_WS_BYTES // 4, device=dev)


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


def rnd(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16)


B = 128
# G forward: ConvT in conv view, backward-data, BatchNorm moments
for H, Ci, Co in ((8, 512, 1024), (16, 256, 512), (32, 128, 256)):
    c = ops.make_conv(B, H, H, Ci, Co, 4, 2, 1)
    x, z = rnd(B, H // 2, H // 2, Co), torch.empty(B, H, H, Ci, device=dev, dtype=torch.bfloat16)
    wp = rnd(ops.pack_bwd_elems(c, dt))
    bias = torch.zeros(Ci, device=dev)
    ep0 = ops.epilogue(bias=bias, splitk_ws=ws)
    nrb = ops.conv_stat_blocks(c, dt, True, ep0)
    stat = torch.empty(2 * Ci * max(nrb, 1), device=dev)
    ep1 = ops.epilogue(bias=bias, splitk_ws=ws, stat_mode=ops.STAT_MOMENTS, stat_out=stat)
    t0, t1 = timeit(lambda: ops.conv_bwd_data(c, dt, x, wp, z, ep0)), timeit(lambda: ops.conv_bwd_data(c, dt, x, wp, z, ep1))
    fl = 2.0 * B * (H // 2) ** 2 * Co * Ci * 16
    print(f"G fwd  H{H} {Co}->{Ci} nrb {nrb}: plain {t0:.1f} us ({fl / t0 / 1e6:.0f} TF/s) | moments {t1:.1f} us")
# G backward: forward conv producing d(a[i-1]) with BatchNorm backward sums
for H, Ci, Co in ((16, 256, 512), (32, 128, 256)):
    c = ops.make_conv(B, H, H, Ci, Co, 4, 2, 1)
    x, out, z = rnd(B, H, H, Ci), torch.empty(B, H // 2, H // 2, Co, device=dev, dtype=torch.bfloat16), rnd(B, H // 2, H // 2, Co)
    wp = rnd(ops.pack_fwd_elems(c, dt))
    ep0 = ops.epilogue(splitk_ws=ws)
    nrb = ops.conv_stat_blocks(c, dt, False, ep0)
    stat = torch.empty(2 * Co * max(nrb, 1), device=dev)
    v = [torch.ones(Co, device=dev) for _ in range(4)]
    ep1 = ops.epilogue(splitk_ws=ws, stat_mode=ops.STAT_BN_BWD, stat_out=stat, stat_aux=z, stat_p=v, stat_act=ops.ACT_RELU)
    t0, t1 = timeit(lambda: ops.conv_fwd(c, dt, x, wp, out, ep0)), timeit(lambda: ops.conv_fwd(c, dt, x, wp, out, ep1))
    print(f"G bwd  H{H} {Ci}->{Co} nrb {nrb}: plain {t0:.1f} us | bn-bwd sums {t1:.1f} us")
# D backward: backward-data with mask, T tapes, bias / coefficient sums
for T in (2, 3):
    for H, Ci, Co in ((8, 512, 1024), (16, 256, 512), (32, 128, 256)):
        c = ops.make_conv(T * B, H, H, Ci, Co, 4, 2, 1)
        x, out, a = rnd(T * B, H // 2, H // 2, Co), torch.empty(T * B, H, H, Ci, device=dev, dtype=torch.bfloat16), rnd(T * B, H, H, Ci)
        wp = rnd(ops.pack_bwd_elems(c, dt))
        sigma = torch.ones(T, device=dev)
        kw = dict(sigma=sigma, sigma_rows=B * (H // 2) ** 2, mask=a, mask_act=ops.ACT_LRELU, mask_slope=0.1, splitk_ws=ws)
        ep0 = ops.epilogue(**kw)
        nrb = ops.conv_stat_blocks(c, dt, True, ep0)
        stat = torch.empty(Ci * max(nrb, 1) + max(nrb, 1) * (Ci // 128), device=dev)
        ep1 = ops.epilogue(stat_mode=ops.STAT_SN_BIAS, stat_out=stat, stat_p=(torch.zeros(Ci, device=dev),), stat_slope=0.1, **kw)
        t0, t1 = timeit(lambda: ops.conv_bwd_data(c, dt, x, wp, out, ep0)), timeit(lambda: ops.conv_bwd_data(c, dt, x, wp, out, ep1))
        print(f"D bwd T{T} H{H} {Co}->{Ci} nrb {nrb}: plain {t0:.1f} us | sn-bias sums {t1:.1f} us")
