"""optimizer.step() + panel refresh of one convolution layer: flat Adam launch + eg_pack_conv against the fused eg_adam_pack_conv (and the
1x1-input layer: eg_pack_strided against eg_adam_pack_rows).  usage: python profiles/scripts/adam_pack_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops
dev = torch.device("cuda:0")
dt = ops.EG_BF16


def timeit(fn, iters=50):
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


step = torch.ones(1, device=dev, dtype=torch.int32)
for cout, cin in ((1024, 512), (512, 256), (256, 128)):
    c = ops.make_conv(128, 8, 8, cin, cout, 4, 2, 1)
    n = cout * cin * 16
    p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v = v.abs()
    wf = torch.empty(ops.pack_fwd_elems(c, dt), device=dev, dtype=torch.bfloat16)
    wb = torch.empty(ops.pack_bwd_elems(c, dt), device=dev, dtype=torch.bfloat16)
    t_adam = timeit(lambda: ops.adam_step_zero(p, g, m, v, n, 1e-3, 0.5, 0.999, 1e-8, step, False, True))
    t_pack = timeit(lambda: ops.pack_conv(c, dt, p, wf, wb))
    t_fused = timeit(lambda: ops.adam_pack_conv(c, dt, p, g, m, v, 1e-3, 0.5, 0.999, 1e-8, step, True, wf, wb))
    print(f"conv {cout}x{cin}x4x4 ({n / 1e6:.1f} M): adam {t_adam:.1f} us ({n * 32 / t_adam / 1e6:.2f} TB/s) + pack {t_pack:.1f} us = {t_adam + t_pack:.1f} us | fused {t_fused:.1f} us "
          f"({n * 36 / t_fused / 1e6:.2f} TB/s)")
K, N, Kpad = 218, 16384, 224
n = K * N
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v = v.abs()
wp = torch.zeros(N * Kpad, device=dev, dtype=torch.bfloat16)
t_adam = timeit(lambda: ops.adam_step_zero(p, g, m, v, n, 1e-3, 0.5, 0.999, 1e-8, step, False, True))
t_pack = timeit(lambda: ops.pack_strided(dt, p, wp, N, K, Kpad, 1024, 1, 16, 16 * 1024))
t_fused = timeit(lambda: ops.adam_pack_rows(dt, p, g, m, v, wp, K, N, Kpad, 16, 1024, 1e-3, 0.5, 0.999, 1e-8, step, True))
print(f"rows {K}x{N} ({n / 1e6:.1f} M): adam {t_adam:.1f} us + pack_strided {t_pack:.1f} us = {t_adam + t_pack:.1f} us | fused {t_fused:.1f} us ({n * 34 / t_fused / 1e6:.2f} TB/s)")
