"""The NT kernels as a plain GEMM (1x1 convolution: M = B*H*W rows, K = Cin, N = Cout) on random bf16 data, back to back -- how far the K loop of
each variant gets when prologue, epilogue and round quantisation are amortised (the microarchitecture guide's GEMM numbers are for 8192^3).
usage: python profiles/scripts/gemm_square.py [--sizes 4096,8192] [--configs 4:1,2:1]"""
import argparse
import ctypes
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,8192")
    ap.add_argument("--configs", default="4:1,2:1")
    a = ap.parse_args()
    dt, dev = 1, "cuda"
    tdt = ops.torch_dtype(dt)
    lib = eg._lib.lib()
    g = torch.Generator(device=dev).manual_seed(1)
    for n in [int(v) for v in a.sizes.split(",")]:
        c = ops.make_conv(n // 1024, 32, 32, n, n, 1, 1, 0)          # M = n rows
        w = (torch.rand(n, n, 1, 1, device=dev, generator=g) - 0.5) * 0.05
        wp = torch.empty(ops.pack_fwd_elems(c, dt), device=dev, dtype=tdt)
        ops.pack_fwd(c, dt, w, wp)
        x = (torch.rand(n // 1024, 32, 32, n, device=dev, generator=g) * 2 - 1).to(tdt)
        y = torch.empty(n // 1024, 32, 32, n, device=dev, dtype=tdt)
        flops = 2.0 * n * n * n
        cells = []
        for v, s in [tuple(int(t) for t in cfg.split(":")) for cfg in a.configs.split(",")]:
            lab = lib.query("eg_igemm_nt_tile", ctypes.byref(c), dt, 0, v, s)
            if lab < 0:
                cells.append(f"v{v}: n/a")
                continue
            run = lambda: ops.conv_fwd(c, dt, x, wp, y, ops.epilogue(nt_variant=v, nt_splitk=s, splitk_ws=None))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e-3 / 5)
            t = statistics.median(ts)
            cells.append(f"v{v}({lab % 1000}): {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TFLOP/s")
        # one spot check against torch on a slice
        ref = (x.view(n, n)[:64].float() @ w.view(n, n).to(tdt).float().t())
        err = (y.view(n, n)[:64].float() - ref).abs().max().item() / ref.abs().max().item()
        print(f"M=N=K={n}  " + "   ".join(cells) + f"   (rel err of the last variant vs torch on 64 rows: {err:.1e})", flush=True)


if __name__ == "__main__":
    main()
