"""The image-side layer as a 1x1 convolution over im2col patches (K = 48 padded to 64, N = 128; celebA/EAD-GAN_celebA.py:110 first D conv):
forced NT variants back to back.  usage: python profiles/scripts/img_layer.py [--T 1,2,3]"""
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
    ap.add_argument("--T", default="1,2,3")
    ap.add_argument("--configs", default="0:0,3:1,4:1,2:1,1:1")
    a = ap.parse_args()
    dt, dev = 1, "cuda"
    tdt = ops.torch_dtype(dt)
    lib = eg._lib.lib()
    g = torch.Generator(device=dev).manual_seed(1)
    cfgs = [tuple(int(v) for v in c.split(":")) for c in a.configs.split(",")]
    for T in [int(t) for t in a.T.split(",")]:
        B = 128 * T
        c = ops.make_conv(B, 32, 32, 64, 128, 1, 1, 0)
        w = (torch.rand(128, 64, 1, 1, device=dev, generator=g) - 0.5) * 0.1
        wp = torch.empty(ops.pack_fwd_elems(c, dt), device=dev, dtype=tdt)
        ops.pack_fwd(c, dt, w, wp)
        x = (torch.rand(B, 32, 32, 64, device=dev, generator=g) * 2 - 1).to(tdt)
        bias = torch.rand(128, device=dev, generator=g) - 0.5
        sig = torch.rand(T, device=dev, generator=g) + 0.5
        y = torch.empty(B, 32, 32, 128, device=dev, dtype=tdt)
        ref = None
        cells = []
        for v, s in cfgs:
            lab = lib.query("eg_igemm_nt_tile", ctypes.byref(c), dt, 0, v, s)
            if lab < 0:
                cells.append(f"v{v}: n/a")
                continue
            run = lambda: ops.conv_fwd(c, dt, x, wp, y, ops.epilogue(bias=bias, sigma=sig, sigma_rows=B * 1024 // T, act=ops.ACT_LRELU, slope=0.1, nt_variant=v, nt_splitk=s))
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            else:
                assert torch.equal(y, ref), (v, (y.float() - ref.float()).abs().max())
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 100.0)
            us = statistics.median(ts)
            mb = (x.numel() + y.numel()) * 2 / 1e6
            cells.append(f"v{v}({lab % 1000}): {us:6.1f} us {mb / us * 1e-6 * 1e6 / 1e6:5.2f} TB/s")
        print(f"T={T}  " + "   ".join(cells), flush=True)


if __name__ == "__main__":
    main()
