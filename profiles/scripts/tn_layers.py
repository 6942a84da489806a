"""Weight-gradient GEMM timings on the CelebA B=128 layers (conv k4 s2 p1; T tapes batched along M): TFLOP/s of eg_conv_wgrad alone and
with its slab reduction.  EG_TN8=0 in the environment selects the per-tap kernel (igemm_tn_kernel) for an A/B in two processes.
usage: python profiles/scripts/tn_layers.py [--T 1,2,3]"""
import argparse
import importlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops
LAYERS = [(32, 128, 256), (16, 256, 512), (8, 512, 1024)]     # H (input), Cin, Cout: discriminator layers 2-4 == generator layers in conv view


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", default="1,2,3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inner", type=int, default=10)
    a = ap.parse_args()
    dt, dev = 1, "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    print(f"EG_TN8={os.environ.get('EG_TN8', '1')}")
    for H, Cin, Cout in LAYERS:
        for T in [int(t) for t in a.T.split(",")]:
            B = 128 * T
            c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
            x = (torch.rand(B, H, H, Cin, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
            dy = (torch.rand(B, H // 2, H // 2, Cout, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
            slab = torch.empty(ops.conv_wgrad_ws_bytes(c, dt) // 4, device=dev)
            grad = torch.zeros(Cout, Cin, 4, 4, device=dev)
            flops = 2.0 * B * (H // 2) ** 2 * Cout * Cin * 16
            res = {}
            for what in ("gemm", "gemm+reduce"):
                ts = []
                for r in range(a.rounds + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.inner):
                        ns = ops.conv_wgrad(c, dt, x, dy, slab)
                        if what != "gemm":
                            ops.wgrad_reduce(slab, ns, Cout, Cout, Cin, 16, grad, accumulate=True)
                    e1.record()
                    e1.synchronize()
                    if r:
                        ts.append(e0.elapsed_time(e1) * 1e-3 / a.inner)
                res[what] = statistics.median(ts)
            print(f"Cin {Cin:4d} Cout {Cout:4d} H {H:2d} T={T}  splits {ns:3d}  gemm {res['gemm'] * 1e6:7.1f} us = {flops / res['gemm'] / 1e12:6.0f} TF/s   "
                  f"with reduce {res['gemm+reduce'] * 1e6:7.1f} us = {flops / res['gemm+reduce'] / 1e12:6.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
