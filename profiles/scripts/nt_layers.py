"""Per-layer A/B of the NT implicit-GEMM variants on the CelebA B=128 bf16 launches (celebA/EAD-GAN_celebA.py:78-90,110-120).

Every launch of the step is one of six shapes x T tapes batched along M (all 34.4 GFLOP x T):
  A  fwd  M=32768T N=256  K=2048 (16 taps x 128)      D  bwd-data 4 phases M=2048T  N=512 K=4096 (4 taps x 1024)
  B  fwd  M=8192T  N=512  K=4096 (16 taps x 256)      E  bwd-data 4 phases M=8192T  N=256 K=2048 (4 taps x 512)
  C  fwd  M=2048T  N=1024 K=8192 (16 taps x 512)      F  bwd-data 4 phases M=32768T N=128 K=1024 (4 taps x 256)
Variants are forced per call (eg_epilogue.nt_variant / nt_splitk); rounds are interleaved in ONE process, inputs are uniform
random, each timing is HIP events around `inner` back-to-back launches.  Prints TFLOP/s (median and best round) and checks the
unsplit variants bit for bit against the register-staged kernel.

usage: python profiles/scripts/nt_layers.py [--shapes ABCDEF] [--T 1,2,3] [--configs v:s,v:s,...] [--rounds 5] [--dtype 1]
       config v:s = nt_variant : nt_splitk   (0:0 = planner; 2 = 128x128, 4 waves; 4 = 256x128, 8 waves (igemm_nt8s); 5 = 4 + input patch)
"""
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

SHAPES = {  # name: (kind, H, Cin, Cout)   conv k4 s2 p1 geometry; batch = 128 * T
    "A": ("fwd", 32, 128, 256), "B": ("fwd", 16, 256, 512), "C": ("fwd", 8, 512, 1024),
    "D": ("bwd", 8, 512, 1024), "E": ("bwd", 16, 256, 512), "F": ("bwd", 32, 128, 256),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="ABCDEF")
    ap.add_argument("--T", default="1,2,3")
    ap.add_argument("--configs", default="2:1,4:1,2:0,4:0,0:0")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inner", type=int, default=10)
    ap.add_argument("--dtype", type=int, default=1)
    ap.add_argument("--check", type=int, default=1)
    a = ap.parse_args()
    dt = a.dtype
    tdt = ops.torch_dtype(dt)
    dev = "cuda"
    ws = torch.zeros(96 << 20, device=dev, dtype=torch.float32)        # 384 MiB of split-K scratch (zeroed: its tail holds arrival counters)
    cfgs = [tuple(int(v) for v in c.split(":")) for c in a.configs.split(",")]
    lib = eg._lib.lib()
    g = torch.Generator(device=dev).manual_seed(1)
    print(f"{'shape':8s} " + " ".join(f"{'v%d:s%d' % c:>22s}" for c in cfgs))
    for name in a.shapes:
        kind, H, Cin, Cout = SHAPES[name]
        for T in [int(t) for t in a.T.split(",")]:
            B = 128 * T
            c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
            w = (torch.rand(Cout, Cin, 4, 4, device=dev, generator=g) - 0.5) * 0.1
            bias = torch.rand(Cout if kind == "fwd" else Cin, device=dev, generator=g) - 0.5
            sig = torch.rand(T, device=dev, generator=g) + 0.5
            if kind == "fwd":
                wp = torch.empty(ops.pack_fwd_elems(c, dt), device=dev, dtype=tdt)
                ops.pack_fwd(c, dt, w, wp)
                x = (torch.rand(B, H, H, Cin, device=dev, generator=g) * 2 - 1).to(tdt)
                y = torch.empty(B, H // 2, H // 2, Cout, device=dev, dtype=tdt)
                rows = B * (H // 2) ** 2 // T
                M, N, K, nph, C = B * (H // 2) ** 2, Cout, 16 * Cin, 1, Cin

                def run(v, s, out=y):
                    ops.conv_fwd(c, dt, x, wp, out, ops.epilogue(bias=bias, sigma=sig, sigma_rows=rows, act=ops.ACT_LRELU, slope=0.1,
                                                                 nt_variant=v, nt_splitk=s, splitk_ws=ws))
            else:
                wp = torch.empty(ops.pack_bwd_elems(c, dt), device=dev, dtype=tdt)
                ops.pack_bwd(c, dt, w, wp)
                x = (torch.rand(B, H // 2, H // 2, Cout, device=dev, generator=g) * 2 - 1).to(tdt)
                mask = (torch.rand(B, H, H, Cin, device=dev, generator=g) * 2 - 1).to(tdt)
                y = torch.empty(B, H, H, Cin, device=dev, dtype=tdt)
                rows = B * (H // 2) ** 2 // T
                M, N, K, nph, C = B * (H // 2) ** 2, Cin, 4 * Cout, 4, Cout

                def run(v, s, out=y):
                    ops.conv_bwd_data(c, dt, x, wp, out, ops.epilogue(sigma=sig, sigma_rows=rows, mask=mask, mask_act=ops.ACT_LRELU,
                                                                      mask_slope=0.1, nt_variant=v, nt_splitk=s, splitk_ws=ws))
            flops = 2.0 * M * N * K * nph
            labels = [lib.query("eg_igemm_nt_tile", ctypes.byref(c), dt, int(kind == "bwd"), v, s) for v, s in cfgs]
            if a.check:
                ref = torch.empty_like(y)
                run(1, 1, ref)
                for (v, s), lab in zip(cfgs, labels):
                    if lab < 0:
                        continue
                    out = torch.full_like(y, 7.0)
                    run(v, s, out)
                    torch.cuda.synchronize()
                    split = lab % 1000 in (132, 148, 149, 150, 152)  # 149: other K order
                    if split:
                        err = (out.float() - ref.float()).abs().max().item()
                        assert err < 0.05 * ref.float().abs().max().item() + 1e-3, (name, T, v, s, err)
                    else:
                        assert torch.equal(out, ref), (name, T, v, s, (out.float() - ref.float()).abs().max().item())
            times = {cfg: [] for cfg in cfgs}
            cfgs_ok = [cfg for cfg, lab in zip(cfgs, labels) if lab >= 0]
            for cfg in cfgs_ok:                                # warm-up
                run(*cfg)
            torch.cuda.synchronize()
            for _ in range(a.rounds):
                for cfg in cfgs_ok:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.inner):
                        run(*cfg)
                    e1.record()
                    e1.synchronize()
                    times[cfg].append(e0.elapsed_time(e1) * 1e-3 / a.inner)
            cells = []
            for cfg, lab in zip(cfgs, labels):
                if lab < 0:
                    cells.append("n/a")
                    continue
                med, best = statistics.median(times[cfg]), min(times[cfg])
                cells.append(f"{lab % 1000:3d} {flops / med / 1e12:6.0f}/{flops / best / 1e12:4.0f} {med * 1e6:6.1f}us")
            print(f"{name} T={T:<3d} " + " ".join(f"{c_:>22s}" for c_ in cells), flush=True)


if __name__ == "__main__":
    main()
