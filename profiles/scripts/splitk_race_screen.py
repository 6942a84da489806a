"""Race screen of igemm_nt8s's in-kernel K-split reduction (cross-workgroup hand-off through sc1 stores / loads and an arrival counter): the same
launch repeated many times, every output compared bit for bit with the first one and with the two-launch path (EG semantics: partial tiles +
epilogue kernel are a different summation ORDER only when nsplit > 2, so the reference here is the first in-kernel result), with other GEMMs
interleaved to vary the timing.  usage: python profiles/scripts/splitk_race_screen.py [--reps 300]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
eg = importlib.import_module("ead-gan_amd")
ops = eg.ops

SHAPES = {"B": ("fwd", 16, 256, 512), "C": ("fwd", 8, 512, 1024), "D": ("bwd", 8, 512, 1024)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    a = ap.parse_args()
    dt, dev = 1, "cuda"
    tdt = ops.torch_dtype(dt)
    ws = torch.zeros(48 << 20, device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev).manual_seed(5)
    bad_total = 0
    for name, (kind, H, Cin, Cout) in SHAPES.items():
        for splits in (2, 4, 8):
            B = 128
            c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
            w = (torch.rand(Cout, Cin, 4, 4, device=dev, generator=g) - 0.5) * 0.1
            if kind == "fwd":
                wp = torch.empty(ops.pack_fwd_elems(c, dt), device=dev, dtype=tdt)
                ops.pack_fwd(c, dt, w, wp)
                x = (torch.rand(B, H, H, Cin, device=dev, generator=g) * 2 - 1).to(tdt)
                y = torch.empty(B, H // 2, H // 2, Cout, device=dev, dtype=tdt)
                bias = torch.rand(Cout, device=dev, generator=g) - 0.5
                run = lambda out: ops.conv_fwd(c, dt, x, wp, out, ops.epilogue(bias=bias, act=ops.ACT_LRELU, slope=0.1, nt_variant=4, nt_splitk=splits, splitk_ws=ws))
            else:
                wp = torch.empty(ops.pack_bwd_elems(c, dt), device=dev, dtype=tdt)
                ops.pack_bwd(c, dt, w, wp)
                x = (torch.rand(B, H // 2, H // 2, Cout, device=dev, generator=g) * 2 - 1).to(tdt)
                y = torch.empty(B, H, H, Cin, device=dev, dtype=tdt)
                mask = (torch.rand(B, H, H, Cin, device=dev, generator=g) * 2 - 1).to(tdt)
                run = lambda out: ops.conv_bwd_data(c, dt, x, wp, out, ops.epilogue(mask=mask, mask_act=ops.ACT_LRELU, mask_slope=0.1, nt_variant=4, nt_splitk=splits, splitk_ws=ws))
            ref = torch.empty_like(y)
            run(ref)
            torch.cuda.synchronize()
            bad = torch.zeros(1, device=dev, dtype=torch.int64)
            filler_a = torch.randn(2048, 2048, device=dev)
            for r in range(a.reps):
                out = torch.full_like(y, 3.0)
                if r % 3 == 1:
                    filler_a @ filler_a                 # another kernel in front: different arrival pattern of the workgroups
                run(out)
                bad += (out != ref).any().to(torch.int64)
            torch.cuda.synchronize()
            cnt = int(ws[-1024:].view(torch.int32).abs().sum())
            print(f"{name} {kind} splits {splits}: {int(bad)} of {a.reps} launches differ; counters nonzero afterwards: {cnt}", flush=True)
            bad_total += int(bad) + cnt
    print("RACE SCREEN", "PASS" if bad_total == 0 else "FAIL")
    sys.exit(0 if bad_total == 0 else 1)


if __name__ == "__main__":
    main()
