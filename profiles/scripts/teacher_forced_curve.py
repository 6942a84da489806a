"""Teacher-forced loss curves, fp32: N iterations of a loop, each started from the CPU oracle's state (tests/test_gpu_celeba.py::
teacher_forced_curve for CelebA, tests/teacher_forced.py for the other loops); prints the largest |loss - oracle loss| per block of 100
iterations.  usage: python profiles/scripts/teacher_forced_curve.py [steps=1000] [B=4] [family=celeba|mnist|dsprites|colored]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
family = sys.argv[3] if len(sys.argv) > 3 else "celeba"


def block(i, dev):
    if (i + 1) % 100 == 0 or i + 1 == steps:
        a = i // 100 * 100
        blk = dev[a:i + 1]
        print(f"  iterations {a:4d}..{i:4d}   max " + " ".join(f"{x:.2e}" for x in blk.max(axis=0)) + "   median " + " ".join(f"{x:.2e}" for x in np.median(blk, axis=0)), flush=True)


if family == "celeba":
    import test_gpu_celeba as tce      # noqa: E402
    tce.setup_module(tce)
    names = ("g_loss", "d_loss", "info_loss")
    print(f"teacher-forced CelebA fp32, B={B}, {steps} iterations: max |loss - oracle| per block {names}", flush=True)
    dev = tce.teacher_forced_curve(steps, B=B, progress=block)
else:
    import torch
    import teacher_forced             # noqa: E402
    torch.set_num_threads(16)
    print(f"teacher-forced {family} fp32, B={B}, {steps} iterations: max |loss - oracle| per block", flush=True)
    dev, names = teacher_forced.curve(family, steps, B=B, progress=block)
    print("  losses:", names)
print(f"overall max {dev.max():.3e} (bound 1e-3): {'PASS' if dev.max() < 1e-3 else 'FAIL'}")
