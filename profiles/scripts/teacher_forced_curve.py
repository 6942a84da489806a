"""Teacher-forced loss curve, CelebA fp32 (tests/test_gpu_celeba.py::teacher_forced_curve): N iterations, each started from the CPU
oracle's state; prints the largest |loss - oracle loss| per block of 100 iterations for g / d / info.
usage: python profiles/scripts/teacher_forced_curve.py [steps=1000] [B=4]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_celeba as tce      # noqa: E402

tce.setup_module(tce)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print(f"teacher-forced CelebA fp32, B={B}, {steps} iterations: max |loss - oracle| per block (g_loss, d_loss, info_loss)", flush=True)


def block(i, dev):
    if (i + 1) % 100 == 0 or i + 1 == steps:
        a = i // 100 * 100
        blk = dev[a:i + 1]
        print(f"  iterations {a:4d}..{i:4d}   max {blk.max(axis=0)[0]:.2e} {blk.max(axis=0)[1]:.2e} {blk.max(axis=0)[2]:.2e}   "
              f"median {np.median(blk, axis=0)[0]:.2e} {np.median(blk, axis=0)[1]:.2e} {np.median(blk, axis=0)[2]:.2e}", flush=True)


dev = tce.teacher_forced_curve(steps, B=B, progress=block)
print(f"overall max {dev.max():.3e} (bound 1e-3): {'PASS' if dev.max() < 1e-3 else 'FAIL'}")
