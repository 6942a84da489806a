"""Toy 3: does a wait on an OLD lane event pick up work captured on that lane LATER?  Main chain M0..M9; lane L runs L1 behind M1 and an
event E is recorded behind L1; behind M5 the lane gets a long chain L2 (4 kernels); main waits for E in front of M7.
  variant a: L2 is captured BEFORE main's wait for E      variant b: L2 is captured AFTER M7..M9 (same dependencies)
In both, M7 may start as soon as M6 is done (L1 finished long ago).  The trace shows whether it does.
usage: rocprofv3 --kernel-trace --output-format csv -d out -- python3 profiles/scripts/queue_toy3.py <a|b>"""
import sys

import torch

variant = sys.argv[1]
dev = torch.device("cuda")
N = 16 * 1024 * 1024
main_t = [torch.zeros(N + 1024 * i, device=dev) for i in range(10)]
lane_t = [torch.zeros(2 * N + 1024 * i, device=dev) for i in range(5)]
L = torch.cuda.Stream(dev)
main = torch.cuda.Stream(dev)


def rec(s=None):
    e = torch.cuda.Event()
    e.record(s) if s is not None else e.record()
    return e


torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    with torch.cuda.graph(g, stream=main):
        E = None
        fork5 = None
        for i in range(10):
            if i == 7:
                if variant == "a":
                    L.wait_event(fork5)
                    with torch.cuda.stream(L):
                        for k in range(1, 5):
                            lane_t[k].add_(1.0)
                main.wait_event(E)
            main_t[i].add_(1.0)
            if i == 1:
                L.wait_event(rec())
                with torch.cuda.stream(L):
                    lane_t[0].add_(1.0)
                E = rec(L)
            if i == 5:
                fork5 = rec()
        if variant == "b":
            L.wait_event(fork5)
            with torch.cuda.stream(L):
                for k in range(1, 5):
                    lane_t[k].add_(1.0)
        main.wait_event(rec(L))
        main_t[0].add_(1.0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
