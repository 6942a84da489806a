"""Toy 2: persistent lanes.  Main chain of 12 kernels; lane X gets a chain at main kernels 1, 4, 7, lane Y at 2, 5, 8; main joins X at kernel 6
(mid-graph) and everything at the end.  Variants:
  plain      : side-first at every fork (what the trainer did)
  once       : side-first only at a lane's FIRST fork, main-first afterwards
  once_noop  : like `once`, plus a no-op kernel on the lane right after every event main waits for (so the main join node is not the lane
               tail's first child)
usage: rocprofv3 --kernel-trace --output-format csv -d out -- python3 profiles/scripts/queue_toy2.py <variant>"""
import sys

import torch

variant = sys.argv[1]
dev = torch.device("cuda")
main_t = [torch.zeros(256 * 1024 * (i + 1), device=dev) for i in range(12)]
side_t = {"X": [torch.zeros(256 * 1024 * (40 + i), device=dev) for i in range(6)], "Y": [torch.zeros(256 * 1024 * (60 + i), device=dev) for i in range(6)]}
noop_t = torch.zeros(1024, device=dev)
S = {c: torch.cuda.Stream(dev) for c in "XY"}
main = torch.cuda.Stream(dev)
used = {"X": 0, "Y": 0}
entered = set()


def rec(s=None):
    e = torch.cuda.Event()
    e.record(s) if s is not None else e.record()
    return e


def chain(c):
    with torch.cuda.stream(S[c]):
        for _ in range(2):
            side_t[c][used[c]].add_(1.0)
            used[c] += 1


def lane_event(c):
    e = rec(S[c])
    if variant == "once_noop":
        with torch.cuda.stream(S[c]):
            noop_t.add_(1.0)
    return e


torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    with torch.cuda.graph(g, stream=main):
        pend = []
        for i in range(12):
            if i == 6:
                main.wait_event(lane_event("X"))
            main_t[i].add_(1.0)
            for (c, e) in pend:
                S[c].wait_event(e)
                chain(c)
            pend = []
            c = {1: "X", 4: "X", 7: "X", 2: "Y", 5: "Y", 8: "Y"}.get(i)
            if c:
                if variant == "plain" or c not in entered:
                    entered.add(c)
                    S[c].wait_event(rec())
                    chain(c)
                else:
                    pend.append((c, rec()))
        for c in "XY":
            main.wait_event(lane_event(c))
        main_t[0].add_(1.0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
