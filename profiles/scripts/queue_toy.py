"""How does hipGraph map captured stream forks onto HW queues?  Toy graph: a main chain of 8 kernels with three side chains forked from it;
every kernel works on a tensor of a distinct size, so the rocprofv3 kernel trace identifies it by grid size.
usage: rocprofv3 --kernel-trace --output-format csv -d out -- python3 profiles/scripts/queue_toy.py <variant>
variants: side_first | main_first | trampoline"""
import sys

import torch

variant = sys.argv[1]
dev = torch.device("cuda")
SZ = lambda i: 256 * 1024 * (i + 1)                       # main kernels: grid ~ (i+1) * 256 blocks of 1024 elements
main_t = [torch.zeros(SZ(i), device=dev) for i in range(8)]
side_t = {c: [torch.zeros(256 * 1024 * (20 + 10 * k + j), device=dev) for j in range(3)] for k, c in enumerate("XYZ")}
tr_t = [torch.zeros(1024 * (3 + i), device=dev) for i in range(8)]
S = {c: torch.cuda.Stream(dev) for c in "XYZ"}
main = torch.cuda.Stream(dev)


def rec(s=None):
    e = torch.cuda.Event()
    e.record(s) if s is not None else e.record()
    return e


torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    with torch.cuda.graph(g, stream=main):
        pend = []
        tri = 0
        for i in range(8):
            main_t[i].add_(1.0)
            c = {1: "X", 3: "Y", 5: "Z"}.get(i)
            if variant != "side_first":
                for (cc, e) in pend:                       # issue the side chain only now: the main stream's next kernel exists already
                    s = S[cc]
                    if variant == "trampoline" and cc in "YZ":
                        # depth control: Y hangs off a trampoline node on X's stream, Z off one on Y's stream
                        host = S["X"] if cc == "Y" else S["Y"]
                        host.wait_event(e)
                        with torch.cuda.stream(host):
                            tr_t[tri].add_(1.0); e_t = rec(); tr_t[tri + 1].add_(1.0)
                        tri += 2
                        s.wait_event(e_t)
                    else:
                        s.wait_event(e)
                    with torch.cuda.stream(s):
                        for t in side_t[cc]:
                            t.add_(1.0)
                pend = []
            if c:
                if variant == "side_first":
                    S[c].wait_event(rec())
                    with torch.cuda.stream(S[c]):
                        for t in side_t[c]:
                            t.add_(1.0)
                else:
                    pend.append((c, rec()))
        for s in S.values():
            main.wait_event(rec(s))
        main_t[0].add_(1.0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("done", variant)
