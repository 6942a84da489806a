"""Do hipGraphs launched on two streams overlap?  A (stream 1) -> B (stream 2, behind A) and C (stream 1, behind A by stream order) -> D.
Kernels: elementwise passes over 64 MB (about 40 us each).  Prints when C starts relative to B's end, for B as ONE chain and for B as TWO
branches (fork / join inside the capture).  usage: python profiles/scripts/graph_streams_toy.py"""
import torch

dev = "cuda"
xs = [torch.ones(16 << 20, device=dev) for _ in range(8)]


def k(i, n=1):
    for _ in range(n):
        xs[i].mul_(1.0001)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin()
        fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    return g


side = torch.cuda.Stream()


def b_two_branches():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    k(2, 4)                                     # branch 1 (created first): 4 kernels
    with torch.cuda.stream(side):
        k(3, 8)                                 # branch 2: 8 kernels
        e2 = torch.cuda.Event(); e2.record()
    cur.wait_event(e2)


def a_branches():
    cur = torch.cuda.current_stream()
    k(0, 1)
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        k(4, 2)
        e2 = torch.cuda.Event(); e2.record()
    k(0, 2)
    cur.wait_event(e2)


CASES = (("A chain,    B chain of 4        ", lambda: k(0, 3), lambda: k(2, 4)),
         ("A chain,    B two branches (4|8)", lambda: k(0, 3), b_two_branches),
         ("A branches, B chain of 8        ", a_branches, lambda: k(2, 8)),
         ("A branches, no B                ", a_branches, None))
for name, afn, bfn in CASES:
    for f in (afn, bfn, lambda: k(1, 1), lambda: k(0, 1)):
        if f is not None:
            f()
    torch.cuda.synchronize()
    gA, gB, gC, gD = capture(afn), (capture(bfn) if bfn else None), capture(lambda: k(1, 6)), capture(lambda: k(0, 2))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    T = lambda: torch.cuda.Event(enable_timing=True)
    recs = []
    torch.cuda.synchronize()
    for rep in range(8):                             # back to back, the host runs ahead (as a training loop does)
        t0, a1, b0, b1, c0, c1, d1 = T(), T(), T(), T(), T(), T(), T()
        recs.append((t0, a1, b0, b1, c0, c1, d1))
        with torch.cuda.stream(s1):
            t0.record(); gA.replay(); a1.record()
        s2.wait_event(a1)
        with torch.cuda.stream(s2):
            b0.record()
            if gB is not None:
                gB.replay()
            b1.record()
        with torch.cuda.stream(s1):
            c0.record(); gC.replay(); c1.record()
            s1.wait_event(b1)
            gD.replay(); d1.record()
    torch.cuda.synchronize()
    for rep in (1, 4, 7):
        t0, a1, b0, b1, c0, c1, d1 = recs[rep]
        e = lambda ev: t0.elapsed_time(ev) * 1e3
        print(f"{name}: A ends {e(a1):6.0f}  B {e(b0):6.0f}..{e(b1):6.0f}  C {e(c0):6.0f}..{e(c1):6.0f}  D ends {e(d1):6.0f} us")
