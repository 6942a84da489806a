"""Which stream/event topologies does hipGraph capture accept on this ROCm?  Each pattern runs in its own child process (a crash in
hipStreamEndCapture is a segfault, not an exception).  usage: python profiles/scripts/capture_patterns.py"""
import subprocess
import sys

PATTERNS = ["fork_join", "lane_to_lane", "delayed_wait", "gather", "lane_waits_opt_after_gather", "fresh_lane_waits_opt", "ungathered_lane_waits_opt", "two_gathers", "full_with_prep_lane", "full"]

CHILD = r'''
import sys, torch
pat = sys.argv[1]
dev = torch.device("cuda")
x = [torch.zeros(1 << 16, device=dev) for _ in range(4)]
main = torch.cuda.Stream(dev)
A, B, O, P = (torch.cuda.Stream(dev) for _ in range(4))
def rec(s=None):
    e = torch.cuda.Event()
    e.record(s) if s is not None else e.record()
    return e
def work(i):
    x[i].add_(1.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    with torch.cuda.graph(g, stream=main):
        work(0)
        if pat == "fork_join":
            A.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            main.wait_event(rec(A))
        elif pat == "lane_to_lane":
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1); ea = rec()
            with torch.cuda.stream(B): B.wait_event(ea); work(2)
            main.wait_event(rec(A)); main.wait_event(rec(B))
        elif pat == "delayed_wait":
            A.wait_event(rec())
            with torch.cuda.stream(A): work(1); ea = rec(); work(2)
            work(3)
            main.wait_event(ea)
            work(1)
            main.wait_event(rec(A))
        elif pat == "gather":
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3)
            for s in (A, B, O): main.wait_event(rec(s))
        elif pat == "lane_waits_opt_after_gather":
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3); eo = rec(); work(3)
            B.wait_event(rec())
            with torch.cuda.stream(B): B.wait_event(eo); work(2)
            for s in (A, B, O): main.wait_event(rec(s))
        elif pat == "fresh_lane_waits_opt":
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3); eo = rec(); work(3)
            P.wait_event(rec())
            with torch.cuda.stream(P): P.wait_event(eo); work(2)
            for s in (A, B, O, P): main.wait_event(rec(s))
        elif pat == "ungathered_lane_waits_opt":
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A))
            with torch.cuda.stream(O): work(3); eo = rec(); work(3)
            with torch.cuda.stream(B): B.wait_event(eo); work(2)
            for s in (A, B, O): main.wait_event(rec(s))
        elif pat == "two_gathers":
            for k in range(2):
                A.wait_event(rec()); B.wait_event(rec())
                with torch.cuda.stream(A): work(1)
                with torch.cuda.stream(B): work(2)
                O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
                with torch.cuda.stream(O): work(3)
                work(0)
            for s in (A, B, O): main.wait_event(rec(s))
        elif pat == "full_with_prep_lane":
            P.wait_event(rec())
            with torch.cuda.stream(P): work(2); ep = rec()
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3); eg_ = rec()
            main.wait_event(ep); work(0)
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3); ed = rec(); work(3)
            P.wait_event(rec())
            with torch.cuda.stream(P): P.wait_event(ed); work(2)
            main.wait_event(eg_); work(0)
            for s in (A, B, O, P): main.wait_event(rec(s))
            work(0)
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3)
            A.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3)
            for s in (A, B, O, P): main.wait_event(rec(s))
        elif pat == "full":
            # the pipelined step's skeleton: prepare on B with a mark, gather to O with a mark, main waits marks late, second gather
            B.wait_event(rec())
            with torch.cuda.stream(B): work(2); ep = rec()
            A.wait_event(rec()); B.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            with torch.cuda.stream(B): work(2)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3); eg_ = rec()
            main.wait_event(ep); work(0)
            A.wait_event(rec())
            with torch.cuda.stream(A): work(1)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3); ed = rec(); work(3)
            B.wait_event(rec())
            with torch.cuda.stream(B): B.wait_event(ed); work(2)
            main.wait_event(eg_); work(0)
            for s in (A, B, O): main.wait_event(rec(s))
            work(0)
            O.wait_event(rec()); O.wait_event(rec(A)); O.wait_event(rec(B))
            with torch.cuda.stream(O): work(3)
            for s in (A, B, O): main.wait_event(rec(s))
        work(0)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
print("OK", pat, [float(t[0]) for t in x], flush=True)
'''

for pat in PATTERNS:
    r = subprocess.run([sys.executable, "-c", CHILD, pat], capture_output=True, text=True, timeout=120)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Fatal" in l][:2]
    print(f"{pat:32s} rc={r.returncode:4d}  {tail}  {err}", flush=True)
