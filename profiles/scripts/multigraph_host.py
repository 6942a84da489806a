"""Host-side timing of the CelebA iteration replayed as several hipGraphs (engine.MultiGraph): how long each hipGraphLaunch takes on the host
and how far the host runs ahead of the GPU.  usage: python profiles/scripts/multigraph_host.py"""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
eg = importlib.import_module("ead-gan_amd")
dev = torch.device("cuda")
B = 128
torch.manual_seed(0)
G = eg.celeba.Generator(dtype="bf16").to(dev)
D = eg.celeba.Discriminator(dtype="bf16").to(dev)
tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16")
g = torch.Generator(device=dev).manual_seed(1)
tr.load_inputs(torch.rand((B, 3, 64, 64), device=dev, generator=g) * 2 - 1, torch.randn((B, 200), device=dev, generator=g),
               torch.rand((B, 8), device=dev, generator=g) * 2 - 1, torch.randint(0, 10, (B,), device=dev, generator=g))
tr.inputs = eg.celeba.DeviceInputs(torch.randint(0, 256, (4096, 3, 64, 64), device=dev, dtype=torch.uint8, generator=g), seed=1)
tr.step_resident()
tr.capture(inputs=tr.inputs)
mg = tr.graph
for _ in range(5):
    tr.step_resident()
torch.cuda.synchronize()
if not hasattr(mg, "segments"):
    raise SystemExit("one hipGraph (EG_MULTI_GRAPH=0?)")
n = 30
acc = [0.0] * len(mg.segments)
main = torch.cuda.current_stream()
t_all0 = time.perf_counter()
for _ in range(n):
    streams = (main, mg.second)
    for i, (gr, s, after) in enumerate(mg.segments):
        st = streams[s]
        for j in after:
            st.wait_event(mg._done[j])
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            gr.replay()
        acc[i] += time.perf_counter() - t0
        mg._done[i].record(st)
    main.wait_event(mg._done[1])
t_enq = time.perf_counter() - t_all0
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all0
print(f"host enqueue {t_enq / n * 1e3:.3f} ms/iter, GPU drained {t_all / n * 1e3:.3f} ms/iter")
for i, a in enumerate(acc):
    print(f"segment {i} (stream {mg.segments[i][1]}): hipGraphLaunch {a / n * 1e6:.0f} us on the host")

# GPU-side: when does each segment start and end relative to the end of the first segment (timing events on the segments' streams)?
T = lambda: torch.cuda.Event(enable_timing=True)
recs = []
for it in range(12):
    streams = (main, mg.second)
    ev = {}
    for i, (gr, s, after) in enumerate(mg.segments):
        st = streams[s]
        for j in after:
            st.wait_event(mg._done[j])
        with torch.cuda.stream(st):
            b, e = T(), T()
            b.record()
            gr.replay()
            e.record()
        ev[i] = (b, e)
        mg._done[i].record(st)
    main.wait_event(mg._done[1])
    recs.append(ev)
torch.cuda.synchronize()
for it in (5, 8, 11):
    ev = recs[it]
    a_end = ev[0][1]
    print("iteration", it, " ".join(f"seg{i}: {a_end.elapsed_time(ev[i][0]) * 1e3:+7.0f}..{a_end.elapsed_time(ev[i][1]) * 1e3:+7.0f} us" for i in range(1, len(mg.segments))),
          f"(segment 0 took {ev[0][0].elapsed_time(ev[0][1]) * 1e3:.0f} us)")
