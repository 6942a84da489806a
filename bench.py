#!/usr/bin/env python3
"""Headline benchmark: imgs/sec of one full EAD-GAN CelebA train iteration (G adversarial step + D step + info/affine
step, three Adams; celebA/EAD-GAN_celebA.py:299-401) at 64x64, batch 128 per GPU, bf16 MFMA compute with fp32 master
weights, synthetic data, on N MI355X of one node (weak scaling, RCCL all-reduce of the gradient arenas).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : dominant kernel (by time) of the step, algorithmic FLOPs per launch / measured launch duration
                 (HIP events on the launch stream, in this process) against the dense bf16 MFMA peak; `algorithmic_bytes` per
                 launch (inputs + weights + outputs once); `traffic` = HBM bytes per launch from the committed PMC passes
                 (`traffic_source` names the file: a profile of this code, not a measurement of this run);
  cpu_baseline : the CPU oracle (oracle/celeba_oracle.py, a port pinned to the reference) timed on the host cores
                 on a bounded sample of the same workload (rank 0, N=1 only): median of 5 iterations after 2 warm-ups.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

GFLOP_PER_IMG = 18.80          # algorithmic FLOPs per image per iteration, dead work excluded
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0          # HBM3E peak (spec), MI355X_MICROARCH.md; ~6300 GB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step (BASELINE config: 128)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying one hipGraph")
    ap.add_argument("--graph-dist", action="store_true", help="N > 1: capture the iteration (RCCL collectives included) into one hipGraph.  Default at "
                    "N > 1 is eager launches: the captured multi-rank path could only be rehearsed with a 1-rank group on the build's one-GPU box, "
                    "a capture that fails in the runtime cannot always be caught from Python, and eager costs 2 %% (4.73 vs 4.63 ms at N = 1)")
    ap.add_argument("--no-probe", action="store_true", help="skip the child-process capture probe (see decide_graph) and capture directly; a capture "
                    "failure then ends this process with a non-zero exit code")
    ap.add_argument("--probe-capture", action="store_true", help=argparse.SUPPRESS)     # internal: this process IS the probe child
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--force-dist", action="store_true", help="initialise a process group and run the all-reduce hooks even with one rank "
                                                              "(rehearses the N>1 code path, incl. RCCL capture into the hipGraph, on one GPU)")
    ap.add_argument("--resident-inputs", action="store_true", help="replay ONE pre-loaded batch instead of drawing every batch on the device "
                    "inside the captured step (default: uint8 dataset in HBM + counter-based z / code / labels, i.e. the timed iteration includes "
                    "the whole input pipeline and every step sees new inputs)")
    ap.add_argument("--sync-bn", action="store_true", help="data parallel: BatchNorm statistics over the global batch (dp.SyncBN); default per-rank")
    ap.add_argument("--wire", default="f32", choices=["f32", "bf16"], help="data parallel: element type of the gradient buckets on the links")
    ap.add_argument("--no-overlap", action="store_true", help="single stream: no side lanes for weight-gradient chains / re-packing")
    ap.add_argument("--workload", default="celeba", choices=["celeba", "mnist", "dsprites", "colored", "pxy"],
                    help="celeba = the headline metric (default); mnist = BASELINE config[1] (use --batch 256 --dtype f32); "
                         "dsprites = config[2] (--batch 128); colored = config[4] (--batch 512)")
    return ap.parse_args()


def roofline_pass(eg, trainer, dtype, workload="celeba", iters=3):
    """Runs `iters` eager iterations with every implicit-GEMM launch bracketed by HIP events on the launch stream (the
    launches sit in their real place in the step, so cache state is the real one) and returns the per-kernel table +
    the roofline object of the dominant kernel (largest total time)."""
    ops = eg.ops
    # single-stream pass: the timed region runs the weight-gradient chains on side streams beside the backward-data GEMMs; a launch
    # timed while another GEMM shares the GPU measures the sharing, not the kernel, so the side streams are folded into the main
    # stream here (same kernels, same arguments, same order inside every chain).  `--no-overlap` runs the whole bench that way:
    # profiles/ holds the rocprofv3 summaries of both commands; the per-kernel averages of the --no-overlap one agree with this table.
    # Rank 0 runs this pass alone: no collectives inside it (the other ranks are not calling them) -- gradient all-reduce AND
    # synchronised BatchNorm are switched off for its duration.
    saved = {k: getattr(trainer, k, None) for k in ("side", "allreduce", "sync_bn", "overlap")}      # overlap: the small-network trainers' two chains
    for k in saved:
        if hasattr(trainer, k):
            setattr(trainer, k, None)
    ops.RECORDER = []
    try:
        for _ in range(iters):
            trainer._step_body()
        torch.cuda.synchronize()
    finally:
        rec, ops.RECORDER = ops.RECORDER, None
        for k, v in saved.items():
            if hasattr(trainer, k):
                setattr(trainer, k, v)
    table, detail = {}, {}
    for label, flops, e0, e1, shape, nbytes in rec:
        ms = e0.elapsed_time(e1)
        t = table.setdefault(label, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        t["launches"] += 1
        t["ms"] += ms
        t["flops"] += flops
        t["bytes"] += nbytes
        d = detail.setdefault((label, shape), [0, 0.0, flops])
        d[0] += 1
        d[1] += ms
    for t in table.values():
        for k in ("launches", "ms", "flops", "bytes"):
            t[k] /= iters
    if os.environ.get("EG_BENCH_DETAIL"):
        for (label, shape), (n, ms, fl) in sorted(detail.items(), key=lambda kv: -kv[1][1]):
            print(f"# {label:34s} {shape:44s} x{n / iters:4.1f}  {ms / n * 1e3:8.1f} us/launch  {fl / (ms / n * 1e-3) / 1e12:7.1f} TF/s  total {ms / iters:6.3f} ms/iter", file=sys.stderr)
    if not table:
        return None, table
    dom = max(table, key=lambda k: table[k]["ms"])
    d = table[dom]
    peak = PEAK_F32_TFLOPS if dtype == "f32" else PEAK_BF16_TFLOPS        # f16 and bf16 MFMA run at the same rate
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    # HBM bytes per launch: NOT measured in this run -- read from the committed PMC passes of this code (rocprofv3 --pmc, FETCH_SIZE and
    # WRITE_SIZE in separate passes, gfx950 corrections per the microarchitecture guide), if this kernel is in them
    traffic = source = None
    for rnd in ("r03", "r02"):                           # PMC passes are per workload; the newest committed one that lists this kernel
        name = f"{rnd}_pmc_traffic.json" if workload == "celeba" else f"{rnd}_pmc_traffic_{workload}.json"
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except OSError:
            continue
        if dom in pmc:
            traffic, source = pmc[dom].get("hbm_bytes_per_launch"), "profiles/" + name
            break
    alg_bytes = d["bytes"] / d["launches"]
    avg_us = d["ms"] * 1e3 / d["launches"]
    common = {"kernel": dom, "mode": "single-stream eager pass", "traffic": traffic, "traffic_source": source, "algorithmic_bytes": round(alg_bytes),
              "launches_per_step": round(d["launches"], 1), "avg_launch_us": round(avg_us, 2), "gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3)}
    # which roof bounds the kernel: its algorithmic intensity (FLOP per byte moved once) against the machine balance peak FLOP/s / 8 TB/s.
    # The 64-channel launches of the small networks sit far below it (7-45 FLOP/B against ~300): they are priced against HBM, and at a
    # few microseconds per launch mostly against launch latency -- said so in `note`
    intensity = d["flops"] / max(d["bytes"], 1.0)
    if intensity >= peak * 1e12 / (PEAK_HBM_GBS * 1e9):
        roof = dict(common, bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4))
    else:
        gbs = alg_bytes / (avg_us * 1e-6) / 1e9
        roof = dict(common, bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                    note=f"{intensity:.0f} FLOP per algorithmic byte (machine balance {peak * 1e12 / (PEAK_HBM_GBS * 1e9):.0f}); {achieved:.1f} TFLOP/s; "
                         f"launches of {avg_us:.1f} us are launch-latency bound rather than bandwidth bound")
    return roof, table


def kernel_table(table):
    return {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} for k, v in table.items()}


def want_graph(a, world):
    """one GPU (and the 1-rank --force-dist rehearsal): hipGraph replay unless --no-graph; several ranks: eager unless --graph-dist"""
    if a.no_graph:
        return False
    return world == 1 or a.graph_dist


def decide_graph(a, world):
    """hipGraph replay or eager launches?  Decided BEFORE this process touches the GPU.  A failed hipGraph capture cannot be recovered from
    inside the process on ROCm 7.2 (engine.CaptureFailed: streams forked into the invalidated capture crash the runtime later), so when a
    graph is wanted the same command is first run as a child process with --probe-capture: it builds the trainer, runs one eager
    iteration, captures, replays twice and exits 0 -- or prints the reason and exits non-zero.  Only after a clean probe does this process
    capture; otherwise it launches eagerly and says so in config.workload.  (N > 1 with --graph-dist: every rank probes with its own
    child, the children form their own process group on MASTER_PORT + 1, and the ranks agree on the minimum of their verdicts.)
    Returns (use_graph, note)."""
    if not want_graph(a, world):
        return False, ""
    if a.no_probe or a.probe_capture:
        return True, ""
    import subprocess
    env = dict(os.environ)
    if "MASTER_PORT" in env:
        env["MASTER_PORT"] = str(int(env["MASTER_PORT"]) + 1)
    argv = [x for x in sys.argv[1:] if x not in ("--no-cpu-baseline", "--no-roofline")]
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--probe-capture", "--no-cpu-baseline", "--no-roofline"]
    try:
        rc = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, timeout=600).returncode
    except subprocess.TimeoutExpired:
        rc = -999
    if rc == 0:
        return True, ""
    print(f"[bench] capture probe exited with code {rc}: launching eagerly", file=sys.stderr, flush=True)
    return False, f" (hipGraph capture failed in the probe process, rc {rc})"


def agree_on_graph(eg, use_graph, dev):
    """all ranks capture, or none does"""
    if not torch.distributed.is_initialized() or torch.distributed.get_world_size() == 1:
        return use_graph
    t = torch.tensor([1.0 if use_graph else 0.0], device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def capture_or_exit(eg, tr, **kw):
    """hipGraph capture of the resident step (RCCL collectives included when world > 1).  A failure ends the process with a non-zero exit
    code and the reason on stderr (engine.exit_after_capture_failure): there is no in-process fallback."""
    try:
        tr.capture(**kw)
    except eg.engine.CaptureFailed as exc:
        eg.engine.exit_after_capture_failure(exc)
    return True


def probe_exit(tr):
    """the probe child's job is done once two replays have completed"""
    tr.step_resident()
    tr.step_resident()
    torch.cuda.synchronize()
    print("[bench] capture probe ok", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os._exit(0)


def cpu_baseline(B, steps, warm=2):
    from oracle import celeba_oracle as co          # the checker, timed as the reported CPU baseline
    torch.set_num_threads(min(16, os.cpu_count() or 1))   # the GPU box's CPU share for one GPU is 16 cores
    orc = co.CelebAOracle(seed=0)
    rng = np.random.RandomState(0)
    real = co.synthetic_real(B, seed=1)
    times = []
    for i in range(steps + warm):
        z, code, labels = co.draw_step_inputs(rng, B)
        t0 = time.perf_counter()
        orc.train_step(real, z, code, labels)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[warm:]))
    return {"value": round(B / t, 2), "unit": "imgs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {steps} timed iterations of the same workload (B={B}, fp32, torch-CPU oracle) after {warm} warm-ups"}


def synthetic_sprites(n, dev, gen):
    """uint8 {0,1} [n,64,64]: one filled axis-aligned box per image (stands in for the dSprites .npz, which is not in the container)"""
    r = lambda lo, hi: torch.randint(lo, hi, (n, 1, 1), device=dev, generator=gen)
    cy, cx, hh, hw = r(20, 44), r(20, 44), r(4, 12), r(4, 12)
    yy, xx = torch.arange(64, device=dev).view(1, 64, 1), torch.arange(64, device=dev).view(1, 1, 64)
    return (((yy - cy).abs() < hh) & ((xx - cx).abs() < hw)).to(torch.uint8)


def main_mnist(a, eg, rank, world, local, dev):
    """secondary line: MNIST/EAD-GAN_rpqmnxy.py iteration (1.403 GFLOP/img necessary); launch-bound, not MFMA-bound."""
    B = a.batch
    torch.manual_seed(0)
    eg.mnist.load_approximator(eg.mnist.Affine_classifier().state_dict())      # seeded stand-in of the frozen rpqmnxy_approximator.pt (not in the container)
    G, D, E = eg.mnist.Generator(dtype=a.dtype).to(dev), eg.mnist.Discriminator(dtype=a.dtype).to(dev), eg.mnist.Encoder(dtype=a.dtype).to(dev)
    for m in (G, D, E):
        m.apply(eg.mnist.weights_init_normal)
    tr = eg.mnist.MnistTrainer(G, D, E, B, dtype=a.dtype, allreduce=eg.dp.GradAllReduce(world, wire=a.wire) if world > 1 else None,
                               sync_bn=eg.dp.SyncBN(world, rank) if (a.sync_bn and world > 1) else None)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    tr.load_inputs(torch.rand((B, 1, 32, 32), device=dev, generator=g) * 2 - 1, torch.randn((B, 62), device=dev, generator=g),
                   torch.rand((B, 7), device=dev, generator=g) * 2 - 1, torch.randint(0, 10, (B,), device=dev, generator=g))
    inputs = None
    if not a.resident_inputs:                            # synthetic uint8 "dataset" resident in HBM (64k digits = 67 MB), batches drawn on the device
        inputs = eg.mnist.DeviceInputs(torch.randint(0, 256, (65536, 1, 32, 32), device=dev, dtype=torch.uint8, generator=g), seed=1000 + rank)
        tr.inputs = inputs
    tr.step_resident()
    use_graph = agree_on_graph(eg, a.use_graph, dev) and capture_or_exit(eg, tr, inputs=inputs)
    if a.probe_capture:
        probe_exit(tr)
    for _ in range(max(a.warmup - 1, 0)):
        tr.step_resident()
    eg.dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step_resident()
    torch.cuda.synchronize()
    eg.dp.barrier()
    dt = eg.dp.max_over_ranks(time.perf_counter() - t0, dev)
    roof = table = None
    if not a.no_roofline and rank == 0:
        roof, table = roofline_pass(eg, tr, a.dtype, "mnist")
    if rank == 0:
        ips = B * world * a.steps / dt
        peak = PEAK_F32_TFLOPS if a.dtype == "f32" else PEAK_BF16_TFLOPS
        print(json.dumps({"metric": "imgs/sec per G+D+E train step, MNIST 32x32", "value": round(ips, 1), "unit": "imgs/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic" + ("" if a.resident_inputs else " (every batch drawn on the device inside the timed step)"),
                          "config": {"workload": f"EAD-GAN MNIST 32x32x1 full train iteration (G + D + info/affine over G+E), batch {B}/GPU, "
                                                 f"{'hipGraph replay' if use_graph else 'eager launches' + a.graph_note}", "per_gpu_batch": B, "parallelism": f"dp{world}"},
                          "whole_step_mfma_frac": round(ips / world * 1.403 / 1e3 / peak, 5), "roofline": roof, "cpu_baseline": None,
                          "final_losses": [round(x, 4) for x in tr.losses.tolist()[:3]],
                          "kernel_table": kernel_table(table) if table else None}), flush=True)


def main_sprites(a, eg, rank, world, local, dev):
    """secondary lines: dSprites/rp.py (0.485 GFLOP/img) and colored_dSprites/rp_color.py (0.533 GFLOP/img) iterations."""
    B, color = a.batch, a.workload == "colored"
    mod = eg.colored if color else eg.dsprites
    torch.manual_seed(0)
    P, G, D, E = mod.Encoder_pxy(dtype=a.dtype).to(dev), mod.Generator(dtype=a.dtype).to(dev), mod.Discriminator(dtype=a.dtype).to(dev), mod.Encoder(dtype=a.dtype).to(dev)
    # (Encoder_pxy stays at its seeded default init: a stand-in of the frozen stage-1 checkpoint encoder_pxy_*.pt, which is not in the container)
    ar = eg.dp.GradAllReduce(world, wire=a.wire) if world > 1 else None
    tr = (mod.ColoredTrainer if color else mod.DspritesTrainer)(P, G, D, E, B, dtype=a.dtype, allreduce=ar, overlap=not a.no_overlap,
                                                                 sync_bn=eg.dp.SyncBN(world, rank) if (a.sync_bn and world > 1) else None)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    sprites = synthetic_sprites(B, dev, g)
    cd = 7 if color else 4
    mk = lambda: (torch.rand((B, cd), device=dev, generator=g) * 2 - 1, torch.randint(0, 3, (B,), device=dev, generator=g))
    c1, l1 = mk()
    c2, l2 = mk()
    if color:
        tr.load_inputs(sprites, torch.rand((B, 3), device=dev, generator=g) * 0.5 + 0.5, c1, l1, c2, l2)
    else:
        tr.load_inputs(sprites, c1, l1, c2, l2)
    inputs = None
    if not a.resident_inputs:                            # synthetic uint8 sprite array resident in HBM (32k sprites = 134 MB), batches drawn on the device
        inputs = mod.DeviceInputs(synthetic_sprites(32768, dev, g), seed=1000 + rank)
        tr.inputs = inputs
    tr.step_resident()
    use_graph = agree_on_graph(eg, a.use_graph, dev) and capture_or_exit(eg, tr, inputs=inputs)
    if a.probe_capture:
        probe_exit(tr)
    for _ in range(max(a.warmup - 1, 0)):
        tr.step_resident()
    eg.dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step_resident()
    torch.cuda.synchronize()
    eg.dp.barrier()
    dt = eg.dp.max_over_ranks(time.perf_counter() - t0, dev)
    roof = table = None
    if not a.no_roofline and rank == 0:
        roof, table = roofline_pass(eg, tr, a.dtype, a.workload)
    if rank == 0:
        ips = B * world * a.steps / dt
        peak = PEAK_F32_TFLOPS if a.dtype == "f32" else PEAK_BF16_TFLOPS
        gf = 0.533 if color else 0.485
        print(json.dumps({"metric": f"imgs/sec per train step, {'colored ' if color else ''}dSprites 64x64", "value": round(ips, 1), "unit": "imgs/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic" + ("" if a.resident_inputs else " (every batch drawn on the device inside the timed step)"),
                          "config": {"workload": f"EAD-GAN {'colored ' if color else ''}dSprites full train iteration (D step + joint info/affine/G step), batch {B}/GPU, "
                                                 f"{'hipGraph replay' if use_graph else 'eager launches' + a.graph_note}", "per_gpu_batch": B, "parallelism": f"dp{world}"},
                          "whole_step_mfma_frac": round(ips / world * gf / 1e3 / peak, 5), "roofline": roof, "cpu_baseline": None,
                          "final_losses": [round(x, 4) for x in tr.losses.tolist()[:5]],
                          "kernel_table": kernel_table(table) if table else None}), flush=True)


def main_pxy(a, eg, rank, world, local, dev):
    """secondary line: dSprites/pxy.py (stage-1 trainer of Encoder_pxy): 2 encoder forwards + backward per image, ~0.094 GFLOP/img."""
    B = a.batch
    torch.manual_seed(0)
    P = eg.dsprites.Encoder_pxy(dtype=a.dtype).to(dev)
    tr = eg.dsprites.PxyTrainer(P, B, dtype=a.dtype, allreduce=eg.dp.GradAllReduce(world, wire=a.wire) if world > 1 else None)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    tr.load_inputs(synthetic_sprites(B, dev, g), torch.rand((B, 3), device=dev, generator=g) * 2 - 1)
    tr.step_resident()
    use_graph = agree_on_graph(eg, a.use_graph, dev) and capture_or_exit(eg, tr)
    if a.probe_capture:
        probe_exit(tr)
    for _ in range(max(a.warmup - 1, 0)):
        tr.step_resident()
    eg.dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step_resident()
    torch.cuda.synchronize()
    eg.dp.barrier()
    dt = eg.dp.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({"metric": "imgs/sec per train step, dSprites stage-1 (Encoder_pxy)", "value": round(B * world * a.steps / dt, 1), "unit": "imgs/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                          "config": {"workload": f"dSprites/pxy.py iteration (E(img), E(warp(img)), affine regulariser, Adam), batch {B}/GPU, "
                                                 f"{'hipGraph replay' if use_graph else 'eager launches' + a.graph_note}", "per_gpu_batch": B, "parallelism": f"dp{world}"},
                          "roofline": None, "cpu_baseline": None, "final_losses": [round(tr.losses.tolist()[0], 4)]}), flush=True)


def main():
    a = parse()
    # graph or eager: decided by a child process before this one loads the HIP library or initialises the GPU (decide_graph)
    a.use_graph, a.graph_note = decide_graph(a, int(os.environ.get("WORLD_SIZE", "1")))
    eg = importlib.import_module("ead-gan_amd")
    if a.force_dist and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        torch.cuda.set_device(0)
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    rank, world, local = eg.dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = a.batch
    if a.workload == "mnist":
        return main_mnist(a, eg, rank, world, local, dev)
    if a.workload in ("dsprites", "colored"):
        return main_sprites(a, eg, rank, world, local, dev)
    if a.workload == "pxy":
        return main_pxy(a, eg, rank, world, local, dev)

    torch.manual_seed(0)                                 # identical replicas on every rank
    G = eg.celeba.Generator(dtype=a.dtype).to(dev)
    D = eg.celeba.Discriminator(dtype=a.dtype).to(dev)
    allreduce = eg.dp.GradAllReduce(world, force=a.force_dist, wire=a.wire) if (world > 1 or a.force_dist) else None
    if allreduce is None and os.environ.get("EG_DP_SCHEDULE_ONLY"):
        allreduce = eg.dp.GradAllReduce(1)               # diagnostic: the data-parallel schedule (communication stream, events) with no-op collectives
    sync = eg.dp.SyncBN(world, rank) if (a.sync_bn and world > 1) else None
    tr = eg.celeba.CelebATrainer(G, D, B, dtype=a.dtype, allreduce=allreduce, overlap=not a.no_overlap, sync_bn=sync)

    # synthetic inputs resident in HBM before the timed region: per-rank shard of the global batch
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    real = torch.rand((B, 3, 64, 64), device=dev, generator=g) * 2 - 1
    z = torch.randn((B, 200), device=dev, generator=g)
    code = torch.rand((B, 8), device=dev, generator=g) * 2 - 1
    labels = torch.randint(0, 10, (B,), device=dev, generator=g)
    tr.load_inputs(real, z, code, labels)

    inputs = None
    if not a.resident_inputs:                            # synthetic uint8 "dataset" resident in HBM (16k images = 197 MB)
        inputs = eg.celeba.DeviceInputs(torch.randint(0, 256, (16384, 3, 64, 64), device=dev, dtype=torch.uint8, generator=g), seed=1000 + rank)
        tr.inputs = inputs
    tr.step_resident()                                   # first eager iteration: loads kernels, sizes workspaces (and RCCL channels)
    use_graph = agree_on_graph(eg, a.use_graph, dev) and capture_or_exit(eg, tr, inputs=inputs)     # RCCL collectives are captured into the same hipGraph
    if a.probe_capture:
        probe_exit(tr)
    for _ in range(max(a.warmup - 1, 0)):
        tr.step_resident()
    eg.dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step_resident()
    torch.cuda.synchronize()
    eg.dp.barrier()
    dt = time.perf_counter() - t0
    dt = eg.dp.max_over_ranks(dt, dev)
    losses = tr.losses.tolist()

    roof = table = None
    if not a.no_roofline and rank == 0:
        roof, table = roofline_pass(eg, tr, a.dtype)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(B, a.cpu_steps)

    if rank == 0:
        ips = B * world * a.steps / dt
        peak = PEAK_F32_TFLOPS if a.dtype == "f32" else PEAK_BF16_TFLOPS
        out = {
            "metric": "imgs/sec per G+D+E train step, CelebA 64x64 bs=128",
            "value": round(ips, 1), "unit": "imgs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic" + ("" if a.resident_inputs else " (every batch drawn on the device inside the timed step: uint8 gather + flip + normalise, Philox z / code / labels)"),
            "config": {"workload": f"EAD-GAN CelebA 64x64x3 full train iteration (G adv + D + info/affine, 3 Adams), batch {B}/GPU, "
                                   f"{'hipGraph replay' if use_graph else 'eager launches' + a.graph_note}, data-parallel x{world}",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "whole_step_mfma_frac": round(ips / world * GFLOP_PER_IMG / 1e3 / peak, 4),
            "roofline": roof, "cpu_baseline": cpu,
            "final_losses": {"g": round(losses[0], 4), "d": round(losses[1], 4), "info": round(losses[2], 4)},
        }
        if table:
            out["kernel_table"] = kernel_table(table)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
