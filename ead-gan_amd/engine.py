"""Host-side building blocks shared by the per-dataset models: flat parameter arenas, the shared device
workspace, and conv-layer records (geometry + packed weight panels) over the C ABI in :mod:`ops`."""
from __future__ import annotations

import contextlib
import os

import torch

from . import ops
from .ops import EG_BF16, EG_F16, EG_F32


def parse_dtype(dtype) -> int:
    if dtype in (EG_F32, "f32", "fp32", "float32", torch.float32):
        return EG_F32
    if dtype in (EG_BF16, "bf16", "bfloat16", torch.bfloat16):
        return EG_BF16
    if dtype in (EG_F16, "f16", "fp16", "float16", "half", torch.float16):
        return EG_F16
    raise ValueError(f"unsupported compute dtype {dtype!r} (use 'f32', 'bf16' or 'f16')")


class Arena:
    """All parameters of a module re-homed into ONE flat fp32 tensor (reference ``.parameters()`` order)
    with a matching flat gradient tensor: one fused Adam launch per optimizer, one all-reduce per pass."""

    def __init__(self, module: torch.nn.Module):
        params = list(module.parameters())
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.slices = {}
        off = 0
        for name, p in module.named_parameters():
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            self.slices[name] = (off, k)
            off += k
        self.numel = n

    def grad_of(self, name, arena_grad=None):
        off, k = self.slices[name]
        return (self.grad if arena_grad is None else arena_grad)[off:off + k]


SPLITK_WS_BYTES = 72 << 20


class Workspace:
    """Scratch shared by every engine on a device (calls are stream-ordered, so one copy suffices).
    Grown only while engines are being built -- never inside a training step."""

    _per_device = {}

    def __init__(self, device, splitk=True, register=True):
        """``register=False``: a second chain's private workspace (own split-K scratch too) -- it does not become the device default"""
        self.device = device
        self._retired = []
        # chip share hint for the weight-gradient GEMMs launched with this scratch (ops.conv_wgrad): 0 = the whole chip; a side lane's
        # workspace says 128 workgroups, its launches run beside the main chain's GEMMs (profiles/r02_i_ab_tn8_target.txt)
        self.wgs_target = 0
        self.slab = torch.empty(0, device=device, dtype=torch.float32)
        self.gtmp = torch.empty(0, device=device, dtype=torch.float32)
        self.small = torch.empty(0, device=device, dtype=torch.float32)
        self.partials = torch.empty(max(2048, ops.sn_partials()), device=device, dtype=torch.float32)
        self.sums = torch.empty(0, device=device, dtype=torch.float32)
        # split-K partial tiles of the NT kernel: the planner splits up to ~2x512 tiles of 128x128 fp32 (64 KiB each)
        if splitk:
            self.splitk = torch.zeros(SPLITK_WS_BYTES // 4, device=device, dtype=torch.float32)      # zeroed: its tail holds arrival counters
            if register:
                ops.set_splitk_workspace(self.splitk)

    _scoped = None

    @classmethod
    @contextlib.contextmanager
    def scope(cls, ws: "Workspace"):
        """engines built inside take ``ws`` as their scratch (a chain that runs beside the main one must not share its scratch)"""
        prev, cls._scoped = cls._scoped, ws
        try:
            yield ws
        finally:
            cls._scoped = prev

    @contextlib.contextmanager
    def active(self):
        """launches enqueued inside split K into THIS workspace's scratch (ops.epilogue's default), not the device default"""
        prev = ops.SPLITK_OVERRIDE
        ops.SPLITK_OVERRIDE = self.splitk
        try:
            yield self
        finally:
            ops.SPLITK_OVERRIDE = prev

    @classmethod
    def get(cls, device) -> "Workspace":
        if cls._scoped is not None:
            return cls._scoped
        key = (device.type, device.index)
        if key not in cls._per_device:
            cls._per_device[key] = cls(device)
        return cls._per_device[key]

    def _grow(self, name, floats):
        """Workspaces only ever grow, and an outgrown buffer is kept alive: a hipGraph captured earlier (another trainer, another batch
        size) has its address baked in, and memory handed back to the caching allocator would be given to somebody else."""
        t = getattr(self, name)
        if t.numel() < floats:
            if t.numel():
                self._retired.append(t)
            setattr(self, name, torch.empty(int(floats), device=self.device, dtype=torch.float32))

    def need_slab(self, nbytes):
        self._grow("slab", (nbytes + 3) // 4)

    def need_gtmp(self, floats):
        self._grow("gtmp", floats)

    def need_small(self, floats):      # bn / sn / bias-grad partial buffers
        self._grow("small", floats)

    def need_sums(self, floats):
        self._grow("sums", floats)


LANE_WGS_TARGET = int(os.environ.get("EG_LANE_WGS", "64"))      # 64: profiles/r03_o_lane_wgs_sweep.txt (128 in round 2)


class _Lane:
    NAMES = ("slab", "gtmp", "small", "sums", "partials")

    def __init__(self, device, like: "Workspace"):
        self.stream = torch.cuda.Stream(device)
        self.ws = Workspace(device, splitk=False)
        self.ws.wgs_target = LANE_WGS_TARGET
        self.like = like
        self.ensure()

    def ensure(self):
        """Lane scratch follows the device workspace: an engine built after the lanes (a second trainer, a larger batch) may have grown
        it, and work forked onto this lane is sized by the device workspace's reservations."""
        for name in self.NAMES:
            self.ws._grow(name, getattr(self.like, name).numel())

    def __enter__(self):
        self._ctx = torch.cuda.stream(self.stream)
        return self._ctx.__enter__()

    def __exit__(self, *a):
        return self._ctx.__exit__(*a)


class SideStream:
    """Extra HIP streams ("lanes", each with its own scratch) for work that does not feed the critical path of a training
    step: the weight-gradient GEMMs with their reductions and bias-gradient sums run beside the backward-data GEMMs of
    the layers below (consecutive layers on alternating lanes, so one layer's HBM-bound reductions overlap the next
    layer's GEMM), weight re-packing after an optimizer step and the spectral-norm power iterations of the next sub-step
    run beside the current one.  ``fork(i)``: lane i waits for everything enqueued on the current stream so far;
    ``join()``: the current stream waits for every lane.  Both are event record/wait pairs, capturable into the step's
    hipGraph.  Lanes never launch NT convolutions (the split-K scratch belongs to the main stream)."""

    def __init__(self, device, like: "Workspace", lanes: int = 2):
        self.lanes = [_Lane(device, like) for _ in range(lanes)]
        # optimizer lane: (gradient all-reduce ->) Adam -> gradient zeroing -> panel re-packing of one network, behind ALL of its
        # weight-gradient chains, while the main stream is already in the next sub-step.  Not part of the round robin.
        self.opt = _Lane(device, like)
        # preparation lane: the next sub-step's power iterations and patch rows.  It waits for optimizer-lane events (new weights), so
        # the optimizer lane must never wait for it: hipStreamEndCapture of ROCm 7.2 segfaults on such a stream-level back edge
        # (X waited for Y, later Y waits for X) even though the node graph is acyclic -- profiles/scripts/capture_patterns.py.  The same
        # holds for longer cycles: the waits among the non-origin streams must form a DAG (profiles/r01_timeline_notes.md item 10).
        self.prep = _Lane(device, like)
        # communication stream (data parallel, eager launches): a gradient bucket's all-reduce is started from here as soon as the lane
        # chain that completes the bucket has fired its event -- the main stream never waits for a lane on the collectives' behalf
        self.comm = torch.cuda.Stream(device)
        self._pending = []
        self.done = {}                                 # tag -> event behind a tagged lane chain (defer(..., tag=))
        self.done_lane = {}                            # tag -> the lane that ran the chain
        # tag -> main-stream event behind the last main-stream kernel that reads the tagged bucket's parameters or panels: the event of
        # the NEXT tagged fork (the backward passes fork a layer's chain, then launch that layer's backward-data GEMM, then move on to
        # the layer below), or close_tags() for the last one
        self.free = {}
        self._last_tag = None
        self._flushing = False
        # issue order of forked work (see defer()): "0" at once (side work captured before the main stream's next kernel), "1" always
        # after it, "once" at once for a lane's first fork of the step and after it from then on
        self.deferred = os.environ.get("EG_DEFER", "0")
        self._entered = set()
        # lanes forked since the last cut(): an iteration captured as SEVERAL hipGraphs (capture_segments) may only join -- record an
        # event on -- the lanes that are part of the current capture; None: every lane (one capture / eager launches)
        self._live = None
        # True: deferred work runs on the CURRENT stream at once (no fork): a capture segment that has to stay ONE chain -- a hipGraph with
        # branches, launched on a second stream, holds back every later graph launch on the first (profiles/scripts/graph_streams_toy.py)
        self.inline = False

    def lane(self, i: int) -> _Lane:
        return self.lanes[i % len(self.lanes)]

    def _touch(self, ln):
        if self._live is not None:
            self._live.append(ln) if ln not in self._live else None

    def _joinable(self, lanes):
        return [ln for ln in lanes if self._live is None or ln in self._live]

    def cut(self):
        """a new capture segment begins: no lane belongs to it yet (the caller has joined them all)"""
        assert not self._pending, "cut() with deferred work pending"
        self._live = []
        self.done.clear()
        self.done_lane.clear()
        self.free.clear()
        self._last_tag = None

    def fork(self, i: int = 0) -> _Lane:
        ln = self.lane(i)
        ln.ensure()
        ev = torch.cuda.Event()
        ev.record()
        ln.stream.wait_event(ev)
        self._touch(ln)
        return ln

    # Deferred issue.  hipGraph's executor keeps the FIRST-created child of a node on the node's own HW queue; every other child moves to
    # another queue behind a cross-queue wait (10-18 us on MI355X, measured: profiles/r01_timeline_notes.md).  A fork therefore records
    # its event at once but its launches are issued by the next flush(), which the caller places right AFTER the next main-stream
    # kernel: the critical chain stays on one queue and the hop lands on the side work.  Dependencies are unchanged (the events).
    def defer(self, i: int, fn, tag=None):
        """lane i runs fn(lane_workspace) behind everything enqueued on the current stream so far.  ``tag``: an event is recorded on the
        lane behind fn and kept in ``self.done[tag]`` (data parallel: the gradient bucket this chain completes can be reduced as soon
        as the event fires, see CelebATrainer)."""
        ev = self.mark()
        if tag is not None:
            if self._last_tag is not None:
                self.free[self._last_tag] = ev
            self._last_tag = tag
        self._pending.append(("lane", self.lane(i), ev, fn, tag))
        self._issue(("lane", i % len(self.lanes)))

    def close_tags(self):
        """the current position of the current stream is behind every reader of the last tagged bucket"""
        if self._last_tag is not None:
            self.free[self._last_tag] = self.mark()
            self._last_tag = None

    def defer_opt_after(self, tags, fn):
        """the optimizer lane runs fn(ws) behind the tagged chains ``tags`` and the main-stream readers of their buckets ONLY (not behind
        the rest of the backward pass): bucket-wise optimizer updates start while the layers below are still in their backward pass"""
        self._pending.append(("optb", self.opt, tuple(tags), fn, None))
        self._issue("opt")

    def defer_opt(self, fn):
        """the optimizer lane runs fn(ws) behind the current stream AND every weight-gradient chain forked so far"""
        self._pending.append(("opt", self.opt, self.mark(), fn, None))
        self._issue("opt")

    def defer_prep(self, fn):
        self._pending.append(("prep", self.prep, self.mark(), fn, None))
        self._issue("prep")

    def _issue(self, key):
        if self.deferred == "0" or (self.deferred == "once" and key not in self._entered):
            self._entered.add(key)
            self.flush()

    def begin_step(self):
        self._entered.clear()

    def flush(self):
        if self._flushing or not self._pending:
            return
        self._flushing = True
        try:
            while self._pending:
                kind, ln, ev, fn, tag = self._pending.pop(0)
                ln.ensure()
                if self.inline:
                    fn(ln.ws)                           # (the events of an "optb" entry belong to chains of this very stream)
                    if tag is not None:
                        self.done[tag] = self.mark()
                        self.done_lane[tag] = ln
                    continue
                if kind == "optb":
                    for t in ev:
                        d = self.done.pop(t, None)          # None: the caller already waited for it (data parallel)
                        if d is not None:
                            ln.stream.wait_event(d)
                        ln.stream.wait_event(self.free.pop(t))
                else:
                    ln.stream.wait_event(ev)
                self._touch(ln)
                if kind == "opt":
                    for other in self._joinable(self.lanes):
                        e2 = torch.cuda.Event()
                        e2.record(other.stream)
                        ln.stream.wait_event(e2)
                with ln:
                    fn(ln.ws)
                    if tag is not None:
                        self.done[tag] = self.mark()
                        self.done_lane[tag] = ln
        finally:
            self._flushing = False

    def fork_prep(self) -> _Lane:
        self.prep.ensure()
        ev = torch.cuda.Event()
        ev.record()
        self.prep.stream.wait_event(ev)
        self._touch(self.prep)
        return self.prep

    def fork_opt(self) -> _Lane:
        """the optimizer lane waits for the current stream AND for everything the weight-gradient lanes hold so far"""
        self.opt.ensure()
        ev = torch.cuda.Event()
        ev.record()
        self.opt.stream.wait_event(ev)
        self._touch(self.opt)
        for ln in self._joinable(self.lanes):
            ev = torch.cuda.Event()
            ev.record(ln.stream)
            self.opt.stream.wait_event(ev)
        return self.opt

    @staticmethod
    def mark() -> "torch.cuda.Event":
        """an event at the current position of the current stream (inside ``with lane:`` -- of that lane)"""
        ev = torch.cuda.Event()
        ev.record()
        return ev

    @staticmethod
    def wait(ev):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def join_lanes(self):
        """the current stream waits for the weight-gradient lanes only (not for the optimizer / preparation lanes)"""
        self.flush()
        cur = torch.cuda.current_stream()
        for ln in self._joinable(self.lanes):
            ev = torch.cuda.Event()
            ev.record(ln.stream)
            cur.wait_event(ev)

    def join(self):
        self.flush()
        cur = torch.cuda.current_stream()
        for ln in self._joinable(self.lanes + [self.opt, self.prep]):
            ev = torch.cuda.Event()
            ev.record(ln.stream)
            cur.wait_event(ev)


class SyncScratch:
    """Per-engine scratch of synchronised BatchNorm: one [3*C] statistics block and one [2*C] backward-sum block per layer."""

    def __init__(self, channels, device):
        self.stats = [torch.empty(3 * c, device=device, dtype=torch.float32) for c in channels]
        self.sums = [torch.empty(2 * c, device=device, dtype=torch.float32) for c in channels]


def bn_train_forward(dt, x, y, M, C, bn, mean, invstd, ws_small, act, slope=0.0, sync=None, stats=None):
    """nn.BatchNorm2d in training mode (batch statistics, running statistics updated).  ``sync`` (a dp.SyncBN): statistics over the
    global batch of all ranks -- local (n, mean, M2) -> one exchange of 3*C floats -> Chan-combined mean / variance."""
    if sync is None:
        ops.bn_fwd_train(dt, x, y, M, C, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.num_batches_tracked, mean, invstd,
                         ws_small, act, slope)
        return
    ops.bn_stats_local(dt, x, M, C, ws_small, stats)
    allst = sync.gather_stats(stats)
    ops.bn_fwd_from_stats(dt, x, y, M, C, allst, sync.world, M * sync.world, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean, bn.running_var,
                          bn.num_batches_tracked, mean, invstd, ws_small, act, slope)


def bn_train_backward(dt, z, da, dz, M, C, bn, mean, invstd, act, slope, dgamma, dbeta, ws, sync=None, sums=None):
    """Backward of bn_train_forward (``ws``: a Workspace).  ``sync``: the two per-channel sums over the global batch (one exchange of
    2*C floats); dgamma / dbeta are this rank's share -- the gradient all-reduce averages them like every other gradient."""
    if sync is None:
        ops.bn_bwd(dt, z, da, dz, M, C, bn.weight, bn.bias, mean, invstd, act, slope, dgamma, dbeta, ws.sums, ws.small)
        return
    ops.bn_bwd_sums_local(dt, z, da, M, C, bn.weight, bn.bias, mean, invstd, act, slope, dgamma, dbeta, sums, ws.small)
    sync.reduce_sums(sums)
    ops.bn_bwd_from_sums(dt, z, da, dz, M, C, sums, M * sync.world, bn.weight, bn.bias, mean, invstd, act, slope, ws.small)


FUSE_DRAWS = os.environ.get("EG_FUSE_INPUTS", "1") != "0"      # (the switch of celeba.FUSE_INPUTS)


class DeviceSampler:
    """Shared part of the trainers' device-side input pipelines: a uint8 dataset resident in HBM, a device step counter and
    counter-based draws (ops.rng_fill: reproducible per (seed, step, stream), the reference's distributions, NOT numpy's stream --
    parity tests keep feeding host draws through ``load_inputs``).  Sampling follows ``DataLoader(shuffle=True)``: a keyed permutation of the
    dataset per epoch, every image exactly once (``sampling="replacement"``: independent uniform draws).  Subclasses implement ``enqueue(trainer)``: fill the trainer's static input slots, then tick
    the counter; inside ``trainer.capture(inputs=...)`` those launches are part of the iteration's hipGraph."""

    def __init__(self, dataset_u8: torch.Tensor, seed: int = 0, sampling: str = "permutation"):
        if not dataset_u8.is_cuda or dataset_u8.dtype != torch.uint8:
            raise ValueError("dataset must be a uint8 device tensor")
        if sampling not in ("permutation", "replacement"):
            raise ValueError("sampling must be 'permutation' (DataLoader(shuffle=True): every image once per epoch) or 'replacement'")
        self.data = dataset_u8.contiguous()
        self.sampling = sampling
        self.seed = int(seed)
        self.step = torch.zeros(1, device=dataset_u8.device, dtype=torch.int32)
        self._buf = {}

    def buf(self, name, shape, dtype):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self._buf[name] = torch.empty(shape, device=self.data.device, dtype=dtype)
        return t

    # Draws of one iteration as ONE launch (eg_rng_fill_multi: the values of one eg_rng_fill per draw -- a value depends on (element, step,
    # stream id, seed) only, not on the launch it comes from): between begin_draws() and end_draws() the draw methods below only collect.
    _draws = None

    def begin_draws(self):
        if FUSE_DRAWS:
            self._draws = []

    def end_draws(self):
        draws, self._draws = self._draws, None
        if draws:
            ops.rng_fill_multi(draws, self.seed, self.step)

    def draw(self, kind, out, a, b, stream_id):
        if self._draws is not None:
            self._draws.append((kind, out, a, b, stream_id))
            return
        ops.rng_fill(kind, out, a, b, self.seed, self.step, stream_id)

    def sample_indices(self, B, stream_id=1):
        idx = self.buf("idx", (B,), torch.int64)
        if self.sampling == "permutation":               # a fresh permutation of the dataset per epoch, as the reference's DataLoader draws
            self.draw(ops.RNG_EPOCH_PERM, idx, self.data.shape[0], 0, stream_id)
        else:
            self.draw(ops.RNG_RANDINT, idx, 0, self.data.shape[0], stream_id)
        return idx

    def labels_onehot(self, name, onehot, n_classes, stream_id, lab=None):
        if lab is None:
            lab = self.buf(name, (onehot.shape[0],), torch.int64)
        if self._draws is not None:
            self._draws.append((ops.RNG_RANDINT, lab, 0, n_classes, stream_id, onehot))      # the one-hot rows written by the same launch
            return lab
        self.draw(ops.RNG_RANDINT, lab, 0, n_classes, stream_id)
        ops.onehot(lab, onehot, onehot.shape[0], n_classes)
        return lab

    def tick(self):
        ops.counter_add(self.step, 1)


class ResidentStep:
    """capture / replay plumbing shared by the small-network trainers (``_step_body`` = one iteration on the static input slots)"""

    inputs = None
    graph = None

    def _step_with_inputs(self):
        if self.inputs is not None:
            self.inputs.enqueue(self)
        self._step_body()

    def capture(self, warmup=False, inputs=None):
        """Capture the iteration into one hipGraph; with ``inputs`` (a DeviceSampler of this trainer's module) the graph first draws
        the batch on the device, so a replay is a complete loop iteration without host work."""
        if warmup:
            self._step_body()
        if inputs is not None:
            self.inputs = inputs
        return capture_step(self, self._step_with_inputs)

    def step_resident(self):
        check_usable(self)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step_with_inputs()
        return self.losses


class CaptureFailed(RuntimeError):
    """A hipGraph capture of a training iteration failed.  THE PROCESS IS NOT USABLE FOR GPU WORK AFTERWARDS: on ROCm 7.2 every stream
    that was forked into the capture (weight-gradient lanes, optimizer / preparation lanes, a second chain, the input sampler's and a
    collective's internal streams) stays attached to the invalidated capture, and launching on them, destroying them or merely letting
    them be garbage-collected crashes the runtime a little later (observed: SIGSEGV three seconds after an `in-process recovery' that
    restored the stream, drained the sticky error and replaced the lanes -- gpurun_out/r02_z_dp2.err).  There is no in-process
    recovery; the contract is: report the reason and end the process (``engine.exit_after_capture_failure``).  Callers that want an
    eager fallback decide graph-vs-eager BEFORE they touch the GPU, by probing the capture in a child process (bench.py does)."""


def exit_after_capture_failure(exc, code=3):
    """Print the reason and end the process at once (os._exit: no destructors run over streams that still belong to the invalidated
    capture -- that is where the crash comes from)."""
    import sys
    print(f"[ead-gan_amd] hipGraph capture failed: {exc}\n[ead-gan_amd] this process cannot use the GPU any more; exiting with code {code} "
          f"(launch eagerly instead: a trainer without capture(), bench.py --no-graph)", file=sys.stderr, flush=True)
    os._exit(code)


def capture_step(trainer, body):
    """Capture ``body()`` (one whole training iteration) into a hipGraph and store it as ``trainer.graph``.  A capture that fails
    raises :class:`CaptureFailed` and marks the trainer unusable (``step_resident`` refuses to launch): see the exception's text for
    why no eager fallback is offered inside the same process."""
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            body()
    except Exception as exc:
        trainer.graph = None
        trainer.capture_failed = f"{type(exc).__name__}: {exc}"
        raise CaptureFailed(trainer.capture_failed) from exc
    trainer.graph = graph
    return trainer


class MultiGraph:
    """One training iteration as several hipGraphs replayed on TWO real streams, ordered by events (an experiment, see celeba.MULTI_GRAPH:
    it did not remove the delay it was built against).  ``replay()`` is what ``torch.cuda.CUDAGraph.replay()`` is to a one-graph trainer."""

    def __init__(self, device):
        self.segments = []                              # (graph, stream index 0 = the caller's stream / 1 = the second stream, after)
        self.second = torch.cuda.Stream(device)
        self._done = []

    def add(self, graph, stream, after):
        self.segments.append((graph, int(stream), tuple(after)))
        self._done.append(torch.cuda.Event())

    def replay(self):
        main = torch.cuda.current_stream()
        streams = (main, self.second)
        for i, (g, s, after) in enumerate(self.segments):
            st = streams[s]
            for j in after:                             # segments on the other stream this one reads from
                st.wait_event(self._done[j])
            if s == 0:
                g.replay()
            else:
                with torch.cuda.stream(st):
                    g.replay()
            self._done[i].record(st)
        for i, (_, s, _) in enumerate(self.segments):   # the caller's stream ends behind every segment
            if s != 0:
                main.wait_event(self._done[i])


def capture_segments(trainer, body):
    """Like capture_step, for a ``body`` that calls ``trainer._cut(stream, after)`` between segments (every lane joined): each segment
    becomes its own hipGraph; ``trainer.graph`` is a MultiGraph.  ``after``: indices of earlier segments on the OTHER stream whose results
    the next segment reads (same-stream order is implicit)."""
    torch.cuda.synchronize()
    dev = torch.cuda.current_device()
    mg = MultiGraph(dev)
    pool = torch.cuda.graph_pool_handle()
    cap = torch.cuda.Stream(dev)
    cap.wait_stream(torch.cuda.current_stream())
    state = {"g": None, "stream": 0, "after": ()}

    def begin():
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=pool)
        state["g"] = g

    def end():
        state["g"].capture_end()
        mg.add(state["g"], state["stream"], state["after"])
        state["g"] = None

    def cut(stream=0, after=()):
        end()
        state["stream"], state["after"] = stream, after
        begin()

    trainer._cut = cut
    try:
        with torch.cuda.stream(cap):
            begin()
            body()
            end()
    except Exception as exc:
        trainer.graph = None
        trainer.capture_failed = f"{type(exc).__name__}: {exc}"
        raise CaptureFailed(trainer.capture_failed) from exc
    finally:
        trainer._cut = None
    torch.cuda.current_stream().wait_stream(cap)
    trainer.graph = mg
    return trainer


def check_usable(trainer):
    why = getattr(trainer, "capture_failed", None)
    if why is not None:
        raise CaptureFailed(f"this trainer's hipGraph capture failed earlier ({why}); the process must not launch GPU work any more")


class ConvRec:
    """One conv-view layer at a fixed batch size: geometry, packed panels, workspace reservations."""

    def __init__(self, dtype, B, H, W, Cin, Cout, k, stride, pad, up=0, device=None, want_fwd=True, want_bwd=True,
                 want_wgrad=True, ws: Workspace | None = None):
        self.dtype = dtype
        self.c = ops.make_conv(B, H, W, Cin, Cout, k, stride, pad, up)
        self.B, self.H, self.W, self.Cin, self.Cout, self.k = B, H, W, Cin, Cout, k
        self.OH = ((H << up) + 2 * pad - k) // stride + 1
        self.OW = ((W << up) + 2 * pad - k) // stride + 1
        tdt = ops.torch_dtype(dtype)
        self.wp_fwd = torch.empty(ops.pack_fwd_elems(self.c, dtype), device=device, dtype=tdt) if want_fwd else None
        self.wp_bwd = torch.empty(ops.pack_bwd_elems(self.c, dtype), device=device, dtype=tdt) if want_bwd else None
        self.Kpad_fwd = ops.round_up(k * k * Cin, ops.bk(dtype))
        if ws is not None:
            if want_wgrad:
                ws.need_slab(ops.conv_wgrad_ws_bytes(self.c, dtype))
                ws.need_gtmp(Cout * Cin * k * k)
            ws.need_small(ops.bias_grad_ws_floats(B * max(self.OH * self.OW, (H << up) * (W << up)), max(Cin, Cout)))

    def pack(self, w_master):
        if self.wp_fwd is not None or self.wp_bwd is not None:
            ops.pack_conv(self.c, self.dtype, w_master, self.wp_fwd, self.wp_bwd)


def import_adam_moments(opt, arenas):
    """exp_avg / exp_avg_sq of a ``torch.optim.Adam`` built over the parameters of one or more modules in ``.parameters()`` order (the
    reference's optimizers: celebA/EAD-GAN_celebA.py:211-217, MNIST/EAD-GAN_rpqmnxy.py:206-217, dSprites/rp.py:270-279) into the flat
    moment tensors of a fused trainer.  ``arenas``: [(number of parameters of the module, m, v)] in the optimizer's parameter order.
    Returns the optimizer's step count (0 if it has not stepped yet)."""
    params = opt.param_groups[0]["params"]
    step, i = 0, 0
    for count, m, v in arenas:
        off = 0
        for p in params[i:i + count]:
            st = opt.state.get(p, {})
            n = p.numel()
            if st:
                m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = int(st["step"])
            off += n
        assert off == m.numel(), "optimizer parameters do not tile the arena"
        i += count
    assert i == len(params)
    return step
