"""Thin torch-tensor front end of the C ABI (include/eadgan_hip.h).

Every function enqueues hand-written HIP kernels on torch's current stream and returns immediately.
torch is used for device memory and streams only.  No function here has a CPU or eager fallback.
"""
from __future__ import annotations

import ctypes
import os

import torch

from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, EG_BF16, EG_F16, EG_F32, NT_AUTO, NT_BUF128,
                   NT_PERS, NT_REG, NT_S8, NT_S8H, NT_S8P, OUT_NCHW_F32, OUT_NHWC, STAT_BN_BWD, STAT_MOMENTS, STAT_NONE, STAT_SN_BIAS, EgConv, EgEpilogue,
                   EgHead, EgRngSeg, EgSnLayer, lib)

__all__ = ["EG_F32", "EG_BF16", "EG_F16", "ACT_NONE", "ACT_LRELU", "ACT_RELU", "ACT_TANH", "ACT_SIGMOID", "OUT_NHWC",
           "OUT_NCHW_F32"]


def torch_dtype(dtype: int):
    return {EG_F32: torch.float32, EG_BF16: torch.bfloat16, EG_F16: torch.float16}[dtype]


def vec(dtype: int) -> int:
    return 4 if dtype == EG_F32 else 8


def bk(dtype: int) -> int:
    return 32 if dtype == EG_F32 else 64


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def make_conv(B, H, W, Cin, Cout, k, stride, pad, up=0) -> EgConv:
    return EgConv(B, H, W, Cin, Cout, k, stride, pad, up)


# scratch lent to every conv launch for split-K (engine.Workspace owns and sizes it while engines are built)
SPLITK_WS = {}
SPLITK_OVERRIDE = None          # set while a second chain enqueues its launches (engine.Workspace.active)


def set_splitk_workspace(t):
    SPLITK_WS[t.device.index] = t


def conv_splitk_ws_bytes(c, dtype, bwd):
    return lib().query("eg_conv_splitk_ws_bytes", ctypes.byref(c), dtype, int(bwd))


# experiment switches (A/B runs of a whole step): default hints of every launch that does not pass its own
_ENV_VARIANT = int(os.environ.get("EG_NT_VARIANT", "0"))
_ENV_SPLITK = int(os.environ.get("EG_NT_SPLITK", "0"))


def epilogue(bias=None, bias_mod=0, sigma=None, act=ACT_NONE, slope=0.0, mask=None, mask_act=ACT_NONE,
             mask_slope=0.0, out_mode=OUT_NHWC, sigma_rows=0, nt_variant=_ENV_VARIANT, nt_splitk=_ENV_SPLITK, splitk_ws="default",
             stat_mode=STAT_NONE, stat_out=None, stat_aux=None, stat_p=(), stat_act=ACT_NONE, stat_slope=0.0) -> EgEpilogue:
    """``nt_variant`` / ``nt_splitk``: per-call kernel hints (NT_* in _lib.py; 0 = the planner decides).  ``splitk_ws``: scratch
    tensor lent for K splits (default: the device's registered workspace; None = never split).  ``stat_*``: column statistics of
    the stored tile fused into the epilogue (eg_epilogue in the header; only where ``conv_stat_blocks`` answers > 0)."""
    ws = splitk_ws
    if isinstance(ws, str):
        ws = SPLITK_OVERRIDE if SPLITK_OVERRIDE is not None else (SPLITK_WS.get(torch.cuda.current_device()) if SPLITK_WS else None)
    sp = [_p(t) for t in stat_p] + [None] * (4 - len(stat_p))
    return EgEpilogue(_p(bias), bias_mod, _p(sigma), act, slope, _p(mask), mask_act, mask_slope, out_mode, sigma_rows,
                      _p(ws), ws.numel() * ws.element_size() if ws is not None else 0, nt_variant, nt_splitk,
                      stat_mode, _p(stat_out), _p(stat_aux), sp[0], sp[1], sp[2], sp[3], stat_act, stat_slope)


def nt_tile(c, dtype, bwd, ep=None) -> int:
    """BM * 1000 + kernel code of the kernel THIS call (its hints, its split-K scratch) is dispatched to (eg_igemm_nt_tile in the header)"""
    if ep is None:
        ep = epilogue()
    return lib().query("eg_igemm_nt_tile_ep", ctypes.byref(c), dtype, int(bwd), ctypes.byref(ep))


def conv_stat_blocks(c, dtype, bwd, ep=None) -> int:
    """row blocks of the fused column statistics this exact call (geometry, hints, scratch of ``ep``) would write; 0 = it cannot fuse them"""
    if ep is None:
        ep = epilogue()
    return lib().query("eg_conv_stat_blocks", ctypes.byref(c), dtype, int(bwd), ctypes.byref(ep))


# ---- implicit-GEMM family ------------------------------------------------------------------------
def pack_fwd_elems(c, dtype):
    return lib().query("eg_pack_fwd_elems", ctypes.byref(c), dtype)


def pack_bwd_elems(c, dtype):
    return lib().query("eg_pack_bwd_elems", ctypes.byref(c), dtype)


def pack_fwd(c, dtype, w_master, wp):
    lib().call("eg_pack_fwd", ctypes.byref(c), dtype, _p(w_master), _p(wp), _stream())


def pack_bwd(c, dtype, w_master, wp):
    lib().call("eg_pack_bwd", ctypes.byref(c), dtype, _p(w_master), _p(wp), _stream())


def pack_conv(c, dtype, w_master, wp_fwd, wp_bwd):
    lib().call("eg_pack_conv", ctypes.byref(c), dtype, _p(w_master), _p(wp_fwd), _p(wp_bwd), _stream())


def pack_strided(dtype, w, wp, N, K, Kpad, n_div, s_hi, s_lo, s_k):
    lib().call("eg_pack_strided", dtype, _p(w), _p(wp), N, K, Kpad, n_div, s_hi, s_lo, s_k, _stream())


def pack_strided2(dtype, w, wp, N, K, Kpad, n_div, s_hi, s_lo, k_div, s_khi, s_klo):
    lib().call("eg_pack_strided2", dtype, _p(w), _p(wp), N, K, Kpad, n_div, s_hi, s_lo, k_div, s_khi, s_klo, _stream())


# Optional launch recorder (bench.py's roofline pass): when set to a list, every implicit-GEMM launch is bracketed by HIP
# events on the launch stream and appends (kernel label, algorithmic FLOPs, start event, end event, shape string).
# Never set on the training path.
RECORDER = None


_SPACER_CYCLES = None


def _spacer(us=25.0):
    """A spin kernel of ~25 us in front of a timed launch (torch.cuda._sleep, calibrated once): the start event, the launch and the end event
    are enqueued while the GPU is still spinning, so the bracket holds the kernel and not the host's launch time -- eager launches behind a
    run of short kernels find the GPU idle otherwise (bench.py's table read 77 us for a 44 us launch, profiles/r03_no_overlap_gemm_by_grid.txt).
    It touches no memory: the cache state the timed kernel sees is the step's own."""
    global _SPACER_CYCLES
    if _SPACER_CYCLES is None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(100000)
        torch.cuda.synchronize()
        a.record()
        torch.cuda._sleep(1000000)
        b.record()
        torch.cuda.synchronize()
        _SPACER_CYCLES = max(1000, int(us * 1e-3 * 1000000 / max(a.elapsed_time(b), 1e-3)))
    torch.cuda._sleep(_SPACER_CYCLES)


def _out_hw(c):
    return ((c.H << c.up) + 2 * c.pad - c.k) // c.stride + 1, ((c.W << c.up) + 2 * c.pad - c.k) // c.stride + 1


def _timed(kind, c, dtype, args, ep=None):
    """RECORDER entry: (kernel label, algorithmic FLOPs, start event, end event, shape string, algorithmic bytes).  Algorithmic bytes =
    every input element, weight and output element of the launch once, in its storage type (plus the activation-gradient mask the
    epilogue reads): what the launch would move with perfect reuse."""
    oh, ow = _out_hw(c)
    M = c.B * oh * ow
    flops = 2.0 * M * c.Cout * c.Cin * c.k * c.k
    es = 4 if dtype == EG_F32 else 2
    x_elems, y_elems, w_elems = c.B * c.H * c.W * c.Cin, M * c.Cout, c.Cout * c.Cin * c.k * c.k
    tname = {EG_F32: "float", EG_BF16: "bf16", EG_F16: "f16"}[dtype]
    if kind == "tn":
        label = f"igemm_tn8_kernel<{tname}>" if lib().query("eg_conv_wgrad_variant", ctypes.byref(c), dtype) == 2 else f"igemm_tn_kernel<{tname}>"
        nbytes = (x_elems + y_elems) * es + w_elems * 4                     # activations + output gradients in, fp32 weight gradient out
    else:
        tile = nt_tile(c, dtype, kind == "bwd", ep)
        bm, bn = tile // 1000, tile % 1000
        label = {131: f"igemm_nt_buf_kernel<{tname}>", 132: f"igemm_nt_buf_kernel<{tname}>+splitk",
                 135: f"igemm_nt_pers_kernel<{tname}>",
                 147: f"igemm_nt8s_kernel<{tname},im2col>", 148: f"igemm_nt8s_kernel<{tname},im2col>+splitk",
                 149: f"igemm_nt8s_kernel<{tname},patch>", 150: f"igemm_nt8s_kernel<{tname},patch>+splitk",
                 151: f"igemm_nt8h_kernel<{tname}>", 152: f"igemm_nt8h_kernel<{tname}>+splitk"}.get(bn, f"igemm_nt_kernel<{tname},{bm},{bn}>")
        nbytes = (x_elems + y_elems + w_elems) * es
        if ep is not None and ep.mask:
            nbytes += (x_elems if kind == "bwd" else y_elems) * es
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _spacer()
    e0.record()
    lib().call(*args, _stream())
    e1.record()
    RECORDER.append((label, flops, e0, e1, f"{kind} B{c.B} H{c.H} Cin{c.Cin} Cout{c.Cout} k{c.k} s{c.stride}", float(nbytes)))


def conv_fwd(c, dtype, X, wp, Y, ep=None):
    if ep is None:
        ep = epilogue()          # carries the split-K scratch
    args = ("eg_conv_fwd", ctypes.byref(c), dtype, _p(X), _p(wp), _p(Y), ctypes.byref(ep) if ep is not None else None)
    if RECORDER is not None:
        return _timed("fwd", c, dtype, args, ep)
    lib().call(*args, _stream())


def conv_bwd_data(c, dtype, dY, wp, dX, ep=None):
    if ep is None:
        ep = epilogue()
    args = ("eg_conv_bwd_data", ctypes.byref(c), dtype, _p(dY), _p(wp), _p(dX), ctypes.byref(ep) if ep is not None else None)
    if RECORDER is not None:
        return _timed("bwd", c, dtype, args, ep)
    lib().call(*args, _stream())


def conv_wgrad_ws_bytes(c, dtype):
    return lib().query("eg_conv_wgrad_ws_bytes", ctypes.byref(c), dtype)


def conv_wgrad(c, dtype, X, dY, slab, wgs_target=0) -> int:
    """``wgs_target``: workgroups the parity-class kernel aims for (0: one per CU; 128 for launches forked beside the main chain)"""
    ns = ctypes.c_int(0)
    args = ("eg_conv_wgrad_target", ctypes.byref(c), dtype, _p(X), _p(dY), _p(slab), ctypes.addressof(ns), wgs_target)
    if RECORDER is not None:
        _timed("tn", c, dtype, args)
    else:
        lib().call(*args, _stream())
    return ns.value


def conv_wgrad_variant(c, dtype) -> int:
    """2: the parity-class kernel (igemm_tn8) runs this weight gradient, 1: the per-tap kernel"""
    return lib().query("eg_conv_wgrad_variant", ctypes.byref(c), dtype)


def wgrad_reduce(slab, nsplit, n_slab, n_rows, C, T, grad, accumulate=True):
    lib().call("eg_wgrad_reduce", _p(slab), nsplit, n_slab, n_rows, C, T, _p(grad), int(accumulate), _stream())


def wgrad_reduce_perm(slab, nsplit, n_slab, n_rows, C, T, grad, row_div=0, row_mul=0, c_row=0):
    lib().call("eg_wgrad_reduce_perm", _p(slab), nsplit, n_slab, n_rows, C, T, _p(grad), row_div, row_mul, c_row, _stream())


def gather_add(out, src, n, div, s_div, s_mod):
    lib().call("eg_gather_add", _p(out), _p(src), n, div, s_div, s_mod, _stream())


def sumpool2x2(dtype, x, y, B, H, W, C):
    lib().call("eg_sumpool2x2", dtype, _p(x), _p(y), B, H, W, C, _stream())


def wgrad_c1_ok(dtype, C, H, W, Cout, k, stride, pad) -> bool:
    return bool(lib().query("eg_wgrad_c1_ok", dtype, C, H, W, Cout, k, stride, pad))


def wgrad_c1_splits(B, H) -> int:
    return lib().query("eg_wgrad_c1_splits", B, H)


def wgrad_c1(dtype, x, dy, slab, B, H, W, C) -> int:
    """weight gradient of Conv2d(64, 1, 3, 1, 1) with the activation read once -> number of slabs [split][9][64] in ``slab``"""
    ns = ctypes.c_int(0)
    lib().call("eg_wgrad_c1", dtype, _p(x), _p(dy), _p(slab), B, H, W, C, ctypes.byref(ns), _stream())
    return ns.value


def up3_expand(w3, w4t, Cout, Cin):
    """effective ConvTranspose2d(4, 2, 1) master [Cin][Cout][4][4] of Upsample(2) + Conv2d(Cin -> Cout, 3, 1, 1) (MNIST/EAD-GAN_rpqmnxy.py:81-82)"""
    lib().call("eg_up3_expand", _p(w3), _p(w4t), Cout, Cin, _stream())


def up3_contract(dw4t, dw3, Cout, Cin, accumulate=True):
    """the transposed convolution's weight gradient [Cin][Cout][4][4] back onto the 3x3 master's gradient [Cout][Cin][3][3]"""
    lib().call("eg_up3_contract", _p(dw4t), _p(dw3), Cout, Cin, int(accumulate), _stream())


def sn_partials():
    return lib().query("eg_sn_partials")


def wgrad_reduce_sn(c, slab, nsplit, w_orig, sigma, u, v, gtmp, partials, grad):
    lib().call("eg_wgrad_reduce_sn", ctypes.byref(c), _p(slab), nsplit, _p(w_orig), _p(sigma), _p(u), _p(v), _p(gtmp), _p(partials), _p(grad), _stream())


def bias_grad_ws_floats(rows, N):
    return lib().query("eg_bias_grad_ws_floats", rows, N)


def bias_grad(dtype, dY, rows, N, partials, gb, bias_mod=0):
    lib().call("eg_bias_grad", dtype, _p(dY), rows, N, bias_mod, _p(partials), _p(gb), _stream())


def bias_grad_sn_ws_floats(rows, N, rows_per_tape):
    return lib().query("eg_bias_grad_sn_ws_floats", rows, N, rows_per_tape)


def bias_grad_sn(dtype, dzs, a, bias, rows, N, rows_per_tape, sigma, slope, ws, gb, coef):
    lib().call("eg_bias_grad_sn", dtype, _p(dzs), _p(a), _p(bias), rows, N, rows_per_tape, _p(sigma), slope, _p(ws), _p(gb), _p(coef), _stream())


def bias_grad_sn_fused(stat, nrb, N, tiles_m, tiles_per_tape, ntapes, sigma, gb, coef):
    lib().call("eg_bias_grad_sn_fused", _p(stat), nrb, N, tiles_m, tiles_per_tape, ntapes, _p(sigma), _p(gb), _p(coef), _stream())


def wgrad_reduce_rank1(slab, nsplit, n_slab, n_rows, C, T, grad, ntapes, coef, u, v, c_row=0):
    lib().call("eg_wgrad_reduce_rank1", _p(slab), nsplit, n_slab, n_rows, C, T, _p(grad), ntapes, _p(coef), _p(u), _p(v), c_row, _stream())


# ---- image side / heads ----------------------------------------------------------------------------
def conv_img_fwd(dtype, img, w_master, out, B, CI, H, W, N, k, stride, pad, ep=None):
    lib().call("eg_conv_img_fwd", dtype, _p(img), _p(w_master), _p(out), B, CI, H, W, N, k, stride, pad,
               ctypes.byref(ep) if ep is not None else None, _stream())


def conv_img_wgrad_ws_bytes(B, CI, N, k):
    return lib().query("eg_conv_img_wgrad_ws_bytes", B, CI, N, k)


def conv_img_wgrad(dtype, dz, img, slab, B, CI, H, W, N, k, stride, pad):
    lib().call("eg_conv_img_wgrad", dtype, _p(dz), _p(img), _p(slab), B, CI, H, W, N, k, stride, pad, _stream())


def im2col_img(dtype, img, out, B, CI, H, W, k, stride, pad, Kp):
    lib().call("eg_im2col_img", dtype, _p(img), _p(out), B, CI, H, W, k, stride, pad, Kp, _stream())


def conv_img_mfma_ok(dtype, C, H, W, N, k, stride, pad) -> bool:
    return bool(lib().query("eg_conv_img_mfma_ok", dtype, C, H, W, N, k, stride, pad))


def conv_img_mfma_stat_blocks(B, H, W, ntapes=1) -> int:
    return lib().query("eg_conv_img_mfma_stat_blocks", B, H, W, ntapes)


def conv_img_mfma(dtype, imgs, wp, out, B, C, H, W, ep=None, gates=None, gate_act=ACT_NONE, gate_slope=0.0, N=128):
    """Conv2d(C -> N = 128 / 64 / 32, 4, 2, 1) of up to three fp32 NCHW image tensors (tapes) straight on the MFMA units, no patch rows in HBM"""
    im = [_p(t) for t in imgs] + [None] * (3 - len(imgs))
    ga = [_p(t) for t in (gates or [])] + [None] * (3 - len(gates or []))
    lib().call("eg_conv_img_mfma_n", dtype, im[0], im[1], im[2], ga[0], ga[1], ga[2], len(imgs), _p(wp), _p(out), B, C, H, W, N,
               ctypes.byref(ep) if ep is not None else None, gate_act, gate_slope, _stream())


def wgrad_img_ok(dtype, C, H, W, N, k, stride, pad) -> bool:
    return bool(lib().query("eg_wgrad_img_ok", dtype, C, H, W, N, k, stride, pad))


def wgrad_img_splits(images, N=32) -> int:
    return lib().query("eg_wgrad_img_splits_n", images, N)


def wgrad_img(dtype, imgs, P, slab, B, C, H, W, N) -> int:
    """weight gradient of an image-side 4x4 / stride-2 layer straight from the fp32 images of up to three tapes (no patch rows in HBM) ->
    number of slabs [N][16 C] written to ``slab``"""
    im = [_p(t) for t in imgs] + [None] * (3 - len(imgs))
    ns = ctypes.c_int(0)
    lib().call("eg_wgrad_img", dtype, im[0], im[1], im[2], len(imgs), _p(P), _p(slab), B, C, H, W, N, ctypes.byref(ns), _stream())
    return ns.value


def convt_img_mfma_ok(dtype, C, Hin, Win, K, k, stride, pad) -> bool:
    return bool(lib().query("eg_convt_img_mfma_ok", dtype, C, Hin, Win, K, k, stride, pad))


def convt_img_mfma(dtype, a, wp, bias, out, B, C, Hin, Win, act=ACT_NONE, slope=0.0, K=128):
    """ConvTranspose2d(K = 128 / 64 -> C <= 3, 4, 2, 1) (+ bias + activation) from 16-bit NHWC activations to an fp32 NCHW image in one launch"""
    lib().call("eg_convt_img_mfma_k", dtype, _p(a), _p(wp), _p(bias), _p(out), B, C, Hin, Win, K, act, slope, _stream())


def cast_pad(dtype, src, dst, rows, n, npad):
    lib().call("eg_cast_pad", dtype, _p(src), _p(dst), rows, n, npad, _stream())


def act_grad_mul_bias_nchw(g, a, out, B, C, HW, act, slope, partial, gb):
    lib().call("eg_act_grad_mul_bias_nchw", _p(g), _p(a), _p(out), B, C, HW, act, slope, _p(partial), _p(gb), _stream())


def head_fused_ok(dtype, T, K, N):
    return bool(lib().cdll.eg_head_fused_ok(dtype, T, K, N))


PACK_BATCH = os.environ.get("EG_PACK_BATCH", "1") != "0"      # 0: every pack its own launch (A/B runs)


class PackBatch:
    """The pack launches issued by ``fn()`` as ONE launch (eg_pack_record_begin / _end / eg_pack_multi).  Eager calls record again (host
    work only) and re-upload the job table if a pointer or a geometry changed; inside a hipGraph capture the table of the last eager call
    is launched as it is (the capture contract: an eager iteration has run on the same buffers)."""
    MAX_JOBS = 64

    def __init__(self):
        self.host = None
        self.dev = None
        self.n = self.nb = 0

    def run(self, fn):
        if not PACK_BATCH:
            return fn()
        if not (self.dev is not None and torch.cuda.is_current_stream_capturing()):
            jb = lib().cdll.eg_pack_job_bytes()
            cap = self.MAX_JOBS * jb
            buf = ctypes.create_string_buffer(cap)
            n, nb = ctypes.c_int(0), ctypes.c_int(0)
            lib().call("eg_pack_record_begin")
            try:
                fn()
            finally:
                lib().call("eg_pack_record_end", buf, cap, ctypes.byref(n), ctypes.byref(nb))
            host = buf.raw[:n.value * jb]
            if host != self.host:
                self.host, self.n, self.nb = host, n.value, nb.value
                self.dev = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(torch.device("cuda", torch.cuda.current_device())) if host else None
        if self.dev is not None and self.n:
            lib().call("eg_pack_multi", _p(self.dev), self.n, self.nb, _stream())


def batched_packs(fn):
    """decorator of an engine's ``repack...`` method: its pack launches run as one (PackBatch per method and argument tuple)"""
    def wrapper(self, *a):
        batches = self.__dict__.setdefault("_pack_batches", {})
        key = (fn.__name__,) + a
        b = batches.get(key)
        if b is None:
            b = batches[key] = PackBatch()
        return b.run(lambda: fn(self, *a))
    wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
    return wrapper


def dense_small_fwd_slices(dtype, x, wp, B, K, Kpad, N, ws):
    """the K-slice sums of dense_small_fwd (no combine launch) -> number of slices in ``ws`` [slice][B][N]"""
    ns = ctypes.c_int(0)
    lib().call("eg_dense_small_fwd_slices", dtype, _p(x), _p(wp), B, K, Kpad, N, _p(ws), ws.numel(), ctypes.byref(ns), _stream())
    return ns.value


def head_fused(dtype, x, wp, bias, partials, nslice, y, dout, dx, sigma, B, T, K, Kpad, N, loss, terms, counter, mask_act, mask_slope, targets=None,
               scales=None, info=None):
    """Behind dense_small_fwd_slices: slice combine, losses and head input gradient of the T rows of every sample in one launch
    (eg_head_fused).  ``targets`` / ``scales``: the adversarial BCE term of each tape; ``info`` = (c_cont, n_cont, n_cat, code, labels, lcat,
    lcon, laff): the info step's three losses (tapes: generated, transformed, real)."""
    h = EgHead()
    h.x, h.wp, h.bias, h.y, h.dout, h.dx, h.sigma = _p(x), _p(wp), _p(bias), _p(y), _p(dout), _p(dx), _p(sigma)
    h.partials, h.nslice = _p(partials), nslice
    h.B, h.T, h.K, h.Kpad, h.N = B, T, K, Kpad, N
    h.loss, h.terms, h.counter, h.mask_act, h.mask_slope = _p(loss), _p(terms), _p(counter), mask_act, float(mask_slope)
    if info is None:
        h.mode = 0
        for t in range(T):
            h.target[t], h.scale[t] = float(targets[t]), float(scales[t])
    else:
        h.mode = 1
        c_cont, n_cont, n_cat, code, labels, lcat, lcon, laff = info
        h.c_cont, h.n_cont, h.n_cat, h.code, h.ldc, h.labels = c_cont, n_cont, n_cat, _p(code), code.stride(0), _p(labels)
        h.lcat, h.lcon, h.laff = float(lcat), float(lcon), float(laff)
    lib().call("eg_head_fused", dtype, ctypes.byref(h), _stream())


def dense_small_bgrad(dy, gb, B, N):
    lib().call("eg_dense_small_bgrad", _p(dy), _p(gb), B, N, _stream())


def flat_reduce(slab, nslab, total, grad, accumulate=True):
    lib().call("eg_flat_reduce", _p(slab), nslab, total, _p(grad), int(accumulate), _stream())


def flat_reduce_sn(slab, nslab, rows, Kdim, w_orig, sigma, u, v, gtmp, partials, grad):
    lib().call("eg_flat_reduce_sn", _p(slab), nslab, rows, Kdim, _p(w_orig), _p(sigma), _p(u), _p(v), _p(gtmp), _p(partials), _p(grad), _stream())


def bias_grad_nchw(x, B, C, HW, gb):
    lib().call("eg_bias_grad_nchw", _p(x), B, C, HW, _p(gb), _stream())


def dense_small_fwd(dtype, x, wp, bias, y, B, K, Kpad, N, ws=None):
    lib().call("eg_dense_small_fwd", dtype, _p(x), _p(wp), _p(bias), _p(y), B, K, Kpad, N, _p(ws), ws.numel() if ws is not None else 0, _stream())


def dense_small_fwd_sn(dtype, x, wp, bias, y, B, K, Kpad, N, sigma, sigma_rows, ws=None):
    lib().call("eg_dense_small_fwd_sn", dtype, _p(x), _p(wp), _p(bias), _p(y), B, K, Kpad, N, _p(sigma), sigma_rows,
               _p(ws), ws.numel() if ws is not None else 0, _stream())


def head_prep_sn(dtype, dy, ldy, y, ldyy, bias, rows, N, sigma, rows_per_tape, dys, npad, col0, gb, coef, dys32=None, ld32=0):
    lib().call("eg_head_prep_sn", dtype, _p(dy), ldy, _p(y), ldyy, _p(bias), rows, N, _p(sigma), rows_per_tape, _p(dys), npad, col0, _p(gb), _p(coef),
               _p(dys32), ld32, _stream())


def dense_small_bwd(dtype, dy, wp, mask, dx, B, K, Kpad, N, mask_act=ACT_NONE, mask_slope=0.0, sigma=None, sigma_rows=0):
    lib().call("eg_dense_small_bwd", dtype, _p(dy), _p(wp), _p(mask), _p(dx), B, K, Kpad, N, mask_act, mask_slope, _p(sigma), sigma_rows, _stream())


def dense_small_wgrad(dtype, dy, x, gw, gb, B, K, N, Cin, taps):
    lib().call("eg_dense_small_wgrad", dtype, _p(dy), _p(x), _p(gw), _p(gb), B, K, N, Cin, taps, _stream())


# ---- norm / spectral norm / optimiser -----------------------------------------------------------------
def bn_ws_floats(M, C):
    return lib().query("eg_bn_ws_floats", M, C)


def bn_fwd_eval(dtype, x, y, M, C, gamma, beta, eps, running_mean, running_var, ws, act=ACT_NONE, slope=0.0):
    lib().call("eg_bn_fwd_eval", dtype, _p(x), _p(y), M, C, _p(gamma), _p(beta), float(eps), _p(running_mean), _p(running_var), _p(ws),
               act, float(slope), _stream())


def bn_stats_local(dtype, x, M, C, ws, stats):
    lib().call("eg_bn_stats_local", dtype, _p(x), M, C, _p(ws), _p(stats), _stream())


def bn_fwd_from_stats(dtype, x, y, M_local, C, stats_all, nranks, M_global, gamma, beta, eps, momentum, rmean, rvar, nbt, save_mean, save_invstd, ws,
                      act=ACT_NONE, slope=0.0):
    lib().call("eg_bn_fwd_from_stats", dtype, _p(x), _p(y), M_local, C, _p(stats_all), nranks, M_global, _p(gamma), _p(beta), float(eps),
               float(momentum if momentum is not None else 0.1), _p(rmean), _p(rvar), _p(nbt), _p(save_mean), _p(save_invstd), _p(ws), act, float(slope),
               _stream())


def bn_bwd_sums_local(dtype, z, da, M, C, gamma, beta, save_mean, save_invstd, act, slope, dgamma, dbeta, sums, ws):
    lib().call("eg_bn_bwd_sums_local", dtype, _p(z), _p(da), M, C, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd), act, float(slope),
               _p(dgamma), _p(dbeta), _p(sums), _p(ws), _stream())


def bn_bwd_from_sums(dtype, z, da, dz, M_local, C, sums_global, M_global, gamma, beta, save_mean, save_invstd, act, slope, ws):
    lib().call("eg_bn_bwd_from_sums", dtype, _p(z), _p(da), _p(dz), M_local, C, _p(sums_global), M_global, _p(gamma), _p(beta), _p(save_mean),
               _p(save_invstd), act, float(slope), _p(ws), _stream())


def bn_fwd_train(dtype, x, y, M, C, gamma, beta, eps, momentum, rmean, rvar, nbt, save_mean, save_invstd, ws, act=ACT_NONE, slope=0.0):
    lib().call("eg_bn_fwd_train", dtype, _p(x), _p(y), M, C, _p(gamma), _p(beta), eps, momentum, _p(rmean), _p(rvar), _p(nbt),
               _p(save_mean), _p(save_invstd), _p(ws), act, slope, _stream())


def bn_fwd_train_fused(dtype, x, y, M, C, stat, nrb, rows_per_block, gamma, beta, eps, momentum, rmean, rvar, nbt, save_mean, save_invstd, ws,
                       act=ACT_NONE, slope=0.0):
    lib().call("eg_bn_fwd_train_fused", dtype, _p(x), _p(y), M, C, _p(stat), nrb, rows_per_block, _p(gamma), _p(beta), eps, momentum, _p(rmean), _p(rvar),
               _p(nbt), _p(save_mean), _p(save_invstd), _p(ws), act, slope, _stream())


def bn_bwd_fused(dtype, z, dy, dz, M, C, stat, nrb, gamma, beta, save_mean, save_invstd, dgamma, dbeta, sums, ws):
    lib().call("eg_bn_bwd_fused", dtype, _p(z), _p(dy), _p(dz), M, C, _p(stat), nrb, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
               _p(dgamma), _p(dbeta), _p(sums), _p(ws), _stream())


def bn_bwd(dtype, z, da, dz, M, C, gamma, beta, save_mean, save_invstd, act, slope, dgamma, dbeta, sums, ws):
    lib().call("eg_bn_bwd", dtype, _p(z), _p(da), _p(dz), M, C, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd), act, slope,
               _p(dgamma), _p(dbeta), _p(sums), _p(ws), _stream())


def bn_bwd_post(dtype, z, da, dz, M, C, gamma, beta, save_mean, save_invstd, dgamma, dbeta, sums, ws, post_act, post_slope, post_sigma):
    lib().call("eg_bn_bwd_post", dtype, _p(z), _p(da), _p(dz), M, C, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd), _p(dgamma), _p(dbeta),
               _p(sums), _p(ws), post_act, post_slope, _p(post_sigma), _stream())


def sn_ws_floats(R, Kd):
    return lib().query("eg_sn_ws_floats", R, Kd)


def sn_power_iter(w_orig, R, Kd, u, v, sigma, u_snap, v_snap, ws, training=True, eps=1e-12):
    lib().call("eg_sn_power_iter", _p(w_orig), R, Kd, _p(u), _p(v), _p(sigma), _p(u_snap), _p(v_snap), _p(ws), int(training), eps, _stream())


def sn_layers(entries):
    """entries: list of (w_orig, u, v, sigma, u_snap, v_snap) tensors -> ctypes array of eg_sn_layer (keep it alive)."""
    arr = (EgSnLayer * len(entries))()
    for i, (w, u, v, sg, us, vs) in enumerate(entries):
        arr[i] = EgSnLayer(_p(w), _p(u), _p(v), _p(sg), _p(us), _p(vs), w.shape[0], w.numel() // w.shape[0])
    return arr


def sn_multi_ws_floats(arr):
    return lib().query("eg_sn_multi_ws_floats", arr, len(arr))


# EXPERIMENT (default off): the power iteration of a network's layers as two launches instead of four (eg_sn_power_iter_multi2: each stage's
# per-layer finish by the last workgroup to arrive).  Same bits, but slower in the step: CelebA 4.33 -> 4.38 ms, dSprites 1.413 -> 1.424
# (profiles/r03_zf_ab_sn2.txt) -- the wide first stage runs 1024-thread workgroups and every layer's finish waits for its last row.
SN_TWO_LAUNCHES = os.environ.get("EG_SN2", "0") != "0"


def sn_power_iter_multi(arr, ws, training=True, eps=1e-12, counters=None):
    """``counters``: int32 tensor of >= 2 * len(arr) zeros owned by the caller (one per engine and stream of use): the iteration runs as two
    launches instead of four, same bits (eg_sn_power_iter_multi2)"""
    if counters is None or not SN_TWO_LAUNCHES:
        lib().call("eg_sn_power_iter_multi", arr, len(arr), _p(ws), int(training), eps, _stream())
    else:
        lib().call("eg_sn_power_iter_multi2", arr, len(arr), _p(ws), _p(counters), int(training), eps, _stream())


def adam_step(p, g, m, v, n, lr, b1, b2, eps, step, tick=True):
    lib().call("eg_adam_step", _p(p), _p(g), _p(m), _p(v), n, lr, b1, b2, eps, _p(step), int(tick), _stream())


def adam_step_zero(p, g, m, v, n, lr, b1, b2, eps, step, tick=True, zero_grad=False):
    """Adam on a slice (views of the arena tensors), optionally clearing the gradient slice in the same pass"""
    lib().call("eg_adam_step_zero", _p(p), _p(g), _p(m), _p(v), n, lr, b1, b2, eps, _p(step), int(tick), int(zero_grad), _stream())


def adam_tick(step):
    lib().call("eg_adam_tick", _p(step), _stream())


def adam_pack_conv_ok(c, dtype, has_fwd, has_bwd) -> bool:
    return bool(lib().query("eg_adam_pack_conv_ok", ctypes.byref(c), dtype, int(has_fwd), int(has_bwd)))


def adam_pack_conv(c, dtype, p, g, m, v, lr, b1, b2, eps, step, zero_grad, wp_fwd, wp_bwd):
    """optimizer.step() on one convolution weight (slices of the four arenas) + refresh of its packed panels, one launch"""
    lib().call("eg_adam_pack_conv", ctypes.byref(c), dtype, _p(p), _p(g), _p(m), _p(v), lr, b1, b2, eps, _p(step), int(zero_grad), _p(wp_fwd), _p(wp_bwd),
               _stream())


def adam_pack_rows(dtype, p, g, m, v, wp, K, N, Kpad, n_mod, n_mul, lr, b1, b2, eps, step, zero_grad):
    lib().call("eg_adam_pack_rows", dtype, _p(p), _p(g), _p(m), _p(v), _p(wp), K, N, Kpad, n_mod, n_mul, lr, b1, b2, eps, _p(step), int(zero_grad), _stream())


def clear_errors():
    return lib().query("eg_clear_errors")


def fill_f32(t, val=0.0):
    lib().call("eg_fill_f32", _p(t), t.numel(), val, _stream())


# ---- utilities ------------------------------------------------------------------------------------------
def concat_cast(dtype, a, b, c, out, B, Cpad):
    wa = a.shape[1]
    wb = b.shape[1] if b is not None else 0
    wc = c.shape[1] if c is not None else 0
    lib().call("eg_concat_cast", dtype, _p(a), wa, _p(b), wb, _p(c), wc, B, Cpad, _p(out), _stream())


def act_grad_mul_f32(g, a, out, act, slope=0.0):
    lib().call("eg_act_grad_mul_f32", _p(g), _p(a), _p(out), g.numel(), act, slope, _stream())


def nchw_to_nhwc(dtype, x, y, B, C, HW, Cpad):
    lib().call("eg_nchw_to_nhwc", dtype, _p(x), _p(y), B, C, HW, Cpad, _stream())


def nhwc_to_nchw(dtype, x, y, B, C, HW, Cpad):
    lib().call("eg_nhwc_to_nchw", dtype, _p(x), _p(y), B, C, HW, Cpad, _stream())


# ---- affine / warp / losses --------------------------------------------------------------------------------
def theta_rpqxy(code, ldc, B, theta):
    lib().call("eg_theta_rpqxy", _p(code), ldc, B, _p(theta), _stream())


def warp_affine(img, theta, out, B, C, H, W):
    lib().call("eg_warp_affine", _p(img), _p(theta), _p(out), B, C, H, W, _stream())


def warp_affine_rpqxy(img, code, ldc, theta_out, out, B, C, H, W, zero=None):
    """theta_rpqxy + warp_affine in one launch; ``zero``: a small fp32 tensor the launch clears first"""
    lib().call("eg_warp_affine_rpqxy", _p(img), _p(code), ldc, _p(theta_out), _p(out), B, C, H, W, _p(zero), zero.numel() if zero is not None else 0, _stream())


def mlp_rpqmnxy_floats():
    return lib().query("eg_mlp_rpqmnxy_floats")


def theta_rpqmnxy(code, ldc, B, theta):
    lib().call("eg_theta_rpqmnxy", _p(code), ldc, B, _p(theta), _stream())


def loss_affine_rpqmnxy(o_real, o_trans, ld, c0, B, code, ldc, mlp, scale, loss, d_real, d_trans, pred_out, ws):
    lib().call("eg_loss_affine_rpqmnxy", _p(o_real), _p(o_trans), ld, c0, B, _p(code), ldc, _p(mlp), scale, _p(loss), _p(d_real), _p(d_trans),
               _p(pred_out), _p(ws), _stream())


def theta_rp(code, ldc, B, theta):
    lib().call("eg_theta_rp", _p(code), ldc, B, _p(theta), _stream())


def theta_pxy_align_inv(code, ldc, B, theta):
    lib().call("eg_theta_pxy_align_inv", _p(code), ldc, B, _p(theta), _stream())


def loss_affine_rp(o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out=None):
    lib().call("eg_loss_affine_rp", _p(o_real), _p(o_trans), ld, c0, B, _p(code), ldc, scale, _p(loss), _p(d_real), _p(d_trans), _p(pred_out), _stream())


def loss_mutual_info(o, ld, c0, n, B, tgt, ldt, t0, target_logits, scale, loss, dout):
    lib().call("eg_loss_mutual_info", _p(o), ld, c0, n, B, _p(tgt), ldt, t0, int(target_logits), scale, _p(loss), _p(dout), _stream())


def u8_colorize(sprites, gain, out, B, C, HW):
    lib().call("eg_u8_colorize", _p(sprites), _p(gain), _p(out), B, C, HW, _stream())


def color_scale(inp, code, ldc, c0, factor, divide, out, B, C, HW):
    lib().call("eg_color_scale", _p(inp), _p(code), ldc, c0, factor, int(divide), _p(out), B, C, HW, _stream())


def loss_affine_rp_color(o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out=None):
    lib().call("eg_loss_affine_rp_color", _p(o_real), _p(o_trans), ld, c0, B, _p(code), ldc, scale, _p(loss), _p(d_real), _p(d_trans), _p(pred_out), _stream())


def add_f32(out, a, b):
    lib().call("eg_add_f32", _p(out), _p(a), _p(b), out.numel(), _stream())


def u8_to_f32(x, y):
    lib().call("eg_u8_to_f32", _p(x), _p(y), y.numel(), _stream())


def loss_bce_sigmoid(o, ld, col, B, target, scale, loss, dout, zero_rows=True):
    lib().call("eg_loss_bce_sigmoid", _p(o), ld, col, B, target, scale, _p(loss), _p(dout), int(zero_rows), _stream())


def loss_mse(o, ld, col0, n, B, tgt, ldt, tconst, scale, loss, dout, zero_rows=True):
    lib().call("eg_loss_mse", _p(o), ld, col0, n, B, _p(tgt), ldt, tconst, scale, _p(loss), _p(dout), int(zero_rows), _stream())


def loss_ce_softmaxed(o, ld, c0, n, B, labels, scale, loss, dout):
    lib().call("eg_loss_ce_softmaxed", _p(o), ld, c0, n, B, _p(labels), scale, _p(loss), _p(dout), _stream())


def loss_affine_rpqxy(o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out=None):
    lib().call("eg_loss_affine_rpqxy", _p(o_real), _p(o_trans), ld, c0, B, _p(code), ldc, scale, _p(loss), _p(d_real), _p(d_trans), _p(pred_out), _stream())


def loss_info_rpqxy(o_gen, o_trans, o_real, ld, c_cont, n_cont, n_cat, B, code, ldc, labels, lcat, lcon, laff, loss, d_gen, d_trans, d_real):
    """loss_mse + loss_ce_softmaxed + loss_affine_rpqxy of the CelebA info step as one launch (same numbers)"""
    lib().call("eg_loss_info_rpqxy", _p(o_gen), _p(o_trans), _p(o_real), ld, c_cont, n_cont, n_cat, B, _p(code), ldc, _p(labels), lcat, lcon, laff,
               _p(loss), _p(d_gen), _p(d_trans), _p(d_real), _stream())


# ---- device-side input pipeline -------------------------------------------------------------------
RNG_UNIFORM, RNG_NORMAL, RNG_RANDINT, RNG_BERNOULLI, RNG_EPOCH_PERM = 0, 1, 2, 3, 4


def rng_fill(kind, out, a, b, seed, step, stream_id):
    """out <- kind(a, b) from Philox4x32-10 keyed by ``seed``, counted by (element, device counter ``step``, ``stream_id``)"""
    lib().call("eg_rng_fill", kind, _p(out), out.numel(), float(a), float(b), int(seed), _p(step), int(stream_id), _stream())


def rng_fill_multi(draws, seed, step):
    """several draws in ONE launch, the values of one rng_fill each: draws = [(kind, out, a, b, stream_id[, onehot tensor])]"""
    arr = (EgRngSeg * len(draws))()
    for i, d in enumerate(draws):
        kind, out, a, b, sid = d[:5]
        oh = d[5] if len(d) > 5 else None
        arr[i] = EgRngSeg(kind, _p(out), out.numel(), float(a), float(b), int(sid), _p(oh), oh.shape[1] if oh is not None else 0)
    lib().call("eg_rng_fill_multi", arr, len(draws), int(seed), _p(step), _stream())


def counter_add(counter, v=1):
    lib().call("eg_counter_add", _p(counter), int(v), _stream())


def gather_u8_images(data, idx, flip, out, B, C, H, W, scale, shift, tick=None):
    """``tick``: device step counter incremented by this launch (the iteration's draws, earlier on the stream, have read it)"""
    if tick is None:
        lib().call("eg_gather_u8_images", _p(data), _p(idx), _p(flip), _p(out), B, C, H, W, float(scale), float(shift), _stream())
    else:
        lib().call("eg_gather_u8_images_tick", _p(data), _p(idx), _p(flip), _p(out), B, C, H, W, float(scale), float(shift), _p(tick), _stream())


def resample_u8(src, dst, planes, in_h, in_w, axis, bounds, kk, ksize, o0, on, c0, cn):
    lib().call("eg_resample_u8", _p(src), _p(dst), planes, in_h, in_w, axis, _p(bounds), _p(kk), ksize, o0, on, c0, cn, _stream())


def onehot(labels, out, B, n):
    lib().call("eg_onehot", _p(labels), _p(out), B, n, _stream())


# ---- stage-1 (Encoder_pxy) trainer ------------------------------------------------------------------
def theta_pxy(code, ldc, B, theta):
    lib().call("eg_theta_pxy", _p(code), ldc, B, _p(theta), _stream())


def loss_affine_pxy(o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out=None, ncol=0):
    lib().call("eg_loss_affine_pxy", _p(o_real), _p(o_trans), ld, c0, B, _p(code), ldc, ncol, float(scale), _p(loss), _p(d_real), _p(d_trans),
               _p(pred_out), _stream())


def warp_affine_zeros(img, theta, out, B, C, H, W):
    lib().call("eg_warp_affine_zeros", _p(img), _p(theta), _p(out), B, C, H, W, _stream())


def affine_para_rpqmnxy(code, ldc, B, para):
    lib().call("eg_affine_para_rpqmnxy", _p(code), ldc, B, _p(para), _stream())


def make_grid(img, B, C, H, W, nrow, padding, pad_value, rng2, grid):
    lib().call("eg_make_grid", _p(img), B, C, H, W, nrow, padding, pad_value, _p(rng2), _p(grid), _stream())


def minmax_ws_floats():
    return lib().query("eg_minmax_ws_floats")


def minmax_f32(x, n, ws, out2):
    lib().call("eg_minmax_f32", _p(x), n, _p(ws), _p(out2), _stream())


def quantize_u8(x, C, H, W, rng2, out):
    lib().call("eg_quantize_u8", _p(x), C, H, W, _p(rng2), _p(out), _stream())


def col2im_img(dtype, cols, B, C, Hin, Win, k, stride, pad, bias, act, slope, out):
    lib().call("eg_col2im_img", dtype, _p(cols), B, C, Hin, Win, k, stride, pad, _p(bias), act, slope, _p(out), _stream())
