"""CelebA 64x64 path of EAD-GAN on MI355X: drop-in ``Generator`` / ``Discriminator`` / ``transformation_2D`` /
``get_matrix`` / ``affine_regularzier`` (names and signatures of celebA/EAD-GAN_celebA.py:67-158 and
celebA/utils_rpqxy.py:59-116) plus the fused train-loop entry :class:`CelebATrainer` (loop body :299-401).

Everything below the Python class surface runs as hand-written HIP kernels through the C ABI
(include/eadgan_hip.h).  ``nn.ConvTranspose2d`` / ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects are kept ONLY as
parameter containers so that ``state_dict()`` keys, shapes and default initialisation are the reference's
(incl. ``weight_orig`` / ``weight_u`` / ``weight_v``); their ``forward`` is never called.
"""
from __future__ import annotations

import argparse
import os

import numpy as np

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops
from .engine import Arena, ConvRec, SideStream, Workspace, capture_segments, capture_step, check_usable, parse_dtype
from .ops import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, EG_BF16, EG_F32, OUT_NCHW_F32)

# module-level hyper-parameters, mirroring the reference's global ``opt`` (argparse defaults, :39-51)
opt = argparse.Namespace(n_epochs=50, batch_size=16, lr=0.0002, b1=0.5, b2=0.999, n_cpu=8, latent_dim=200, code_dim=8,
                         n_classes=10, img_size=64, channels=3, sample_interval=4000)

# image-side transposed convolutions (128 -> C) as one GEMM + col2im gather (1) or as the 4-phase implicit GEMM (0)
IMG_GEMM = os.environ.get("EG_IMG_GEMM", "1") == "1"

# the iteration's head (device-side draws, gather, affine matrix + warp) as 3 launches instead of 11 (same values)
FUSE_INPUTS = os.environ.get("EG_FUSE_INPUTS", "1") != "0"

# image-side convolutions (first D layer forward, input gradient of G's last layer) straight from the fp32 images on the MFMA units
# (ops.conv_img_mfma) instead of patch rows in HBM + a K = 64 GEMM; the patch rows remain for the weight gradients, off the main chain
IMG_DIRECT = os.environ.get("EG_IMG_DIRECT", "1") != "0"

# column statistics (BatchNorm batch statistics / backward sums, bias gradient + spectral-norm coefficient) taken from the epilogue of the
# convolution that produces the tensor (eg_epilogue.stat_mode) instead of by kernels that re-read it; 0: the stand-alone kernels (A/B runs)
FUSE_STATS = os.environ.get("EG_FUSE_STATS", "1") != "0"

# optimizer.step() of a convolution weight and the refresh of its packed panels as ONE launch per layer (ops.adam_pack_conv / adam_pack_rows);
# 0: one Adam launch over the arena (or bucket) followed by the re-packing launches (A/B runs; same bits either way)
# the discriminator's head with the sub-step's losses and the head's input gradient in two launches (K-sliced dense head; eg_head_fused: slice
# combine + losses + dense backward, the affine term's Jacobian spread over lanes) instead of four or five on the main chain; same bits;
# EG_FUSE_HEAD=0: the separate launches
FUSE_HEAD = os.environ.get("EG_FUSE_HEAD", "1") != "0"
FUSE_ADAM = os.environ.get("EG_FUSE_ADAM", "1") != "0"
# ... per optimizer update of the pipelined step (g1, d2, d3, g3 as in EG_BUCKET_OPT; "all")
FUSE_ADAM_AT = os.environ.get("EG_FUSE_ADAM_AT", "g3")


def adam_bucket(eng, arena, tag, lo, hi, m, v, lr, betas, step, zero, fuse=True):
    """optimizer.step() (+ zero_grad) on one gradient bucket [lo, hi) of ``arena`` and the refresh of the packed panels of its layers.  The
    bucket's big convolution weight (``eng.fused_weight(tag)``) is updated AND re-packed by one launch; the remaining parameters of the
    bucket (biases, BatchNorm affine) by plain Adam launches on their slices.  The step counter has been ticked by the caller."""
    b1, b2 = betas
    fw = eng.fused_weight(tag) if (FUSE_ADAM and fuse) else None
    if fw is None:
        ops.adam_step_zero(arena.flat[lo:hi], arena.grad[lo:hi], m[lo:hi], v[lo:hi], hi - lo, lr, b1, b2, 1e-8, step, False, zero)
        eng.repack_bucket(tag)
        return
    name, launch = fw
    off, k = arena.slices[name]
    assert lo <= off and off + k <= hi
    launch(arena.flat[off:off + k], arena.grad[off:off + k], m[off:off + k], v[off:off + k], lr, b1, b2, step, zero)
    for a, b in ((lo, off), (off + k, hi)):
        if b > a:
            ops.adam_step_zero(arena.flat[a:b], arena.grad[a:b], m[a:b], v[a:b], b - a, lr, b1, b2, 1e-8, step, False, zero)


G_WIDTHS = (1024, 512, 256, 128)
D_WIDTHS = (128, 256, 512, 1024)
LRELU_SLOPE = 0.1
SN_EPS = 1e-12


def to_categorical(y, num_columns, device=None):
    """one-hot float tensor (celebA/EAD-GAN_celebA.py:56-62)."""
    y = torch.as_tensor(np.asarray(y), dtype=torch.int64, device=device)
    return torch.nn.functional.one_hot(y, num_columns).to(torch.float32)


def _require_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("ead-gan_amd runs on MI355X only: tensors must be on a HIP device (no CPU fallback)")


# ================================================================================================
# Generator
# ================================================================================================
class _GenEngine:
    """Static-shape forward/backward of the generator at one batch size."""

    def __init__(self, gen: "Generator", B: int, dtype: int):
        self.gen, self.B, self.dtype = gen, B, dtype
        dev = gen.arena.flat.device
        tdt = ops.torch_dtype(dtype)
        self.ws = ws = Workspace.get(dev)
        self.cin = gen.input_dim
        self.cpad = ops.round_up(self.cin, 8)
        W = G_WIDTHS
        s = gen.init_size                                              # 4
        # L0: ConvTranspose2d(cin,1024,4,1,0) on a 1x1 input == GEMM [B,cin] x [cin, 16*1024]
        self.l0 = ConvRec(dtype, B, 1, 1, self.cpad, 16 * W[0], 1, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.l0w = ConvRec(dtype, B, 1, 1, 16 * W[0], self.cpad, 1, 1, 0, device=dev, want_fwd=False, want_bwd=False, ws=ws)
        # L1..L3: ConvTranspose2d(k4,s2,p1) in conv view (Cout_cv = ConvT in-channels)
        self.mid = [ConvRec(dtype, B, s * 2 ** (i + 1), s * 2 ** (i + 1), W[i + 1], W[i], 4, 2, 1, device=dev, ws=ws) for i in range(3)]
        self.l4 = ConvRec(dtype, B, s * 16, s * 16, gen.channels, W[3], 4, 2, 1, device=dev, want_fwd=False, want_wgrad=False, ws=ws)
        # the same layer seen from the image side as a 1x1 conv over im2col patches (K = C*16): input/weight gradients on MFMA
        self.kp = gen.channels * 16
        self.l4p = ConvRec(dtype, B, s * 8, s * 8, self.kp, W[3], 1, 1, 0, device=dev, want_bwd=False, ws=ws)
        self.patches = torch.empty(B * (s * 8) ** 2, self.kp, device=dev, dtype=tdt)
        # forward of the same layer as ONE GEMM over the input pixels with N = 16 taps x C columns + a col2im gather (eg_col2im_img): every
        # activation is read once instead of 16 times (4 phases x 4 taps of the implicit transposed convolution)
        self.l4g = ConvRec(dtype, B, s * 8, s * 8, W[3], self.kp, 1, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.cols4 = torch.empty(B * (s * 8) ** 2, self.kp, device=dev, dtype=tdt)
        ws.need_small(B * gen.channels)
        # activations
        e = lambda *shape, dt=tdt: torch.empty(shape, device=dev, dtype=dt)
        self.inp = e(B, self.cpad)
        self.h0 = e(B, s, s, W[0])
        self.z = [e(B, s * 2 ** (i + 1), s * 2 ** (i + 1), W[i + 1]) for i in range(3)]
        self.a = [torch.empty_like(t) for t in self.z]
        self.mean = [e(W[i + 1], dt=torch.float32) for i in range(3)]
        self.invstd = [e(W[i + 1], dt=torch.float32) for i in range(3)]
        self.bn_stats = [e(3 * W[i + 1], dt=torch.float32) for i in range(3)]      # synchronised BatchNorm: local (n, mean, M2) / local sums
        self.bn_sums = [e(2 * W[i + 1], dt=torch.float32) for i in range(3)]
        self.img = e(B, gen.channels, s * 16, s * 16, dt=torch.float32)
        # gradient scratch (ping-pong between layers)
        self.dimg_z = torch.empty_like(self.img)
        self.da = [torch.empty_like(t) for t in self.z]
        self.dz = [torch.empty_like(t) for t in self.z]
        self.dh0 = torch.empty_like(self.h0)
        for i in range(3):
            ws.need_small(ops.bn_ws_floats(self.z[i].numel() // W[i + 1], W[i + 1]))
        ws.need_small(ops.bias_grad_ws_floats(B * 16, W[0]))
        ws.need_sums(2 * max(W))
        self._stat = {}                                                # (kind, layer) -> (row blocks, buffer) of the fused column statistics
        self.repack()

    def _p(self, idx, kind):
        return getattr(self.gen.conv_blocks[idx], kind)

    def _stat_buf(self, key, c, bwd, ep, C):
        """(nrb, buffer) if the launch described by (c, bwd, ep) can take its column statistics in the epilogue, else (0, None).  Asked
        once per layer with the very epilogue the launch uses (same hints, same split-K scratch)."""
        if key not in self._stat:
            nrb = ops.conv_stat_blocks(c, self.dtype, bwd, ep) if FUSE_STATS else 0
            self._stat[key] = (nrb, torch.empty(2 * C * nrb, device=self.inp.device, dtype=torch.float32) if nrb else None)
        return self._stat[key]

    # gradient buckets in the order the backward pass completes them: (tag of the lane chain that writes it last, parameter names)
    BUCKETS = (("G4", ("conv_blocks.10.weight", "conv_blocks.10.bias")),
               ("G3", ("conv_blocks.7.weight", "conv_blocks.7.bias", "conv_blocks.8.weight", "conv_blocks.8.bias")),
               ("G2", ("conv_blocks.4.weight", "conv_blocks.4.bias", "conv_blocks.5.weight", "conv_blocks.5.bias")),
               ("G1", ("conv_blocks.1.weight", "conv_blocks.1.bias", "conv_blocks.2.weight", "conv_blocks.2.bias")),
               ("G0", ("conv_blocks.0.weight", "conv_blocks.0.bias")))

    def repack(self):
        for tag, _ in self.BUCKETS:
            self.repack_bucket(tag)

    def fused_weight(self, tag):
        """(parameter name, launcher) of the bucket's convolution weight if Adam + re-packing of it run as one launch, else None"""
        dt = self.dtype
        if tag == "G0":
            # [cin][1024][4][4] seen as [cin][16384]: panel row (t, co) = column co * 16 + t
            return ("conv_blocks.0.weight", lambda p, g, m, v, lr, b1, b2, step, zero: ops.adam_pack_rows(
                dt, p, g, m, v, self.l0.wp_fwd, self.cin, 16 * G_WIDTHS[0], self.l0.Kpad_fwd, 16, G_WIDTHS[0], lr, b1, b2, 1e-8, step, zero))
        if tag in ("G1", "G2", "G3"):
            r = self.mid[int(tag[1]) - 1]
            if ops.adam_pack_conv_ok(r.c, dt, True, True):
                return (f"conv_blocks.{(1, 4, 7)[int(tag[1]) - 1]}.weight", lambda p, g, m, v, lr, b1, b2, step, zero: ops.adam_pack_conv(
                    r.c, dt, p, g, m, v, lr, b1, b2, 1e-8, step, zero, r.wp_fwd, r.wp_bwd))
        return None

    def repack_bucket(self, tag):
        """re-pack the panels of one gradient bucket's layers (the optimizer lane updates bucket by bucket)"""
        g, dt = self.gen, self.dtype
        if tag == "G0":
            w0 = self._p(0, "weight")                                  # [cin][1024][4][4]
            ops.pack_strided(dt, w0, self.l0.wp_fwd, 16 * G_WIDTHS[0], self.cin, self.l0.Kpad_fwd, G_WIDTHS[0], 1, 16, 16 * G_WIDTHS[0])
        elif tag in ("G1", "G2", "G3"):
            i = int(tag[1]) - 1
            self.mid[i].pack(self._p((1, 4, 7)[i], "weight"))
        else:
            self.l4.pack(self._p(10, "weight"))
            ops.pack_strided(dt, self._p(10, "weight"), self.l4p.wp_fwd, G_WIDTHS[3], self.kp, self.l4p.Kpad_fwd, 1, self.kp, 0, 1)
            # wp[t*C + c][ci] = W[ci][c][t]   (master [128][C][4][4])
            ops.pack_strided(dt, self._p(10, "weight"), self.l4g.wp_fwd, self.kp, G_WIDTHS[3], self.l4g.Kpad_fwd, g.channels, 1, 16, self.kp)

    def forward(self, noise, labels, code, training=True, sync=None, ws=None):
        """``training=False``: BatchNorm with the running statistics, nothing updated (module.eval()).  ``sync`` (a dp.SyncBN):
        batch statistics over all ranks (synchronised BatchNorm).  ``ws``: scratch to use instead of the engine's (a forward that runs
        on its own stream beside other main-stream work; the caller also activates that workspace's split-K scratch)."""
        dt, B, W = self.dtype, self.B, G_WIDTHS
        small = (ws or self.ws).small
        ops.concat_cast(dt, noise, labels, code, self.inp, B, self.cpad)
        ops.conv_fwd(self.l0.c, dt, self.inp, self.l0.wp_fwd, self.h0, ops.epilogue(bias=self._p(0, "bias"), bias_mod=W[0], nt_variant=G0_VARIANT))
        x = self.h0
        for i, idx in enumerate((1, 4, 7)):
            r = self.mid[i]
            bn = self.gen.conv_blocks[idx + 1]
            M = self.z[i].numel() // W[i + 1]
            nrb, stat = self._stat_buf(("fwd", i), r.c, True, ops.epilogue(bias=self._p(idx, "bias")), W[i + 1]) if (training and sync is None) else (0, None)
            if nrb:
                # BatchNorm batch statistics from the transposed convolution's epilogue: z is read once (apply) instead of twice
                ops.conv_bwd_data(r.c, dt, x, r.wp_bwd, self.z[i], ops.epilogue(bias=self._p(idx, "bias"), stat_mode=ops.STAT_MOMENTS, stat_out=stat))
                ops.bn_fwd_train_fused(dt, self.z[i], self.a[i], M, W[i + 1], stat, nrb, M // nrb, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean,
                                       bn.running_var, bn.num_batches_tracked, self.mean[i], self.invstd[i], small, ACT_RELU)
                x = self.a[i]
                continue
            ops.conv_bwd_data(r.c, dt, x, r.wp_bwd, self.z[i], ops.epilogue(bias=self._p(idx, "bias")))
            if training and sync is not None:
                ops.bn_stats_local(dt, self.z[i], M, W[i + 1], small, self.bn_stats[i])
                allst = sync.gather_stats(self.bn_stats[i])
                ops.bn_fwd_from_stats(dt, self.z[i], self.a[i], M, W[i + 1], allst, sync.world, M * sync.world, bn.weight, bn.bias, bn.eps, bn.momentum,
                                      bn.running_mean, bn.running_var, bn.num_batches_tracked, self.mean[i], self.invstd[i], small, ACT_RELU)
            elif training:
                ops.bn_fwd_train(dt, self.z[i], self.a[i], M, W[i + 1], bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean,
                                 bn.running_var, bn.num_batches_tracked, self.mean[i], self.invstd[i], small, ACT_RELU)
            else:
                ops.bn_fwd_eval(dt, self.z[i], self.a[i], M, W[i + 1], bn.weight, bn.bias, bn.eps, bn.running_mean, bn.running_var, small, ACT_RELU)
            x = self.a[i]
        if IMG_GEMM and IMG_DIRECT and ops.convt_img_mfma_ok(dt, self.gen.channels, self.l4g.H, self.l4g.W, G_WIDTHS[3], 4, 2, 1):
            # the last transposed convolution + Tanh in one launch: the GEMM's 48 columns per pixel stay in LDS
            ops.convt_img_mfma(dt, x, self.l4g.wp_fwd, self._p(10, "bias"), self.img, B, self.gen.channels, self.l4g.H, self.l4g.W, ACT_TANH, 0.0)
        elif IMG_GEMM:
            ops.conv_fwd(self.l4g.c, dt, x, self.l4g.wp_fwd, self.cols4, None)
            ops.col2im_img(dt, self.cols4, B, self.gen.channels, self.l4g.H, self.l4g.W, 4, 2, 1, self._p(10, "bias"), ACT_TANH, 0.0, self.img)
        else:
            ops.conv_bwd_data(self.l4.c, dt, x, self.l4.wp_bwd, self.img,
                              ops.epilogue(bias=self._p(10, "bias"), act=ACT_TANH, out_mode=OUT_NCHW_F32))
        return self.img

    def backward(self, dimg, grad, side=None, sync=None):
        """Accumulates d(loss)/d(params) into the flat gradient tensor ``grad`` (arena layout).  ``sync``: as in forward.  With ``side`` (an
        engine.SideStream) the weight/bias-gradient work of every layer is enqueued there, behind a fork taken right after
        the layer's output gradient exists; the caller joins before it reads ``grad``."""
        dt, B, W, ws, gen = self.dtype, self.B, G_WIDTHS, self.ws, self.gen
        gof = lambda name: gen.arena.grad_of(name, grad)
        C, S = gen.channels, self.img.shape[-1]
        def wgrad_side(fn, lane, tag=None):             # issued by the flush() that follows the next main-stream kernel (engine.SideStream)
            if side is None:
                fn(ws)
            else:
                side.defer(lane, fn, tag)
        flush = side.flush if side is not None else (lambda: None)

        direct = IMG_DIRECT and ops.conv_img_mfma_ok(dt, C, S, S, W[3], 4, 2, 1)
        if not direct:
            # tanh backward fused with the bias gradient of the last ConvTranspose2d
            ops.act_grad_mul_bias_nchw(dimg, self.img, self.dimg_z, B, C, S * S, ACT_TANH, 0.0, ws.small, gof("conv_blocks.10.bias"))
            # L4 = ConvTranspose2d(128 -> C): weight / input gradients as 1x1-conv GEMMs over im2col patches of d(img)
            ops.im2col_img(dt, self.dimg_z, self.patches, B, C, S, S, 4, 2, 1, self.kp)

        def l4_wgrad(wsw):
            if direct:                                  # the patch rows only feed the weight gradient: built here, beside the main chain
                ops.act_grad_mul_bias_nchw(dimg, self.img, self.dimg_z, B, C, S * S, ACT_TANH, 0.0, wsw.small, gof("conv_blocks.10.bias"))
                ops.im2col_img(dt, self.dimg_z, self.patches, B, C, S, S, 4, 2, 1, self.kp)
            ns = ops.conv_wgrad(self.l4p.c, dt, self.patches, self.a[2], wsw.slab, wsw.wgs_target)
            ops.wgrad_reduce(wsw.slab, ns, W[3], W[3], self.kp, 1, gof("conv_blocks.10.weight"))
        wgrad_side(l4_wgrad, 0, "G4")
        fused = (0, None)                               # (row blocks, sums) if da[i] came with its BatchNorm backward sums (then it holds dy)
        if direct:
            # d(a) = Conv2d(C -> 128, 4, 2, 1) of d(img) * tanh'(img): straight from the two fp32 images, no patch rows -- and, in the same
            # launch, dy = da * relu'(bn(z)) with the two sums of the last BatchNorm layer's backward
            ep4 = None
            if FUSE_STATS and sync is None:
                if "bwd2" not in self._stat:
                    nrb = ops.conv_img_mfma_stat_blocks(B, S, S)
                    self._stat["bwd2"] = (nrb, torch.empty(2 * W[3] * nrb, device=self.inp.device, dtype=torch.float32))
                fused = self._stat["bwd2"]
                bn3 = gen.conv_blocks[8]
                ep4 = ops.epilogue(stat_mode=ops.STAT_BN_BWD, stat_out=fused[1], stat_aux=self.z[2], stat_p=(self.mean[2], self.invstd[2], bn3.weight, bn3.bias),
                                   stat_act=ACT_RELU)
            ops.conv_img_mfma(dt, [dimg], self.l4p.wp_fwd, self.da[2], B, C, S, S, ep4, gates=[self.img], gate_act=ACT_TANH)
        else:
            ops.conv_fwd(self.l4p.c, dt, self.patches, self.l4p.wp_fwd, self.da[2], None)
        flush()
        # L3..L1
        for i, idx in ((2, 7), (1, 4), (0, 1)):
            r = self.mid[i]
            bn = gen.conv_blocks[idx + 1]
            M = self.z[i].numel() // W[i + 1]
            if fused[0]:
                ops.bn_bwd_fused(dt, self.z[i], self.da[i], self.dz[i], M, W[i + 1], fused[1], fused[0], bn.weight, bn.bias, self.mean[i], self.invstd[i],
                                 gof(f"conv_blocks.{idx + 1}.weight"), gof(f"conv_blocks.{idx + 1}.bias"), ws.sums, ws.small)
            elif sync is not None:
                ops.bn_bwd_sums_local(dt, self.z[i], self.da[i], M, W[i + 1], bn.weight, bn.bias, self.mean[i], self.invstd[i], ACT_RELU, 0.0,
                                      gof(f"conv_blocks.{idx + 1}.weight"), gof(f"conv_blocks.{idx + 1}.bias"), self.bn_sums[i], ws.small)
                sync.reduce_sums(self.bn_sums[i])
                ops.bn_bwd_from_sums(dt, self.z[i], self.da[i], self.dz[i], M, W[i + 1], self.bn_sums[i], M * sync.world, bn.weight, bn.bias,
                                     self.mean[i], self.invstd[i], ACT_RELU, 0.0, ws.small)
            else:
                ops.bn_bwd(dt, self.z[i], self.da[i], self.dz[i], M, W[i + 1], bn.weight, bn.bias, self.mean[i], self.invstd[i], ACT_RELU, 0.0,
                           gof(f"conv_blocks.{idx + 1}.weight"), gof(f"conv_blocks.{idx + 1}.bias"), ws.sums, ws.small)
            x_in = self.a[i - 1] if i > 0 else self.h0

            def mid_wgrad(wsw, i=i, idx=idx, r=r, M=M, x_in=x_in):
                ns = ops.conv_wgrad(r.c, dt, self.dz[i], x_in, wsw.slab, wsw.wgs_target)
                ops.wgrad_reduce(wsw.slab, ns, r.Cout, r.Cout, r.Cin, 16, gof(f"conv_blocks.{idx}.weight"))
                ops.bias_grad(dt, self.dz[i], M, W[i + 1], wsw.small, gof(f"conv_blocks.{idx}.bias"))
            wgrad_side(mid_wgrad, i + 1, f"G{i + 1}")
            fused = (0, None)
            if i > 0 and sync is None:
                # the launch that produces d(a[i-1]) also forms dy = da * relu'(bn(z)) and the two sums of that layer's BatchNorm backward
                fused = self._stat_buf(("bwd", i - 1), r.c, False, ops.epilogue(), W[i])
            if fused[0]:
                bnl = gen.conv_blocks[(1, 4, 7)[i - 1] + 1]
                ops.conv_fwd(r.c, dt, self.dz[i], r.wp_fwd, self.da[i - 1],
                             ops.epilogue(stat_mode=ops.STAT_BN_BWD, stat_out=fused[1], stat_aux=self.z[i - 1],
                                          stat_p=(self.mean[i - 1], self.invstd[i - 1], bnl.weight, bnl.bias), stat_act=ACT_RELU))
            else:
                ops.conv_fwd(r.c, dt, self.dz[i], r.wp_fwd, self.da[i - 1] if i > 0 else self.dh0, None)
            flush()

        # L0
        def l0_wgrad(wsw):
            ns = ops.conv_wgrad(self.l0w.c, dt, self.dh0, self.inp, wsw.slab, wsw.wgs_target)
            ops.wgrad_reduce(wsw.slab, ns, self.cpad, self.cin, W[0], 16, gof("conv_blocks.0.weight"))
            ops.bias_grad(dt, self.dh0, B * 16, W[0], wsw.small, gof("conv_blocks.0.bias"))
        wgrad_side(l0_wgrad, 0, "G0")                   # stays pending: the caller's next main-stream kernel goes first


class _HipModule(nn.Module):
    """Shared plumbing: flat arena, per-batch engines, invalidation when the module is moved."""

    compute_dtype = EG_F32

    def _init_engine_state(self, dtype):
        self.compute_dtype = parse_dtype(dtype)
        self._arena = None
        self._engines = {}

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._arena, self._engines = None, {}
        return out

    @property
    def arena(self) -> Arena:
        if self._arena is None:
            p = next(self.parameters())
            _require_cuda(p)
            self._arena = Arena(self)
            self._engines = {}
        return self._arena

    def set_compute_dtype(self, dtype):
        self.compute_dtype = parse_dtype(dtype)

    def repack(self):
        """Refresh the packed (kernel-layout, compute-dtype) weight panels after the fp32 masters changed."""
        for e in self._engines.values():
            e.repack()

    def fresh_engine(self, B):
        """The engine for batch size B with panels re-packed from the masters: the drop-in forward path.  The kernels read packed
        copies of the convolution weights, and the masters can change behind the module's back in ways nothing reports --
        ``optimizer.step()`` bumps a version counter, ``module.apply(weights_init_normal)`` writes through ``.data`` and does not -- so
        every drop-in forward re-packs first (2 small launches per layer, asynchronous).  The fused trainers keep masters and panels in
        step themselves and never come through here."""
        eng = self.engine(B)
        eng.repack()
        return eng

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.repack()
        return out


class Generator(_HipModule):
    """Drop-in for celebA/EAD-GAN_celebA.py:67-102.  ``forward(noise, labels, code) -> img [B,3,64,64]``."""

    def __init__(self, latent_dim=None, code_dim=None, n_classes=None, img_size=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.latent_dim, self.code_dim, self.n_classes = g(latent_dim, "latent_dim"), g(code_dim, "code_dim"), g(n_classes, "n_classes")
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        if self.img_size != 64 or self.channels > 4:
            raise ValueError("the CelebA generator is defined for 64x64 images (reference: img_size // 2**4 == 4)")
        self.input_dim = self.latent_dim + self.code_dim + self.n_classes
        self.init_size = self.img_size // 2 ** 4
        W = G_WIDTHS
        self.conv_blocks = nn.Sequential(
            nn.ConvTranspose2d(self.input_dim, W[0], 4, 1, 0),
            nn.ConvTranspose2d(W[0], W[1], 4, stride=2, padding=1), nn.BatchNorm2d(W[1]), nn.ReLU(),
            nn.ConvTranspose2d(W[1], W[2], 4, stride=2, padding=1), nn.BatchNorm2d(W[2]), nn.ReLU(),
            nn.ConvTranspose2d(W[2], W[3], 4, stride=2, padding=1), nn.BatchNorm2d(W[3]), nn.ReLU(),
            nn.ConvTranspose2d(W[3], self.channels, 4, stride=2, padding=1), nn.Tanh())
        self._init_engine_state(dtype)

    def engine(self, B) -> _GenEngine:
        self.arena
        key = (B, self.compute_dtype)
        if key not in self._engines:
            self._engines[key] = _GenEngine(self, B, self.compute_dtype)
        return self._engines[key]

    def forward(self, noise, labels, code):
        _require_cuda(noise)
        eng = self.fresh_engine(noise.shape[0])
        if not self.training:
            # inference (generate_image.py:146-154, gen_imgs.py:106-120): running-stat BatchNorm, no autograd graph
            with torch.no_grad():
                return eng.forward(noise.float().contiguous(), labels.float().contiguous(), code.float().contiguous(), training=False).clone()
        params = [p for p in self.parameters()]
        return _GenFn.apply(eng, noise.float().contiguous(), labels.float().contiguous(), code.float().contiguous(), *params)


class _GenFn(torch.autograd.Function):
    """autograd bridge so the drop-in modules compose with torch losses/optimizers (eager mode)."""

    @staticmethod
    def forward(ctx, eng, noise, labels, code, *params):
        ctx.eng = eng
        return eng.forward(noise, labels, code).clone()

    @staticmethod
    def backward(ctx, dimg):
        eng = ctx.eng
        scratch = torch.zeros_like(eng.gen.arena.grad)
        eng.backward(dimg.contiguous(), scratch)
        grads = [scratch[off:off + k].view(p.shape) for p, (off, k) in zip(eng.gen.parameters(), eng.gen.arena.slices.values())]
        return (None, None, None, None, *grads)


# ================================================================================================
# Discriminator / Q network
# ================================================================================================
class _DiscEngine:
    """Static-shape forward/backward of the discriminator for up to NT "tapes" (independent forwards, each with its
    own spectral-norm power iteration).  D has no BatchNorm, so the tapes of one sub-step are batched along M into
    single launches: epilogues pick 1/sigma per tape, ONE weight-gradient GEMM runs over all tapes on gradients that
    already carry 1/sigma_tape (dzs = dz/sigma), and the rank-1 spectral-norm terms  -<G_t,W>/sigma_t^2 u_t v_t^T
    are added by a single-pass reduction whose coefficients come from activations
    (<G_t,W>/sigma_t^2 = sum dzs_t * (z_t - bias)), never from a second sweep over the weights."""

    NT = 3     # the info step runs three forwards before one backward (celebA/EAD-GAN_celebA.py:380-388)
    # gradient buckets in the order the backward pass completes them (see _GenEngine.BUCKETS)
    BUCKETS = (("D4", ("main.8.weight", "main.8.bias")), ("D3", ("main.6.bias", "main.6.weight_orig")), ("D2", ("main.4.bias", "main.4.weight_orig")),
               ("D1", ("main.2.bias", "main.2.weight_orig")), ("D0", ("main.0.bias", "main.0.weight_orig")))

    def __init__(self, disc: "Discriminator", B: int, dtype: int):
        self.disc, self.B, self.dtype = disc, B, dtype
        dev = disc.arena.flat.device
        self.ws = ws = Workspace.get(dev)
        W, C, S, NT = D_WIDTHS, disc.channels, disc.img_size, self.NT
        self.C, self.S = C, S
        self.nout = disc.n_out
        tdt = ops.torch_dtype(dtype)
        self.kp = C * 16
        self.cin = [self.kp, W[0], W[1], W[2]]                 # gathered channels of layer i (layer 0: im2col patches)
        self.hw = [S >> (i + 1) for i in range(4)]             # output extent of layer i
        # packed panels live in per-layer records built for the largest batch; geometry structs per tape count
        self.l1 = ConvRec(dtype, B, S, S, C, W[0], 4, 2, 1, device=dev, want_fwd=False, want_wgrad=False, ws=ws)       # d(img), tape 0 only
        self.l1p = ConvRec(dtype, NT * B, S // 2, S // 2, self.kp, W[0], 1, 1, 0, device=dev, want_bwd=False, ws=ws)  # image side as 1x1 conv over patches
        # d(img) as one GEMM over the layer-0 lattice with N = 16 taps x C columns + eg_col2im_img (see _GenEngine.l4g)
        self.l1g = ConvRec(dtype, B, S // 2, S // 2, W[0], self.kp, 1, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.cols1 = torch.empty(B * (S // 2) ** 2, self.kp, device=dev, dtype=tdt)
        self.mid = [ConvRec(dtype, NT * B, S >> (i + 1), S >> (i + 1), W[i], W[i + 1], 4, 2, 1, device=dev, ws=ws) for i in range(3)]
        self.head = ConvRec(dtype, NT * B, 4, 4, W[3], self.nout, 4, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.headw = ConvRec(dtype, NT * B, 1, 1, 16 * W[3], 32, 1, 1, 0, device=dev, want_fwd=False, want_bwd=False, ws=ws)
        self.geo = {}
        for T in range(1, NT + 1):
            g = {"l1p": ops.make_conv(T * B, S // 2, S // 2, self.kp, W[0], 1, 1, 0),
                 "mid": [ops.make_conv(T * B, S >> (i + 1), S >> (i + 1), W[i], W[i + 1], 4, 2, 1) for i in range(3)],
                 "headw": ops.make_conv(T * B, 1, 1, 16 * W[3], 32, 1, 1, 0)}
            self.geo[T] = g
        for i in range(4):
            ws.need_small(ops.sn_ws_floats(W[i], self.cin[i] * (16 if i else 1)))
            ws.need_small(ops.bias_grad_sn_ws_floats(NT * B * self.hw[i] ** 2, W[i], B * self.hw[i] ** 2))
        # tapes: contiguous along the batch axis
        self.patches = torch.empty(NT * B * (S // 2) ** 2, self.kp, device=dev, dtype=tdt)
        self.a = [torch.empty(NT * B, self.hw[i], self.hw[i], W[i], device=dev, dtype=tdt) for i in range(4)]
        self.out = torch.empty(NT * B, self.nout, device=dev, dtype=torch.float32)
        self.dz = [torch.empty_like(t) for t in self.a]
        self.dout_t = torch.empty(NT * B, 32, device=dev, dtype=tdt)
        self._head_scratch = None                       # head_losses: per-sample loss terms + arrival counter
        self.dimg = torch.empty(B, C, S, S, device=dev, dtype=torch.float32)
        kd = [self.kp] + [W[i] * 16 for i in range(3)]
        self.sigma = [torch.ones(NT, device=dev, dtype=torch.float32) for _ in range(4)]
        self.u = [torch.zeros(NT, W[i], device=dev, dtype=torch.float32) for i in range(4)]
        self.v = [torch.zeros(NT, kd[i], device=dev, dtype=torch.float32) for i in range(4)]
        self.coef = [torch.zeros(4, device=dev, dtype=torch.float32) for _ in range(4)]
        self.imgs = [None] * NT
        self.patch_ok = [False] * NT                    # tape has its patch rows (the weight gradient of layer 0 reads them)
        self.img_direct = IMG_DIRECT and ops.conv_img_mfma_ok(dtype, C, S, S, W[0], 4, 2, 1)
        # ... and the first layer's weight gradient from the images too (ops.wgrad_img): no patch rows in HBM at all for that layer
        self.wgrad_direct = self.img_direct and WGRAD_IMG and self.kp == 16 * C and ops.wgrad_img_ok(dtype, C, S, S, W[0], 4, 2, 1)
        if self.wgrad_direct:
            self.ws.need_slab(ops.wgrad_img_splits(NT * B, W[0]) * W[0] * self.kp * 4)
        self._stat = {}                                 # (layer, T) -> (row blocks, buffer) of the fused bias-gradient / coefficient sums
        self._sn_arrays = None
        self._sn(0)
        self.repack()

    def _sn(self, t):
        """eg_sn_layer array of tape t (pointers are stable: module buffers and tape snapshots never move)."""
        if self._sn_arrays is None:
            self._sn_arrays = []
            for tt in range(self.NT):
                ent = [(self._m(i).weight_orig, self._m(i).weight_u, self._m(i).weight_v, self.sigma[i][tt:tt + 1], self.u[i][tt], self.v[i][tt])
                       for i in range(4)]
                self._sn_arrays.append(ops.sn_layers(ent))
            # own scratch: the power iterations may run on a side stream while the main stream uses the shared workspace
            self.sn_scratch = torch.empty(ops.sn_multi_ws_floats(self._sn_arrays[0]), device=self.ws.device, dtype=torch.float32)
            self.sn_counters = torch.zeros(16, device=self.ws.device, dtype=torch.int32)      # arrival counters of the two-launch iteration
        return self._sn_arrays[t]

    def _m(self, i):
        return self.disc.main[2 * i]

    def repack(self):
        for tag, _ in self.BUCKETS:
            self.repack_bucket(tag)

    def fused_weight(self, tag):
        """see _GenEngine.fused_weight"""
        if tag in ("D1", "D2", "D3"):
            i = int(tag[1])
            r = self.mid[i - 1]
            if ops.adam_pack_conv_ok(r.c, self.dtype, True, True):
                return (f"main.{2 * i}.weight_orig", lambda p, g, m, v, lr, b1, b2, step, zero: ops.adam_pack_conv(
                    r.c, self.dtype, p, g, m, v, lr, b1, b2, 1e-8, step, zero, r.wp_fwd, r.wp_bwd))
        return None

    def repack_bucket(self, tag):
        """re-pack the panels of one gradient bucket's layer (see _GenEngine.repack_bucket)"""
        if tag == "D0":
            self.l1.pack(self._m(0).weight_orig)
            # wp[t*C + c][co] = W[co][c][t]   (Conv2d master [128][C][4][4]; the transposed convolution reads it as [in = 128][out = C])
            ops.pack_strided(self.dtype, self._m(0).weight_orig, self.l1g.wp_fwd, self.kp, D_WIDTHS[0], self.l1g.Kpad_fwd, self.C, 1, 16, self.kp)
            ops.pack_strided(self.dtype, self._m(0).weight_orig, self.l1p.wp_fwd, D_WIDTHS[0], self.kp, self.l1p.Kpad_fwd, 1, self.kp, 0, 1)
        elif tag in ("D1", "D2", "D3"):
            i = int(tag[1]) - 1
            self.mid[i].pack(self._m(i + 1).weight_orig)
        else:
            self.head.pack(self._m(4).weight)

    def rows(self, i):
        """lattice rows of one tape at the output of layer i"""
        return self.B * self.hw[i] ** 2

    def _stat_buf(self, i, T, c, ep):
        """fused sums of layer i's bias gradient / spectral-norm coefficient in the backward-data launch that produces dzs_i (T tapes):
        (row blocks, buffer), (0, None) where that launch cannot take them (see _GenEngine._stat_buf)"""
        key = (i, T)
        if key not in self._stat:
            ok = FUSE_STATS and self.rows(i + 1) % 256 == 0
            nrb = ops.conv_stat_blocks(c, self.dtype, True, ep) if ok else 0
            N = D_WIDTHS[i]
            self._stat[key] = (nrb, torch.empty(N * nrb + nrb * (N // 128), device=self.out.device, dtype=torch.float32) if nrb else None)
        return self._stat[key]

    def _sn_tape(self, t, training=True):
        ops.sn_power_iter_multi(self._sn(t), self.sn_scratch, training, SN_EPS, self.sn_counters)
        if not training:
            for i in range(4):
                self.u[i][t].copy_(self._m(i).weight_u)
                self.v[i][t].copy_(self._m(i).weight_v)

    def _im2col_tape(self, t, img):
        npix = self.B * (self.S // 2) ** 2
        ops.im2col_img(self.dtype, img, self.patches[t * npix:(t + 1) * npix], self.B, self.C, self.S, self.S, 4, 2, 1, self.kp)
        self.patch_ok[t] = True

    def prepare(self, t0, imgs, training=True):
        """Image-independent head start of ``forward(imgs, t0, prepared=...)``: the tapes' power iterations (in list order, like
        consecutive D(...) calls) and the patch rows of the images that already exist (``None`` entries are left to forward).
        The trainer runs this on a side stream while the previous sub-step is still busy."""
        for k, img in enumerate(imgs):
            self._sn_tape(t0 + k, training)
            if img is not None:
                self._im2col_tape(t0 + k, img)

    def forward(self, imgs, t0=0, training=True, prepared=None, head=True):
        """Runs len(imgs) forwards as tapes t0.. (power iterations in list order, like consecutive D(...) calls).
        ``head=False``: stop in front of the head (``head_losses`` runs it together with the losses).
        ``prepared``: per-tape flags of a preceding ``prepare`` call (power iterations done for all tapes, patch rows done where
        the flag is set).  Returns the head outputs [len(imgs)*B, 19] (a view of the tape buffer)."""
        dt, B, W, ws = self.dtype, self.B, D_WIDTHS, self.ws
        T = len(imgs)
        assert 1 <= T and t0 + T <= self.NT
        for k, img in enumerate(imgs):
            t = t0 + k
            self.imgs[t] = img
            if prepared is None:
                self._sn_tape(t, training)
            if prepared is None or not prepared[k]:
                if self.img_direct:
                    self.patch_ok[t] = False            # built by the weight-gradient chain if a backward pass wants them (backward)
                else:
                    self._im2col_tape(t, img)
        g = self.geo[T]
        sl = lambda buf, i: buf[t0 * (buf.shape[0] // self.NT):]
        ep = lambda i: ops.epilogue(bias=self._m(i).bias, sigma=self.sigma[i][t0:], sigma_rows=self.rows(i), act=ACT_LRELU, slope=LRELU_SLOPE)
        if self.img_direct:
            # first layer straight from the fp32 images of the T tapes (one launch, 1/sigma per tape): no patch rows on the main chain
            ops.conv_img_mfma(dt, [im.contiguous() for im in imgs], self.l1p.wp_fwd, sl(self.a[0], 0), B, self.C, self.S, self.S, ep(0))
        else:
            ops.conv_fwd(g["l1p"], dt, sl(self.patches, 0), self.l1p.wp_fwd, sl(self.a[0], 0), ep(0))
        for i in range(3):
            ops.conv_fwd(g["mid"][i], dt, sl(self.a[i], i), self.mid[i].wp_fwd, sl(self.a[i + 1], i + 1), ep(i + 1))
        K = 16 * W[3]
        out = self.out[t0 * B:(t0 + T) * B]
        if head:
            ops.dense_small_fwd(dt, sl(self.a[3], 3), self.head.wp_fwd, self._m(4).bias, out, T * B, K, self.head.Kpad_fwd, self.nout, ws.small)
        return out

    def head_fused_ok(self, T):
        return FUSE_HEAD and ops.head_fused_ok(self.dtype, T, 16 * D_WIDTHS[3], self.nout)

    def head_losses(self, t0, T, dout, loss, targets=None, scales=None, info=None):
        """The head of tapes t0..t0+T-1 (after ``forward(..., head=False)``), their losses (added to ``loss[0]``), ``dout`` = d(loss)/d(head
        output) and the head's input gradient dzs_3 -- ``backward(..., head_done=True)`` continues from there -- in TWO launches (the K-sliced
        dense head; slice combine + losses + dense backward) with the bits of the five they replace."""
        B, K = self.B, 16 * D_WIDTHS[3]
        sl = lambda buf: buf[t0 * (buf.shape[0] // self.NT):]
        if self._head_scratch is None:
            self._head_scratch = (torch.empty(3 * B, device=self.out.device, dtype=torch.float32),
                                  torch.zeros(1, device=self.out.device, dtype=torch.int32))
        terms, counter = self._head_scratch
        out = self.out[t0 * B:(t0 + T) * B]
        ns = ops.dense_small_fwd_slices(self.dtype, sl(self.a[3]), self.head.wp_fwd, T * B, K, self.head.Kpad_fwd, self.nout, self.ws.small)
        ops.head_fused(self.dtype, sl(self.a[3]), self.head.wp_fwd, self._m(4).bias, self.ws.small, ns, out, dout, sl(self.dz[3]), self.sigma[3][t0:],
                       B, T, K, self.head.Kpad_fwd, self.nout, loss, terms, counter, ACT_LRELU, LRELU_SLOPE, targets=targets, scales=scales, info=info)
        return out

    def backward(self, t0, T, dout, grad, need_wgrad=True, need_dimg=False, side=None, head_done=False):
        """``dout``: d(loss)/d(head output) of tapes t0..t0+T-1, [T*B,19] fp32.  Accumulates into flat ``grad`` (arena
        layout); returns d(loss)/d(img) of tape t0 when ``need_dimg``.  With ``side`` (engine.SideStream) the weight- and
        bias-gradient work is enqueued there (see _GenEngine.backward); the caller joins before it reads ``grad``."""
        dt, B, W, ws, disc = self.dtype, self.B, D_WIDTHS, self.ws, self.disc
        gof = lambda name: disc.arena.grad_of(name, grad)
        g = self.geo[T]
        sl = lambda buf: buf[t0 * (buf.shape[0] // self.NT):]
        K = 16 * W[3]
        def wgrad_side(fn, lane, tag=None):             # issued by the flush() that follows the next main-stream kernel (engine.SideStream)
            if side is None:
                fn(ws)
            else:
                side.defer(lane, fn, tag)
        flush = side.flush if side is not None else (lambda: None)

        if need_wgrad:
            def head_wgrad(wsw):
                ops.cast_pad(dt, dout, self.dout_t, T * B, self.nout, 32)
                ns = ops.conv_wgrad(g["headw"], dt, sl(self.a[3]), self.dout_t, wsw.slab, wsw.wgs_target)
                ops.wgrad_reduce(wsw.slab, ns, 32, self.nout, W[3], 16, gof("main.8.weight"))
                ops.dense_small_bgrad(dout, gof("main.8.bias"), T * B, self.nout)
            wgrad_side(head_wgrad, 0, "D4")
        # dzs_3 = (W5^T dout) * lrelu'(a3) / sigma_3[tape]
        if not head_done:
            ops.dense_small_bwd(dt, dout, self.head.wp_fwd, sl(self.a[3]), sl(self.dz[3]), T * B, K, self.head.Kpad_fwd, self.nout, ACT_LRELU, LRELU_SLOPE,
                                self.sigma[3][t0:], B)
        flush()
        fused = (0, None)                               # (row blocks, sums) if dzs_i came with its column sums
        for i in (3, 2, 1, 0):
            m = self._m(i)
            geo = g["mid"][i - 1] if i > 0 else g["l1p"]
            x_in = sl(self.a[i - 1]) if i > 0 else sl(self.patches)
            if need_wgrad:
                def layer_wgrad(wsw, i=i, m=m, geo=geo, x_in=x_in, fused=fused):
                    direct = i == 0 and self.wgrad_direct       # straight from the tapes' images: patch rows expanded in LDS (eg_wgrad_img)
                    if i == 0 and not direct:
                        for t in range(t0, t0 + T):     # patch rows of tapes whose forward ran straight from the image
                            if not self.patch_ok[t]:
                                self._im2col_tape(t, self.imgs[t])

                    def gemm():
                        if direct:
                            return ops.wgrad_img(dt, [self.imgs[t] for t in range(t0, t0 + T)], sl(self.dz[0]), wsw.slab, B, self.C, self.S, self.S, W[0])
                        return ops.conv_wgrad(geo, dt, x_in, sl(self.dz[i]), wsw.slab, wsw.wgs_target)
                    # the weight-gradient GEMM first: the column sums below (only the slab reduce needs their coefficient) do not fit on a CU
                    # beside a resident GEMM workgroup and used to hold the chain's GEMM back by ~50 us
                    if WGRAD_FIRST:
                        ns = gemm()
                    if fused[0]:
                        tiles_m = fused[0] // 4         # row blocks (of 256 or 128 lattice rows: the kernel's tile height) per sub-pixel phase
                        ops.bias_grad_sn_fused(fused[1], fused[0], W[i], tiles_m, tiles_m // T, T, self.sigma[i][t0:], gof(f"main.{2 * i}.bias"), self.coef[i])
                    else:
                        ops.bias_grad_sn(dt, sl(self.dz[i]), sl(self.a[i]), m.bias, T * self.rows(i), W[i], self.rows(i), self.sigma[i][t0:], LRELU_SLOPE,
                                         wsw.small, gof(f"main.{2 * i}.bias"), self.coef[i])
                    if not WGRAD_FIRST:
                        ns = gemm()
                    taps = 16 if i > 0 else 1
                    ops.wgrad_reduce_rank1(wsw.slab, ns, W[i], W[i], self.cin[i], taps, gof(f"main.{2 * i}.weight_orig"), T, self.coef[i],
                                           self.u[i][t0:], self.v[i][t0:])
                wgrad_side(layer_wgrad, i + 1, f"D{i}")
            if i > 0:
                # dzs_{i-1} = conv^T(dzs_i, W_i) * lrelu'(a_{i-1}) / sigma_{i-1}[tape]
                kw = dict(sigma=self.sigma[i - 1][t0:], sigma_rows=self.rows(i), mask=sl(self.a[i - 1]), mask_act=ACT_LRELU, mask_slope=LRELU_SLOPE)
                # ... and, from the same tile, layer i-1's bias-gradient column sums and spectral-norm coefficient
                fused = self._stat_buf(i - 1, T, geo, ops.epilogue(**kw)) if need_wgrad else (0, None)
                if fused[0]:
                    kw.update(stat_mode=ops.STAT_SN_BIAS, stat_out=fused[1], stat_p=(self._m(i - 1).bias,), stat_slope=LRELU_SLOPE)
                ops.conv_bwd_data(geo, dt, sl(self.dz[i]), self.mid[i - 1].wp_bwd, sl(self.dz[i - 1]), ops.epilogue(**kw))
                flush()
        if need_dimg:
            if IMG_GEMM and IMG_DIRECT and ops.convt_img_mfma_ok(dt, self.C, self.S // 2, self.S // 2, W[0], 4, 2, 1):
                ops.convt_img_mfma(dt, sl(self.dz[0]), self.l1g.wp_fwd, None, self.dimg, B, self.C, self.S // 2, self.S // 2, ACT_NONE, 0.0)
            elif IMG_GEMM:
                ops.conv_fwd(self.l1g.c, dt, sl(self.dz[0]), self.l1g.wp_fwd, self.cols1, None)
                ops.col2im_img(dt, self.cols1, B, self.C, self.S // 2, self.S // 2, 4, 2, 1, None, ACT_NONE, 0.0, self.dimg)
            else:
                ops.conv_bwd_data(self.l1.c, dt, sl(self.dz[0]), self.l1.wp_bwd, self.dimg, ops.epilogue(out_mode=OUT_NCHW_F32))
            flush()
            return self.dimg
        return None                                     # layer 0's chain stays pending: the caller's next main-stream kernel goes first


class Discriminator(_HipModule):
    """Drop-in for celebA/EAD-GAN_celebA.py:105-138.  ``forward(img) -> (cat, cont, validity)``; the one
    19-channel head plays discriminator and Q-network (cat = softmax(out[:,9:19]), cont = out[:,1:9],
    validity = sigmoid(out[:,0]))."""

    def __init__(self, code_dim=None, n_classes=None, img_size=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.code_dim, self.n_classes = g(code_dim, "code_dim"), g(n_classes, "n_classes")
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        if self.img_size != 64 or self.channels > 4:
            raise ValueError("the CelebA discriminator is defined for 64x64 images")
        self.n_out = 1 + self.n_classes + self.code_dim
        W = D_WIDTHS
        self.main = nn.Sequential(
            spectral_norm(nn.Conv2d(self.channels, W[0], 4, 2, 1)), nn.LeakyReLU(LRELU_SLOPE, inplace=True),
            spectral_norm(nn.Conv2d(W[0], W[1], 4, 2, 1)), nn.LeakyReLU(LRELU_SLOPE, inplace=True),
            spectral_norm(nn.Conv2d(W[1], W[2], 4, 2, 1)), nn.LeakyReLU(LRELU_SLOPE, inplace=True),
            spectral_norm(nn.Conv2d(W[2], W[3], 4, 2, 1)), nn.LeakyReLU(LRELU_SLOPE, inplace=True),
            nn.Conv2d(W[3], self.n_out, 4, 1, 0))
        self._init_engine_state(dtype)
        self._next_tape = 0

    def engine(self, B) -> _DiscEngine:
        self.arena
        key = (B, self.compute_dtype)
        if key not in self._engines:
            self._engines[key] = _DiscEngine(self, B, self.compute_dtype)
        return self._engines[key]

    def raw_forward(self, img):
        """head output [B,19] with autograd support (eager path)."""
        _require_cuda(img)
        eng = self.fresh_engine(img.shape[0])
        t = self._next_tape
        self._next_tape = (t + 1) % _DiscEngine.NT
        return _DiscFn.apply(eng, t, self.training, img.float().contiguous(), *list(self.parameters()))

    def forward(self, img):
        out = self.raw_forward(img)
        cd, nc = self.code_dim, self.n_classes
        validity = torch.sigmoid(out[:, 0])
        cat = torch.softmax(out[:, cd + 1: cd + 1 + nc], dim=1)
        cont = out[:, 1: cd + 1]
        return cat, cont, validity


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, t, training, img, *params):
        ctx.eng, ctx.t = eng, t
        ctx.need_w = any(p.requires_grad for p in params)
        ctx.need_img = img.requires_grad
        return eng.forward([img], t, training).clone()

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.eng
        scratch = torch.zeros_like(eng.disc.arena.grad)
        dimg = eng.backward(ctx.t, 1, dout.contiguous(), scratch, need_wgrad=ctx.need_w, need_dimg=ctx.need_img)
        grads = [scratch[off:off + k].view(p.shape) for p, (off, k) in zip(eng.disc.parameters(), eng.disc.arena.slices.values())]
        return (None, None, None, dimg.clone() if dimg is not None else None, *grads)


# ================================================================================================
# affine utilities (celebA/utils_rpqxy.py) and the STN warp
# ================================================================================================
def get_matrix(code_input_raw):
    """[B,>=5] latent codes -> [B,3,3] affine matrices R(theta) Z(p,q) T(x,y)  (utils_rpqxy.py:59-80)."""
    _require_cuda(code_input_raw)
    c = code_input_raw.float().contiguous()
    B = c.shape[0]
    theta = torch.empty(B, 2, 3, device=c.device, dtype=torch.float32)
    ops.theta_rpqxy(c, c.shape[1], B, theta)
    A = torch.zeros(B, 3, 3, device=c.device, dtype=torch.float32)
    A[:, :2] = theta
    A[:, 2, 2] = 1.0
    return A


class transformation_2D(nn.Module):
    """affine_grid + grid_sample(bilinear, border, align_corners=False) as one HIP kernel (:144-158)."""

    def stn(self, x, matrix_2D):
        _require_cuda(x)
        x = x.float().contiguous()
        out = torch.empty_like(x)
        B, C, H, W = x.shape
        ops.warp_affine(x, matrix_2D.float().contiguous(), out, B, C, H, W)
        return out

    def forward(self, img, matrix_2D):
        return self.stn(img, matrix_2D)


class _AffineRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real_code, trans_code):
        B, ld = real_code.shape
        pred = torch.empty(B, 5, device=real_code.device, dtype=torch.float32)
        zero_code = torch.zeros(B, 5, device=real_code.device, dtype=torch.float32)
        ops.loss_affine_rpqxy(real_code, trans_code, ld, 0, B, zero_code, 5, 1.0, None, None, None, pred)
        ctx.save_for_backward(real_code, trans_code)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        # d/dcode of sum_j dpred_j * pred_j  ==  gradient of the MSE kernel with target = pred - dpred*(5B/2)
        real_code, trans_code = ctx.saved_tensors
        B, ld = real_code.shape
        pred = torch.empty(B, 5, device=real_code.device, dtype=torch.float32)
        zero_code = torch.zeros(B, 5, device=real_code.device, dtype=torch.float32)
        ops.loss_affine_rpqxy(real_code, trans_code, ld, 0, B, zero_code, 5, 1.0, None, None, None, pred)
        tgt = pred - dpred.float() * (5.0 * B / 2.0)
        d_real = torch.empty_like(real_code)
        d_trans = torch.empty_like(trans_code)
        ops.loss_affine_rpqxy(real_code, trans_code, ld, 0, B, tgt.contiguous(), 5, 1.0, None, d_real, d_trans, None)
        return d_real, d_trans


def affine_regularzier(real_code, trans_code):
    """closed-form relative-transform recovery (utils_rpqxy.py:82-116); spelling follows the reference."""
    _require_cuda(real_code)
    return _AffineRegFn.apply(real_code.float().contiguous(), trans_code.float().contiguous())


# ================================================================================================
# fused train-loop entry
# ================================================================================================
def pil_bilinear_tables(in_size: int, out_size: int):
    """Coefficient tables of PIL's antialiased bilinear resample of an 8-bit image from ``in_size`` to ``out_size`` pixels (what
    transforms.Resize does to a PIL image, celebA/EAD-GAN_celebA.py:194): Pillow's precompute_coeffs + normalize_coeffs_8bpc
    (libImaging/Resample.c) in the same double arithmetic -> (bounds int32 [out,2], kk int32 [out,ksize], ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                          # bilinear filter support 1.0
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(xmax, dtype=np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0
        for x in range(xmax):                            # left-to-right double sum, as in C
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << 22)
            kk[xx, x] = int(v - 0.5) if w[x] < 0 else int(v + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_center_crop_u8(images_u8: torch.Tensor, size: int = 64) -> torch.Tensor:
    """transforms.Resize(size) + transforms.CenterCrop(size) (celebA/EAD-GAN_celebA.py:194-196) of a uint8 [N,C,H,W] device tensor, on the
    device: the smaller edge goes to ``size`` with PIL's antialiased bilinear filter (horizontal pass, 8-bit intermediate, vertical pass,
    bit-exact against Image.resize), the centre ``size`` x ``size`` window is kept.  Returns uint8 [N,C,size,size] -- what DeviceInputs
    takes.  Only the rows / columns the crop keeps are computed."""
    _require_cuda(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
        raise ValueError("images must be a uint8 [N, C, H, W] device tensor")
    N, C, H, W = images_u8.shape
    if W <= H:
        ow, oh = size, int(size * H / W)                 # torchvision: the smaller edge matches `size`
    else:
        oh, ow = size, int(size * W / H)
    top, left = int(round((oh - size) / 2.0)), int(round((ow - size) / 2.0))
    dev = images_u8.device
    src = images_u8.contiguous()
    bh, kh, ksh = pil_bilinear_tables(W, ow)
    bv, kv, ksv = pil_bilinear_tables(H, oh)
    # horizontal pass: only the source rows the vertical pass of the kept window reads; PIL skips the pass when the size does not change
    r0 = int(bv[top, 0])
    r1 = int(bv[top + size - 1, 0] + bv[top + size - 1, 1])
    tmp = torch.empty(N * C, r1 - r0, size, device=dev, dtype=torch.uint8)
    if ow != W:
        ops.resample_u8(src[:, :, r0:r1].contiguous(), tmp, N * C, r1 - r0, W, 1, torch.from_numpy(bh).to(dev), torch.from_numpy(kh).to(dev), ksh,
                        left, size, 0, r1 - r0)
    else:
        tmp.copy_(src[:, :, r0:r1, left:left + size].reshape(N * C, r1 - r0, size))
    out = torch.empty(N, C, size, size, device=dev, dtype=torch.uint8)
    if oh != H:
        bv2 = bv.copy()
        bv2[:, 0] -= r0                                  # the intermediate starts at source row r0
        ops.resample_u8(tmp, out, N * C, r1 - r0, size, 0, torch.from_numpy(bv2).to(dev), torch.from_numpy(kv).to(dev), ksv, top, size, 0, size)
    else:
        out.copy_(tmp[:, top - r0:top - r0 + size].reshape(N, C, size, size))
    return out


class DeviceInputs:
    """Device-side replacement of the loop's host input work (celebA/EAD-GAN_celebA.py:194-206 DataLoader + RandomHorizontalFlip +
    ToTensor + Normalize(0.5, 0.5); :308-317 numpy draws of z ~ N(0,1), code ~ U(-1,1), labels ~ randint): a uint8 [N,3,64,64] dataset
    resident in HBM (resized / cropped once on the way in: ``resize_center_crop_u8``) and a counter-based generator.  ``enqueue(trainer)`` fills the trainer's
    static input slots with six launches and ticks the device step counter; inside ``trainer.capture(inputs=...)`` they become part
    of the iteration's hipGraph, so a replay needs no host work at all.  Draws are reproducible per (seed, step) and have the
    reference's distributions; they are NOT numpy's stream (parity tests keep using ``load_inputs`` with host draws)."""

    def __init__(self, dataset_u8: torch.Tensor, seed: int = 0, flip: bool = True, sampling: str = "permutation"):
        """``sampling``: "permutation" (default) = DataLoader(shuffle=True) of the reference (:204-206): a keyed permutation of the dataset
        per epoch, every image exactly once, batches of constant size straddle epoch ends; "replacement" = independent uniform draws"""
        _require_cuda(dataset_u8)
        if dataset_u8.dtype != torch.uint8 or dataset_u8.dim() != 4:
            raise ValueError("dataset must be a uint8 [N, C, H, W] device tensor")
        if sampling not in ("permutation", "replacement"):
            raise ValueError("sampling must be 'permutation' or 'replacement'")
        self.data = dataset_u8.contiguous()
        self.sampling = sampling
        self.seed, self.flip = int(seed), flip
        dev = dataset_u8.device
        self.step = torch.zeros(1, device=dev, dtype=torch.int32)
        self.idx = None
        self.flips = None

    def enqueue(self, tr: "CelebATrainer"):
        B = tr.B
        N, C, H, W = self.data.shape
        if self.idx is None or self.idx.numel() != B:
            self.idx = torch.empty(B, device=self.data.device, dtype=torch.int64)
            self.flips = torch.empty(B, device=self.data.device, dtype=torch.uint8)
        if FUSE_INPUTS:
            # the five draws + the one-hot labels as one launch, the counter tick inside the gather: 3 launches instead of 8 at the head of
            # the iteration's critical chain (the same values: a draw depends on (element, step, stream id, seed) only)
            first = (ops.RNG_EPOCH_PERM, self.idx, N, 0, 1) if self.sampling == "permutation" else (ops.RNG_RANDINT, self.idx, 0, N, 1)
            ops.rng_fill_multi([first,                                                      # which images: epoch permutation / uniform draws
                                (ops.RNG_BERNOULLI, self.flips, 0.5, 0.0, 2),               # RandomHorizontalFlip(p=0.5)
                                (ops.RNG_NORMAL, tr.z, 0.0, 1.0, 3), (ops.RNG_UNIFORM, tr.code, -1.0, 1.0, 4),
                                (ops.RNG_RANDINT, tr.labels, 0, tr.G.n_classes, 5, tr.onehot)], self.seed, self.step)
            def gather():                               # ToTensor + Normalize(.5,.5)
                ops.gather_u8_images(self.data, self.idx, self.flips if self.flip else None, tr.real, B, C, H, W, 2.0 / 255.0, -1.0, tick=self.step)
            if INPUTS_ON_PREP and tr.side is not None:
                tr._inputs_tail = gather                # the pipelined body runs it on its preparation lane, in front of the warp
            else:
                gather()
            return
        if self.sampling == "permutation":
            ops.rng_fill(ops.RNG_EPOCH_PERM, self.idx, N, 0, self.seed, self.step, 1)
        else:
            ops.rng_fill(ops.RNG_RANDINT, self.idx, 0, N, self.seed, self.step, 1)
        ops.rng_fill(ops.RNG_BERNOULLI, self.flips, 0.5, 0.0, self.seed, self.step, 2)     # RandomHorizontalFlip(p=0.5)
        ops.gather_u8_images(self.data, self.idx, self.flips if self.flip else None, tr.real, B, C, H, W, 2.0 / 255.0, -1.0)   # ToTensor + Normalize(.5,.5)
        ops.rng_fill(ops.RNG_NORMAL, tr.z, 0.0, 1.0, self.seed, self.step, 3)
        ops.rng_fill(ops.RNG_UNIFORM, tr.code, -1.0, 1.0, self.seed, self.step, 4)
        ops.rng_fill(ops.RNG_RANDINT, tr.labels, 0, tr.G.n_classes, self.seed, self.step, 5)
        ops.onehot(tr.labels, tr.onehot, B, tr.G.n_classes)
        ops.counter_add(self.step, 1)


# 1: D(gen) of the generator step batched with the discriminator step's D(scaled), D(gen) as ONE three-tape forward (same weights: the
# discriminator is not updated in between; each tape keeps its own power iteration in the reference's order).  Same results (tested with
# the switch on), but SLOWER in the overlapped step, 4.75 -> 4.94 ms (profiles/r02_x_ab_batch_d12.txt): the discriminator step's forward
# no longer runs beside the generator's weight-gradient lanes and update.  Default 0: two forwards (T = 1, T = 2).
BATCH_D12 = os.environ.get("EG_BATCH_D12", "0") != "0"
# optimizer updates bucket by bucket (each bucket behind its own weight-gradient chain) instead of one update behind all chains: "0" never,
# "1" every update, "3" the info step's two, or a comma list of g1 (generator step), d2 (discriminator step), d3, g3 (info step: D, then G)
# data parallel: gradient buckets that cross the links as one message, in completion order (contiguous in the arenas)
# EXPERIMENT (default off): the captured iteration as FOUR hipGraphs on two streams (engine.MultiGraph): [inputs, steps 1 and 2] -> [D's
# update + step 3's power iterations and patch rows, ONE chain] beside [step 3's generator forward] -> [rest of step 3].  In ONE hipGraph
# the generator forward starts ~200 us after step 2's main chain ends, although the node graph lets it start at once.  Cut into graphs on
# real streams the delay stays (4.35 -> 4.46 ms, profiles/r03_x_ab_multi_graph.txt; toys: profiles/scripts/graph_streams_toy.py -- a graph
# WITH branches on a second stream holds back later launches on the first, a one-chain graph does not; more HSA queues make it far worse,
# profiles/r03_y_ab_hwq.txt).  Same bits (tests/test_gpu_celeba.py).  Single process only.
MULTI_GRAPH = os.environ.get("EG_MULTI_GRAPH", "0") != "0"
# the image gather and the affine warp at the head of the iteration on the preparation lane, beside the generator forward (which needs the
# draws only), instead of in front of it on the main chain; same bits; EG_INPUTS_ON_PREP=0: on the main chain
INPUTS_ON_PREP = os.environ.get("EG_INPUTS_ON_PREP", "1") != "0"
LAZY_PATCHES = os.environ.get("EG_LAZY_PATCHES", "1") != "0"
# EXPERIMENT (default off here; on in the small networks' trunks): the first D layer's weight gradient straight from the images (ops.wgrad_img,
# N = 128) instead of lazily built patch rows + the per-tap GEMM.  Same gradient within fp32 summation order (tests), 7 launches fewer, but
# SLOWER in the overlapped step, 4.27 -> 4.32 ms (profiles/r03_zzf_ab_wgrad_img_celeba.txt): its 59 KiB of LDS per workgroup do not fit on a CU
# beside a resident 8-wave GEMM workgroup, so the lane chain waits for tiles to retire where the 12-18 us patch-row launches slipped in
WGRAD_IMG = os.environ.get("EG_WGRAD_IMG_CELEBA", "0") != "0"
WGRAD_FIRST = os.environ.get("EG_WGRAD_FIRST", "1") != "0"     # D's lane chains: the weight-gradient GEMM before the bias-gradient sums
# kernel hint of the generator's first layer (ONE 128-row tile x 128 column tiles, 4 K steps: 1 GFLOP): the register-staged kernel (1) runs it
# in ~10 us where the planner's persistent pipeline (0) takes 22-24; same bits; step -0.6 % (profiles/r03_zh_ab_g0_variant.txt)
G0_VARIANT = int(os.environ.get("EG_G0_VARIANT", "1"))
# EXPERIMENT (default off): step 3's generator forward between the forward and the backward of step 2 (pipelined body).  Same bits, but
# slower, 4.46 -> 4.61 ms (profiles/r03_zb_ab_g3_mid.txt): behind step 2's backward the discriminator's update and step 3's power
# iterations then run with nothing beside them -- that chain (update -> three power iterations -> patch rows), not the generator forward,
# is what step 3's discriminator forward waits for.
G3_MID = os.environ.get("EG_G3_MID", "0") != "0"
ZERO_ON_PREP = os.environ.get("EG_ZERO_ON_PREP", "1") != "0"     # gradient zeroing of steps 1 / 2 on the preparation lane (0: on the main stream)
SPLIT2_SET = set(filter(None, os.environ.get("EG_SPLIT2", "").split(",")))      # updates done in two pieces (early layers / rest): g1, d2, d3, g3
DP_START = os.environ.get("EG_DP_START", "lane")
COMM_GROUPS = {"G": (("G4", "G3", "G2"), ("G1", "G0")), "D": (("D4", "D3"), ("D2", "D1", "D0"))}
BUCKET_OPT = os.environ.get("EG_BUCKET_OPT", "g3")
BUCKET_SET = {"0": set(), "1": {"g1", "d2", "d3", "g3"}, "3": {"d3", "g3"}}.get(BUCKET_OPT, set(BUCKET_OPT.split(",")))


class CelebATrainer:
    """One call of :meth:`train_step` == one iteration of the reference loop body
    (celebA/EAD-GAN_celebA.py:299-401): G adversarial step, D step, info+affine step, three Adams
    (lr 1e-3 / 2e-4 / 2e-4, betas (.5,.999), :211-217) -- hand-scheduled over the C ABI with dead work removed
    (no D weight gradients in the G step) and, optionally, replayed from one hipGraph.

    ``allreduce``: optional callable(flat_grad_tensor) applied after each backward pass (data parallel)."""

    def __init__(self, generator: Generator, discriminator: Discriminator, batch_size: int, dtype="bf16", allreduce=None,
                 lr_g=1e-3, lr_d=2e-4, lr_info=2e-4, betas=(0.5, 0.999), lambda_cat=1.0, lambda_con=1.0, lambda_affine=1.0, overlap=True,
                 sync_bn=None):
        self.G, self.D, self.B = generator, discriminator, batch_size
        dt = parse_dtype(dtype)
        generator.set_compute_dtype(dt)
        discriminator.set_compute_dtype(dt)
        self.ge, self.de = generator.engine(batch_size), discriminator.engine(batch_size)
        dev = generator.arena.flat.device
        self.dev = dev
        self.allreduce = allreduce
        # optional dp.SyncBN: the generator's BatchNorm statistics over the global batch (N ranks == 1 rank at equal global batch);
        # default (None): per-rank statistics, like torch DistributedDataParallel without SyncBatchNorm
        self.sync_bn = sync_bn
        self.lr = (lr_g, lr_d, lr_info)
        self.betas = betas
        self.lam = (lambda_cat, lambda_con, lambda_affine)
        ga, da = generator.arena, discriminator.arena
        z = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)
        self.mG, self.vG = z(ga.numel), z(ga.numel)                     # optimizer_G
        self.mD, self.vD = z(da.numel), z(da.numel)                     # optimizer_D
        self.miG, self.viG, self.miD, self.viD = z(ga.numel), z(ga.numel), z(da.numel), z(da.numel)   # optimizer_info
        self.steps = torch.zeros(3, device=dev, dtype=torch.int32)
        self.losses = torch.zeros(4, device=dev, dtype=torch.float32)   # g, d, info
        B = batch_size
        C, S = generator.channels, generator.img_size
        self.theta = torch.empty(B, 2, 3, device=dev, dtype=torch.float32)
        self.scaled = torch.empty(B, C, S, S, device=dev, dtype=torch.float32)
        self.dout = torch.empty(3 * B, 19, device=dev, dtype=torch.float32)
        # static input slots (a captured graph reads these)
        self.real = torch.empty(B, C, S, S, device=dev, dtype=torch.float32)
        self.z = torch.empty(B, generator.latent_dim, device=dev, dtype=torch.float32)
        self.code = torch.empty(B, generator.code_dim, device=dev, dtype=torch.float32)
        self.onehot = torch.empty(B, generator.n_classes, device=dev, dtype=torch.float32)
        self.labels = torch.empty(B, device=dev, dtype=torch.int64)
        self.graph = None
        self.inputs = None
        # weight-gradient chains and re-packing run on a second stream beside the backward-data chain (same arithmetic, same order
        # inside every chain -> bit-identical results with and without)
        self.side = SideStream(dev, Workspace.get(dev), lanes=int(os.environ.get("EG_LANES", "4"))) if overlap else None
        # experiment (EG_G3_EARLY=1): the info step's generator forward on a stream of its own, as soon as the generator's first update is
        # done, beside the discriminator step -- it needs nothing the discriminator step produces
        self.g3_early = overlap and os.environ.get("EG_G3_EARLY", "0") != "0" and sync_bn is None
        if self.g3_early:
            self.g3_stream = torch.cuda.Stream(dev)
            self.g3_ws = Workspace(dev, register=False)
            self.g3_ws._grow("small", Workspace.get(dev).small.numel())

    # -- the hot path ---------------------------------------------------------------------------------
    def _buckets(self, arena):
        """(tag, lo, hi) element ranges of `arena` per layer bucket, in completion order; the ranges tile the arena"""
        eng = self.ge if arena is self.G.arena else self.de
        out, covered = [], 0
        for tag, names in eng.BUCKETS:
            offs = [arena.slices[n] for n in names]
            lo, hi = min(o for o, _ in offs), max(o + k for o, k in offs)
            assert sum(k for _, k in offs) == hi - lo, (tag, "bucket parameters are not contiguous in the arena")
            out.append((tag, lo, hi))
            covered += hi - lo
        assert covered == arena.numel, "buckets do not tile the arena"
        return out

    def _adam(self, arena, m, v, lr, slot, tick):
        """optimizer.step() of one network + refresh of its packed panels (serial body)"""
        eng = self.ge if arena is self.G.arena else self.de
        if not FUSE_ADAM:
            ops.adam_step(arena.flat, arena.grad, m, v, arena.numel, lr, self.betas[0], self.betas[1], 1e-8, self.steps[slot:slot + 1], tick)
            eng.repack()
            return
        if tick:
            ops.adam_tick(self.steps[slot:slot + 1])
        for tag, lo, hi in self._buckets(arena):
            adam_bucket(eng, arena, tag, lo, hi, m, v, lr, self.betas, self.steps[slot:slot + 1], False)

    def _step_body(self):
        if self.side is not None:
            return self._step_body_pipelined()
        return self._step_body_serial()

    def _inputs_head(self):
        G, B = self.G, self.B
        # A = get_matrix(code[:, :5]); scaled = trans_2D(real, A[:, 0:2])           (:325-327)
        if FUSE_INPUTS and (G.img_size * G.img_size) % 256 == 0:
            ops.warp_affine_rpqxy(self.real, self.code, G.code_dim, self.theta, self.scaled, B, G.channels, G.img_size, G.img_size, zero=self.losses)
            return
        ops.fill_f32(self.losses)
        ops.theta_rpqxy(self.code, G.code_dim, B, self.theta)
        ops.warp_affine(self.real, self.theta, self.scaled, B, G.channels, G.img_size, G.img_size)

    def _reduce(self, flat):
        """gradient average over the ranks, on the CURRENT stream (the optimizer lane: only that lane waits for the collective)"""
        ar = self.allreduce
        if ar is None:
            return
        if hasattr(ar, "start"):
            ar.finish(ar.start(flat))
        else:
            ar(flat)

    def _sn_d12(self):
        """power iterations of the three D forwards of steps 1 and 2 in the reference's order: D(gen) of step 1 (tape 2), then D(scaled),
        D(gen) of step 2 (tapes 0, 1); all read the same weights"""
        for t in (2, 0, 1):
            self.de._sn_tape(t)

    def _forward_d12(self, gen, prepared=False):
        """ONE forward over tapes 0 = D(scaled), 1 = D(gen) [step 2], 2 = D(gen) [step 1]  (celebA/EAD-GAN_celebA.py:338,357-358)"""
        if not prepared:
            self._sn_d12()
            self.de._im2col_tape(0, self.scaled)
        return self.de.forward([self.scaled, gen, gen], 0, prepared=(True, False, False))

    def _step_body_serial(self):
        """one stream, program order of the reference loop body (overlap=False; the pipelined body below is bit-identical)"""
        G, D, ge, de, B = self.G, self.D, self.ge, self.de, self.B
        ga, da = G.arena, D.arena
        cd, nc = G.code_dim, G.n_classes
        lcat, lcon, laff = self.lam
        fh1 = not BATCH_D12 and de.head_fused_ok(2)      # head + losses + head backward in one launch (steps 1 and 2)
        fh3 = de.head_fused_ok(3) and cd >= 5            # ... of the info step
        self._inputs_head()
        # ---- 1) generator adversarial step (:334-345) ----
        ops.fill_f32(ga.grad)
        gen = ge.forward(self.z, self.onehot, self.code, sync=self.sync_bn)
        if BATCH_D12:
            out12 = self._forward_d12(gen)
            out = out12[2 * B:]
        else:
            out = de.forward([gen], 2, head=not fh1)
        if fh1:
            de.head_losses(2, 1, self.dout[2 * B:], self.losses[0:1], targets=(1.0,), scales=(1.0,))
        else:
            ops.loss_bce_sigmoid(out, 19, 0, B, 1.0, 1.0, self.losses[0:1], self.dout[2 * B:])
        dimg = de.backward(2, 1, self.dout[2 * B:], da.grad, need_wgrad=False, need_dimg=True, head_done=fh1)
        ge.backward(dimg, ga.grad, None, sync=self.sync_bn)
        self._reduce(ga.grad)
        self._adam(ga, self.mG, self.vG, self.lr[0], 0, True)
        # ---- 2) discriminator step (:353-366); gen is the (detached) output of step 1; D(scaled) then D(gen), batched ----
        ops.fill_f32(da.grad)
        out = out12 if BATCH_D12 else de.forward([self.scaled, gen], 0, head=not fh1)
        if fh1:
            de.head_losses(0, 2, self.dout[:2 * B], self.losses[1:2], targets=(1.0, 0.0), scales=(0.5, 0.5))
        else:
            ops.loss_bce_sigmoid(out[:B], 19, 0, B, 1.0, 0.5, self.losses[1:2], self.dout[:B])
            ops.loss_bce_sigmoid(out[B:], 19, 0, B, 0.0, 0.5, self.losses[1:2], self.dout[B:2 * B])
        de.backward(0, 2, self.dout[:2 * B], da.grad, head_done=fh1)
        self._reduce(da.grad)
        self._adam(da, self.mD, self.vD, self.lr[1], 1, True)
        ops.fill_f32(da.grad)
        # ---- 3) info + affine step (:375-401): D(gen), D(scaled), D(real) batched as tapes 0,1,2 ----
        ops.fill_f32(ga.grad)
        gen = ge.forward(self.z, self.onehot, self.code, sync=self.sync_bn)
        out = de.forward([gen, self.scaled, self.real], 0, head=not fh3)
        if fh3:
            de.head_losses(0, 3, self.dout, self.losses[2:3], info=(1, cd, nc, self.code, self.labels, lcat, lcon, laff))
        else:
            o_gen, o_trans, o_real = out[:B], out[B:2 * B], out[2 * B:]
            ops.loss_mse(o_gen, 19, 1, cd, B, self.code, cd, 0.0, lcon, self.losses[2:3], self.dout[:B])
            ops.loss_ce_softmaxed(o_gen, 19, cd + 1, nc, B, self.labels, lcat, self.losses[2:3], self.dout[:B])
            ops.loss_affine_rpqxy(o_real, o_trans, 19, 1, B, self.code, cd, laff, self.losses[2:3], self.dout[2 * B:], self.dout[B:2 * B])
        dimg = de.backward(0, 3, self.dout, da.grad, need_dimg=True, head_done=fh3)
        ge.backward(dimg, ga.grad, None, sync=self.sync_bn)
        self._reduce(da.grad)
        self._reduce(ga.grad)
        self._adam(da, self.miD, self.viD, self.lr[2], 2, True)     # optimizer_info's step counter is shared by both arenas
        self._adam(ga, self.miG, self.viG, self.lr[2], 2, False)

    def _step_body_pipelined(self):
        """The same iteration on five streams.  Main stream: the forward / backward-data chain of the three sub-steps, back to back.
        Lanes 0,1: weight-gradient GEMMs with their reductions.  Preparation lane: the next sub-step's power iterations and patch
        rows.  Optimizer lane:
        per network, behind ALL of its weight-gradient chains: (all-reduce over the ranks ->) Adam -> gradient zeroing -> re-packing.
        The main stream never waits for an optimizer update it does not need: step 2 neither reads nor writes G, step 3 starts with
        the generator forward, which does not read D; it waits for single events (``mark`` / ``wait``) instead of joining every lane."""
        G, D, ge, de, B = self.G, self.D, self.ge, self.de, self.B
        ga, da = G.arena, D.arena
        cd, nc = G.code_dim, G.n_classes
        lcat, lcon, laff = self.lam
        side = self.side
        side.begin_step()
        evs = {}
        fh1 = not BATCH_D12 and de.head_fused_ok(2)      # head + losses + head backward in one launch (steps 1 and 2)
        fh3 = de.head_fused_ok(3) and cd >= 5            # ... of the info step

        ar = self.allreduce
        ar_async = ar is not None and hasattr(ar, "start")

        def update(arena, m, v, lr, slot, tick, zero, eng, key=None, key_w=None, where=""):
            """Queue one network's optimizer update on the optimizer lane, bucket by bucket in the order the backward pass completes them:
            a bucket's Adam (+ gradient zeroing in the same pass) and panel re-packing wait only for the lane chain that completes the
            bucket's gradients and for the main-stream kernels that still read its parameters (engine.SideStream.free), so the update of
            the upper layers runs beside the backward pass of the lower ones and only the last bucket's update is behind all of it.
            Data parallel: each bucket's all-reduce is STARTED here, on the main stream, once its chain is done (RCCL's stream must only
            ever wait for the capture's origin stream: a lane that RCCL waited for and that later waits for RCCL is the stream-level back
            edge hipStreamEndCapture crashes on), and FINISHED on the optimizer lane, so the main stream waits neither for the
            collective nor for Adam / re-packing."""
            if side.deferred != "1":
                side.flush()
            side.close_tags()
            fuse = FUSE_ADAM and (FUSE_ADAM_AT == "all" or where in FUSE_ADAM_AT.split(","))
            buckets = self._buckets(arena)
            hs, finished = {}, set()

            def finish(tag):
                h = hs.get(tag)
                if h is not None and id(h) not in finished:
                    finished.add(id(h))
                    ar.finish(h)
            if ar is not None:
                capturing = torch.cuda.is_current_stream_capturing() and os.environ.get("EG_COMM_CAPTURE", "0") == "0"
                if ar_async and not capturing:
                    # eager launches (the default at N > 1): the gradient arena crosses the links as TWO messages per update (COMM_GROUPS:
                    # the layers whose gradients are complete early / the rest -- every collective costs the host ~40 us and a ring its
                    # latency, five per update bought nothing), each started from the communication stream behind the lane chains that
                    # complete it and finished on the optimizer lane: the main stream waits for neither
                    span = {tag: (lo, hi) for tag, lo, hi in buckets}
                    for grp in COMM_GROUPS["G" if arena is ga else "D"]:
                        # the stream the message is started from: DP_START = "lane": the lane that ran the group's last chain (no further
                        # stream: eight busy streams on four hardware queues stall each other), "comm": a communication stream
                        src = side.done_lane[grp[-1]].stream if DP_START == "lane" else side.comm
                        for tag in grp:
                            src.wait_event(side.done.pop(tag))              # KeyError: a bucket whose chain was never forked
                        lo, hi = min(span[t][0] for t in grp), max(span[t][1] for t in grp)
                        assert sum(span[t][1] - span[t][0] for t in grp) == hi - lo, "a communication group must be contiguous in the arena"
                        with torch.cuda.stream(src):
                            h = ar.start(arena.grad[lo:hi])
                            started = side.mark()
                        side.opt.stream.wait_event(started)                 # (the collective's own handle orders the optimizer lane behind it too)
                        for tag in grp:
                            hs[tag] = h                 # whichever bucket of the group is updated first finishes the message
                    buckets_ar = ()
                else:
                    buckets_ar = buckets
                for tag, lo, hi in buckets_ar:
                    done = side.done.pop(tag)           # KeyError: a bucket whose chain was never forked
                    # inside a hipGraph capture RCCL's stream may only ever wait for the capture's origin stream (a lane that RCCL waited
                    # for and that later waits for RCCL is the stream-level back edge hipStreamEndCapture crashes on): started from the
                    # main stream, which therefore waits for the bucket's chain
                    side.wait(done)
                    if ar_async:
                        hs[tag] = ar.start(arena.grad[lo:hi])
                    else:
                        ar(arena.grad[lo:hi])
            last = buckets[-1][0]
            if where in SPLIT2_SET and ar is None:
                # the update in TWO pieces: the layers whose gradients are complete early (COMM_GROUPS' first group: for D 77 % of the
                # parameters) as soon as their chains are done, the rest behind all chains -- the path from the end of the backward pass
                # to "weights are new" (the next power iteration waits for it) is an Adam over the small remainder only
                span = {tag: (lo, hi) for tag, lo, hi in buckets}
                early = COMM_GROUPS["G" if arena is ga else "D"][0]
                late = tuple(t for t, _, _ in buckets if t not in early)

                def piece(tags, first):
                    lo, hi = min(span[t][0] for t in tags), max(span[t][1] for t in tags)
                    assert sum(span[t][1] - span[t][0] for t in tags) == hi - lo

                    def fn(_ws):
                        ops.adam_step_zero(arena.flat[lo:hi], arena.grad[lo:hi], m[lo:hi], v[lo:hi], hi - lo, lr, self.betas[0], self.betas[1], 1e-8,
                                           self.steps[slot:slot + 1], tick and first, zero)
                        if not first and key_w:
                            evs[key_w] = side.mark()
                        for t in tags:
                            eng.repack_bucket(t)
                        if not first and key:
                            evs[key] = side.mark()
                    return fn
                side.defer_opt_after(early, piece(early, True))
                for t in late:
                    side.done.pop(t, None)
                    side.free.pop(t, None)
                side.defer_opt(piece(late, False))
                return
            if where not in BUCKET_SET:                 # the whole arena in one update behind ALL chains
                def whole(_ws):
                    for tag in hs:
                        finish(tag)
                    if fuse:
                        if tick:
                            ops.adam_tick(self.steps[slot:slot + 1])
                        for tag, lo, hi in buckets:
                            adam_bucket(eng, arena, tag, lo, hi, m, v, lr, self.betas, self.steps[slot:slot + 1], zero)
                        if key_w:
                            evs[key_w] = side.mark()
                    else:
                        ops.adam_step_zero(arena.flat, arena.grad, m, v, arena.numel, lr, self.betas[0], self.betas[1], 1e-8, self.steps[slot:slot + 1],
                                           tick, zero)
                        if key_w:
                            evs[key_w] = side.mark()
                        eng.repack()
                    if key:
                        evs[key] = side.mark()
                side.done.clear()
                side.free.clear()
                side.defer_opt(whole)
                return
            for k, (tag, lo, hi) in enumerate(buckets):
                def fn(_ws, tag=tag, lo=lo, hi=hi, first=(k == 0)):
                    finish(tag)
                    if fuse:
                        if tick and first:
                            ops.adam_tick(self.steps[slot:slot + 1])
                        adam_bucket(eng, arena, tag, lo, hi, m, v, lr, self.betas, self.steps[slot:slot + 1], zero)
                        if key_w and tag == last:
                            evs[key_w] = side.mark()    # master weights are new
                    else:
                        ops.adam_step_zero(arena.flat[lo:hi], arena.grad[lo:hi], m[lo:hi], v[lo:hi], hi - lo, lr, self.betas[0], self.betas[1], 1e-8,
                                           self.steps[slot:slot + 1], tick and first, zero)
                        if key_w and tag == last:
                            evs[key_w] = side.mark()    # master weights are new
                        eng.repack_bucket(tag)
                    if key and tag == last:
                        evs[key] = side.mark()          # panels are new, gradients zeroed
                side.defer_opt_after((tag,), fn)

        # the generator forward needs the draws only: the image gather (DeviceInputs) and the warp go to the preparation lane, in front of
        # step 1's power iteration -- everything that reads them (step 2, step 3, the losses the warp launch zeroes) is behind the wait
        # for that lane's event below
        tail, self._inputs_tail = getattr(self, "_inputs_tail", None), None
        on_prep = INPUTS_ON_PREP
        if not on_prep:
            if tail is not None:
                tail()
            self._inputs_head()
        # ---- 1) generator adversarial step (:334-345); D(gen) lives in tape slot 2 so that step 2's tapes can be prepared meanwhile ----

        def sn1(_ws):                                   # the power iterations only need D's weights: beside the generator forward
            if on_prep:
                if tail is not None:
                    tail()
                self._inputs_head()
            if BATCH_D12:
                self._sn_d12()
                de._im2col_tape(0, self.scaled)
            else:
                de._sn_tape(2)
            # optimizer.zero_grad() of the generator and discriminator steps (the last updates of an iteration leave their gradients in
            # place: callers and tests read them): here, beside the generator forward -- the first gradient write is a backward pass away
            if ZERO_ON_PREP:
                ops.fill_f32(ga.grad)
                ops.fill_f32(da.grad)
            evs["sn1"] = side.mark()
        side.defer_prep(sn1)
        if not ZERO_ON_PREP:
            ops.fill_f32(ga.grad)
        gen = ge.forward(self.z, self.onehot, self.code, sync=self.sync_bn)
        side.wait(evs["sn1"])
        if BATCH_D12:
            out12 = self._forward_d12(gen, prepared=True)
            out = out12[2 * B:]
        else:
            out = de.forward([gen], 2, prepared=(False,), head=not fh1)

            def prep2(_ws):                             # step 2's power iterations (after step 1's in the u/v chain) and patch rows
                de.prepare(0, [self.scaled, gen])
                evs["prep2"] = side.mark()
            side.defer_prep(prep2)
        if fh1:
            de.head_losses(2, 1, self.dout[2 * B:], self.losses[0:1], targets=(1.0,), scales=(1.0,))
        else:
            ops.loss_bce_sigmoid(out, 19, 0, B, 1.0, 1.0, self.losses[0:1], self.dout[2 * B:])
        side.flush()
        dimg = de.backward(2, 1, self.dout[2 * B:], da.grad, need_wgrad=False, need_dimg=True, head_done=fh1)
        ge.backward(dimg, ga.grad, side, sync=self.sync_bn)
        update(ga, self.mG, self.vG, self.lr[0], 0, True, True, ge, key="g", where="g1")                # beside the whole of step 2
        if self.g3_early:
            side.flush()
            self.g3_stream.wait_event(evs["g"])         # new panels, gradients zeroed
            if not BATCH_D12:
                self.g3_stream.wait_event(evs["prep2"])     # step 2's patch rows of the step-1 image exist: the image buffer may be rewritten
            with torch.cuda.stream(self.g3_stream), self.g3_ws.active():
                ge.forward(self.z, self.onehot, self.code, ws=self.g3_ws)
                evs["g3fwd"] = side.mark()
        # ---- 2) discriminator step (:353-366); gen is the (detached) output of step 1; D(scaled) then D(gen), batched ----
        if not ZERO_ON_PREP:
            ops.fill_f32(da.grad)
        side.flush()
        if BATCH_D12:
            out = out12
        else:
            side.wait(evs["prep2"])
            out = de.forward([self.scaled, gen], 0, prepared=(True, True), head=not fh1)
        if fh1:
            de.head_losses(0, 2, self.dout[:2 * B], self.losses[1:2], targets=(1.0, 0.0), scales=(0.5, 0.5))
        else:
            ops.loss_bce_sigmoid(out[:B], 19, 0, B, 1.0, 0.5, self.losses[1:2], self.dout[:B])
            ops.loss_bce_sigmoid(out[B:], 19, 0, B, 0.0, 0.5, self.losses[1:2], self.dout[B:2 * B])
        cut = getattr(self, "_cut", None)               # capture_segments: the iteration is being captured as several hipGraphs
        g3_mid = G3_MID and cut is None and not self.g3_early and not BATCH_D12
        if g3_mid:
            # Step 3's generator forward HERE, between the discriminator step's forward and backward: it reads only G (new since step
            # 1's update, long done) and writes only G's activations and the image buffer, whose step-1 content step 2 no longer needs
            # (the discriminator read it in its forward; the patch rows its weight gradient reads were built by prep2).  Placed where
            # the loop has it -- behind step 2's backward -- it starts ~200 us late and runs beside the power iterations at 2/3 speed
            # (DESIGN.md 6.0); here it is plain main-chain work between two GEMM sequences.  Same kernels, same operands: same bits.
            side.wait(evs["g"])
            gen3 = ge.forward(self.z, self.onehot, self.code, sync=self.sync_bn)
        de.backward(0, 2, self.dout[:2 * B], da.grad, side=side, head_done=fh1)
        g3_first = os.environ.get("EG_G3_FIRST", "0") != "0"
        keep = side.deferred
        if cut is not None:
            side.join()                                 # every chain of steps 1 and 2 (and the generator's update) ends in this segment
            side.cut()
            cut(1, (0,))                                # second stream, behind the first segment
            side.inline = True                          # ... as ONE chain: update, then step 3's power iterations and patch rows
        if g3_first:
            side.deferred = "1"                         # layer 0's chain, D's update and step 3's preparation are CAPTURED behind the generator forward's launches
        update(da, self.mD, self.vD, self.lr[1], 1, True, True, de, key_w="dw", where="d2")             # beside step 3's generator forward

        # the first layer reads the images themselves (IMG_DIRECT): the patch rows are only the weight gradient's operand and are built by
        # its own chain in step 3's backward -- not here, on the chain step 3's discriminator forward waits for (EG_LAZY_PATCHES=0: here)
        lazy = LAZY_PATCHES and de.img_direct

        def prep3(_ws):                                 # step 3's three power iterations (new weights) [, patch rows of scaled / real]
            side.wait(evs["dw"])
            de.prepare(0, [None, None, None] if lazy else [None, self.scaled, self.real])
        side.defer_prep(prep3)
        if cut is not None:
            side.flush()
            side.inline = False
            side.cut()
            cut(0, ())                                  # the caller's stream again: behind the first segment by stream order
        # ---- 3) info + affine step (:375-401): D(gen), D(scaled), D(real) batched as tapes 0,1,2 ----
        if self.g3_early:
            side.wait(evs["g3fwd"])
            gen = ge.img
        elif g3_mid:
            gen = gen3
        else:
            if cut is None:
                side.wait(evs["g"])                     # G's panels and zeroed gradients (optimizer lane, step 1)
            gen = ge.forward(self.z, self.onehot, self.code, sync=self.sync_bn)
        side.deferred = keep
        if cut is not None:
            side.join()
            side.cut()
            cut(0, (1,))                                # behind the second stream's segment: D's panels, power iterations, patch rows
        else:
            side.join()                                 # D's panels, power iterations, patch rows
        out = de.forward([gen, self.scaled, self.real], 0, prepared=(False, False, False) if lazy else (False, True, True), head=not fh3)
        if fh3:
            de.head_losses(0, 3, self.dout, self.losses[2:3], info=(1, cd, nc, self.code, self.labels, lcat, lcon, laff))
        else:
            o_gen, o_trans, o_real = out[:B], out[B:2 * B], out[2 * B:]
            ops.loss_info_rpqxy(o_gen, o_trans, o_real, 19, 1, cd, nc, B, self.code, cd, self.labels, lcat, lcon, laff, self.losses[2:3], self.dout[:B],
                                self.dout[B:2 * B], self.dout[2 * B:])           # the three losses in one launch
        dimg = de.backward(0, 3, self.dout, da.grad, need_dimg=True, side=side, head_done=fh3)
        # D's update beside the generator backward; it ticks optimizer_info's counter (shared by both arenas), G's does not
        update(da, self.miD, self.viD, self.lr[2], 2, True, False, de, where="d3")
        ge.backward(dimg, ga.grad, side, sync=self.sync_bn)
        update(ga, self.miG, self.viG, self.lr[2], 2, False, False, ge, where="g3")
        side.join()

    # -- public API -----------------------------------------------------------------------------------
    def import_adam_state(self, opt_G, opt_D, opt_info):
        """Load exp_avg / exp_avg_sq / step of three ``torch.optim.Adam`` objects built like the reference's
        (celebA/EAD-GAN_celebA.py:211-217: G params | D params | G+D params, in ``.parameters()`` order)."""
        def fill(opt, params, m, v, off0=0):
            off = off0
            step = 0
            for p in params:
                st = opt.state.get(p, {})
                n = p.numel()
                if st:
                    m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    step = int(st["step"])
                off += n
            return step
        gp, dp = opt_G.param_groups[0]["params"], opt_D.param_groups[0]["params"]
        ip = opt_info.param_groups[0]["params"]
        s0 = fill(opt_G, gp, self.mG, self.vG)
        s1 = fill(opt_D, dp, self.mD, self.vD)
        s2 = fill(opt_info, ip[:len(gp)], self.miG, self.viG)
        fill(opt_info, ip[len(gp):], self.miD, self.viD)
        self.steps.copy_(torch.tensor([s0, s1, s2], dtype=torch.int32))

    def load_inputs(self, real_imgs, z, code, labels):
        self.real.copy_(real_imgs, non_blocking=True)
        self.z.copy_(z, non_blocking=True)
        self.code.copy_(code, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.onehot.zero_()
        self.onehot.scatter_(1, self.labels.view(-1, 1), 1.0)

    def capture(self, warmup: bool = False, inputs: "DeviceInputs | None" = None):
        """Capture the whole iteration into one hipGraph (inputs are read from the static slots; with ``inputs`` -- a DeviceInputs --
        the graph first draws them on the device, so a replay is a complete loop iteration without host work).

        At least one eager iteration must have run before (it loads every kernel and sizes the workspace);
        ``warmup=True`` runs that iteration here -- note that it IS a real training step on the current inputs."""
        if warmup:
            self._step_body()
        if inputs is not None:
            self.inputs = inputs
        if MULTI_GRAPH and self.side is not None and self.allreduce is None and not self.g3_early and not BATCH_D12:
            self.side._live = []
            try:
                return capture_segments(self, self._step_with_inputs)
            finally:
                self.side._live = None
                self.side.inline = False
        return capture_step(self, self._step_with_inputs)

    def _step_with_inputs(self):
        if getattr(self, "inputs", None) is not None:
            self.inputs.enqueue(self)
        self._step_body()

    def step_resident(self):
        """Run one iteration on whatever is in the static input slots (or on fresh device-side draws if the trainer was captured /
        configured with a DeviceInputs); returns the device loss tensor [g,d,info,_]."""
        check_usable(self)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step_with_inputs()
        return self.losses

    def train_step(self, real_imgs, z, code, labels):
        """train-loop entry: real_imgs [B,3,64,64] in [-1,1], z [B,200], code [B,8], labels int64 [B]."""
        self.load_inputs(real_imgs, z, code, labels)
        l = self.step_resident().tolist()
        return {"g_loss": l[0], "d_loss": l[1], "info_loss": l[2]}
