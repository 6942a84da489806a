"""MNIST 32x32 path of EAD-GAN on MI355X: drop-in ``Generator`` / ``Discriminator`` / ``Encoder`` / ``transformation_2D`` /
``weights_init_normal`` / ``to_categorical`` (MNIST/EAD-GAN_rpqmnxy.py:54-192), ``get_matrix`` / ``affine_regularizer`` and the
latent<->affine scalings (MNIST/utils_rpqmnxy.py:46-134), plus the fused train-loop entry :class:`MnistTrainer`
(loop body :338-446).  All arithmetic runs in hand-written HIP kernels behind the C ABI; torch modules are parameter
containers only (reference ``state_dict`` keys: ``l1.0.*``, ``conv_blocks.N.*``, ``adv_layer.0.weight_orig`` ...).
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops
from .celeba import _HipModule, _require_cuda, transformation_2D      # noqa: F401  (same STN warp in both scripts)
from .engine import (FUSE_DRAWS, Arena, ConvRec, DeviceSampler, ResidentStep, SideStream, SyncScratch, Workspace, bn_train_backward, bn_train_forward, capture_step, check_usable,
                     parse_dtype)
from .ops import ACT_LRELU, ACT_NONE, ACT_TANH, EG_F32, OUT_NCHW_F32
from .trunk import Head, TrunkEngine

opt = argparse.Namespace(n_epochs=200, batch_size=128, lr=0.0001, b1=0.5, b2=0.999, n_cpu=8, latent_dim=62, code_dim=7, n_classes=10,
                         img_size=32, channels=1, sample_interval=4000)          # argparse defaults, :35-48
TRUNK = (16, 32, 64, 128)
SLOPE = 0.2
# the generator's two Upsample(2) + Conv2d(3, 1, 1) blocks (:81-82, 85-86) as transposed 4x4 / stride-2 convolutions with summed taps: 4 taps per
# output pixel instead of 9 in the forward, the input gradient (no upsampled gradient + sum-pool) and the weight gradient (ops.up3_expand /
# up3_contract; the same sums up to fp32 rounding of the tap sums).  EG_UP3_CONVT=0: the 3x3 convolution over the upsampled lattice
UP3_AS_CONVT = os.environ.get("EG_UP3_CONVT", "1") != "0"
# the last layer's weight gradient (Conv2d(64, 1, 3, 1, 1), :88) by the kernel that reads the activation once (ops.wgrad_c1) instead of the
# per-tap GEMM over the output padded to 8 channels (nine passes over 67 MB at B = 256); EG_C1_DIRECT=0: the GEMM
C1_DIRECT = os.environ.get("EG_C1_DIRECT", "1") != "0"


def weights_init_normal(m):
    """:54-60 -- Conv weights N(0,.02) (reaches spectral-norm ``weight_orig`` through shared storage), BatchNorm N(1,.02)/0."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


def to_categorical(y, num_columns, device=None):
    y = torch.as_tensor(np.asarray(y), dtype=torch.int64, device=device)
    return torch.nn.functional.one_hot(y, num_columns).to(torch.float32)


# ================================================================================================
# Generator: Linear -> view[B,128,8,8] -> BN -> Up -> Conv3x3 -> BN(.8) -> LReLU -> Up -> Conv3x3 -> BN(.8) -> LReLU -> Conv3x3 -> Tanh
# ================================================================================================
class _GenEngine:
    def __init__(self, gen: "Generator", B: int, dtype: int):
        self.gen, self.B, self.dtype = gen, B, dtype
        dev = gen.arena.flat.device
        tdt = ops.torch_dtype(dtype)
        self.ws = ws = Workspace.get(dev)
        s = gen.init_size                                            # 8
        self.cin = gen.input_dim                                     # 79
        self.cpad = ops.round_up(self.cin, 8)
        self.nl1 = 128 * s * s
        self.l1 = ConvRec(dtype, B, 1, 1, self.cpad, self.nl1, 1, 1, 0, device=dev, want_bwd=False, ws=ws)
        self.up3 = UP3_AS_CONVT
        if self.up3:
            # Upsample(2) + Conv2d(3, 1, 1) as the transposed 4x4 / stride-2 convolution with summed taps (ops.up3_expand): conv view of
            # ConvTranspose2d(Cin -> Cout, 4, 2, 1) = Conv2d(Cout -> Cin, 4, 2, 1) on the OUTPUT side, as in the dSprites / CelebA generators
            self.c1 = ConvRec(dtype, B, 2 * s, 2 * s, 128, 128, 4, 2, 1, device=dev, ws=ws)
            self.c2 = ConvRec(dtype, B, 4 * s, 4 * s, 64, 128, 4, 2, 1, device=dev, ws=ws)
            self.w4 = [torch.empty(128, 128, 4, 4, device=dev), torch.empty(128, 64, 4, 4, device=dev)]      # effective masters [in][out][4][4]
            self.dw4 = [torch.empty_like(w) for w in self.w4]
        else:
            self.c1 = ConvRec(dtype, B, s, s, 128, 128, 3, 1, 1, up=1, device=dev, ws=ws)
            self.c2 = ConvRec(dtype, B, 2 * s, 2 * s, 128, 64, 3, 1, 1, up=1, device=dev, ws=ws)
        self.CH = gen.channels
        # last conv (64 -> channels): forward with the real N; input/weight gradients with the output side padded to 8 channels
        self.c3f = ConvRec(dtype, B, 4 * s, 4 * s, 64, self.CH, 3, 1, 1, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.c3 = ConvRec(dtype, B, 4 * s, 4 * s, 64, 8, 3, 1, 1, device=dev, want_fwd=False, ws=ws)
        self.w3pad = torch.zeros(8, 64, 3, 3, device=dev, dtype=torch.float32)
        self.c1_direct = C1_DIRECT and ops.wgrad_c1_ok(dtype, 64, 4 * s, 4 * s, self.CH, 3, 1, 1)
        if self.c1_direct:
            ws.need_slab(ops.wgrad_c1_splits(B, 4 * s) * 9 * 64 * 4)
        e = lambda *shape, dt=tdt: torch.empty(shape, device=dev, dtype=dt)
        f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        self.inp = e(B, self.cpad)
        self.bias_perm = f(self.nl1)
        self.gb_perm = f(self.nl1)
        self.h = e(B, s, s, 128)
        self.a0 = e(B, s, s, 128)
        self.z1, self.a1 = e(B, 2 * s, 2 * s, 128), e(B, 2 * s, 2 * s, 128)
        self.z2, self.a2 = e(B, 4 * s, 4 * s, 64), e(B, 4 * s, 4 * s, 64)
        self.img = f(B, self.CH, 4 * s, 4 * s)
        self.mean = [f(128), f(128), f(64)]
        self.invstd = [f(128), f(128), f(64)]
        self.sync_scratch = SyncScratch((128, 128, 64), dev)
        self.dimg_z = torch.empty_like(self.img)
        self.p8 = e(B * (4 * s) ** 2, 8)
        self.da2, self.dz2 = torch.empty_like(self.a2), torch.empty_like(self.a2)
        self.dup = None if self.up3 else e(B, 4 * s, 4 * s, 128)      # gradient at an upsampled resolution (largest: 32x32x128)
        self.da1, self.dz1 = torch.empty_like(self.a1), torch.empty_like(self.a1)
        self.da0, self.dh = torch.empty_like(self.a0), torch.empty_like(self.h)
        for M, C in ((B * s * s, 128), (B * 4 * s * s, 128), (B * 16 * s * s, 64)):
            ws.need_small(ops.bn_ws_floats(M, C))
        ws.need_small(ops.bias_grad_ws_floats(B, self.nl1))
        ws.need_small(B * self.CH)
        ws.need_sums(2 * 128)
        self.repack()

    def _p(self, path):
        m = self.gen
        for part in path.split("."):
            m = m[int(part)] if part.isdigit() else getattr(m, part)
        return m

    def repack(self):
        self._repack_masters()
        if self.up3:                                    # (plain launches: a batched method's body does not run inside a capture)
            ops.up3_expand(self.gen.conv_blocks[2].weight, self.w4[0], 128, 128)
            ops.up3_expand(self.gen.conv_blocks[6].weight, self.w4[1], 64, 128)
        self._repack_derived()

    @ops.batched_packs
    def _repack_derived(self):
        """panels packed from copies that ``_repack_masters`` has just written (their own launch behind it)"""
        self.c3.pack(self.w3pad)
        if self.up3:
            self.c1.pack(self.w4[0])
            self.c2.pack(self.w4[1])

    @ops.batched_packs
    def _repack_masters(self):
        dt, g = self.dtype, self.gen
        hw = g.init_size ** 2
        w, b = g.l1[0].weight, g.l1[0].bias
        # NHWC row n' = hw*128 + c  <-  master row f = c*hw_count + hw  (the reference views the Linear output as [B,128,8,8])
        ops.pack_strided(dt, w, self.l1.wp_fwd, self.nl1, self.cin, self.l1.Kpad_fwd, 128, self.cin, hw * self.cin, 1)
        ops.pack_strided(EG_F32, b, self.bias_perm, self.nl1, 1, 1, 128, 1, hw, 0)
        if not self.up3:
            self.c1.pack(g.conv_blocks[2].weight)
            self.c2.pack(g.conv_blocks[6].weight)
        self.c3f.pack(g.conv_blocks[9].weight)
        ops.pack_strided(EG_F32, g.conv_blocks[9].weight, self.w3pad, self.CH, 576, 576, 1, 576, 0, 1)

    def forward(self, noise, labels, code, training=True, sync=None):
        """``training=False``: BatchNorm with the running statistics, nothing updated (module.eval()).  ``sync`` (a dp.SyncBN):
        batch statistics over all ranks."""
        dt, B, g, ws = self.dtype, self.B, self.gen, self.ws
        cb = g.conv_blocks
        ops.concat_cast(dt, noise, labels, code, self.inp, B, self.cpad)
        ops.conv_fwd(self.l1.c, dt, self.inp, self.l1.wp_fwd, self.h, ops.epilogue(bias=self.bias_perm))

        def bn(x, y, mod, i, M, C, act, slope=0.0):
            if not training:
                ops.bn_fwd_eval(dt, x, y, M, C, mod.weight, mod.bias, mod.eps, mod.running_mean, mod.running_var, ws.small, act, slope)
                return
            bn_train_forward(dt, x, y, M, C, mod, self.mean[i], self.invstd[i], ws.small, act, slope, sync, self.sync_scratch.stats[i])
        s = g.init_size
        bn(self.h, self.a0, cb[0], 0, B * s * s, 128, ACT_NONE)
        up_conv = ops.conv_bwd_data if self.up3 else ops.conv_fwd      # (transposed convolution = backward-data of its conv view)
        up_w = (lambda r: r.wp_bwd) if self.up3 else (lambda r: r.wp_fwd)
        up_conv(self.c1.c, dt, self.a0, up_w(self.c1), self.z1, ops.epilogue(bias=cb[2].bias))
        bn(self.z1, self.a1, cb[3], 1, B * 4 * s * s, 128, ACT_LRELU, SLOPE)
        up_conv(self.c2.c, dt, self.a1, up_w(self.c2), self.z2, ops.epilogue(bias=cb[6].bias))
        bn(self.z2, self.a2, cb[7], 2, B * 16 * s * s, 64, ACT_LRELU, SLOPE)
        ops.conv_fwd(self.c3f.c, dt, self.a2, self.c3f.wp_fwd, self.img, ops.epilogue(bias=cb[9].bias, act=ACT_TANH, out_mode=OUT_NCHW_F32))
        return self.img

    def backward(self, dimg, grad, side=None, sync=None):
        """Accumulates d(loss)/d(params) into ``grad``.  With ``side`` (engine.SideStream) every layer's weight-gradient chain (TN GEMM, slab
        reduce, bias sums) is forked onto a lane as soon as the layer's output gradient exists, beside the backward-data chain of the
        layers below; the caller joins before it reads ``grad``.  Same kernels, same order inside every chain: bit-identical."""
        dt, B, g, ws = self.dtype, self.B, self.gen, self.ws
        cb = g.conv_blocks
        gof = lambda name: g.arena.grad_of(name, grad)
        s = g.init_size
        S = 4 * s

        def wgrad_side(fn, lane):
            if side is None:
                fn(ws)
            else:
                side.defer(lane, fn)
        flush = side.flush if side is not None else (lambda: None)

        ops.act_grad_mul_bias_nchw(dimg, self.img, self.dimg_z, B, self.CH, S * S, ACT_TANH, 0.0, ws.small, gof("conv_blocks.9.bias"))
        ops.cast_pad(dt, self.dimg_z, self.p8, B * S * S, self.CH, 8)          # [M][1] fp32 (== NCHW with C=1) -> [M][8]

        def c3_wgrad(wsw):
            if self.c1_direct:                          # the activation read once (lane = channel) instead of nine per-tap GEMM passes
                ns = ops.wgrad_c1(dt, self.a2, self.dimg_z, wsw.slab, B, S, S, 64)
                ops.wgrad_reduce(wsw.slab, ns, 1, 1, 64, 9, gof("conv_blocks.9.weight"))
                return
            ns = ops.conv_wgrad(self.c3.c, dt, self.a2, self.p8, wsw.slab, wsw.wgs_target)
            ops.wgrad_reduce(wsw.slab, ns, 8, self.CH, 64, 9, gof("conv_blocks.9.weight"))
        wgrad_side(c3_wgrad, 0)
        ops.conv_bwd_data(self.c3.c, dt, self.p8, self.c3.wp_bwd, self.da2, None)
        flush()

        def bn_bwd(z, da, dz, mod, i, M, C, act, name):
            bn_train_backward(dt, z, da, dz, M, C, mod, self.mean[i], self.invstd[i], act, SLOPE, gof(name + ".weight"), gof(name + ".bias"), ws,
                              sync, self.sync_scratch.sums[i])
        # conv2 (128 -> 64 on the 2x-upsampled a1)
        bn_bwd(self.z2, self.da2, self.dz2, cb[7], 2, B * S * S, 64, ACT_LRELU, "conv_blocks.7")

        def c2_wgrad(wsw):
            if self.up3:
                ns = ops.conv_wgrad(self.c2.c, dt, self.dz2, self.a1, wsw.slab, 0)     # (one split count on a lane and on the main stream: same bits)
                ops.wgrad_reduce(wsw.slab, ns, 128, 128, 64, 16, self.dw4[1], accumulate=False)
                ops.up3_contract(self.dw4[1], gof("conv_blocks.6.weight"), 64, 128)
            else:
                ns = ops.conv_wgrad(self.c2.c, dt, self.a1, self.dz2, wsw.slab, wsw.wgs_target)
                ops.wgrad_reduce(wsw.slab, ns, 64, 64, 128, 9, gof("conv_blocks.6.weight"))
            ops.bias_grad(dt, self.dz2, B * S * S, 64, wsw.small, gof("conv_blocks.6.bias"))
        wgrad_side(c2_wgrad, 1)
        if self.up3:
            ops.conv_fwd(self.c2.c, dt, self.dz2, self.c2.wp_fwd, self.da1, None)      # the input gradient at the low resolution: no sum-pool
            flush()
        else:
            ops.conv_bwd_data(self.c2.c, dt, self.dz2, self.c2.wp_bwd, self.dup, None)
            flush()
            ops.sumpool2x2(dt, self.dup, self.da1, B, 2 * s, 2 * s, 128)
        # conv1 (128 -> 128 on the 2x-upsampled a0)
        bn_bwd(self.z1, self.da1, self.dz1, cb[3], 1, B * 4 * s * s, 128, ACT_LRELU, "conv_blocks.3")

        def c1_wgrad(wsw):
            if self.up3:
                ns = ops.conv_wgrad(self.c1.c, dt, self.dz1, self.a0, wsw.slab, 0)
                ops.wgrad_reduce(wsw.slab, ns, 128, 128, 128, 16, self.dw4[0], accumulate=False)
                ops.up3_contract(self.dw4[0], gof("conv_blocks.2.weight"), 128, 128)
            else:
                ns = ops.conv_wgrad(self.c1.c, dt, self.a0, self.dz1, wsw.slab, wsw.wgs_target)
                ops.wgrad_reduce(wsw.slab, ns, 128, 128, 128, 9, gof("conv_blocks.2.weight"))
            ops.bias_grad(dt, self.dz1, B * 4 * s * s, 128, wsw.small, gof("conv_blocks.2.bias"))
        wgrad_side(c1_wgrad, 2)
        if self.up3:
            ops.conv_fwd(self.c1.c, dt, self.dz1, self.c1.wp_fwd, self.da0, None)
            flush()
        else:
            ops.conv_bwd_data(self.c1.c, dt, self.dz1, self.c1.wp_bwd, self.dup, None)     # [B,16,16,128] in the front of dup
            flush()
            ops.sumpool2x2(dt, self.dup, self.da0, B, s, s, 128)
        bn_bwd(self.h, self.da0, self.dh, cb[0], 0, B * s * s, 128, ACT_NONE, "conv_blocks.0")
        # l1: dW[f][k] = sum_b dh[b][n'(f)] * x[b][k]
        hw = s * s

        def l1_wgrad(wsw):
            ns = ops.conv_wgrad(self.l1.c, dt, self.inp, self.dh, wsw.slab, wsw.wgs_target)
            ops.wgrad_reduce_perm(wsw.slab, ns, self.nl1, self.nl1, self.cpad, 1, gof("l1.0.weight"), 128, hw, self.cin)
            ops.fill_f32(self.gb_perm)
            ops.bias_grad(dt, self.dh, B, self.nl1, wsw.small, self.gb_perm)
            ops.gather_add(gof("l1.0.bias"), self.gb_perm, self.nl1, hw, 1, 128)
        wgrad_side(l1_wgrad, 3)
        flush()


class Generator(_HipModule):
    """Drop-in for MNIST/EAD-GAN_rpqmnxy.py:71-98."""

    def __init__(self, latent_dim=None, code_dim=None, n_classes=None, img_size=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.latent_dim, self.code_dim, self.n_classes = g(latent_dim, "latent_dim"), g(code_dim, "code_dim"), g(n_classes, "n_classes")
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        if self.img_size != 32:
            raise ValueError("the MNIST networks need 32x32 inputs (img_size // 2**4 must equal the 2x2 trunk output, SURVEY 0.3)")
        self.input_dim = self.latent_dim + self.n_classes + self.code_dim
        self.init_size = self.img_size // 4
        self.l1 = nn.Sequential(nn.Linear(self.input_dim, 128 * self.init_size ** 2))
        self.conv_blocks = nn.Sequential(
            nn.BatchNorm2d(128), nn.Upsample(scale_factor=2), nn.Conv2d(128, 128, 3, stride=1, padding=1), nn.BatchNorm2d(128, 0.8),
            nn.LeakyReLU(SLOPE, inplace=True), nn.Upsample(scale_factor=2), nn.Conv2d(128, 64, 3, stride=1, padding=1), nn.BatchNorm2d(64, 0.8),
            nn.LeakyReLU(SLOPE, inplace=True), nn.Conv2d(64, self.channels, 3, stride=1, padding=1), nn.Tanh())
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
        if not self.training:       # inference (MNIST/generate_image.py): running-stat BatchNorm, no autograd graph
            with torch.no_grad():
                return eng.forward(noise.float().contiguous(), labels.float().contiguous(), code.float().contiguous(), training=False).clone()
        return _GenFn.apply(eng, noise.float().contiguous(), labels.float().contiguous(), code.float().contiguous(), *list(self.parameters()))


class _GenFn(torch.autograd.Function):
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
# Discriminator / Encoder
# ================================================================================================
class _TrunkModule(_HipModule):
    NT = 3

    def _build_trunk(self, channels, bn):
        layers, cin = [], channels
        for i, c in enumerate(TRUNK):
            layers += [spectral_norm(nn.Conv2d(cin, c, 3, 2, 1)), nn.LeakyReLU(SLOPE, inplace=True)]
            if bn and i > 0:
                layers.append(nn.BatchNorm2d(c, 0.8))
            cin = c
        return nn.Sequential(*layers)

    def _trunk_parts(self):
        convs, names, bns, bnames = [], [], [], []
        mods = list(self.conv_blocks)
        for i, m in enumerate(mods):
            if isinstance(m, nn.Conv2d):
                convs.append(m)
                names.append(f"conv_blocks.{i}")
                nxt = mods[i + 2] if i + 2 < len(mods) and isinstance(mods[i + 2], nn.BatchNorm2d) else None
                bns.append(nxt)
                bnames.append(f"conv_blocks.{i + 2}" if nxt is not None else None)
        return convs, names, bns, bnames

    def engine(self, B) -> TrunkEngine:
        self.arena
        key = (B, self.compute_dtype)
        if key not in self._engines:
            convs, names, bns, bnames = self._trunk_parts()
            self._engines[key] = TrunkEngine(self, convs, names, bns, bnames, self._heads(), self.channels, self.img_size, 3, SLOPE, B,
                                             self.compute_dtype, self.NT)
        return self._engines[key]


class Discriminator(_TrunkModule):
    """Drop-in for :101-134 -- LSGAN critic: 4 x (SN-Conv3x3 s2 + LeakyReLU .2) -> SN-Linear(512,1), no sigmoid."""

    def __init__(self, img_size=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        self.conv_blocks = self._build_trunk(self.channels, bn=False)
        ds = self.img_size // 2 ** 4
        self.adv_layer = nn.Sequential(spectral_norm(nn.Linear(128 * ds ** 2, 1)))
        self._init_engine_state(dtype)
        self._next_tape = 0

    def _heads(self):
        return [Head("adv_layer.0", self.adv_layer[0], sn=True)]

    def forward(self, img):
        _require_cuda(img)
        eng = self.fresh_engine(img.shape[0])
        t = self._next_tape
        self._next_tape = (t + 1) % self.NT
        (out,) = _TrunkFn.apply(eng, t, self.training, ("adv_layer.0",), img.float().contiguous(), *list(self.parameters()))
        return out


class Encoder(_TrunkModule):
    """Drop-in for :137-175 -- trunk blocks are SN-Conv -> LeakyReLU -> BatchNorm(eps .8) (no BN on the first), three SN-Linear
    heads; returns (softmax(label logits), latent_code, noise).  The noise head never receives a gradient in the training
    loop (its output is discarded, :425-432) but its spectral-norm buffers advance on every forward."""

    def __init__(self, latent_dim=None, code_dim=None, n_classes=None, img_size=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.latent_dim, self.code_dim, self.n_classes = g(latent_dim, "latent_dim"), g(code_dim, "code_dim"), g(n_classes, "n_classes")
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        self.conv_blocks = self._build_trunk(self.channels, bn=True)
        ds = self.img_size // 2 ** 4
        self.aux_layer = nn.Sequential(spectral_norm(nn.Linear(128 * ds ** 2, self.n_classes)), nn.Softmax(dim=1))
        self.latent_layer = nn.Sequential(spectral_norm(nn.Linear(128 * ds ** 2, self.code_dim)))
        self.noise_layer = nn.Sequential(spectral_norm(nn.Linear(128 * ds ** 2, self.latent_dim)))
        self._init_engine_state(dtype)
        self._next_tape = 0

    def _heads(self):
        return [Head("aux_layer.0", self.aux_layer[0], sn=True), Head("latent_layer.0", self.latent_layer[0], sn=True),
                Head("noise_layer.0", self.noise_layer[0], sn=True, compute=False, grad=False)]

    def forward(self, img):
        _require_cuda(img)
        eng = self.fresh_engine(img.shape[0])
        t = self._next_tape
        self._next_tape = (t + 1) % self.NT
        logits, latent = _TrunkFn.apply(eng, t, self.training, ("aux_layer.0", "latent_layer.0"), img.float().contiguous(), *list(self.parameters()))
        # the reference also returns the noise head; it is unused by every caller in the hot path, so it is computed on demand only
        return torch.softmax(logits, dim=1), latent, None


class _TrunkFn(torch.autograd.Function):
    """autograd bridge for the eager drop-in path (one tape per call)."""

    @staticmethod
    def forward(ctx, eng, t, training, names, img, *params):
        ctx.eng, ctx.t, ctx.names = eng, t, names
        ctx.need_w = any(p.requires_grad for p in params)
        ctx.need_img = img.requires_grad
        outs = eng.forward([img], t, training)
        return tuple(outs[n].clone() for n in names)

    @staticmethod
    def backward(ctx, *douts):
        eng = ctx.eng
        scratch = torch.zeros_like(eng.owner.arena.grad)
        d = {n: (g.contiguous() if g is not None else torch.zeros_like(eng.outs[n][:eng.B])) for n, g in zip(ctx.names, douts)}
        dimg = eng.backward(ctx.t, 1, d, scratch, need_wgrad=ctx.need_w, need_dimg=ctx.need_img)
        grads = [scratch[off:off + k].view(p.shape) for p, (off, k) in zip(eng.owner.parameters(), eng.owner.arena.slices.values())]
        return (None, None, None, None, dimg.clone() if dimg is not None else None, *grads)


# ================================================================================================
# affine utilities (MNIST/utils_rpqmnxy.py)
# ================================================================================================
_APPROX = {}


def load_approximator(state_dict, device="cuda"):
    """Install the frozen ``Affine_classifier`` weights (``rpqmnxy_approximator.pt``, produced by the reference's
    approximate_rpqmnxy.py) as the device blob the regulariser kernel reads: W1 b1 ... W5 b5 then W2^T W3^T W4^T."""
    f = lambda k: state_dict[k].detach().to(device=device, dtype=torch.float32).contiguous()
    parts = []
    for i in range(5):
        parts += [f(f"fc_block.{2 * i}.weight").reshape(-1), f(f"fc_block.{2 * i}.bias").reshape(-1)]
    parts.append(torch.zeros(1, device=device))                        # b5 is padded to 8 floats
    parts += [f(f"fc_block.{2 * i}.weight").t().contiguous().reshape(-1) for i in (1, 2, 3)]
    blob = torch.cat(parts).contiguous()
    assert blob.numel() == ops.mlp_rpqmnxy_floats(), (blob.numel(), ops.mlp_rpqmnxy_floats())
    _APPROX[torch.device(device).type] = blob
    return blob


def _approx(dev):
    if dev.type not in _APPROX:
        raise RuntimeError("affine_regularizer needs the frozen approximator: call mnist.load_approximator(state_dict) first "
                           "(the reference loads rpqmnxy_approximator.pt at import, MNIST/utils_rpqmnxy.py:36-43)")
    return _APPROX[dev.type]


def get_matrix(code_input_raw):
    """[B,7] codes -> [B,3,3] = R(theta) Z(p,q) S(m,n) T(x,y)  (utils_rpqmnxy.py:87-114)."""
    _require_cuda(code_input_raw)
    c = code_input_raw.float().contiguous()
    B = c.shape[0]
    theta = torch.empty(B, 2, 3, device=c.device, dtype=torch.float32)
    ops.theta_rpqmnxy(c, c.shape[1], B, theta)
    A = torch.zeros(B, 3, 3, device=c.device, dtype=torch.float32)
    A[:, :2] = theta
    A[:, 2, 2] = 1.0
    return A


class _AffineRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real_code, trans_code):
        B, ld = real_code.shape
        dev = real_code.device
        pred = torch.empty(B, 7, device=dev)
        ops.loss_affine_rpqmnxy(real_code, trans_code, ld, 0, B, torch.zeros(B, 7, device=dev), 7, _approx(dev), 1.0, None, None, None, pred,
                                torch.empty(B, device=dev))
        ctx.save_for_backward(real_code, trans_code, pred)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        real_code, trans_code, pred = ctx.saved_tensors
        B, ld = real_code.shape
        dev = real_code.device
        tgt = (pred - dpred.float() * (7.0 * B / 2.0)).contiguous()       # MSE gradient with this target == J^T dpred
        d_real, d_trans = torch.empty_like(real_code), torch.empty_like(trans_code)
        ops.loss_affine_rpqmnxy(real_code, trans_code, ld, 0, B, tgt, 7, _approx(dev), 1.0, None, d_real, d_trans, None, torch.empty(B, device=dev))
        return d_real, d_trans


def affine_regularizer(real_code, trans_code):
    """relative transform -> frozen MLP -> latent units (utils_rpqmnxy.py:117-134)."""
    _require_cuda(real_code)
    return _AffineRegFn.apply(real_code.float().contiguous(), trans_code.float().contiguous())


# ================================================================================================
# fused train-loop entry
# ================================================================================================
class MnistTrainer(ResidentStep):
    """One call of :meth:`train_step` == one iteration of MNIST/EAD-GAN_rpqmnxy.py:340-446: LSGAN G step, D step (lr x2),
    info+affine step over G+E (lambda_cat 1, lambda_con .1, lambda_affine .1, :201-203), three Adams (:249-255)."""

    def __init__(self, generator, discriminator, encoder, batch_size, dtype="f32", allreduce=None, lr=1e-4, betas=(0.5, 0.999),
                 lambda_cat=1.0, lambda_con=0.1, lambda_affine=0.1, lrs=None, overlap=False, sync_bn=None):
        """``sync_bn`` (a dp.SyncBN): the generator's BatchNorm layers use the statistics of the global batch (N ranks == 1 rank at equal
        global batch for the generator); BatchNorm layers of the other networks stay per-rank, as under torch DDP without SyncBatchNorm."""
        self.G, self.D, self.E, self.B = generator, discriminator, encoder, batch_size
        self.sync_bn = sync_bn
        dt = parse_dtype(dtype)
        for m in (generator, discriminator, encoder):
            m.set_compute_dtype(dt)
        self.ge, self.de, self.ee = generator.engine(batch_size), discriminator.engine(batch_size), encoder.engine(batch_size)
        dev = generator.arena.flat.device
        self.allreduce = allreduce
        self.lr = lrs or (lr, 2 * lr, lr)
        self.betas = betas
        self.lam = (lambda_cat, lambda_con, lambda_affine)
        ga, da, ea = generator.arena, discriminator.arena, encoder.arena
        z = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)
        self.mG, self.vG, self.mD, self.vD = z(ga.numel), z(ga.numel), z(da.numel), z(da.numel)
        self.miG, self.viG, self.miE, self.viE = z(ga.numel), z(ga.numel), z(ea.numel), z(ea.numel)
        self.steps = torch.zeros(3, device=dev, dtype=torch.int32)
        self.losses = torch.zeros(4, device=dev, dtype=torch.float32)
        B, C, S = batch_size, generator.channels, generator.img_size
        self.theta = torch.empty(B, 2, 3, device=dev)
        self.scaled = torch.empty(B, C, S, S, device=dev)
        self.dout_d = torch.zeros(2 * B, 1, device=dev)
        self.d_cat = torch.zeros(3 * B, generator.n_classes, device=dev)
        self.d_code = torch.zeros(3 * B, generator.code_dim, device=dev)
        self.ws_aff = torch.empty(B, device=dev)
        self.real = torch.empty(B, C, S, S, device=dev)
        self.z = torch.empty(B, generator.latent_dim, device=dev)
        self.code = torch.empty(B, generator.code_dim, device=dev)
        self.onehot = torch.empty(B, generator.n_classes, device=dev)
        self.labels = torch.empty(B, device=dev, dtype=torch.int64)
        self.mlp = _approx(dev)
        self.graph = None
        self.dev = dev
        # overlap=True: the generator's weight-gradient chains on side lanes beside its backward-data chain (engine.SideStream).  Bit-identical,
        # but measured SLOWER here (fp32 B=256 7.33 -> 7.60 ms, bf16 B=128 3.32 -> 3.64 ms): this backward is almost all GEMM, so forked chains
        # only compete with the main chain and add queue hops -- unlike the CelebA step, whose small-kernel phases they fill.  Default off.
        self.side = SideStream(dev, Workspace.get(dev), lanes=4) if overlap else None

    def _adam(self, arena, m, v, lr, slot, tick):
        ops.adam_step(arena.flat, arena.grad, m, v, arena.numel, lr, self.betas[0], self.betas[1], 1e-8, self.steps[slot:slot + 1], tick)

    def import_adam_state(self, opt_G, opt_D, opt_info):
        """moments and step counts of three ``torch.optim.Adam`` built like the reference's (MNIST/EAD-GAN_rpqmnxy.py:206-217: G | D | G + E
        parameters, ``.parameters()`` order) -- teacher-forced comparisons against a CPU run of the reference loop"""
        from .engine import import_adam_moments
        n = lambda mod: len(list(mod.parameters()))
        s0 = import_adam_moments(opt_G, [(n(self.G), self.mG, self.vG)])
        s1 = import_adam_moments(opt_D, [(n(self.D), self.mD, self.vD)])
        s2 = import_adam_moments(opt_info, [(n(self.G), self.miG, self.viG), (n(self.E), self.miE, self.viE)])
        self.steps.copy_(torch.tensor([s0, s1, s2], dtype=torch.int32))

    def _step_body(self):
        G, D, E, ge, de, ee, B = self.G, self.D, self.E, self.ge, self.de, self.ee, self.B
        ga, da, ea = G.arena, D.arena, E.arena
        cd, nc = G.code_dim, G.n_classes
        lcat, lcon, laff = self.lam
        ops.fill_f32(self.losses)
        ops.theta_rpqmnxy(self.code, cd, B, self.theta)                                  # :365
        ops.warp_affine(self.real, self.theta, self.scaled, B, G.channels, G.img_size, G.img_size)   # :367
        # ---- 1) generator step (:375-388), LSGAN: MSE(validity, 1) ----
        ops.fill_f32(ga.grad)
        gen = ge.forward(self.z, self.onehot, self.code, sync=self.sync_bn)
        out = de.forward([gen])["adv_layer.0"]
        ops.loss_mse(out, 1, 0, 1, B, None, 0, 1.0, 1.0, self.losses[0:1], self.dout_d[:B])
        dimg = de.backward(0, 1, {"adv_layer.0": self.dout_d[:B]}, da.grad, need_wgrad=False, need_dimg=True)
        side = self.side
        join = side.join if side is not None else (lambda: None)
        ge.backward(dimg, ga.grad, side, sync=self.sync_bn)
        join()
        if self.allreduce is not None:
            self.allreduce(ga.grad)
        self._adam(ga, self.mG, self.vG, self.lr[0], 0, True)
        ge.repack()
        # ---- 2) discriminator step (:395-409): D(scaled) then D(gen.detach()), batched ----
        ops.fill_f32(da.grad)
        out = de.forward([self.scaled, gen])["adv_layer.0"]
        ops.loss_mse(out[:B], 1, 0, 1, B, None, 0, 1.0, 0.5, self.losses[1:2], self.dout_d[:B])
        ops.loss_mse(out[B:], 1, 0, 1, B, None, 0, 0.0, 0.5, self.losses[1:2], self.dout_d[B:])
        de.backward(0, 2, {"adv_layer.0": self.dout_d}, da.grad)
        if self.allreduce is not None:
            self.allreduce(da.grad)
        self._adam(da, self.mD, self.vD, self.lr[1], 1, True)
        de.repack()
        # ---- 3) info + affine step (:415-446): E(gen), E(scaled), E(real) (BatchNorm -> one tape per forward) ----
        ops.fill_f32(ga.grad)
        ops.fill_f32(ea.grad)
        gen = ge.forward(self.z, self.onehot, self.code, sync=self.sync_bn)
        outs = ee.forward([gen, self.scaled, self.real])
        cat, lat = outs["aux_layer.0"], outs["latent_layer.0"]
        ops.fill_f32(self.d_cat)
        ops.loss_ce_softmaxed(cat[:B], nc, 0, nc, B, self.labels, lcat, self.losses[2:3], self.d_cat[:B])
        ops.loss_mse(lat[:B], cd, 0, cd, B, self.code, cd, 0.0, lcon, self.losses[2:3], self.d_code[:B])
        ops.loss_affine_rpqmnxy(lat[2 * B:], lat[B:2 * B], cd, 0, B, self.code, cd, self.mlp, laff, self.losses[2:3], self.d_code[2 * B:],
                                self.d_code[B:2 * B], None, self.ws_aff)
        dimg = ee.backward(0, 3, {"aux_layer.0": self.d_cat, "latent_layer.0": self.d_code}, ea.grad, need_dimg=True)
        pending = self.allreduce.start(ea.grad) if (self.allreduce is not None and hasattr(self.allreduce, "start")) else None
        ge.backward(dimg, ga.grad, side, sync=self.sync_bn)    # overlaps with the encoder-gradient all-reduce
        join()
        if self.allreduce is not None:
            self.allreduce(ga.grad)
            if pending is not None:
                self.allreduce.finish(pending)
            elif not hasattr(self.allreduce, "start"):
                self.allreduce(ea.grad)
        self._adam(ga, self.miG, self.viG, self.lr[2], 2, True)
        self._adam(ea, self.miE, self.viE, self.lr[2], 2, False)
        ge.repack()
        ee.repack()

    def load_inputs(self, real_imgs, z, code, labels):
        self.real.copy_(real_imgs, non_blocking=True)
        self.z.copy_(z, non_blocking=True)
        self.code.copy_(code, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.onehot.zero_()
        self.onehot.scatter_(1, self.labels.view(-1, 1), 1.0)

    def train_step(self, real_imgs, z, code, labels):
        self.load_inputs(real_imgs, z, code, labels)
        l = self.step_resident().tolist()
        return {"g_loss": l[0], "d_loss": l[1], "info_loss": l[2]}


class DeviceInputs(DeviceSampler):
    """Device-side replacement of the MNIST loop's host input work (MNIST/EAD-GAN_rpqmnxy.py:233-246 DataLoader + Resize(32) + ToTensor +
    Normalize(.5,.5); :351-357 numpy draws in the reference's order: labels ~ randint(10), z ~ N(0,1), code ~ U(-1,1)).  ``dataset_u8``:
    uint8 [N,1,32,32] (resized once on the way in)."""

    def enqueue(self, tr: "MnistTrainer"):
        B = tr.B
        N, C, H, W = self.data.shape
        self.begin_draws()                              # image indices, labels (+ one-hot rows), z, code: one launch
        idx = self.sample_indices(B, 1)
        self.labels_onehot("labels", tr.onehot, tr.onehot.shape[1], 2, lab=tr.labels)
        self.draw(ops.RNG_NORMAL, tr.z, 0.0, 1.0, 3)
        self.draw(ops.RNG_UNIFORM, tr.code, -1.0, 1.0, 4)
        self.end_draws()
        # ToTensor + Normalize(.5,.5); the step counter ticks in the same launch (every draw of the iteration has read it)
        ops.gather_u8_images(self.data, idx, None, tr.real, B, C, H, W, 2.0 / 255.0, -1.0, tick=self.step if FUSE_DRAWS else None)
        if not FUSE_DRAWS:
            self.tick()


# ================================================================================================
# Fit of the affine-inverse MLP (MNIST/approximate_rpqmnxy.py): produces the rpqmnxy_approximator.pt the MNIST loop loads frozen
# ================================================================================================
class Affine_classifier(nn.Module):
    """Parameter container with the reference's keys (approximate_rpqmnxy.py:12-31 == utils_rpqmnxy.py:12-34): fc_block.{0,2,4,6,8}."""

    def __init__(self):
        super().__init__()
        self.fc_block = nn.Sequential(nn.Linear(6, 256), nn.LeakyReLU(), nn.Linear(256, 256), nn.LeakyReLU(), nn.Linear(256, 256), nn.LeakyReLU(),
                                      nn.Linear(256, 256), nn.LeakyReLU(), nn.Linear(256, 7))


class ApproximatorTrainer:
    """One call == one iteration of approximate_rpqmnxy.py:119-136: code ~ U(-1,1)^7 -> rows 0,1 of R Z S T (6 numbers) -> MLP
    6-256-256-256-256-7 (LeakyReLU 0.01) -> MSE vs the affine parameters -> Adam(lr 2e-4, betas (.5,.999)).  The dense layers run as
    1x1 convolutions over B "pixels" on the implicit-GEMM kernels, the 7-wide head on the dense-head kernels.  After fitting,
    ``state_dict()`` of the module is the checkpoint; ``install()`` hands it to the regulariser kernel (load_approximator)."""

    def __init__(self, mlp: Affine_classifier, batch_size=128, dtype="f32", lr=2e-4, betas=(0.5, 0.999)):
        self.mlp, self.B = mlp, batch_size
        self.dtype = dt = parse_dtype(dtype)
        p = next(mlp.parameters())
        _require_cuda(p)
        dev = p.device
        self.dev = dev
        self.arena = Arena(mlp)
        self.ws = ws = Workspace.get(dev)
        tdt = ops.torch_dtype(dt)
        B = batch_size
        self.kin = 8                                    # 6 inputs padded to the 16-byte vector width
        self.l = [ConvRec(dt, B, 1, 1, self.kin, 256, 1, 1, 0, device=dev, want_bwd=False, ws=ws)] + \
                 [ConvRec(dt, B, 1, 1, 256, 256, 1, 1, 0, device=dev, ws=ws) for _ in range(3)]
        self.head = ConvRec(dt, B, 1, 1, 256, 7, 1, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        z = lambda *n, d=torch.float32: torch.zeros(*n, device=dev, dtype=d)
        self.code, self.theta, self.para = z(B, 7), z(B, 2, 3), z(B, 7)
        self.x = z(B, self.kin, d=tdt)
        self.a = [z(B, 256, d=tdt) for _ in range(4)]
        self.dz = [z(B, 256, d=tdt) for _ in range(4)]
        self.out, self.dout = z(B, 7), z(B, 7)
        self.m, self.v = z(self.arena.numel), z(self.arena.numel)
        self.steps = torch.zeros(1, device=dev, dtype=torch.int32)
        self.losses = z(4)
        self.lr, self.betas = lr, betas
        self.graph = None
        self.repack()

    def _lin(self, i):
        return self.mlp.fc_block[2 * i]

    @ops.batched_packs
    def repack(self):
        ops.pack_strided(self.dtype, self._lin(0).weight, self.l[0].wp_fwd, 256, 6, self.l[0].Kpad_fwd, 1, 6, 0, 1)
        for i in (1, 2, 3):
            self.l[i].pack(self._lin(i).weight)
        self.head.pack(self._lin(4).weight)

    def _step_body(self):
        dt, B, ws, ar = self.dtype, self.B, self.ws, self.arena
        gof = lambda name: ar.grad_of(name)
        ops.fill_f32(self.losses)
        ops.theta_rpqmnxy(self.code, 7, B, self.theta)                 # rows 0,1 of get_matrix_rpqmnxy == cat(A[:,0], A[:,1])  (:127-128)
        ops.affine_para_rpqmnxy(self.code, 7, B, self.para)            # the target is code_input, the AFFINE PARAMETERS (:127,135)
        ops.cast_pad(dt, self.theta, self.x, B, 6, self.kin)
        x = self.x
        for i in range(4):
            ops.conv_fwd(self.l[i].c, dt, x, self.l[i].wp_fwd, self.a[i], ops.epilogue(bias=self._lin(i).bias, act=ACT_LRELU, slope=0.01))
            x = self.a[i]
        ops.dense_small_fwd(dt, x, self.head.wp_fwd, self._lin(4).bias, self.out, B, 256, self.head.Kpad_fwd, 7, ws.small)
        ops.loss_mse(self.out, 7, 0, 7, B, self.para, 7, 0.0, 1.0, self.losses[0:1], self.dout)
        ops.fill_f32(ar.grad)
        ops.dense_small_wgrad(dt, self.dout, self.a[3], gof("fc_block.8.weight"), gof("fc_block.8.bias"), B, 256, 7, 256, 1)
        ops.dense_small_bwd(dt, self.dout, self.head.wp_fwd, self.a[3], self.dz[3], B, 256, self.head.Kpad_fwd, 7, ACT_LRELU, 0.01)
        for i in (3, 2, 1, 0):
            x_in = self.a[i - 1] if i > 0 else self.x
            ops.bias_grad(dt, self.dz[i], B, 256, ws.small, gof(f"fc_block.{2 * i}.bias"))
            ns = ops.conv_wgrad(self.l[i].c, dt, x_in, self.dz[i], ws.slab)
            if i > 0:
                ops.wgrad_reduce(ws.slab, ns, 256, 256, 256, 1, gof(f"fc_block.{2 * i}.weight"))
                ops.conv_bwd_data(self.l[i].c, dt, self.dz[i], self.l[i].wp_bwd, self.dz[i - 1], ops.epilogue(mask=self.a[i - 1], mask_act=ACT_LRELU, mask_slope=0.01))
            else:
                ops.wgrad_reduce_perm(ws.slab, ns, 256, 256, self.kin, 1, gof("fc_block.0.weight"), 0, 0, 6)
        ops.adam_step(ar.flat, ar.grad, self.m, self.v, ar.numel, self.lr, self.betas[0], self.betas[1], 1e-8, self.steps[0:1], True)
        self.repack()

    def capture(self, warmup=False):
        if warmup:
            self._step_body()
        return capture_step(self, self._step_body)

    def step_resident(self):
        check_usable(self)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step_body()
        return self.losses

    def train_step(self, code):
        """code [B,7] ~ U(-1,1) -> {'affine_loss'}"""
        self.code.copy_(code, non_blocking=True)
        return {"affine_loss": float(self.step_resident()[0])}

    def install(self):
        """hand the fitted weights to the regulariser kernel (what loading rpqmnxy_approximator.pt does in the reference)"""
        return load_approximator(self.mlp.state_dict(), device=self.dev)
