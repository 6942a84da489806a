"""dSprites 64x64 stage-2 path of EAD-GAN on MI355X: drop-in ``Encoder_pxy`` / ``Discriminator`` / ``Generator`` / ``Encoder`` /
``transformation_2D`` / ``mutual_info_loss`` / ``to_categorical`` (dSprites/rp.py:61-232), ``get_matrix`` / ``get_matrix_D`` /
``affine_regularzier`` (dSprites/utils_rp.py), ``get_matrix_pxy_align`` (dSprites/utils_pxy.py:69-87) and the fused train-loop
entry :class:`DspritesTrainer` (loop body rp.py:365-482).  torch modules are parameter containers only.
"""
from __future__ import annotations

import os
import argparse

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops
from .celeba import FUSE_STATS, IMG_GEMM, _HipModule, _require_cuda, transformation_2D      # noqa: F401
from .engine import FUSE_DRAWS, Arena, ConvRec, DeviceSampler, ResidentStep, SideStream, SyncScratch, Workspace, bn_train_backward, bn_train_forward, capture_step, check_usable, parse_dtype
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, EG_F32, OUT_NCHW_F32
from .trunk import IMG_DIRECT, WGRAD_IMG, Head, TrunkEngine

opt = argparse.Namespace(n_epochs=100, batch_size=128, lr=0.0001, b1=0.5, b2=0.999, n_cpu=8, latent_dim=200, code_dim=4, n_classes=3,
                         img_size=64, channels=1, sample_interval=1000)           # argparse defaults rp.py:40-51
TRUNK = (32, 32, 64, 64)
# EXPERIMENT (default off): the generator's last-layer backward straight from the image gradient -- eg_wgrad_img (N = 64) for the weight
# gradient, eg_conv_img_mfma (N = 64) with the BatchNorm-backward sums in its epilogue for the input gradient -- instead of patch rows + the
# two GEMMs over them.  Same results within fp32 summation order (tests), one launch fewer, but SLOWER in the step: dSprites 1.305 -> 1.32 ms,
# colored 2.50 -> 2.53 (profiles/r03_zzo_ab_l4_direct.txt): the statistics instantiation of the image kernel runs two workgroups per CU
L4_DIRECT = os.environ.get("EG_L4_DIRECT", "0") != "0"
# EXPERIMENT (default off): two-chain step with the alignment pass behind the first generator forward (beside the second chain's generator
# forward) instead of in front of both chains.  Same kernels on the same operands, and the node graph is shorter by ~7 launches -- but the
# replayed step is far SLOWER, dSprites 1.31 -> 1.73 ms, colored 2.54 -> 3.05 (profiles/r03_zzh_ab_align_late.txt): the second chain then
# forks from a point with main-chain work queued behind it, and hipGraph's stream-to-queue placement (profiles/r01_timeline_notes.md) runs
# the two chains one after the other
ALIGN_LATE = os.environ.get("EG_ALIGN_LATE", "0") != "0"


def to_categorical(y, num_columns, device=None):
    y = torch.as_tensor(np.asarray(y), dtype=torch.int64, device=device)
    return torch.nn.functional.one_hot(y, num_columns).to(torch.float32)


def _trunk(ch, sn, slope):
    f = spectral_norm if sn else (lambda m: m)
    return nn.Sequential(f(nn.Conv2d(ch, 32, 4, 2, 1)), nn.LeakyReLU(slope, inplace=True), f(nn.Conv2d(32, 32, 4, 2, 1)), nn.LeakyReLU(slope, inplace=True),
                         f(nn.Conv2d(32, 64, 4, 2, 1)), nn.LeakyReLU(slope, inplace=True), f(nn.Conv2d(64, 64, 4, 2, 1)), nn.LeakyReLU(slope, inplace=True))


# ================================================================================================
# Encoder_pxy: frozen alignment encoder (forward only; its gradients are dead work in the reference)
# ================================================================================================
class _PxyEngine:
    def __init__(self, mod: "Encoder_pxy", B, dtype):
        self.mod, self.B, self.dtype = mod, B, dtype
        dev = next(mod.parameters()).device
        tdt = ops.torch_dtype(dtype)
        self.ws = ws = Workspace.get(dev)
        C, S = mod.channels, mod.img_size
        self.C, self.S = C, S
        self.k0 = C * 16
        self.kp = ops.round_up(self.k0, 8)
        self.l0 = ConvRec(dtype, B, S // 2, S // 2, self.kp, TRUNK[0], 1, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.mid = [ConvRec(dtype, B, S >> (i + 1), S >> (i + 1), TRUNK[i], TRUNK[i + 1], 4, 2, 1, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
                    for i in range(3)]
        self.nout = mod.fc1.weight.shape[0]
        self.head = ConvRec(dtype, B, 4, 4, TRUNK[3], self.nout, 4, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.patches = torch.zeros(B * (S // 2) ** 2, self.kp, device=dev, dtype=tdt)
        self.a = [torch.empty(B, S >> (i + 1), S >> (i + 1), TRUNK[i], device=dev, dtype=tdt) for i in range(4)]
        self.out = torch.empty(B, self.nout, device=dev, dtype=torch.float32)
        self.repack()

    @ops.batched_packs
    def repack(self):
        cb = self.mod.conv_block
        ops.pack_strided(self.dtype, cb[0].weight, self.l0.wp_fwd, TRUNK[0], self.k0, self.l0.Kpad_fwd, 1, self.k0, 0, 1)
        for i in range(3):
            self.mid[i].pack(cb[2 * (i + 1)].weight)
        self.head.pack(self.mod.fc1.weight)          # Linear over the NCHW-flattened 4x4x64 map == 4x4 conv

    def forward(self, img):
        dt, B, cb = self.dtype, self.B, self.mod.conv_block
        ep0 = ops.epilogue(bias=cb[0].bias, act=ACT_LRELU, slope=0.1)
        if IMG_DIRECT and self.l0.Kpad_fwd == 64 and ops.conv_img_mfma_ok(dt, self.C, self.S, self.S, TRUNK[0], 4, 2, 1):
            ops.conv_img_mfma(dt, [img], self.l0.wp_fwd, self.a[0], B, self.C, self.S, self.S, ep0, N=TRUNK[0])      # no patch rows (frozen: no weight gradient)
        else:
            ops.im2col_img(dt, img, self.patches, B, self.C, self.S, self.S, 4, 2, 1, self.kp)
            ops.conv_fwd(self.l0.c, dt, self.patches, self.l0.wp_fwd, self.a[0], ep0)
        for i in range(3):
            ops.conv_fwd(self.mid[i].c, dt, self.a[i], self.mid[i].wp_fwd, self.a[i + 1], ops.epilogue(bias=cb[2 * (i + 1)].bias, act=ACT_LRELU, slope=0.1))
        ops.dense_small_fwd(dt, self.a[3], self.head.wp_fwd, self.mod.fc1.bias, self.out, B, 16 * TRUNK[3], self.head.Kpad_fwd, self.nout, self.ws.small)
        return self.out


class Encoder_pxy(_HipModule):
    """Drop-in for rp.py:61-87 (inference only on this path: the reference loads ``encoder_pxy_50000.pt`` and freezes it)."""

    def __init__(self, img_size=None, channels=None, n_out=3, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        self.conv_block = _trunk(self.channels, False, 0.1)
        self.fc1 = nn.Linear(1024, n_out)
        self._init_engine_state(dtype)

    @property
    def arena(self):                 # frozen: parameters stay where they are, no gradient arena
        return None

    def engine(self, B) -> _PxyEngine:
        key = (B, self.compute_dtype)
        if key not in self._engines:
            _require_cuda(next(self.parameters()))
            self._engines[key] = _PxyEngine(self, B, self.compute_dtype)
        return self._engines[key]

    def forward(self, img):
        _require_cuda(img)
        return self.fresh_engine(img.shape[0]).forward(img.float().contiguous()).clone()


# ================================================================================================
# Discriminator / Encoder (spectrally-normalised trunks with hidden FC layers)
# ================================================================================================
class _TrunkModule(_HipModule):
    NT = 3

    def _convs(self):
        cb = self.conv_block
        return [cb[i] for i in (0, 2, 4, 6)], [f"conv_block.{i}" for i in (0, 2, 4, 6)]

    def engine(self, B, slot=0) -> TrunkEngine:
        self.arena
        key = (B, self.compute_dtype) if slot == 0 else (B, self.compute_dtype, slot)
        if key not in self._engines:
            convs, names = self._convs()
            fcs, fnames = self._fcs()
            self._engines[key] = TrunkEngine(self, convs, names, [None] * 4, [None] * 4, self._heads(), self.channels, self.img_size, 4, 0.2, B,
                                             self.compute_dtype, self.NT, fcs=fcs, fc_names=fnames)
        return self._engines[key]


class Discriminator(_TrunkModule):
    """Drop-in for rp.py:90-119: SN trunk -> SN-Linear(1024,128)+LeakyReLU -> Linear(128,1) -> sigmoid."""

    def __init__(self, img_size=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        self.conv_block = _trunk(self.channels, True, 0.2)
        self.fc1 = nn.Sequential(spectral_norm(nn.Linear(1024, 128)), nn.LeakyReLU(0.2, inplace=True))
        self.fc2 = nn.Linear(128, 1)
        self._init_engine_state(dtype)
        self._next_tape = 0

    def _fcs(self):
        return [self.fc1[0]], ["fc1.0"]

    def _heads(self):
        return [Head("fc2", self.fc2, sn=False)]

    def forward(self, img):
        from .mnist import _TrunkFn
        _require_cuda(img)
        eng = self.fresh_engine(img.shape[0])
        t = self._next_tape
        self._next_tape = (t + 1) % self.NT
        (logit,) = _TrunkFn.apply(eng, t, self.training, ("fc2",), img.float().contiguous(), *list(self.parameters()))
        return torch.sigmoid(logit)


class Encoder(_TrunkModule):
    """Drop-in for rp.py:160-194: SN trunk -> 2 x (SN-Linear + LeakyReLU) -> softmax(SN-Linear(128,n_classes)), SN-Linear(128,code_dim)."""

    def __init__(self, code_dim=None, n_classes=None, img_size=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.code_dim, self.n_classes = g(code_dim, "code_dim"), g(n_classes, "n_classes")
        self.img_size, self.channels = g(img_size, "img_size"), g(channels, "channels")
        self.conv_block = _trunk(self.channels, True, 0.2)
        self.fc1 = nn.Sequential(spectral_norm(nn.Linear(1024, 128)), nn.LeakyReLU(0.2, inplace=True))
        self.fc2 = nn.Sequential(spectral_norm(nn.Linear(128, 128)), nn.LeakyReLU(0.2, inplace=True))
        self.cat_layer = nn.Sequential(spectral_norm(nn.Linear(128, self.n_classes)), nn.Softmax(dim=1))
        self.cont_layer = nn.Sequential(spectral_norm(nn.Linear(128, self.code_dim)))
        self._init_engine_state(dtype)
        self._next_tape = 0

    def _fcs(self):
        return [self.fc1[0], self.fc2[0]], ["fc1.0", "fc2.0"]

    def _heads(self):
        return [Head("cat_layer.0", self.cat_layer[0], sn=True), Head("cont_layer.0", self.cont_layer[0], sn=True)]

    def forward(self, img):
        from .mnist import _TrunkFn
        _require_cuda(img)
        eng = self.fresh_engine(img.shape[0])
        t = self._next_tape
        self._next_tape = (t + 1) % self.NT
        cat, cont = _TrunkFn.apply(eng, t, self.training, ("cat_layer.0", "cont_layer.0"), img.float().contiguous(), *list(self.parameters()))
        return torch.softmax(cat, dim=1), cont


# ================================================================================================
# Generator: Linear(7,128)+ReLU -> Linear(128,1024)+ReLU -> view[B,64,4,4] -> 3 x (ConvT 4x4 s2 + BN + ReLU) -> ConvT(64,1) -> sigmoid
# ================================================================================================
class _GenEngine:
    def __init__(self, gen: "Generator", B, dtype):
        self.gen, self.B, self.dtype = gen, B, dtype
        dev = gen.arena.flat.device
        tdt = ops.torch_dtype(dtype)
        self.ws = ws = Workspace.get(dev)
        self._stat = {}                                 # fused-statistics buffers per launch (celeba._GenEngine._stat_buf)
        self.cin = gen.n_classes + gen.code_dim
        self.cpad = ops.round_up(self.cin, 8)
        self.CH = gen.channels
        self.f1 = ConvRec(dtype, B, 1, 1, self.cpad, 128, 1, 1, 0, device=dev, want_bwd=False, ws=ws)
        self.f2 = ConvRec(dtype, B, 1, 1, 128, 1024, 1, 1, 0, device=dev, ws=ws)
        self.mid = [ConvRec(dtype, B, 8 << i, 8 << i, 64, 64, 4, 2, 1, device=dev, ws=ws) for i in range(3)]
        self.l4 = ConvRec(dtype, B, 64, 64, self.CH, 64, 4, 2, 1, device=dev, want_fwd=False, want_wgrad=False, ws=ws)
        self.k0 = self.CH * 16
        self.kp = ops.round_up(self.k0, 8)
        self.l4p = ConvRec(dtype, B, 32, 32, self.kp, 64, 1, 1, 0, device=dev, want_bwd=False, ws=ws)
        # forward of the last ConvTranspose2d(64 -> C) as ONE GEMM over the 32x32 lattice with N = 16 taps x C columns + the col2im gather
        # (eg_col2im_img): every activation read once instead of 16 times (see celeba._GenEngine.l4g)
        self.l4g = ConvRec(dtype, B, 32, 32, 64, self.k0, 1, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.l4_direct = (IMG_DIRECT and WGRAD_IMG and L4_DIRECT and self.kp == self.k0 and self.l4p.Kpad_fwd == 64
                          and ops.conv_img_mfma_ok(dtype, self.CH, 64, 64, 64, 4, 2, 1) and ops.wgrad_img_ok(dtype, self.CH, 64, 64, 64, 4, 2, 1))
        if self.l4_direct:
            ws.need_slab(ops.wgrad_img_splits(B, 64) * 64 * self.kp * 4)
        self.cols4 = torch.empty(B * 32 * 32, self.k0, device=dev, dtype=tdt)
        e = lambda *s, dt=tdt: torch.empty(s, device=dev, dtype=dt)
        f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
        self.inp = e(B, self.cpad)
        self.a1 = e(B, 128)
        self.bias2_perm, self.gb2_perm = f(1024), f(1024)
        self.h = e(B, 4, 4, 64)
        self.z = [e(B, 8 << i, 8 << i, 64) for i in range(3)]
        self.a = [torch.empty_like(t) for t in self.z]
        self.mean = [f(64) for _ in range(3)]
        self.invstd = [f(64) for _ in range(3)]
        self.sync_scratch = SyncScratch((64, 64, 64), dev)
        self.img = f(B, self.CH, 64, 64)
        self.dimg_z = torch.empty_like(self.img)
        self.patches = torch.zeros(B * 32 * 32, self.kp, device=dev, dtype=tdt)
        self.da = [torch.empty_like(t) for t in self.z]
        self.dz = [torch.empty_like(t) for t in self.z]
        self.dh = torch.empty_like(self.h)
        self.dz1 = torch.empty_like(self.a1)
        for i in range(3):
            ws.need_small(ops.bn_ws_floats(B * (8 << i) ** 2, 64))
        ws.need_small(ops.bias_grad_ws_floats(B, 1024))
        ws.need_small(B * self.CH)
        ws.need_sums(128)
        self.repack()

    def _stat_buf(self, key, c, bwd, ep, C):
        """(nrb, buffer) if the launch described by (c, bwd, ep) can take its column statistics in the epilogue, else (0, None)"""
        if key not in self._stat:
            nrb = ops.conv_stat_blocks(c, self.dtype, bwd, ep) if FUSE_STATS else 0
            self._stat[key] = (nrb, torch.empty(2 * C * nrb, device=self.inp.device, dtype=torch.float32) if nrb else None)
        return self._stat[key]

    @ops.batched_packs
    def repack(self):
        dt, g = self.dtype, self.gen
        cb = g.conv_block
        w1, w2 = g.fc1[0].weight, g.fc2[0].weight
        ops.pack_strided(dt, w1, self.f1.wp_fwd, 128, self.cin, self.f1.Kpad_fwd, 1, self.cin, 0, 1)
        # fc2 rows in NHWC order n' = hw*64 + c  <-  master row f = c*16 + hw (view [B,64,4,4])
        ops.pack_strided(dt, w2, self.f2.wp_fwd, 1024, 128, self.f2.Kpad_fwd, 64, 128, 16 * 128, 1)
        # backward panel [k][n'] = W2[f(n')][k]
        ops.pack_strided2(dt, w2, self.f2.wp_bwd, 128, 1024, ops.round_up(1024, ops.bk(dt)), 1, 1, 0, 64, 128, 16 * 128)
        ops.pack_strided(EG_F32, g.fc2[0].bias, self.bias2_perm, 1024, 1, 1, 64, 1, 16, 0)
        for i, idx in enumerate((0, 3, 6)):
            self.mid[i].pack(cb[idx].weight)
        self.l4.pack(cb[9].weight)
        ops.pack_strided(dt, cb[9].weight, self.l4p.wp_fwd, 64, self.k0, self.l4p.Kpad_fwd, 1, self.k0, 0, 1)
        ops.pack_strided(dt, cb[9].weight, self.l4g.wp_fwd, self.k0, 64, self.l4g.Kpad_fwd, self.CH, 1, 16, self.k0)     # wp[t*C + c][ci] = W[ci][c][t]

    def forward(self, labels, code, training=True, sync=None):
        """z_c = cat(one-hot labels, code)  (rp.py:404-405).  ``training=False``: running-stat BatchNorm (module.eval()).  ``sync`` (a
        dp.SyncBN): batch statistics over all ranks."""
        dt, B, g, ws = self.dtype, self.B, self.gen, self.ws
        cb = g.conv_block
        ops.concat_cast(dt, labels, code, None, self.inp, B, self.cpad)
        ops.conv_fwd(self.f1.c, dt, self.inp, self.f1.wp_fwd, self.a1, ops.epilogue(bias=g.fc1[0].bias, act=ACT_RELU))
        ops.conv_fwd(self.f2.c, dt, self.a1, self.f2.wp_fwd, self.h, ops.epilogue(bias=self.bias2_perm, act=ACT_RELU))
        x = self.h
        for i, idx in enumerate((0, 3, 6)):
            r = self.mid[i]
            bn = cb[idx + 1]
            M = B * (8 << i) ** 2
            nrb, stat = self._stat_buf(("fwd", i), r.c, True, ops.epilogue(bias=cb[idx].bias), 64) if (training and sync is None) else (0, None)
            if nrb:
                # BatchNorm batch statistics from the transposed convolution's epilogue (celeba._GenEngine.forward): two launches instead of three
                ops.conv_bwd_data(r.c, dt, x, r.wp_bwd, self.z[i], ops.epilogue(bias=cb[idx].bias, stat_mode=ops.STAT_MOMENTS, stat_out=stat))
                ops.bn_fwd_train_fused(dt, self.z[i], self.a[i], M, 64, stat, nrb, M // nrb, bn.weight, bn.bias, bn.eps, bn.momentum, bn.running_mean,
                                       bn.running_var, bn.num_batches_tracked, self.mean[i], self.invstd[i], ws.small, ACT_RELU)
                x = self.a[i]
                continue
            ops.conv_bwd_data(r.c, dt, x, r.wp_bwd, self.z[i], ops.epilogue(bias=cb[idx].bias))
            if training:
                bn_train_forward(dt, self.z[i], self.a[i], B * (8 << i) ** 2, 64, bn, self.mean[i], self.invstd[i], ws.small, ACT_RELU, 0.0, sync,
                                 self.sync_scratch.stats[i])
            else:
                ops.bn_fwd_eval(dt, self.z[i], self.a[i], B * (8 << i) ** 2, 64, bn.weight, bn.bias, bn.eps, bn.running_mean, bn.running_var, ws.small, ACT_RELU)
            x = self.a[i]
        if IMG_GEMM and IMG_DIRECT and self.l4g.Kpad_fwd == 64 and ops.convt_img_mfma_ok(dt, self.CH, 32, 32, 64, 4, 2, 1):
            # GEMM + col2im gather as ONE launch (the columns stay in LDS; celeba._GenEngine.forward): same bits
            ops.convt_img_mfma(dt, x, self.l4g.wp_fwd, cb[9].bias, self.img, B, self.CH, 32, 32, ACT_SIGMOID, 0.0, K=64)
        elif IMG_GEMM:
            ops.conv_fwd(self.l4g.c, dt, x, self.l4g.wp_fwd, self.cols4, None)
            ops.col2im_img(dt, self.cols4, B, self.CH, 32, 32, 4, 2, 1, cb[9].bias, ACT_SIGMOID, 0.0, self.img)
        else:
            ops.conv_bwd_data(self.l4.c, dt, x, self.l4.wp_bwd, self.img, ops.epilogue(bias=cb[9].bias, act=ACT_SIGMOID, out_mode=OUT_NCHW_F32))
        return self.img

    def backward(self, dimg, grad, sync=None, side=None):
        """``side`` (engine.SideStream): the weight- / bias-gradient chains run on side lanes; the caller joins them before reading ``grad``"""
        dt, B, g, ws = self.dtype, self.B, self.gen, self.ws
        cb = g.conv_block
        gof = lambda name: g.arena.grad_of(name, grad)
        nlane = [0]

        def wgrad_side(fn):
            if side is None:
                fn(ws)
            else:
                side.defer(nlane[0], fn)
                nlane[0] += 1
        ops.act_grad_mul_bias_nchw(dimg, self.img, self.dimg_z, B, self.CH, 64 * 64, ACT_SIGMOID, 0.0, ws.small, gof("conv_block.9.bias"))
        # the last layer's backward straight from the image gradient (no patch rows in HBM): weight gradient by eg_wgrad_img, input gradient by
        # eg_conv_img_mfma with the BatchNorm-backward sums of the layer below in its epilogue (celeba._GenEngine.backward)
        direct = self.l4_direct and sync is None and FUSE_STATS
        if not direct:
            ops.im2col_img(dt, self.dimg_z, self.patches, B, self.CH, 64, 64, 4, 2, 1, self.kp)

        def l4_wgrad(wsw):
            if direct:
                ns = ops.wgrad_img(dt, [self.dimg_z], self.a[2], wsw.slab, B, self.CH, 64, 64, 64)
            else:
                ns = ops.conv_wgrad(self.l4p.c, dt, self.patches, self.a[2], wsw.slab, wsw.wgs_target)
            ops.wgrad_reduce_perm(wsw.slab, ns, 64, 64, self.kp, 1, gof("conv_block.9.weight"), 0, 0, self.k0)
        wgrad_side(l4_wgrad)

        def bn_bwd_ep(i, c):
            """(row blocks, sums, epilogue) if the launch that produces da[i] can also form dy = da * relu'(bn(z)) and the two sums of layer
            i's BatchNorm backward (celeba._GenEngine.backward), else (0, None, None)"""
            nrb, stat = self._stat_buf(("bwd", i), c, False, ops.epilogue(), 64) if sync is None else (0, None)
            if not nrb:
                return 0, None, None
            bnl = cb[(0, 3, 6)[i] + 1]
            return nrb, stat, ops.epilogue(stat_mode=ops.STAT_BN_BWD, stat_out=stat, stat_aux=self.z[i], stat_p=(self.mean[i], self.invstd[i], bnl.weight, bnl.bias),
                                           stat_act=ACT_RELU)
        if direct:
            if "bwd_img" not in self._stat:
                nrb = ops.conv_img_mfma_stat_blocks(B, 64, 64)
                self._stat["bwd_img"] = (nrb, torch.empty(2 * 64 * nrb, device=self.inp.device, dtype=torch.float32))
            nrb, stat = self._stat["bwd_img"]
            fused = (nrb, stat, ops.epilogue(stat_mode=ops.STAT_BN_BWD, stat_out=stat, stat_aux=self.z[2], stat_p=(self.mean[2], self.invstd[2], cb[7].weight, cb[7].bias),
                                             stat_act=ACT_RELU))
            ops.conv_img_mfma(dt, [self.dimg_z], self.l4p.wp_fwd, self.da[2], B, self.CH, 64, 64, fused[2], N=64)
        else:
            fused = bn_bwd_ep(2, self.l4p.c)
            ops.conv_fwd(self.l4p.c, dt, self.patches, self.l4p.wp_fwd, self.da[2], fused[2])
        for i, idx in ((2, 6), (1, 3), (0, 0)):
            r = self.mid[i]
            bn = cb[idx + 1]
            M = B * (8 << i) ** 2
            if fused[0]:
                ops.bn_bwd_fused(dt, self.z[i], self.da[i], self.dz[i], M, 64, fused[1], fused[0], bn.weight, bn.bias, self.mean[i], self.invstd[i],
                                 gof(f"conv_block.{idx + 1}.weight"), gof(f"conv_block.{idx + 1}.bias"), ws.sums, ws.small)
            else:
                bn_train_backward(dt, self.z[i], self.da[i], self.dz[i], M, 64, bn, self.mean[i], self.invstd[i], ACT_RELU, 0.0,
                                  gof(f"conv_block.{idx + 1}.weight"), gof(f"conv_block.{idx + 1}.bias"), ws, sync, self.sync_scratch.sums[i])
            x_in = self.a[i - 1] if i > 0 else self.h

            def mid_wgrad(wsw, i=i, idx=idx, r=r, M=M, x_in=x_in):
                ns = ops.conv_wgrad(r.c, dt, self.dz[i], x_in, wsw.slab, wsw.wgs_target)
                ops.wgrad_reduce(wsw.slab, ns, 64, 64, 64, 16, gof(f"conv_block.{idx}.weight"))
                ops.bias_grad(dt, self.dz[i], M, 64, wsw.small, gof(f"conv_block.{idx}.bias"))
            wgrad_side(mid_wgrad)
            if i > 0:
                fused = bn_bwd_ep(i - 1, r.c)
                ops.conv_fwd(r.c, dt, self.dz[i], r.wp_fwd, self.da[i - 1], fused[2])
            else:                                    # into the ReLU output of fc2: mask fused into the epilogue
                ops.conv_fwd(r.c, dt, self.dz[i], r.wp_fwd, self.dh, ops.epilogue(mask=self.h, mask_act=ACT_RELU))

        # fc2: dW[f][k] = sum_b dh[b][n'(f)] a1[b][k]
        def fc2_wgrad(wsw):
            ns = ops.conv_wgrad(self.f2.c, dt, self.a1, self.dh, wsw.slab, wsw.wgs_target)
            ops.wgrad_reduce_perm(wsw.slab, ns, 1024, 1024, 128, 1, gof("fc2.0.weight"), 64, 16, 0)
            ops.fill_f32(self.gb2_perm)
            ops.bias_grad(dt, self.dh, B, 1024, wsw.small, self.gb2_perm)
            ops.gather_add(gof("fc2.0.bias"), self.gb2_perm, 1024, 16, 1, 64)
        wgrad_side(fc2_wgrad)
        ops.conv_bwd_data(self.f2.c, dt, self.dh, self.f2.wp_bwd, self.dz1, ops.epilogue(mask=self.a1, mask_act=ACT_RELU))

        # fc1
        def fc1_wgrad(wsw):
            ns = ops.conv_wgrad(self.f1.c, dt, self.inp, self.dz1, wsw.slab, wsw.wgs_target)
            ops.wgrad_reduce_perm(wsw.slab, ns, 128, 128, self.cpad, 1, gof("fc1.0.weight"), 0, 0, self.cin)
            ops.bias_grad(dt, self.dz1, B, 128, wsw.small, gof("fc1.0.bias"))
        wgrad_side(fc1_wgrad)


class Generator(_HipModule):
    """Drop-in for rp.py:123-157; ``forward(z_c)`` with z_c = cat(one-hot label, code)."""

    def __init__(self, code_dim=None, n_classes=None, channels=None, dtype="f32"):
        super().__init__()
        g = lambda v, name: getattr(opt, name) if v is None else v
        self.code_dim, self.n_classes, self.channels = g(code_dim, "code_dim"), g(n_classes, "n_classes"), g(channels, "channels")
        self.img_size = 64
        self.conv_block = nn.Sequential(
            nn.ConvTranspose2d(64, 64, 4, 2, 1), nn.BatchNorm2d(64), nn.ReLU(), nn.ConvTranspose2d(64, 64, 4, 2, 1), nn.BatchNorm2d(64), nn.ReLU(),
            nn.ConvTranspose2d(64, 64, 4, 2, 1), nn.BatchNorm2d(64), nn.ReLU(), nn.ConvTranspose2d(64, self.channels, 4, 2, 1))
        self.fc1 = nn.Sequential(nn.Linear(self.n_classes + self.code_dim, 128), nn.ReLU())
        self.fc2 = nn.Sequential(nn.Linear(128, 64 * 4 * 4), nn.ReLU())
        self._init_engine_state(dtype)

    def engine(self, B, slot=0) -> _GenEngine:
        """``slot``: independent engines of one batch size (own activations and panels): the trainer's two generator forwards of an
        iteration run on different chains"""
        self.arena
        key = (B, self.compute_dtype) if slot == 0 else (B, self.compute_dtype, slot)
        if key not in self._engines:
            self._engines[key] = _GenEngine(self, B, self.compute_dtype)
        return self._engines[key]

    def forward(self, z_c):
        _require_cuda(z_c)
        z_c = z_c.float().contiguous()
        eng = self.fresh_engine(z_c.shape[0])
        if not self.training:       # inference (dSprites/gen_imgs.py): running-stat BatchNorm, no autograd graph
            with torch.no_grad():
                return eng.forward(z_c[:, :self.n_classes].contiguous(), z_c[:, self.n_classes:].contiguous(), training=False).clone()
        return _GenFn.apply(eng, z_c[:, :self.n_classes].contiguous(), z_c[:, self.n_classes:].contiguous(), *list(self.parameters()))


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, labels, code, *params):
        ctx.eng = eng
        return eng.forward(labels, code).clone()

    @staticmethod
    def backward(ctx, dimg):
        eng = ctx.eng
        scratch = torch.zeros_like(eng.gen.arena.grad)
        eng.backward(dimg.contiguous(), scratch)
        grads = [scratch[off:off + k].view(p.shape) for p, (off, k) in zip(eng.gen.parameters(), eng.gen.arena.slices.values())]
        return (None, None, None, *grads)


# ================================================================================================
# affine utilities / losses
# ================================================================================================
def _theta_to_A(theta):
    B = theta.shape[0]
    A = torch.zeros(B, 3, 3, device=theta.device, dtype=torch.float32)
    A[:, :2] = theta
    A[:, 2, 2] = 1.0
    return A


def get_matrix(code_input_raw):
    """[B,>=4] codes -> [B,3,3] = R(theta) Z(p,p) T(x,y)  (utils_rp.py:94-115)."""
    _require_cuda(code_input_raw)
    c = code_input_raw.float().contiguous()
    theta = torch.empty(c.shape[0], 2, 3, device=c.device)
    ops.theta_rp(c, c.shape[1], c.shape[0], theta)
    return _theta_to_A(theta)


get_matrix_D = get_matrix        # identical formulas (utils_rp.py:38-59)


def get_matrix_pxy_align(code_input_raw):
    """translation-only alignment matrix T(.1 c1, .1 c2)  (utils_pxy.py:69-87)."""
    _require_cuda(code_input_raw)
    c = code_input_raw.float().contiguous()
    theta = torch.empty(c.shape[0], 2, 3, device=c.device)
    ops.theta_pxy_align_inv(c, c.shape[1], c.shape[0], theta)
    theta[:, :, 2] = -theta[:, :, 2]             # the kernel produces the inverse the loop actually uses (rp.py:376)
    return _theta_to_A(theta)


class _AffineRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real_code, trans_code):
        B, ld = real_code.shape
        pred = torch.empty(B, 4, device=real_code.device)
        ops.loss_affine_rp(real_code, trans_code, ld, 0, B, torch.zeros(B, 4, device=real_code.device), 4, 1.0, None, None, None, pred)
        ctx.save_for_backward(real_code, trans_code, pred)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        real_code, trans_code, pred = ctx.saved_tensors
        B, ld = real_code.shape
        tgt = (pred - dpred.float() * (4.0 * B / 2.0)).contiguous()
        d_real, d_trans = torch.empty_like(real_code), torch.empty_like(trans_code)
        ops.loss_affine_rp(real_code, trans_code, ld, 0, B, tgt, 4, 1.0, None, d_real, d_trans, None)
        return d_real, d_trans


def affine_regularzier(real_code, trans_code):
    """closed-form relative-transform recovery over the first 4 codes (utils_rp.py:118-147)."""
    _require_cuda(real_code)
    return _AffineRegFn.apply(real_code.float().contiguous(), trans_code.float().contiguous())


def mutual_info_loss(c_given_x, c):
    """rp.py:225-232 on probabilities (eager torch ops on [B,n] tensors; the fused trainer uses eg_loss_mutual_info on logits)."""
    eps = 1e-8
    return torch.mean(-torch.sum(torch.log(c_given_x + eps) * c, dim=1)) + torch.mean(-torch.sum(torch.log(c + eps) * c, dim=1))


# ================================================================================================
# fused train-loop entry
# ================================================================================================
# ================================================================================================
# Stage-1 trainer: fits Encoder_pxy (dSprites/pxy.py), whose checkpoint the stage-2 loop loads frozen
# ================================================================================================
class _PxyTrainEngine:
    """Encoder_pxy forward + backward for two batched tapes (E(img) and E(warp(img)): plain convs, no BatchNorm, so both forwards
    share every launch).  The image and the warp matrix carry no gradient (pxy.py:166-178), so the backward ends in the first conv."""

    def __init__(self, mod: "Encoder_pxy", B, dtype, arena: Arena):
        self.mod, self.B, self.dtype, self.arena = mod, B, dtype, arena
        dev = arena.flat.device
        tdt = ops.torch_dtype(dtype)
        self.ws = ws = Workspace.get(dev)
        C, S = mod.channels, mod.img_size
        self.C, self.S = C, S
        NB = 2 * B
        self.NB = NB
        self.k0 = C * 16
        self.kp = ops.round_up(self.k0, 8)
        self.l0 = ConvRec(dtype, NB, S // 2, S // 2, self.kp, TRUNK[0], 1, 1, 0, device=dev, want_bwd=False, ws=ws)
        self.mid = [ConvRec(dtype, NB, S >> (i + 1), S >> (i + 1), TRUNK[i], TRUNK[i + 1], 4, 2, 1, device=dev, ws=ws) for i in range(3)]
        self.nout = mod.fc1.weight.shape[0]
        self.head = ConvRec(dtype, NB, 4, 4, TRUNK[3], self.nout, 4, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.patches = torch.zeros(NB * (S // 2) ** 2, self.kp, device=dev, dtype=tdt)
        self.a = [torch.empty(NB, S >> (i + 1), S >> (i + 1), TRUNK[i], device=dev, dtype=tdt) for i in range(4)]
        self.dz = [torch.empty_like(t) for t in self.a]
        self.out = torch.empty(NB, self.nout, device=dev, dtype=torch.float32)
        self.repack()

    @ops.batched_packs
    def repack(self):
        cb = self.mod.conv_block
        ops.pack_strided(self.dtype, cb[0].weight, self.l0.wp_fwd, TRUNK[0], self.k0, self.l0.Kpad_fwd, 1, self.k0, 0, 1)
        for i in range(3):
            self.mid[i].pack(cb[2 * (i + 1)].weight)
        self.head.pack(self.mod.fc1.weight)

    def forward(self, img, trans_img):
        """-> codes [2B, n_out] fp32: rows [:B] of ``img``, rows [B:] of ``trans_img``"""
        dt, B, cb = self.dtype, self.B, self.mod.conv_block
        npix = B * (self.S // 2) ** 2
        ops.im2col_img(dt, img, self.patches[:npix], B, self.C, self.S, self.S, 4, 2, 1, self.kp)
        ops.im2col_img(dt, trans_img, self.patches[npix:], B, self.C, self.S, self.S, 4, 2, 1, self.kp)
        ops.conv_fwd(self.l0.c, dt, self.patches, self.l0.wp_fwd, self.a[0], ops.epilogue(bias=cb[0].bias, act=ACT_LRELU, slope=0.1))
        for i in range(3):
            ops.conv_fwd(self.mid[i].c, dt, self.a[i], self.mid[i].wp_fwd, self.a[i + 1], ops.epilogue(bias=cb[2 * (i + 1)].bias, act=ACT_LRELU, slope=0.1))
        ops.dense_small_fwd(dt, self.a[3], self.head.wp_fwd, self.mod.fc1.bias, self.out, self.NB, 16 * TRUNK[3], self.head.Kpad_fwd, self.nout, self.ws.small)
        return self.out

    def backward(self, dout, grad):
        """dout: d(loss)/d(codes) [2B, n_out] fp32; accumulates into the flat gradient ``grad`` (arena layout)."""
        dt, NB, ws, cb = self.dtype, self.NB, self.ws, self.mod.conv_block
        gof = lambda name: self.arena.grad_of(name, grad)
        K = 16 * TRUNK[3]
        ops.dense_small_wgrad(dt, dout, self.a[3], gof("fc1.weight"), gof("fc1.bias"), NB, K, self.nout, TRUNK[3], 16)
        ops.dense_small_bwd(dt, dout, self.head.wp_fwd, self.a[3], self.dz[3], NB, K, self.head.Kpad_fwd, self.nout, ACT_LRELU, 0.1)
        for i in (3, 2, 1, 0):
            idx = 2 * i
            rec = self.mid[i - 1] if i > 0 else self.l0
            x_in = self.a[i - 1] if i > 0 else self.patches
            rows = NB * (self.S >> (i + 1)) ** 2
            ops.bias_grad(dt, self.dz[i], rows, TRUNK[i], ws.small, gof(f"conv_block.{idx}.bias"))
            ns = ops.conv_wgrad(rec.c, dt, x_in, self.dz[i], ws.slab)
            if i > 0:
                ops.wgrad_reduce(ws.slab, ns, TRUNK[i], TRUNK[i], TRUNK[i - 1], 16, gof(f"conv_block.{idx}.weight"))
                ops.conv_bwd_data(rec.c, dt, self.dz[i], rec.wp_bwd, self.dz[i - 1],
                                  ops.epilogue(mask=self.a[i - 1], mask_act=ACT_LRELU, mask_slope=0.1))
            else:
                ops.wgrad_reduce_perm(ws.slab, ns, TRUNK[0], TRUNK[0], self.kp, 1, gof("conv_block.0.weight"), 0, 0, self.k0)


class PxyTrainer:
    """One call == one iteration of dSprites/pxy.py:156-191 (stage-1 trainer): real_code = E(img); trans_img = warp(img,
    get_matrix_pxy(code)); trans_code = E(trans_img); loss = MSE(affine_regularzier_pxy(real_code, trans_code), code); Adam(lr 2e-4,
    betas (.5, .999), :127) on Encoder_pxy.  Its ``state_dict()`` is the ``encoder_pxy_%d.pt`` the stage-2 loop loads (:205)."""

    def __init__(self, encoder_pxy: "Encoder_pxy", batch_size, dtype="f32", lr=2e-4, betas=(0.5, 0.999), allreduce=None):
        self.P, self.B = encoder_pxy, batch_size
        dt = parse_dtype(dtype)
        encoder_pxy.set_compute_dtype(dt)
        _require_cuda(next(encoder_pxy.parameters()))
        self.arena = Arena(encoder_pxy)                  # trainable here: parameters re-homed into one flat fp32 arena
        encoder_pxy._engines = {}                        # inference engines hold panels of the old storage
        self.eng = _PxyTrainEngine(encoder_pxy, batch_size, dt, self.arena)
        dev = self.arena.flat.device
        self.dev = dev
        self.lr, self.betas, self.allreduce = lr, betas, allreduce
        z = lambda *n: torch.zeros(*n, device=dev, dtype=torch.float32)
        B, C = batch_size, encoder_pxy.channels
        self.m, self.v = z(self.arena.numel), z(self.arena.numel)
        self.steps = torch.zeros(1, device=dev, dtype=torch.int32)
        self.losses = z(4)
        self.img_u8 = torch.zeros(B, 64, 64, device=dev, dtype=torch.uint8)
        self.img, self.trans = z(B, C, 64, 64), z(B, C, 64, 64)
        self.nd = encoder_pxy.fc1.weight.shape[0]           # 3 (p, x, y); the colored variant appends three colour gains
        self.code = z(B, self.nd)
        self.theta = z(B, 2, 3)
        self.dout = z(2 * B, self.nd)
        self.graph = None

    # hooks the colored variant overrides (colored.PxyColorTrainer)
    def _make_image(self):
        ops.u8_to_f32(self.img_u8, self.img)                                           # pxy.py:161-162

    def _transform(self):
        ops.warp_affine(self.img, self.theta, self.trans, self.B, self.P.channels, 64, 64)   # :177 (padding_mode='border')

    def _step_body(self):
        B, eng, ar, nd = self.B, self.eng, self.arena, self.nd
        ops.fill_f32(self.losses)
        self._make_image()
        ops.theta_pxy(self.code, nd, B, self.theta)                                     # :176
        self._transform()
        codes = eng.forward(self.img, self.trans)                                       # :174,178
        ops.loss_affine_pxy(codes[:B], codes[B:], nd, 0, B, self.code, nd, 1.0, self.losses[0:1], self.dout[:B], self.dout[B:], ncol=nd - 3)   # :180-182
        ops.fill_f32(ar.grad)
        eng.backward(self.dout, ar.grad)
        if self.allreduce is not None:
            self.allreduce(ar.grad)
        ops.adam_step(ar.flat, ar.grad, self.m, self.v, ar.numel, self.lr, self.betas[0], self.betas[1], 1e-8, self.steps[0:1], True)
        eng.repack()

    def load_inputs(self, img_u8, code):
        self.img_u8.copy_(img_u8, non_blocking=True)
        self.code.copy_(code, non_blocking=True)

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

    def train_step(self, img_u8, code):
        """train-loop entry: img_u8 uint8 [B,64,64] sprites, code [B,3] ~ U(-1,1) -> {'affine_loss'}"""
        self.load_inputs(img_u8, code)
        return {"affine_loss": float(self.step_resident()[0])}


class DspritesTrainer(ResidentStep):
    """One call == one iteration of dSprites/rp.py:365-482: D step (Adam lr 2e-4), then the joint info + affine + adversarial-G +
    relative-category step over G+E (Adam lr 1e-4).  optimizer_G of the reference is never stepped and is not created.
    Dead work removed: Encoder_pxy backward, the second (identical) alignment pass, D weight gradients in the joint step."""

    def __init__(self, encoder_pxy, generator, discriminator, encoder, batch_size, dtype="f32", allreduce=None, lrs=(2e-4, 1e-4), betas=(0.5, 0.999),
                 sync_bn=None, overlap=True):
        """``sync_bn`` (a dp.SyncBN): the generator's three BatchNorm layers use the statistics of the global batch.
        ``overlap``: the iteration runs as two chains (see _step_body); single-process only -- with ``allreduce`` / ``sync_bn`` the
        collectives stay on one stream and the serial body runs."""
        self.P, self.G, self.D, self.E, self.B = encoder_pxy, generator, discriminator, encoder, batch_size
        self.sync_bn = sync_bn
        self.overlap = bool(overlap) and allreduce is None and sync_bn is None
        dt = parse_dtype(dtype)
        for m in (encoder_pxy, generator, discriminator, encoder):
            m.set_compute_dtype(dt)
        B = batch_size
        self.pe, self.ge, self.de = encoder_pxy.engine(B), generator.engine(B), discriminator.engine(B)
        dev = generator.arena.flat.device
        if self.overlap:
            # second chain: own stream, own scratch (split-K scratch included), own engines -- the encoder, and a second generator engine
            # for the joint step's forward (and backward), so that it does not wait for the D step to release the first one's buffers
            self.ws2 = Workspace(dev, register=False)
            self.chain = torch.cuda.Stream(dev)
            # weight-gradient side lanes under the two chains: slower (profiles/r02_n_ab_small_lanes.txt), off.  "tailN": N lanes for the
            # generator backward only -- the one phase in which a single chain is active
            # (colored dSprites B = 512: 3.47 -> 3.34 ms with tail2; dSprites B = 128: 1.71 -> 1.73: on from batch 256 up,
            #  profiles/r02_z_ab_tail_lanes.txt)
            env = os.environ.get("EG_SMALL_LANES", "tail2" if B >= 256 else "0")
            self.tail_lanes = env.startswith("tail")
            nl = int(env[4:] or 2) if self.tail_lanes else int(env)
            self.side_a = SideStream(dev, Workspace.get(dev), lanes=nl) if nl else None     # weight-gradient lanes of the main chain
            self.side_b = SideStream(dev, self.ws2, lanes=nl) if (nl and not self.tail_lanes) else None    # ... and of the second chain
            with Workspace.scope(self.ws2):
                self.ee, self.ge2 = encoder.engine(B, slot=1), generator.engine(B, slot=1)
        else:
            self.ee, self.ge2 = encoder.engine(B), self.ge
        self.allreduce, self.lr, self.betas = allreduce, lrs, betas
        ga, da, ea = generator.arena, discriminator.arena, encoder.arena
        z = lambda *n: torch.zeros(*n, device=dev, dtype=torch.float32)
        self.mD, self.vD = z(da.numel), z(da.numel)
        self.miG, self.viG, self.miE, self.viE = z(ga.numel), z(ga.numel), z(ea.numel), z(ea.numel)
        self.steps = torch.zeros(2, device=dev, dtype=torch.int32)
        self.losses = z(8)                                    # d, g, info, affine, relative_cat
        C = generator.channels
        self.nc, self.cd = generator.n_classes, generator.code_dim
        self.img = z(B, C, 64, 64)
        self.theta, self.theta2 = z(B, 2, 3), z(B, 2, 3)
        self.align, self.trans1, self.trans2 = z(B, C, 64, 64), z(B, C, 64, 64), z(B, C, 64, 64)
        self.dimg = z(B, C, 64, 64)
        self.dout_d = z(2 * B, 1)
        self.d_cat, self.d_cont = z(3 * B, self.nc), z(3 * B, self.cd)
        self.code1, self.code2 = z(B, self.cd), z(B, self.cd)
        self.onehot1, self.onehot2 = z(B, self.nc), z(B, self.nc)
        self.graph = None

    def _adam(self, arena, m, v, lr, slot, tick):
        ops.adam_step(arena.flat, arena.grad, m, v, arena.numel, lr, self.betas[0], self.betas[1], 1e-8, self.steps[slot:slot + 1], tick)

    def import_adam_state(self, opt_D, opt_info):
        """moments and step counts of the two ``torch.optim.Adam`` the reference steps (dSprites/rp.py:270-279: D | G + E parameters,
        ``.parameters()`` order) -- teacher-forced comparisons against a CPU run of the reference loop"""
        from .engine import import_adam_moments
        n = lambda mod: len(list(mod.parameters()))
        s0 = import_adam_moments(opt_D, [(n(self.D), self.mD, self.vD)])
        s1 = import_adam_moments(opt_info, [(n(self.G), self.miG, self.viG), (n(self.E), self.miE, self.viE)])
        self.steps.copy_(torch.tensor([s0, s1], dtype=torch.int32))

    def _step_body(self):
        """One iteration as TWO chains.  These networks are small (0.5 GFLOP per image): the step is a chain of ~200 launches of a few
        microseconds each, bound by launch latency, not by the chip -- so independent parts run side by side:
          main chain : align -> G(code1) -> D step (D forward x2, loss, backward, Adam, re-pack) -> D(gen2) -> d(img) through D
          2nd chain  : G(code2) [second generator engine, behind G(code1): BatchNorm running statistics keep the reference's order]
                       -> transform(code2) -> E forward x3 -> info / affine / relative-category losses -> E backward -> d(img) through E
          joined     : d(img) sum -> G backward -> Adam(G), Adam(E) -> re-pack.
        The joint step's generator and encoder passes read only G and E, which the D step does not touch (rp.py:404-482).  Same kernels
        on the same operands as the serial body: bit-identical results (tests/test_gpu_dsprites.py)."""
        if not self.overlap:
            return self._step_body_serial()
        B, nc, cd = self.B, self.nc, self.cd
        ge, ge2, de, ee = self.ge, self.ge2, self.de, self.ee
        ga, da, ea = self.G.arena, self.D.arena, self.E.arena
        L = self.losses
        mark = SideStream.mark
        main, chain = torch.cuda.current_stream(), self.chain
        sa, sb = self.side_a, self.side_b
        sa_d = None if getattr(self, "tail_lanes", False) else sa          # lanes of the D step's backward
        join = lambda sd: sd.join_lanes() if sd is not None else None
        ops.fill_f32(L)
        align_late = ALIGN_LATE
        if not align_late:
            self._align()                                                                    # :374-377
        gen = ge.forward(self.onehot1, self.code1)
        chain.wait_event(mark())
        if align_late:
            # the alignment pass (frozen Encoder_pxy forward + warp: ~7 launches) reads only the real images: behind the first generator forward
            # on the main chain it runs beside the second chain's generator forward instead of in front of both chains
            self._align()
        e_align = mark()
        with torch.cuda.stream(chain), self.ws2.active():
            ops.fill_f32(ga.grad)
            ops.fill_f32(ea.grad)
            gen2 = ge2.forward(self.onehot2, self.code2)
            e_gen2 = mark()
            chain.wait_event(e_align)
            self._transform(self.code2, self.trans2, second=True)
            eo = ee.forward([gen2, self.align, self.trans2])
            cat, cont = eo["cat_layer.0"], eo["cont_layer.0"]
            ops.fill_f32(self.d_cat)
            ops.loss_mutual_info(cat[:B], nc, 0, nc, B, self.onehot2, nc, 0, False, 1.0, L[2:3], self.d_cat[:B])
            ops.loss_mse(cont[:B], cd, 0, cd, B, self.code2, cd, 0.0, 1.0, L[2:3], self.d_cont[:B])
            self._affine_loss(cont[B:2 * B], cont[2 * B:], L[3:4], self.d_cont[B:2 * B], self.d_cont[2 * B:])
            ops.loss_mutual_info(cat[2 * B:], nc, 0, nc, B, cat[B:2 * B], nc, 0, True, 1.0, L[4:5], self.d_cat[2 * B:])
            dimg_e = ee.backward(0, 3, {"cat_layer.0": self.d_cat, "cont_layer.0": self.d_cont}, ea.grad, need_dimg=True, side=sb)
            e_dimg = mark()
            evs = {}

            def update_e(_ws):
                # optimizer_info's step counter is shared by G and E: it ticks here, G's update (main chain, behind this event) reads it
                self._adam(ea, self.miE, self.viE, self.lr[1], 1, True)
                ee.repack()
                evs["chain"] = mark()
            if sb is not None:
                # behind the encoder's weight-gradient lanes, on the lanes' optimizer stream: the second chain itself must never wait for
                # its lanes (hipStreamEndCapture crashes on a stream-level cycle that does not pass through the capture's origin stream)
                sb.defer_opt(update_e)
            else:
                update_e(None)
            e_chain = evs["chain"]
        # ---- D step (:404-419): D(trans) then D(gen.detach()) ----
        self._transform(self.code1, self.trans1)                                             # :396-400
        ops.fill_f32(da.grad)
        out = de.forward([self.trans1, gen])["fc2"]
        ops.loss_bce_sigmoid(out[:B], 1, 0, B, 1.0, 0.5, L[0:1], self.dout_d[:B])
        ops.loss_bce_sigmoid(out[B:], 1, 0, B, 0.0, 0.5, L[0:1], self.dout_d[B:])
        de.backward(0, 2, {"fc2": self.dout_d}, da.grad, side=sa_d)
        join(sa_d)
        self._adam(da, self.mD, self.vD, self.lr[0], 0, True)
        de.repack()
        # ---- joint step (:424-482): the generator's adversarial term needs the UPDATED discriminator ----
        main.wait_event(e_gen2)
        g_fake = de.forward([gen2], patches=False)["fc2"]             # (input gradient only: no weight gradient, no patch rows)
        ops.loss_bce_sigmoid(g_fake, 1, 0, B, 1.0, 1.0, L[1:2], self.dout_d[:B])
        dimg_d = de.backward(0, 1, {"fc2": self.dout_d[:B]}, da.grad, need_wgrad=False, need_dimg=True)
        main.wait_event(e_dimg)
        ops.add_f32(self.dimg, dimg_e, dimg_d)
        ge2.backward(self.dimg, ga.grad, side=sa)
        join(sa)
        main.wait_event(e_chain)
        self._adam(ga, self.miG, self.viG, self.lr[1], 1, False)
        ge.repack()
        ge2.repack()

    def _step_body_serial(self):
        B, nc, cd = self.B, self.nc, self.cd
        pe, ge, de, ee = self.pe, self.ge, self.de, self.ee
        ga, da, ea = self.G.arena, self.D.arena, self.E.arena
        C = self.G.channels
        L = self.losses
        ops.fill_f32(L)
        self._align()                                                                        # :374-377
        self._transform(self.code1, self.trans1)                                             # :396-400
        # ---- D step (:404-419): D(trans) then D(gen.detach()) ----
        gen = ge.forward(self.onehot1, self.code1, sync=self.sync_bn)
        ops.fill_f32(da.grad)
        out = de.forward([self.trans1, gen])["fc2"]
        ops.loss_bce_sigmoid(out[:B], 1, 0, B, 1.0, 0.5, L[0:1], self.dout_d[:B])
        ops.loss_bce_sigmoid(out[B:], 1, 0, B, 0.0, 0.5, L[0:1], self.dout_d[B:])
        de.backward(0, 2, {"fc2": self.dout_d}, da.grad)
        if self.allreduce is not None:
            self.allreduce(da.grad)
        self._adam(da, self.mD, self.vD, self.lr[0], 0, True)
        de.repack()
        # ---- joint step (:424-482) ----
        ops.fill_f32(ga.grad)
        ops.fill_f32(ea.grad)
        gen = ge.forward(self.onehot2, self.code2, sync=self.sync_bn)
        self._transform(self.code2, self.trans2)
        eo = ee.forward([gen, self.align, self.trans2])
        cat, cont = eo["cat_layer.0"], eo["cont_layer.0"]
        g_fake = de.forward([gen], patches=False)["fc2"]
        ops.loss_bce_sigmoid(g_fake, 1, 0, B, 1.0, 1.0, L[1:2], self.dout_d[:B])
        dimg_d = de.backward(0, 1, {"fc2": self.dout_d[:B]}, da.grad, need_wgrad=False, need_dimg=True)
        ops.fill_f32(self.d_cat)
        ops.loss_mutual_info(cat[:B], nc, 0, nc, B, self.onehot2, nc, 0, False, 1.0, L[2:3], self.d_cat[:B])
        ops.loss_mse(cont[:B], cd, 0, cd, B, self.code2, cd, 0.0, 1.0, L[2:3], self.d_cont[:B])
        self._affine_loss(cont[B:2 * B], cont[2 * B:], L[3:4], self.d_cont[B:2 * B], self.d_cont[2 * B:])
        ops.loss_mutual_info(cat[2 * B:], nc, 0, nc, B, cat[B:2 * B], nc, 0, True, 1.0, L[4:5], self.d_cat[2 * B:])
        dimg_e = ee.backward(0, 3, {"cat_layer.0": self.d_cat, "cont_layer.0": self.d_cont}, ea.grad, need_dimg=True)
        ops.add_f32(self.dimg, dimg_e, dimg_d)
        pending = self.allreduce.start(ea.grad) if (self.allreduce is not None and hasattr(self.allreduce, "start")) else None
        ge.backward(self.dimg, ga.grad, sync=self.sync_bn)
        if self.allreduce is not None:
            self.allreduce(ga.grad)
            if pending is not None:
                self.allreduce.finish(pending)
            elif not hasattr(self.allreduce, "start"):
                self.allreduce(ea.grad)
        self._adam(ga, self.miG, self.viG, self.lr[1], 1, True)
        self._adam(ea, self.miE, self.viE, self.lr[1], 1, False)
        ge.repack()
        ee.repack()

    # -- dataset-specific pieces (overridden by the colored variant) -------------------------------------------------
    def _align(self):
        """align = warp(img, inverse(T(x,y))[:, :2]) with (p,x,y) from the frozen Encoder_pxy."""
        B, C = self.B, self.G.channels
        pcode = self.pe.forward(self.img)
        ops.theta_pxy_align_inv(pcode, pcode.shape[1], B, self.theta)
        ops.warp_affine(self.img, self.theta, self.align, B, C, 64, 64)

    def _transform(self, code, out, second=False):
        """``second``: called from the second chain -- its own theta scratch"""
        B, C = self.B, self.G.channels
        theta = self.theta2 if second else self.theta
        ops.theta_rp(code, self.cd, B, theta)
        ops.warp_affine(self.align, theta, out, B, C, 64, 64)

    def _affine_loss(self, cont_align, cont_trans, loss, d_align, d_trans):
        ops.loss_affine_rp(cont_align, cont_trans, self.cd, 0, self.B, self.code2, self.cd, 1.0, loss, d_align, d_trans)

    def load_inputs(self, img_u8, code1, labels1, code2, labels2):
        """img_u8: uint8 [B,64,64] sprites (or float [B,C,64,64]); codes [B,code_dim]; labels int64 [B]."""
        if img_u8.dtype == torch.uint8:
            ops.u8_to_f32(img_u8.contiguous(), self.img)
        else:
            self.img.copy_(img_u8.reshape(self.img.shape))
        self.code1.copy_(code1)
        self.code2.copy_(code2)
        for oh, lab in ((self.onehot1, labels1), (self.onehot2, labels2)):
            oh.zero_()
            oh.scatter_(1, lab.view(-1, 1), 1.0)

    def train_step(self, img_u8, code1, labels1, code2, labels2):
        self.load_inputs(img_u8, code1, labels1, code2, labels2)
        l = self.step_resident().tolist()
        return dict(d_loss=l[0], g_loss=l[1], info_loss=l[2], affine_loss=l[3], relative_cat_loss=l[4])


class DeviceInputs(DeviceSampler):
    """Device-side replacement of the dSprites loop's host input work (dSprites/rp.py:236-262 the uint8 sprite array + DataLoader; :389-430
    numpy draws in the reference's order: code ~ U(-1,1), labels ~ randint, then again for the joint step).  ``dataset_u8``: uint8
    [N,64,64] sprites with values {0,1}."""

    def sprites(self, tr, idx=None):
        B = tr.B
        batch = self.buf("sprites", (B,) + tuple(self.data.shape[1:]), torch.uint8)
        torch.index_select(self.data, 0, self.sample_indices(B, 1) if idx is None else idx, out=batch)
        return batch

    def codes_and_labels(self, tr, first_stream):
        self.draw(ops.RNG_UNIFORM, tr.code1, -1.0, 1.0, first_stream)
        self.labels_onehot("labels1", tr.onehot1, tr.nc, first_stream + 1)
        self.draw(ops.RNG_UNIFORM, tr.code2, -1.0, 1.0, first_stream + 2)
        self.labels_onehot("labels2", tr.onehot2, tr.nc, first_stream + 3)

    def enqueue(self, tr: "DspritesTrainer"):
        self.begin_draws()                              # sprite indices, codes and labels (+ one-hot rows): one launch
        idx = self.sample_indices(tr.B, 1)
        self.codes_and_labels(tr, 2)
        self.end_draws()
        if FUSE_DRAWS and self.data.dim() == 3:
            # gather + uint8 -> float + counter tick as one launch (the gather of the MNIST / CelebA samplers; the same {0, 1} values)
            N, H, W = self.data.shape
            ops.gather_u8_images(self.data, idx, None, tr.img, tr.B, 1, H, W, 1.0, 0.0, tick=self.step)
            return
        ops.u8_to_f32(self.sprites(tr, idx), tr.img)
        self.tick()
