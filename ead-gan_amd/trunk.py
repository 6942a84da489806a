"""Generic engine for the reference's discriminator/encoder trunks: a stack of spectrally-normalised stride-2 convolutions
(LeakyReLU, optionally followed by BatchNorm2d) ending in one or more dense heads over the flattened feature map.

Used by MNIST's Discriminator / Encoder (MNIST/EAD-GAN_rpqmnxy.py:101-175).  Same design as the CelebA discriminator
engine: activations NHWC in the compute dtype, W/sigma never materialised (1/sigma folded into epilogues), every forward
("tape") keeps its own sigma/u/v, tapes without BatchNorm are batched along M, gradients travel as dzs = dz/sigma so that
ONE weight-gradient GEMM covers all tapes and the spectral-norm rank-1 terms are added by a single-pass reduce.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import torch

from . import ops
from .engine import ConvRec, Workspace
from .ops import ACT_LRELU, ACT_NONE, EG_F32, OUT_NCHW_F32

IMG_GEMM = os.environ.get("EG_IMG_GEMM", "1") == "1"     # as in celeba.py
FUSE_STATS = os.environ.get("EG_FUSE_STATS", "1") != "0"  # as in celeba.py
# the first layer straight from the fp32 images (ops.conv_img_mfma: 4x4 / stride-2 first layers with 32 / 64 / 128 channels, 16-bit types) instead
# of patch rows + a K = 64 GEMM over them; the patch rows are then only the weight gradient's operand (``forward(..., patches=False)`` skips
# them where no weight gradient follows).  Same bits.  EG_IMG_DIRECT=0: patch rows + GEMM
IMG_DIRECT = os.environ.get("EG_IMG_DIRECT", "1") != "0"
# ... and its weight gradient straight from the images too (ops.wgrad_img: patch rows expanded in LDS): no patch rows in HBM at all for the
# first layer.  EG_WGRAD_IMG=0: patch rows (eg_im2col_img at forward time) + the per-tap GEMM
WGRAD_IMG = os.environ.get("EG_WGRAD_IMG", "1") != "0"
# trunks with BatchNorm (the MNIST discriminator / encoder): the forward's convolutions, hidden layers and heads once over all T tapes with one
# BatchNorm pass per tape in between, instead of T whole passes.  EG_BN_BATCH_TAPES=0: one pass per tape
BN_BATCH_TAPES = os.environ.get("EG_BN_BATCH_TAPES", "1") != "0"

SN_EPS = 1e-12


@dataclass
class Head:
    name: str            # parameter prefix inside the owning module, e.g. "adv_layer.0"
    module: object       # nn.Linear / nn.Conv2d container (spectral-normed or plain)
    sn: bool
    compute: bool = True     # evaluate the head output (False: only its power iteration runs, e.g. MNIST noise_layer)
    grad: bool = True        # takes part in backward
    N: int = 0
    off: int = 0             # row offset inside the combined head panel


class TrunkEngine:
    def __init__(self, owner, convs, conv_names, bns, bn_names, heads, in_ch, size, k, slope, B, dtype, NT, fcs=None, fc_names=None):
        """convs: SN conv modules; bns: per-layer BatchNorm module or None (applied AFTER the LeakyReLU of that layer);
        fcs: optional hidden SN-Linear layers (each followed by LeakyReLU(slope)) between the conv trunk and the heads."""
        self.owner, self.convs, self.conv_names, self.bns, self.bn_names = owner, convs, conv_names, bns, bn_names
        self.fcs, self.fc_names = list(fcs or []), list(fc_names or [])
        self.heads, self.B, self.dtype, self.NT, self.k, self.slope = heads, B, dtype, NT, k, slope
        self._stat = {}                                 # fused-statistics buffers per (layer, tapes) (_sn_stat)
        self.in_ch, self.S = in_ch, size
        dev = owner.arena.flat.device
        self.ws = ws = Workspace.get(dev)
        tdt = ops.torch_dtype(dtype)
        L = len(convs)
        self.L = L
        self.W = [c.weight_orig.shape[0] for c in convs]
        self.hw = [size >> (i + 1) for i in range(L)]
        self.taps = k * k
        self.k0 = in_ch * k * k                                   # real K of the image-side layer
        self.kp = ops.round_up(self.k0, 8)
        self.cin = [self.kp] + self.W[:-1]
        self.has_bn = any(b is not None for b in bns)
        pad = 1
        # records (packed panels) for the largest batch; per-tape-count geometries
        self.l0img = ConvRec(dtype, B, size, size, in_ch, self.W[0], k, 2, pad, device=dev, want_fwd=False, want_wgrad=False, ws=ws)
        self.l0p = ConvRec(dtype, NT * B, size // 2, size // 2, self.kp, self.W[0], 1, 1, 0, device=dev, want_bwd=False, ws=ws)
        # d(img) as ONE GEMM over the layer-0 lattice with N = k*k*C columns + the col2im gather (eg_col2im_img; see celeba._DiscEngine.l1g) where
        # the GEMM engine takes that N (a multiple of 8: the 4x4 dSprites trunks; the 3x3 single-channel MNIST trunk keeps the implicit form)
        self.l0g = None
        if self.k0 % 8 == 0 and in_ch in (1, 3) and size // 2 == (size + 2 * pad - k) // 2 + 1:
            self.l0g = ConvRec(dtype, B, size // 2, size // 2, self.W[0], self.k0, 1, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
            self.cols0 = torch.empty(B * (size // 2) ** 2, self.k0, device=dev, dtype=ops.torch_dtype(dtype))
        self.in_ch, self.k_img, self.pad_img = in_ch, k, pad
        self.mid = [ConvRec(dtype, NT * B, self.hw[i], self.hw[i], self.W[i], self.W[i + 1], k, 2, pad, device=dev, ws=ws) for i in range(L - 1)]
        self.hk = self.hw[-1]
        self.Kc = self.hk * self.hk * self.W[-1]                    # flattened conv-trunk features
        self.fN = [f.weight_orig.shape[0] for f in self.fcs]
        self.fK = [self.Kc] + self.fN[:-1]
        self.K = self.fN[-1] if self.fcs else self.Kc               # features the heads read
        self.head_hk = 1 if self.fcs else self.hk
        self.head_C = self.K if self.fcs else self.W[-1]
        assert not (self.fcs and bns[-1] is not None), "hidden FC layers after a BatchNorm-terminated trunk are not needed by any reference net"
        self.frec = [ConvRec(dtype, NT * B, 1, 1, self.fK[i], self.fN[i], 1, 1, 0, device=dev, ws=ws) for i in range(len(self.fcs))]
        off = 0
        for h in heads:
            w = h.module.weight_orig if h.sn else h.module.weight
            h.N = w.shape[0]
            if h.compute:
                h.off = off
                off += h.N
        self.ncomb = off
        assert self.ncomb <= 32, "combined head panel holds at most 32 outputs"
        self.head_rec = ConvRec(dtype, NT * B, self.head_hk, self.head_hk, self.head_C, 32, self.head_hk, 1, 0, device=dev, want_bwd=False, want_wgrad=False, ws=ws)
        self.head_rec.wp_fwd.zero_()
        self.headw = ConvRec(dtype, NT * B, 1, 1, self.K, 32, 1, 1, 0, device=dev, want_fwd=False, want_bwd=False, ws=ws)
        self.geo = {}
        for T in range(1, NT + 1):
            self.geo[T] = {"l0p": ops.make_conv(T * B, size // 2, size // 2, self.kp, self.W[0], 1, 1, 0),
                           "mid": [ops.make_conv(T * B, self.hw[i], self.hw[i], self.W[i], self.W[i + 1], k, 2, pad) for i in range(L - 1)],
                           "headw": ops.make_conv(T * B, 1, 1, self.K, 32, 1, 1, 0),
                           "fc": [ops.make_conv(T * B, 1, 1, self.fK[i], self.fN[i], 1, 1, 0) for i in range(len(self.fcs))]}
        for i in range(L):
            ws.need_small(ops.bias_grad_sn_ws_floats(NT * B * self.hw[i] ** 2, self.W[i], B * self.hw[i] ** 2))
            if bns[i] is not None:
                ws.need_small(ops.bn_ws_floats(B * self.hw[i] ** 2, self.W[i]))
                ws.need_sums(2 * self.W[i])
        e = lambda *s, dt=tdt: torch.empty(s, device=dev, dtype=dt)
        z = lambda *s: torch.zeros(s, device=dev, dtype=torch.float32)
        self.patches = torch.zeros(NT * B * (size // 2) ** 2, self.kp, device=dev, dtype=tdt)
        self.a = [e(NT * B, self.hw[i], self.hw[i], self.W[i]) for i in range(L)]          # LeakyReLU outputs
        self.y = [e(NT * B, self.hw[i], self.hw[i], self.W[i]) if bns[i] is not None else None for i in range(L)]   # BatchNorm outputs
        self.dz = [torch.empty_like(t) for t in self.a]                                     # dL/dz / sigma
        self.dyb = [torch.empty_like(t) if bns[i] is not None else None for i, t in enumerate(self.a)]   # dL/d(BN output)
        self.mean = [z(NT, self.W[i]) if bns[i] is not None else None for i in range(L)]
        self.invstd = [z(NT, self.W[i]) if bns[i] is not None else None for i in range(L)]
        self.fa = [e(NT * B, n) for n in self.fN]                                            # hidden FC outputs (LeakyReLU)
        self.fdz = [torch.empty_like(t) for t in self.fa]
        for n in self.fN:
            ws.need_small(ops.bias_grad_sn_ws_floats(NT * B, n, B))
        self.fsigma = [torch.ones(NT, device=dev) for _ in self.fcs]
        self.fu = [z(NT, n) for n in self.fN]
        self.fv = [z(NT, kk) for kk in self.fK]
        self.fcoef = [z(4) for _ in self.fcs]
        self.outs = {h.name: z(NT * B, h.N) for h in heads if h.compute}
        self.dys_t = torch.zeros(NT * B, 32, device=dev, dtype=tdt)
        self.dys32 = z(NT * B, max(self.ncomb, 1))
        self.dimg = z(B, in_ch, size, size)
        kd = [self.k0] + [self.W[i] * self.taps for i in range(L - 1)]
        self.sigma = [torch.ones(NT, device=dev) for _ in range(L)]
        self.u = [z(NT, self.W[i]) for i in range(L)]
        self.v = [z(NT, kd[i]) for i in range(L)]
        self.coef = [z(4) for _ in range(L)]
        self.hsigma = {h.name: torch.ones(NT, device=dev) for h in heads if h.sn}
        self.hu = {h.name: z(NT, h.N) for h in heads if h.sn}
        self.hv = {h.name: z(NT, self.K) for h in heads if h.sn}
        self.hcoef = {h.name: z(4) for h in heads}
        self._sn_arrays = []
        for t in range(NT):
            ent = [(c.weight_orig, c.weight_u, c.weight_v, self.sigma[i][t:t + 1], self.u[i][t], self.v[i][t]) for i, c in enumerate(convs)]
            ent += [(f.weight_orig, f.weight_u, f.weight_v, self.fsigma[i][t:t + 1], self.fu[i][t], self.fv[i][t]) for i, f in enumerate(self.fcs)]
            ent += [(h.module.weight_orig, h.module.weight_u, h.module.weight_v, self.hsigma[h.name][t:t + 1], self.hu[h.name][t], self.hv[h.name][t])
                    for h in heads if h.sn]
            self._sn_arrays.append(ops.sn_layers(ent))
        ws.need_small(ops.sn_multi_ws_floats(self._sn_arrays[0]))
        self.sn_counters = torch.zeros(16, device=dev, dtype=torch.int32)       # arrival counters of the two-launch power iteration
        self.imgs = [None] * NT
        self.patch_ok = [False] * NT
        self.img_direct = (IMG_DIRECT and NT <= 3 and self.l0p.Kpad_fwd == 64 and self.kp <= 64
                           and ops.conv_img_mfma_ok(dtype, in_ch, size, size, self.W[0], k, 2, pad))
        self.wgrad_direct = (self.img_direct and WGRAD_IMG and self.kp == self.k0
                             and ops.wgrad_img_ok(dtype, in_ch, size, size, self.W[0], k, 2, pad))
        if self.wgrad_direct:
            ws.need_slab(ops.wgrad_img_splits(NT * B, self.W[0]) * self.W[0] * self.kp * 4)
        self.repack()

    # ------------------------------------------------------------------------------------------------------------------
    @ops.batched_packs
    def repack(self):
        dt = self.dtype
        self.l0img.pack(self.convs[0].weight_orig)
        if self.l0g is not None:                        # wp[t*C + c][co] = W[co][c][t]   (Conv2d master [W0][C][k][k])
            kk = self.k_img * self.k_img
            ops.pack_strided(dt, self.convs[0].weight_orig, self.l0g.wp_fwd, self.k0, self.W[0], self.l0g.Kpad_fwd, self.in_ch, 1, kk, self.k0)
        ops.pack_strided(dt, self.convs[0].weight_orig, self.l0p.wp_fwd, self.W[0], self.k0, self.l0p.Kpad_fwd, 1, self.k0, 0, 1)
        for i in range(self.L - 1):
            self.mid[i].pack(self.convs[i + 1].weight_orig)
        for i, f in enumerate(self.fcs):
            if i == 0:
                # Linear over the NCHW-flattened map: forward panel [N][(kh,kw,ci)] via the k=hk conv view, backward panel [n'][co] with
                # n' = t*C + ci  <-  master column f = ci*T + t
                T = self.hk * self.hk
                ops.pack_fwd(ops.make_conv(self.B, self.hk, self.hk, self.W[-1], self.fN[0], self.hk, 1, 0), dt, f.weight_orig, self.frec[0].wp_fwd)
                ops.pack_strided(dt, f.weight_orig, self.frec[0].wp_bwd, self.Kc, self.fN[0], ops.round_up(self.fN[0], ops.bk(dt)), self.W[-1], 1, T, self.Kc)
            else:
                self.frec[i].pack(f.weight_orig)
        kpad = self.head_rec.Kpad_fwd
        for h in self.heads:
            if not h.compute:
                continue
            w = h.module.weight_orig if h.sn else h.module.weight
            c = ops.make_conv(self.B, self.head_hk, self.head_hk, self.head_C, h.N, self.head_hk, 1, 0)
            ops.pack_fwd(c, dt, w, self.head_rec.wp_fwd[h.off * kpad:])

    def rows(self, i):
        return self.B * self.hw[i] ** 2

    def _sl(self, buf, t0):
        return buf[t0 * (buf.shape[0] // self.NT):]

    def _inp(self, i, t0):
        """input activation of layer i (output of layer i-1 after its optional BatchNorm; patches for i == 0)"""
        if i == 0:
            return self._sl(self.patches, t0)
        return self._sl(self.y[i - 1] if self.bns[i - 1] is not None else self.a[i - 1], t0)

    def _ep(self, i, t0):
        return ops.epilogue(bias=self.convs[i].bias, sigma=self.sigma[i][t0:], sigma_rows=self.rows(i), act=ACT_LRELU, slope=self.slope)

    def _fwd_pass(self, t0, T, training=True, skip0=False):
        dt, B = self.dtype, self.B
        g = self.geo[T]
        for i in range(self.L):
            geo = g["l0p"] if i == 0 else g["mid"][i - 1]
            wp = self.l0p.wp_fwd if i == 0 else self.mid[i - 1].wp_fwd
            if not (i == 0 and skip0):                  # (layer 0 of all tapes came from the images in one launch)
                ops.conv_fwd(geo, dt, self._inp(i, t0), wp, self._sl(self.a[i], t0), self._ep(i, t0))
            bn = self.bns[i]
            if bn is not None:
                # BatchNorm is per forward call: one pass per tape, in tape order (every module's running statistics then see the tapes in the
                # order of the reference's consecutive calls), while the convolutions around it run once over all T tapes
                for t in range(t0, t0 + T):
                    if training:
                        ops.bn_fwd_train(dt, self._sl(self.a[i], t), self._sl(self.y[i], t), self.rows(i), self.W[i], bn.weight, bn.bias, bn.eps, bn.momentum,
                                         bn.running_mean, bn.running_var, bn.num_batches_tracked, self.mean[i][t], self.invstd[i][t], self.ws.small, ACT_NONE)
                    else:   # module.eval() (score/*.load_encoder of the reference): running statistics, nothing updated
                        ops.bn_fwd_eval(dt, self._sl(self.a[i], t), self._sl(self.y[i], t), self.rows(i), self.W[i], bn.weight, bn.bias, bn.eps,
                                        bn.running_mean, bn.running_var, self.ws.small, ACT_NONE)
        x = self._inp(self.L, t0)
        for i, f in enumerate(self.fcs):
            ops.conv_fwd(g["fc"][i], dt, x, self.frec[i].wp_fwd, self.fa[i][t0 * B:],
                         ops.epilogue(bias=f.bias, sigma=self.fsigma[i][t0:], sigma_rows=B, act=ACT_LRELU, slope=self.slope))
            x = self.fa[i][t0 * B:]
        kpad = self.head_rec.Kpad_fwd
        for h in self.heads:
            if not h.compute:
                continue
            out = self.outs[h.name][t0 * B:(t0 + T) * B]
            wp = self.head_rec.wp_fwd[h.off * kpad:]
            if h.sn:
                ops.dense_small_fwd_sn(dt, x, wp, h.module.bias, out, T * B, self.K, kpad, h.N, self.hsigma[h.name][t0:], B, self.ws.small)
            else:
                ops.dense_small_fwd(dt, x, wp, h.module.bias, out, T * B, self.K, kpad, h.N, self.ws.small)

    def forward(self, imgs, t0=0, training=True, patches=True):
        """len(imgs) forwards as tapes t0.. (power iterations in list order).  Returns {head name: [len(imgs)*B, N]} views.
        ``patches=False``: no weight gradient will be asked for these tapes -- their patch rows are not built where the first layer reads the
        images themselves (``img_direct``)."""
        dt, B = self.dtype, self.B
        T = len(imgs)
        assert 1 <= T and t0 + T <= self.NT
        npix = B * (self.S // 2) ** 2
        for kk, img in enumerate(imgs):
            t = t0 + kk
            self.imgs[t] = img
            ops.sn_power_iter_multi(self._sn_arrays[t], self.ws.small, training, SN_EPS, self.sn_counters)
            if not training:
                for i, c in enumerate(self.convs):
                    self.u[i][t].copy_(c.weight_u)
                    self.v[i][t].copy_(c.weight_v)
                for i, f in enumerate(self.fcs):
                    self.fu[i][t].copy_(f.weight_u)
                    self.fv[i][t].copy_(f.weight_v)
                for h in self.heads:
                    if h.sn:
                        self.hu[h.name][t].copy_(h.module.weight_u)
                        self.hv[h.name][t].copy_(h.module.weight_v)
            if not self.img_direct:
                ops.im2col_img(dt, img, self.patches[t * npix:(t + 1) * npix], B, self.in_ch, self.S, self.S, self.k, 2, 1, self.kp)
                self.patch_ok[t] = True
        if self.img_direct:
            ops.conv_img_mfma(dt, list(imgs), self.l0p.wp_fwd, self._sl(self.a[0], t0), B, self.in_ch, self.S, self.S, self._ep(0, t0), N=self.W[0])
            for kk, img in enumerate(imgs):             # the weight gradient's operand, behind the launch the next layer waits for
                if patches and not self.wgrad_direct:
                    ops.im2col_img(dt, img, self.patches[(t0 + kk) * npix:(t0 + kk + 1) * npix], B, self.in_ch, self.S, self.S, self.k, 2, 1, self.kp)
                self.patch_ok[t0 + kk] = bool(patches) and not self.wgrad_direct
        if self.has_bn and not BN_BATCH_TAPES:
            for kk in range(T):
                self._fwd_pass(t0 + kk, 1, training, self.img_direct)
        else:
            self._fwd_pass(t0, T, training, self.img_direct)
        return {h.name: self.outs[h.name][t0 * B:(t0 + T) * B] for h in self.heads if h.compute}

    # ------------------------------------------------------------------------------------------------------------------
    def _sn_stat(self, i, T, c, ep):
        """fused sums of layer i's bias gradient / spectral-norm coefficient in the backward-data launch that produces dzs_i (T tapes): (row
        blocks, buffer), (0, None) where that launch cannot take them (4x4 stride-2 layers, whole 128-row tiles per tape and phase)"""
        key = (i, T)
        if key not in self._stat:
            ok = FUSE_STATS and self.taps == 16 and self.rows(i + 1) % 128 == 0 and self.dtype != EG_F32
            nrb = ops.conv_stat_blocks(c, self.dtype, True, ep) if ok else 0
            if nrb % (4 * T):
                nrb = 0
            N = self.W[i]
            self._stat[key] = (nrb, torch.empty(N * nrb + nrb * max(N // 128, 1), device=self.patches.device, dtype=torch.float32) if nrb else None)
        return self._stat[key]

    def _bwd_pass(self, t0, T, douts, grad, need_wgrad, need_dimg, side=None):
        dt, B, ws, L = self.dtype, self.B, self.ws, self.L
        gof = lambda name: self.owner.arena.grad_of(name, grad)
        nlane = [0]

        def wgrad_side(fn):                              # a layer's weight- / bias-gradient chain: on a side lane when there is one
            if side is None:
                fn(ws)
            else:
                side.defer(nlane[0], fn)
                nlane[0] += 1
        g = self.geo[T]
        kpad = self.head_rec.Kpad_fwd
        rows = T * B
        dys_t, dys32 = self.dys_t[t0 * B:], self.dys32[t0 * B:]
        gheads = [h for h in self.heads if h.compute]
        for h in gheads:
            dout = douts.get(h.name)
            assert dout is not None, f"missing head gradient for {h.name} (pass zeros)"
            out = self.outs[h.name][t0 * B:]
            ops.head_prep_sn(dt, dout, h.N, out, h.N, h.module.bias, rows, h.N, self.hsigma[h.name][t0:] if h.sn else None, B, dys_t, 32, h.off,
                             gof(h.name + ".bias") if (need_wgrad and h.grad) else None, self.hcoef[h.name] if h.sn else None, dys32, self.ncomb)
        nf = len(self.fcs)
        x = self.fa[-1][t0 * B:] if nf else self._inp(L, t0)
        if need_wgrad:
            def head_wgrad(wsw):
                ns = ops.conv_wgrad(g["headw"], dt, x, dys_t, wsw.slab, wsw.wgs_target)
                tk = self.head_hk * self.head_hk
                for h in gheads:
                    if not h.grad:
                        continue
                    slab = wsw.slab[h.off * self.K:]
                    if h.sn:
                        ops.wgrad_reduce_rank1(slab, ns, 32, h.N, self.head_C, tk, gof(h.name + ".weight_orig"), T, self.hcoef[h.name],
                                               self.hu[h.name][t0:], self.hv[h.name][t0:])
                    else:
                        ops.wgrad_reduce(slab, ns, 32, h.N, self.head_C, tk, gof(h.name + ".weight"))
            wgrad_side(head_wgrad)
        last = L - 1
        if nf:
            ops.dense_small_bwd(dt, dys32, self.head_rec.wp_fwd, self.fa[-1][t0 * B:], self.fdz[-1][t0 * B:], rows, self.K, kpad, self.ncomb,
                                ACT_LRELU, self.slope, self.fsigma[-1][t0:], B)
            for i in range(nf - 1, -1, -1):
                f, nm = self.fcs[i], self.fc_names[i]
                xin = self.fa[i - 1][t0 * B:] if i > 0 else self._inp(L, t0)
                if need_wgrad:
                    def fc_wgrad(wsw, i=i, f=f, nm=nm, xin=xin):
                        ops.bias_grad_sn(dt, self.fdz[i][t0 * B:], self.fa[i][t0 * B:], f.bias, rows, self.fN[i], B, self.fsigma[i][t0:], self.slope,
                                         wsw.small, gof(nm + ".bias"), self.fcoef[i])
                        ns = ops.conv_wgrad(g["fc"][i], dt, xin, self.fdz[i][t0 * B:], wsw.slab, wsw.wgs_target)
                        C, tk = (self.W[-1], self.hk * self.hk) if i == 0 else (self.fK[i], 1)
                        ops.wgrad_reduce_rank1(wsw.slab, ns, self.fN[i], self.fN[i], C, tk, gof(nm + ".weight_orig"), T, self.fcoef[i], self.fu[i][t0:],
                                               self.fv[i][t0:])
                    wgrad_side(fc_wgrad)
                if i > 0:
                    ops.conv_bwd_data(g["fc"][i], dt, self.fdz[i][t0 * B:], self.frec[i].wp_bwd, self.fdz[i - 1][t0 * B:],
                                      ops.epilogue(sigma=self.fsigma[i - 1][t0:], sigma_rows=B, mask=self.fa[i - 1][t0 * B:], mask_act=ACT_LRELU, mask_slope=self.slope))
                else:
                    ops.conv_bwd_data(g["fc"][0], dt, self.fdz[0][t0 * B:], self.frec[0].wp_bwd, self._sl(self.dz[last], t0),
                                      ops.epilogue(sigma=self.sigma[last][t0:], sigma_rows=B, mask=self._sl(self.a[last], t0), mask_act=ACT_LRELU,
                                                   mask_slope=self.slope))
        elif self.bns[last] is not None:
            ops.dense_small_bwd(dt, dys32, self.head_rec.wp_fwd, None, self._sl(self.dyb[last], t0), rows, self.K, kpad, self.ncomb)
        else:
            ops.dense_small_bwd(dt, dys32, self.head_rec.wp_fwd, self._sl(self.a[last], t0), self._sl(self.dz[last], t0), rows, self.K, kpad, self.ncomb,
                                ACT_LRELU, self.slope, self.sigma[last][t0:], B)
        fused = (0, None)                               # (row blocks, sums) if dzs_i came with its bias-gradient sums / spectral-norm dots
        for i in range(L - 1, -1, -1):
            bn = self.bns[i]
            if bn is not None:
                nm = self.bn_names[i]
                for t in range(t0, t0 + T):             # per tape, as the forward's statistics were (the convolutions around it take all tapes at once)
                    ops.bn_bwd_post(dt, self._sl(self.a[i], t), self._sl(self.dyb[i], t), self._sl(self.dz[i], t), self.rows(i), self.W[i], bn.weight, bn.bias,
                                    self.mean[i][t], self.invstd[i][t], gof(nm + ".weight") if need_wgrad else None, gof(nm + ".bias") if need_wgrad else None,
                                    ws.sums, ws.small, ACT_LRELU, self.slope, self.sigma[i][t:t + 1])
            geo = g["l0p"] if i == 0 else g["mid"][i - 1]
            if need_wgrad:
                assert i > 0 or self.wgrad_direct or all(self.patch_ok[t0:t0 + T]), "forward(..., patches=False) built no patch rows for these tapes"

                def layer_wgrad(wsw, i=i, geo=geo, nm=self.conv_names[i], fused=fused):
                    if fused[0]:
                        tiles_m = fused[0] // 4         # row blocks (128 lattice rows) per sub-pixel phase of the launch that produced dzs_i
                        ops.bias_grad_sn_fused(fused[1], fused[0], self.W[i], tiles_m, tiles_m // T, T, self.sigma[i][t0:], gof(nm + ".bias"), self.coef[i])
                    else:
                        ops.bias_grad_sn(dt, self._sl(self.dz[i], t0), self._sl(self.a[i], t0), self.convs[i].bias, T * self.rows(i), self.W[i], self.rows(i),
                                         self.sigma[i][t0:], self.slope, wsw.small, gof(nm + ".bias"), self.coef[i])
                    if i == 0 and self.wgrad_direct:    # straight from the tapes' images (they are intact until the caller joins this chain)
                        ns = ops.wgrad_img(dt, [self.imgs[t] for t in range(t0, t0 + T)], self._sl(self.dz[0], t0), wsw.slab, B, self.in_ch, self.S, self.S, self.W[0])
                    else:
                        ns = ops.conv_wgrad(geo, dt, self._inp(i, t0), self._sl(self.dz[i], t0), wsw.slab, wsw.wgs_target)
                    ops.wgrad_reduce_rank1(wsw.slab, ns, self.W[i], self.W[i], self.cin[i], self.taps if i > 0 else 1, gof(nm + ".weight_orig"), T,
                                           self.coef[i], self.u[i][t0:], self.v[i][t0:], self.k0 if i == 0 else 0)
                wgrad_side(layer_wgrad)
            fused = (0, None)
            if i > 0:
                if self.bns[i - 1] is not None:
                    ops.conv_bwd_data(geo, dt, self._sl(self.dz[i], t0), self.mid[i - 1].wp_bwd, self._sl(self.dyb[i - 1], t0), None)
                else:
                    kw = dict(sigma=self.sigma[i - 1][t0:], sigma_rows=self.rows(i), mask=self._sl(self.a[i - 1], t0), mask_act=ACT_LRELU, mask_slope=self.slope)
                    # ... and, from the same tile, layer i-1's bias-gradient column sums and spectral-norm coefficient (celeba._DiscEngine.backward)
                    fused = self._sn_stat(i - 1, T, geo, ops.epilogue(**kw)) if need_wgrad else (0, None)
                    if fused[0]:
                        kw.update(stat_mode=ops.STAT_SN_BIAS, stat_out=fused[1], stat_p=(self.convs[i - 1].bias,), stat_slope=self.slope)
                    ops.conv_bwd_data(geo, dt, self._sl(self.dz[i], t0), self.mid[i - 1].wp_bwd, self._sl(self.dz[i - 1], t0), ops.epilogue(**kw))
        if need_dimg:
            if self.l0g is not None and IMG_GEMM:
                ops.conv_fwd(self.l0g.c, dt, self._sl(self.dz[0], t0), self.l0g.wp_fwd, self.cols0, None)
                ops.col2im_img(dt, self.cols0, B, self.in_ch, self.l0g.H, self.l0g.W, self.k_img, 2, self.pad_img, None, ACT_NONE, 0.0, self.dimg)
            else:
                ops.conv_bwd_data(self.l0img.c, dt, self._sl(self.dz[0], t0), self.l0img.wp_bwd, self.dimg, ops.epilogue(out_mode=OUT_NCHW_F32))
            return self.dimg
        return None

    def backward(self, t0, T, douts, grad, need_wgrad=True, need_dimg=False, side=None):
        """douts: {head name: d(loss)/d(head output) [T*B, N] fp32} for every computed head (zeros where a head carries no
        loss).  Accumulates into flat ``grad``; returns d(loss)/d(img) of tape t0 if ``need_dimg``.  ``side`` (engine.SideStream): each
        layer's weight- / bias-gradient chain runs on a side lane; the caller joins the lanes before it reads ``grad``."""
        if self.has_bn and T > 1 and not BN_BATCH_TAPES:
            dimg = None
            for kk in range(T):
                sub = {k: v[kk * self.B:(kk + 1) * self.B] for k, v in douts.items()}
                r = self._bwd_pass(t0 + kk, 1, sub, grad, need_wgrad, need_dimg and kk == 0, side)
                dimg = r if kk == 0 else dimg
            return dimg
        return self._bwd_pass(t0, T, douts, grad, need_wgrad, need_dimg, side)
