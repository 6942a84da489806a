"""colored-dSprites 64x64x3 stage-2 path of EAD-GAN on MI355X (colored_dSprites/rp_color.py, utils_rp_color.py, utils_pxy.py):
the dSprites networks with 3 channels, 7 codes (theta,p,x,y + RGB gains c*.5+1), a 6-output frozen ``Encoder_pxy`` (p,x,y + RGB
gains c*.1+1), sprites coloured on the fly with U(.5,1) gains, and ``affine_color_regularzier``.  Reuses the dSprites engines."""
from __future__ import annotations

import argparse

import torch

from . import dsprites as ds
from . import ops
from .celeba import _require_cuda, transformation_2D          # noqa: F401
from .dsprites import get_matrix, get_matrix_D, get_matrix_pxy_align, mutual_info_loss, to_categorical      # noqa: F401

opt = argparse.Namespace(n_epochs=100, batch_size=128, lr=0.0002, b1=0.5, b2=0.999, n_cpu=8, latent_dim=200, code_dim=7, n_classes=3,
                         img_size=64, channels=3, sample_interval=1000)           # argparse defaults rp_color.py:40-51


class Encoder_pxy(ds.Encoder_pxy):
    def __init__(self, dtype="f32"):
        super().__init__(img_size=64, channels=3, n_out=6, dtype=dtype)


class Discriminator(ds.Discriminator):
    def __init__(self, dtype="f32"):
        super().__init__(img_size=64, channels=3, dtype=dtype)


class Generator(ds.Generator):
    def __init__(self, dtype="f32"):
        super().__init__(code_dim=7, n_classes=3, channels=3, dtype=dtype)


class Encoder(ds.Encoder):
    def __init__(self, dtype="f32"):
        super().__init__(code_dim=7, n_classes=3, img_size=64, channels=3, dtype=dtype)


def from_latent_vector_2_color_para(code_input_raw):
    """RGB gains c*.5+1 (utils_rp_color.py:38-48)."""
    return code_input_raw * 0.5 + 1


def from_latent_vector_2_color_para_pxy(code_input_raw):
    """alignment gains c*.1+1 (colored_dSprites/utils_pxy.py:48-57)."""
    return code_input_raw * 0.1 + 1


class _AffineColorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real_code, trans_code):
        B, ld = real_code.shape
        pred = torch.empty(B, 7, device=real_code.device)
        ops.loss_affine_rp_color(real_code, trans_code, ld, 0, B, torch.zeros(B, 7, device=real_code.device), 7, 1.0, None, None, None, pred)
        ctx.save_for_backward(real_code, trans_code, pred)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        real_code, trans_code, pred = ctx.saved_tensors
        B, ld = real_code.shape
        tgt = (pred - dpred.float() * (7.0 * B / 2.0)).contiguous()
        d_real, d_trans = torch.empty_like(real_code), torch.empty_like(trans_code)
        ops.loss_affine_rp_color(real_code, trans_code, ld, 0, B, tgt, 7, 1.0, None, d_real, d_trans, None)
        return d_real, d_trans


def affine_color_regularzier(real_code, trans_code):
    """utils_rp_color.py:100-139: 4 affine codes as dSprites + relative RGB gains."""
    _require_cuda(real_code)
    return _AffineColorFn.apply(real_code.float().contiguous(), trans_code.float().contiguous())


class ColoredTrainer(ds.DspritesTrainer):
    """One call == one iteration of colored_dSprites/rp_color.py:365-516 (both Adams lr 2e-4, :274-280)."""

    def __init__(self, encoder_pxy, generator, discriminator, encoder, batch_size, dtype="f32", allreduce=None, lrs=(2e-4, 2e-4), betas=(0.5, 0.999),
                 sync_bn=None, overlap=True):
        super().__init__(encoder_pxy, generator, discriminator, encoder, batch_size, dtype, allreduce, lrs, betas, sync_bn, overlap)
        dev = self.img.device
        self.tmp, self.tmp2 = torch.zeros_like(self.img), torch.zeros_like(self.img)
        self.gains = torch.zeros(batch_size, 3, device=dev)

    def _align(self):
        B = self.B
        pcode = self.pe.forward(self.img)                                   # [B,6] = p,x,y,r,g,b
        ops.theta_pxy_align_inv(pcode, 6, B, self.theta)
        ops.warp_affine(self.img, self.theta, self.tmp, B, 3, 64, 64)
        ops.color_scale(self.tmp, pcode, 6, 3, 0.1, True, self.align, B, 3, 64 * 64)      # :390-394

    def _transform(self, code, out, second=False):
        B = self.B
        theta, tmp = (self.theta2, self.tmp2) if second else (self.theta, self.tmp)
        ops.theta_rp(code, 7, B, theta)
        ops.warp_affine(self.align, theta, tmp, B, 3, 64, 64)
        ops.color_scale(tmp, code, 7, 4, 0.5, False, out, B, 3, 64 * 64)                  # :415-424

    def _affine_loss(self, cont_align, cont_trans, loss, d_align, d_trans):
        ops.loss_affine_rp_color(cont_align, cont_trans, 7, 0, self.B, self.code2, 7, 1.0, loss, d_align, d_trans)

    def load_inputs(self, sprites_u8, gains, code1, labels1, code2, labels2):
        """sprites_u8: uint8 [B,64,64]; gains: [B,3] U(.5,1) colour gains (rp_color.py:368-381)."""
        self.gains.copy_(gains.to(torch.float32))
        ops.u8_colorize(sprites_u8.contiguous(), self.gains, self.img, self.B, 3, 64 * 64)
        self.code1.copy_(code1)
        self.code2.copy_(code2)
        for oh, lab in ((self.onehot1, labels1), (self.onehot2, labels2)):
            oh.zero_()
            oh.scatter_(1, lab.view(-1, 1), 1.0)

    def train_step(self, sprites_u8, gains, code1, labels1, code2, labels2):
        self.load_inputs(sprites_u8, gains, code1, labels1, code2, labels2)
        l = self.step_resident().tolist()
        return dict(d_loss=l[0], g_loss=l[1], info_loss=l[2], affine_loss=l[3], relative_cat_loss=l[4])


# ---- stage-1 trainer of the colored Encoder_pxy (colored_dSprites/pxy_color.py:160-216) ----------------------------------------
class PxyColorTrainer(ds.PxyTrainer):
    """pxy_color.py loop: sprites repeated to 3 channels and multiplied by U(.5,1) gains (:166-177); 6-d code (p, x, y, r, g, b);
    trans = warp(img, get_matrix_pxy(code)) with ZERO padding (:90) times the colour gains 1 + .1 c[3:] (:196-201);
    affine_regularzier_pxy with the colour ratio (utils_pxy.py:150-176); Adam lr 2e-4 on Encoder_pxy(channels=3, n_out=6)."""

    def __init__(self, encoder_pxy, batch_size, dtype="f32", lr=2e-4, betas=(0.5, 0.999), allreduce=None):
        super().__init__(encoder_pxy, batch_size, dtype=dtype, lr=lr, betas=betas, allreduce=allreduce)
        dev = self.dev
        self.gains = torch.ones(batch_size, 3, device=dev, dtype=torch.float32)
        self.warped = torch.zeros(batch_size, 3, 64, 64, device=dev, dtype=torch.float32)

    def _make_image(self):
        ops.u8_colorize(self.img_u8, self.gains, self.img, self.B, 3, 64 * 64)

    def _transform(self):
        ops.warp_affine_zeros(self.img, self.theta, self.warped, self.B, 3, 64, 64)
        ops.color_scale(self.warped, self.code, self.nd, 3, 0.1, False, self.trans, self.B, 3, 64 * 64)

    def load_inputs(self, img_u8, gains, code):
        self.img_u8.copy_(img_u8, non_blocking=True)
        self.gains.copy_(gains.view(-1, 3), non_blocking=True)
        self.code.copy_(code, non_blocking=True)

    def train_step(self, img_u8, gains, code):
        """img_u8 uint8 [B,64,64]; gains [B,3] ~ U(.5,1); code [B,6] ~ U(-1,1) -> {'affine_loss'}"""
        self.load_inputs(img_u8, gains, code)
        return {"affine_loss": float(self.step_resident()[0])}


class DeviceInputs(ds.DeviceInputs):
    """colored_dSprites/rp_color.py:363-381: sprites repeated to three channels times per-image colour gains ~ U(.5, 1) (drawn FIRST,
    :372), then the dSprites draws (:405-446)."""

    def enqueue(self, tr: "ColoredTrainer"):
        self.begin_draws()                              # sprite indices, colour gains, codes and labels (+ one-hot rows): one launch
        idx = self.sample_indices(tr.B, 1)
        self.draw(ops.RNG_UNIFORM, tr.gains, 0.5, 1.0, 2)
        self.codes_and_labels(tr, 3)
        self.end_draws()
        ops.u8_colorize(self.sprites(tr, idx), tr.gains, tr.img, tr.B, 3, 64 * 64)
        self.tick()
