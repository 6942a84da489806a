"""CPU oracle for the CelebA hot path of EAD-GAN  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch-CPU fp32, functional style) of the
algorithm in the reference's ``celebA/EAD-GAN_celebA.py`` and ``celebA/utils_rpqxy.py``.
It is the *checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path (``ead-gan_amd``) never does.

Parity status: PINNED.  ``oracle/make_golden.py`` runs the reference's own loop body (via an AST
harness, in the build container) on seeded synthetic batches and stores losses / parameter probes in
``tests/golden/celeba_*.npz``; ``tests/test_oracle_golden.py`` replays the same draws through this
file and compares.

Reference citations (file:line relative to /root/reference):
  Generator                celebA/EAD-GAN_celebA.py:67-102
  Discriminator            celebA/EAD-GAN_celebA.py:105-138
  transformation_2D        celebA/EAD-GAN_celebA.py:144-158
  losses / lambdas         celebA/EAD-GAN_celebA.py:161-169
  optimizers               celebA/EAD-GAN_celebA.py:211-217
  loop body                celebA/EAD-GAN_celebA.py:297-401
  get_matrix               celebA/utils_rpqxy.py:59-80
  affine_regularzier       celebA/utils_rpqxy.py:82-116
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

LATENT, CODE, NCLS, IMG, CH = 200, 8, 10, 64, 3      # argparse defaults, EAD-GAN_celebA.py:39-51
G_WIDTHS = (1024, 512, 256, 128)                      # :76-90
D_WIDTHS = (128, 256, 512, 1024)                      # :110-120
SN_EPS = 1e-12                                        # torch spectral_norm default
LRELU = 0.1                                           # :111


# ----------------------------------------------------------------------------------------------
# parameter construction (torch default init, reference construction order: G then D, :172-173)
# ----------------------------------------------------------------------------------------------
def init_state(seed: int = 0):
    """Returns (G, D) ordered dicts keyed exactly like the reference's state_dict()s."""
    torch.manual_seed(seed)
    G = OrderedDict()
    cin = LATENT + CODE + NCLS
    # conv_blocks indices in the reference Sequential: 0 | 1,2,(3) | 4,5,(6) | 7,8,(9) | 10,(11)
    layer = nn.ConvTranspose2d(cin, G_WIDTHS[0], 4, 1, 0)
    G["conv_blocks.0.weight"], G["conv_blocks.0.bias"] = layer.weight.detach(), layer.bias.detach()
    idx = 1
    for a, b in zip(G_WIDTHS[:-1], G_WIDTHS[1:]):
        layer = nn.ConvTranspose2d(a, b, 4, 2, 1)
        bn = nn.BatchNorm2d(b)
        G[f"conv_blocks.{idx}.weight"], G[f"conv_blocks.{idx}.bias"] = layer.weight.detach(), layer.bias.detach()
        for k, v in bn.state_dict().items():
            G[f"conv_blocks.{idx + 1}.{k}"] = v.clone()
        idx += 3
    layer = nn.ConvTranspose2d(G_WIDTHS[-1], CH, 4, 2, 1)
    G[f"conv_blocks.{idx}.weight"], G[f"conv_blocks.{idx}.bias"] = layer.weight.detach(), layer.bias.detach()

    D = OrderedDict()
    cprev = CH
    for i, c in enumerate(D_WIDTHS):
        m = nn.utils.spectral_norm(nn.Conv2d(cprev, c, 4, 2, 1))
        sd = m.state_dict()
        for k in ("bias", "weight_orig", "weight_u", "weight_v"):
            D[f"main.{2 * i}.{k}"] = sd[k].clone()
        cprev = c
    m = nn.Conv2d(cprev, 1 + NCLS + CODE, 4, 1, 0)
    D["main.8.weight"], D["main.8.bias"] = m.weight.detach().clone(), m.bias.detach().clone()
    for d in (G, D):
        for k, v in d.items():
            d[k] = v.clone().contiguous()
            if v.dtype.is_floating_point and not _is_buffer(k):
                d[k].requires_grad_(True)
    return G, D


def _is_buffer(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v"))


def trainable(d):
    """Parameters in reference ``.parameters()`` order (registration order == dict order)."""
    return [v for k, v in d.items() if not _is_buffer(k)]


# ----------------------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------------------
def batchnorm_train(x, d, prefix, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d in training mode (biased var to normalise, unbiased var into running_var)."""
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    with torch.no_grad():
        d[prefix + "running_mean"].mul_(1 - momentum).add_(momentum * mean)
        d[prefix + "running_var"].mul_(1 - momentum).add_(momentum * var * n / (n - 1))
        d[prefix + "num_batches_tracked"] += 1
    xhat = (x - mean[None, :, None, None]) * torch.rsqrt(var + eps)[None, :, None, None]
    return xhat * d[prefix + "weight"][None, :, None, None] + d[prefix + "bias"][None, :, None, None]


def spectral_weight(d, prefix, training=True):
    """Hook-style torch spectral_norm: one power iteration (no grad, in place on u/v), then W/sigma
    with gradient flowing through sigma (u, v treated as constants)."""
    w = d[prefix + "weight_orig"]
    u, v = d[prefix + "weight_u"], d[prefix + "weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=SN_EPS))
    uc, vc = u.clone(), v.clone()
    sigma = torch.dot(uc, torch.mv(wm, vc))
    return w / sigma


def generator_forward(G, noise, labels, code, taps=None):
    """EAD-GAN_celebA.py:95-102.  First ConvT has no BN and no activation (:76-78).  ``taps`` (a list) receives the three
    post-ReLU activations (tests use their signs to tell rounding-induced ReLU flips from real gradient errors)."""
    x = torch.cat((noise, labels, code), -1)
    x = x.view(x.size(0), x.size(1), 1, 1)
    x = F.conv_transpose2d(x, G["conv_blocks.0.weight"], G["conv_blocks.0.bias"], 1, 0)
    idx = 1
    for _ in range(3):
        x = F.conv_transpose2d(x, G[f"conv_blocks.{idx}.weight"], G[f"conv_blocks.{idx}.bias"], 2, 1)
        x = batchnorm_train(x, G, f"conv_blocks.{idx + 1}.")
        x = F.relu(x)
        if taps is not None:
            taps.append(x.detach())
        idx += 3
    x = F.conv_transpose2d(x, G[f"conv_blocks.{idx}.weight"], G[f"conv_blocks.{idx}.bias"], 2, 1)
    return torch.tanh(x)


def discriminator_forward(D, img, training=True):
    """EAD-GAN_celebA.py:126-138 -> (cat, cont, validity)."""
    x = img
    for i in range(4):
        w = spectral_weight(D, f"main.{2 * i}.", training)
        x = F.leaky_relu(F.conv2d(x, w, D[f"main.{2 * i}.bias"], 2, 1), LRELU)
    out = F.conv2d(x, D["main.8.weight"], D["main.8.bias"], 1, 0).squeeze()
    validity = torch.sigmoid(out[:, 0])
    cat = F.softmax(out[:, CODE + 1: CODE + 1 + NCLS], dim=1)
    cont = out[:, 1: CODE + 1]
    return cat, cont, validity


def warp(img, theta):
    """transformation_2D.stn (:146-152): affine_grid + bilinear grid_sample, border padding,
    align_corners=False (torch default)."""
    grid = F.affine_grid(theta, list(img.shape), align_corners=False)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=False)


# ----------------------------------------------------------------------------------------------
# affine code algebra  (celebA/utils_rpqxy.py)
# ----------------------------------------------------------------------------------------------
R_FACTOR, PQ_FACTOR, XY_FACTOR = 9.0, 0.2, 0.1        # utils_rpqxy.py:27-29


def latent_to_affine(c):
    """utils_rpqxy.py:25-38: theta = c0*pi/9, p,q = 1+0.2c, x,y = 0.1c."""
    return torch.stack((c[:, 0] * np.pi / R_FACTOR, c[:, 1] * PQ_FACTOR + 1, c[:, 2] * PQ_FACTOR + 1,
                        c[:, 3] * XY_FACTOR, c[:, 4] * XY_FACTOR), dim=1)


def affine_to_latent(a):
    """utils_rpqxy.py:41-55."""
    return torch.stack((a[:, 0] / np.pi * R_FACTOR, (a[:, 1] - 1) / PQ_FACTOR, (a[:, 2] - 1) / PQ_FACTOR,
                        a[:, 3] / XY_FACTOR, a[:, 4] / XY_FACTOR), dim=1)


def get_matrix(code5):
    """utils_rpqxy.py:59-80:  A = Rot(theta) @ diag(p,q,1) @ Trans(x,y)   -> [B,3,3]."""
    a = latent_to_affine(code5)
    B = code5.shape[0]
    c, s = torch.cos(a[:, 0]), torch.sin(a[:, 0])
    one, zero = torch.ones(B), torch.zeros(B)
    rot = torch.stack((c, -s, zero, s, c, zero, zero, zero, one), 1).view(B, 3, 3)
    zoom = torch.stack((a[:, 1], zero, zero, zero, a[:, 2], zero, zero, zero, one), 1).view(B, 3, 3)
    trn = torch.stack((one, zero, a[:, 3], zero, one, a[:, 4], zero, zero, one), 1).view(B, 3, 3)
    return rot @ zoom @ trn


def affine_regularzier(real_code, trans_code):
    """utils_rpqxy.py:82-116 (closed form).  Spelling follows the reference."""
    rel = get_matrix(trans_code[:, :5]) @ torch.inverse(get_matrix(real_code[:, :5]))
    t1 = rel[:, 0, 0] * rel[:, 1, 0] - rel[:, 0, 1] * rel[:, 1, 1]
    t2 = rel[:, 0, 0] ** 2 + rel[:, 1, 1] ** 2 - rel[:, 0, 1] ** 2 - rel[:, 1, 0] ** 2
    th = 0.5 * torch.atan(2 * t1 / t2)
    c, s = torch.cos(th), torch.sin(th)
    p = rel[:, 0, 0] * c + rel[:, 1, 0] * s
    q = -rel[:, 0, 1] * s + rel[:, 1, 1] * c
    x = (rel[:, 0, 2] * c + rel[:, 1, 2] * s) / p
    y = (rel[:, 1, 2] * c - rel[:, 0, 2] * s) / q
    return affine_to_latent(torch.stack((th, p, q, x, y), dim=1)).float()


# ----------------------------------------------------------------------------------------------
# training step
# ----------------------------------------------------------------------------------------------
class CelebAOracle:
    """Holds G/D state + the three Adams (:211-217: lr 1e-3 / 2e-4 / 2e-4, betas (.5,.999))."""

    def __init__(self, seed=0, G=None, D=None, lrs=(1e-3, 2e-4, 2e-4)):
        if G is None:
            G, D = init_state(seed)
        self.G, self.D = G, D
        gp, dp = trainable(G), trainable(D)
        self.opt_G = torch.optim.Adam(gp, lr=lrs[0], betas=(0.5, 0.999))
        self.opt_D = torch.optim.Adam(dp, lr=lrs[1], betas=(0.5, 0.999))
        self.opt_info = torch.optim.Adam(gp + dp, lr=lrs[2], betas=(0.5, 0.999))

    def train_step(self, real_imgs, z, code, labels):
        """One loop body, EAD-GAN_celebA.py:299-401.  ``labels`` int64 [B]; returns dict of floats."""
        G, D = self.G, self.D
        B = real_imgs.shape[0]
        valid, fake = torch.ones(B), torch.zeros(B)                       # :302-303 (1-D)
        onehot = F.one_hot(labels, NCLS).float()                          # to_categorical :56-62
        A = get_matrix(code[:, :5])                                       # :325
        scaled = warp(real_imgs, A[:, 0:2])                               # :327
        # 1) generator adversarial step  :334-345
        self.opt_G.zero_grad()
        gen = generator_forward(G, z, onehot, code)
        _, _, validity = discriminator_forward(D, gen)
        g_loss = F.binary_cross_entropy(validity, valid)
        g_loss.backward()
        self.opt_G.step()
        # 2) discriminator step  :353-366
        self.opt_D.zero_grad()
        _, _, real_pred = discriminator_forward(D, scaled)
        _, _, fake_pred = discriminator_forward(D, gen.detach())
        d_loss = (F.binary_cross_entropy(real_pred, valid) + F.binary_cross_entropy(fake_pred, fake)) / 2
        d_loss.backward()
        self.opt_D.step()
        # 3) info + affine step  :375-401   (softmaxed probabilities fed to CrossEntropy, :383)
        self.opt_info.zero_grad()
        gen = generator_forward(G, z, onehot, code)
        pred_label, pred_code, _ = discriminator_forward(D, gen)
        info1 = F.cross_entropy(pred_label, labels) + F.mse_loss(pred_code, code)
        _, trans_code, _ = discriminator_forward(D, scaled)
        _, real_code, _ = discriminator_forward(D, real_imgs)
        pred_aff = affine_regularzier(real_code, trans_code)
        info_loss = info1 + F.mse_loss(pred_aff, code[:, :5])
        info_loss.backward()
        self.opt_info.step()
        return {"g_loss": float(g_loss), "d_loss": float(d_loss), "info_loss": float(info_loss)}


def draw_step_inputs(rng: np.random.RandomState, B: int):
    """Host draws in the reference's order (:308-317): normal z -> uniform code -> randint labels,
    float64 -> float32 like ``FloatTensor(np_array)``."""
    z = torch.tensor(rng.normal(0, 1, (B, LATENT)), dtype=torch.float32)
    code = torch.tensor(rng.uniform(-1, 1, (B, CODE)), dtype=torch.float32)
    labels = torch.tensor(rng.randint(0, NCLS, B), dtype=torch.int64)
    return z, code, labels


def synthetic_real(B: int, seed: int = 1234):
    """U(-1,1) image batch [B,3,64,64] standing in for normalised CelebA crops."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand((B, CH, IMG, IMG), generator=g) * 2 - 1
