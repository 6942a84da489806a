"""CPU oracle for the dSprites hot path of EAD-GAN  --  TEST INFRASTRUCTURE ONLY (see oracle/celeba_oracle.py).

From-scratch torch-CPU fp32 restatement of ``dSprites/rp.py`` (stage-2 trainer), ``dSprites/utils_rp.py`` and the two
functions of ``dSprites/utils_pxy.py`` it uses.  Parity status: PINNED by ``tests/golden/dsprites_*.npz`` (reference loop
body run through the AST harness; the frozen ``encoder_pxy_50000.pt`` -- produced by the out-of-scope stage-1 trainer
pxy.py -- is a SEEDED stand-in; sprites are synthetic uint8 {0,1} images).

Reference citations (file:line relative to /root/reference):
  Encoder_pxy dSprites/rp.py:61-87   Discriminator :90-119   Generator :123-157   Encoder :160-194   transformation_2D :199-213
  mutual_info_loss :225-232   losses/optimizers :249-282   loop body :363-482
  get_matrix / get_matrix_D / affine_regularzier   dSprites/utils_rp.py:38-59,94-147     get_matrix_pxy_align   dSprites/utils_pxy.py:69-87
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .celeba_oracle import batchnorm_train, spectral_weight, warp

CODE, NCLS, IMG, CH = 4, 3, 64, 1            # argparse defaults rp.py:40-51 (latent_dim is unused by the generator)
CONV_IDX = (0, 2, 4, 6)


def _is_buffer(key):
    return key.endswith(("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v"))


def trainable(d):
    return [v for k, v in d.items() if not _is_buffer(k)]


def _trunk(ch, sn, slope):
    f = nn.utils.spectral_norm if sn else (lambda m: m)
    return nn.Sequential(f(nn.Conv2d(ch, 32, 4, 2, 1)), nn.LeakyReLU(slope), f(nn.Conv2d(32, 32, 4, 2, 1)), nn.LeakyReLU(slope),
                         f(nn.Conv2d(32, 64, 4, 2, 1)), nn.LeakyReLU(slope), f(nn.Conv2d(64, 64, 4, 2, 1)), nn.LeakyReLU(slope))


def _containers(ch=CH, code=CODE, ncls=NCLS, pxy_out=3):
    sn = nn.utils.spectral_norm
    P = nn.ModuleDict({"conv_block": _trunk(ch, False, 0.1), "fc1": nn.Linear(1024, pxy_out)})
    E = nn.ModuleDict({"conv_block": _trunk(ch, True, 0.2), "fc1": nn.Sequential(sn(nn.Linear(1024, 128))), "fc2": nn.Sequential(sn(nn.Linear(128, 128))),
                       "cat_layer": nn.Sequential(sn(nn.Linear(128, ncls))), "cont_layer": nn.Sequential(sn(nn.Linear(128, code)))})
    D = nn.ModuleDict({"conv_block": _trunk(ch, True, 0.2), "fc1": nn.Sequential(sn(nn.Linear(1024, 128))), "fc2": nn.Linear(128, 1)})
    gblock = nn.Sequential(nn.ConvTranspose2d(64, 64, 4, 2, 1), nn.BatchNorm2d(64), nn.ReLU(), nn.ConvTranspose2d(64, 64, 4, 2, 1), nn.BatchNorm2d(64),
                           nn.ReLU(), nn.ConvTranspose2d(64, 64, 4, 2, 1), nn.BatchNorm2d(64), nn.ReLU(), nn.ConvTranspose2d(64, ch, 4, 2, 1))
    G = nn.ModuleDict({"conv_block": gblock, "fc1": nn.Sequential(nn.Linear(ncls + code, 128)), "fc2": nn.Sequential(nn.Linear(128, 64 * 4 * 4))})
    return P, E, D, G


def _to_dict(m):
    d = OrderedDict()
    for k, v in m.state_dict().items():
        d[k] = v.detach().clone().contiguous()
        if v.dtype.is_floating_point and not _is_buffer(k):
            d[k].requires_grad_(True)
    return d


def init_state(seed=0, ch=CH, code=CODE, pxy_out=3):
    """(E, D, G) in the reference's construction order encoder_pxy, encoder, discriminator, generator (rp.py:253-256); the
    encoder_pxy constructor consumes the RNG first and is then overwritten from its checkpoint."""
    torch.manual_seed(seed)
    P, E, D, G = _containers(ch=ch, code=code, pxy_out=pxy_out)
    return _to_dict(E), _to_dict(D), _to_dict(G)


def make_encoder_pxy(seed=321, ch=CH, pxy_out=3):
    """Seeded stand-in for encoder_pxy_50000.pt (state dict of Encoder_pxy, rp.py:61-87)."""
    torch.manual_seed(seed)
    P = _containers(ch=ch, pxy_out=pxy_out)[0]
    return OrderedDict((k, v.detach().clone()) for k, v in P.state_dict().items())


# ---- networks ---------------------------------------------------------------------------------------------------------
def encoder_pxy_forward(P, img):
    x = img
    for i in CONV_IDX:
        x = F.leaky_relu(F.conv2d(x, P[f"conv_block.{i}.weight"], P[f"conv_block.{i}.bias"], 2, 1), 0.1)
    return F.linear(x.view(x.shape[0], -1), P["fc1.weight"], P["fc1.bias"])


def _sn_trunk(S, img):
    x = img
    for i in CONV_IDX:
        x = F.leaky_relu(F.conv2d(x, spectral_weight(S, f"conv_block.{i}."), S[f"conv_block.{i}.bias"], 2, 1), 0.2)
    return x.view(x.shape[0], -1)


def discriminator_logit(D, img):
    x = F.leaky_relu(F.linear(_sn_trunk(D, img), spectral_weight(D, "fc1.0."), D["fc1.0.bias"]), 0.2)
    return F.linear(x, D["fc2.weight"], D["fc2.bias"])


def discriminator_forward(D, img):
    return torch.sigmoid(discriminator_logit(D, img))


def encoder_logits(E, img):
    x = F.leaky_relu(F.linear(_sn_trunk(E, img), spectral_weight(E, "fc1.0."), E["fc1.0.bias"]), 0.2)
    x = F.leaky_relu(F.linear(x, spectral_weight(E, "fc2.0."), E["fc2.0.bias"]), 0.2)
    cat = F.linear(x, spectral_weight(E, "cat_layer.0."), E["cat_layer.0.bias"])
    cont = F.linear(x, spectral_weight(E, "cont_layer.0."), E["cont_layer.0.bias"])
    return cat, cont


def encoder_forward(E, img):
    cat, cont = encoder_logits(E, img)
    return F.softmax(cat, dim=1), cont


def generator_forward(G, z_c):
    x = F.relu(F.linear(z_c, G["fc1.0.weight"], G["fc1.0.bias"]))
    x = F.relu(F.linear(x, G["fc2.0.weight"], G["fc2.0.bias"])).view(z_c.shape[0], 64, 4, 4)
    for i in (0, 3, 6):
        x = F.conv_transpose2d(x, G[f"conv_block.{i}.weight"], G[f"conv_block.{i}.bias"], 2, 1)
        x = F.relu(batchnorm_train(x, G, f"conv_block.{i + 1}."))
    return torch.sigmoid(F.conv_transpose2d(x, G["conv_block.9.weight"], G["conv_block.9.bias"], 2, 1))


def mutual_info_loss(c_given_x, c):
    eps = 1e-8
    return torch.mean(-torch.sum(torch.log(c_given_x + eps) * c, dim=1)) + torch.mean(-torch.sum(torch.log(c + eps) * c, dim=1))


# ---- affine algebra ------------------------------------------------------------------------------------------------------
def get_matrix(code4):
    """A = Rot(theta) @ diag(p,p,1) @ Trans(x,y); theta = c0*pi/9, p = 1+.2c1, x,y = .1c  (utils_rp.py:94-115, == get_matrix_D :38-59)."""
    B = code4.shape[0]
    th, p, x, y = code4[:, 0] * np.pi / 9, code4[:, 1] * 0.2 + 1, code4[:, 2] * 0.1, code4[:, 3] * 0.1
    c, s = torch.cos(th), torch.sin(th)
    one, zero = torch.ones(B), torch.zeros(B)
    mk = lambda *e: torch.stack(e, 1).view(B, 3, 3)
    return mk(c, -s, zero, s, c, zero, zero, zero, one) @ mk(p, zero, zero, zero, p, zero, zero, zero, one) @ mk(one, zero, x, zero, one, y, zero, zero, one)


def get_matrix_pxy_align(code3):
    """translation only: T(x,y) with x,y = .1 c1, .1 c2  (utils_pxy.py:69-87)."""
    B = code3.shape[0]
    one, zero = torch.ones(B), torch.zeros(B)
    return torch.stack((one, zero, code3[:, 1] * 0.1, zero, one, code3[:, 2] * 0.1, zero, zero, one), 1).view(B, 3, 3)


def affine_regularzier(real_code, trans_code):
    rel = get_matrix(trans_code[:, :4]) @ torch.inverse(get_matrix(real_code[:, :4]))
    th = torch.atan((rel[:, 1, 0] - rel[:, 0, 1]) / (rel[:, 0, 0] + rel[:, 1, 1]))
    c, s = torch.cos(th), torch.sin(th)
    p = 0.5 * (c * (rel[:, 0, 0] + rel[:, 1, 1]) + s * (rel[:, 1, 0] - rel[:, 0, 1]))
    x = (rel[:, 0, 2] * c + rel[:, 1, 2] * s) / p
    y = (rel[:, 1, 2] * c - rel[:, 0, 2] * s) / p
    return torch.stack((th / np.pi * 9, (p - 1) / 0.2, x / 0.1, y / 0.1), dim=1).float()


# ---- training step ---------------------------------------------------------------------------------------------------------
class DspritesOracle:
    """optimizer_G exists in the reference but is never stepped (rp.py:276,417-419,480-482): only D (lr 2e-4) and info (G+E, lr 1e-4)."""

    def __init__(self, seed=0, pxy=None, lrs=(2e-4, 1e-4)):
        self.E, self.D, self.G = init_state(seed)
        self.P = pxy if pxy is not None else make_encoder_pxy()
        self.opt_D = torch.optim.Adam(trainable(self.D), lr=lrs[0], betas=(0.5, 0.999))
        self.opt_info = torch.optim.Adam(trainable(self.G) + trainable(self.E), lr=lrs[1], betas=(0.5, 0.999))

    def train_step(self, img_u8, code1, labels1, code2, labels2):
        """One loop body (rp.py:365-482).  img_u8: uint8 [B,64,64] sprites."""
        E, D, G, P = self.E, self.D, self.G, self.P
        img = img_u8.unsqueeze(1).float()
        B = img.shape[0]
        valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
        with torch.no_grad():                          # frozen; its gradients are dead work in the reference
            align = warp(img, torch.inverse(get_matrix_pxy_align(encoder_pxy_forward(P, img)))[:, 0:2])
        trans = warp(align, get_matrix(code1)[:, 0:2])
        gen = generator_forward(G, torch.cat((F.one_hot(labels1, NCLS).float(), code1), dim=1))
        d_real = discriminator_forward(D, trans)                 # real first, then fake: the order fixes which power iteration each one gets
        d_fake = discriminator_forward(D, gen.detach())
        d_loss = (F.binary_cross_entropy(d_fake, fake) + F.binary_cross_entropy(d_real, valid)) / 2
        self.opt_D.zero_grad()
        d_loss.backward()
        self.opt_D.step()
        onehot2 = F.one_hot(labels2, NCLS).float()
        gen = generator_forward(G, torch.cat((onehot2, code2), dim=1))
        rec_cat, rec_cont = encoder_forward(E, gen)
        g_loss = F.binary_cross_entropy(discriminator_forward(D, gen), valid)
        info_loss = mutual_info_loss(rec_cat, onehot2) + F.mse_loss(rec_cont, code2)
        trans2 = warp(align, get_matrix(code2)[:, 0:2])
        align_cat, align_cont = encoder_forward(E, align)
        trans_cat, trans_cont = encoder_forward(E, trans2)
        affine_loss = F.mse_loss(affine_regularzier(align_cont, trans_cont), code2)
        relative_cat_loss = mutual_info_loss(trans_cat, align_cat.detach())
        total = info_loss + affine_loss + g_loss + relative_cat_loss
        self.opt_info.zero_grad()
        for v in trainable(D):                         # the reference leaves dead gradients on D here; cleared so they cannot leak into tests
            v.grad = None
        total.backward()
        self.opt_info.step()
        return {k: float(v.detach()) for k, v in dict(d_loss=d_loss, g_loss=g_loss, info_loss=info_loss, affine_loss=affine_loss,
                                                     relative_cat_loss=relative_cat_loss).items()}


def draw_step_inputs(rng: np.random.RandomState, B: int, code_dim=CODE, ncls=NCLS):
    """reference order (rp.py:389-393,424-429): uniform code, randint labels, uniform code, randint labels (float64 -> float32)."""
    out = []
    for _ in range(2):
        out.append(torch.tensor(rng.uniform(-1, 1, (B, code_dim)), dtype=torch.float32))
        out.append(torch.tensor(rng.randint(0, ncls, B), dtype=torch.int64))
    return tuple(out)


def synthetic_sprites(B, seed=99):
    """uint8 {0,1} images [B,64,64]: a filled axis-aligned box per sample (stands in for the dSprites .npz, which is not in the container)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.zeros(B, IMG, IMG, dtype=torch.uint8)
    cx, cy = torch.randint(20, 44, (B,), generator=g), torch.randint(20, 44, (B,), generator=g)
    hw = torch.randint(4, 12, (B, 2), generator=g)
    for b in range(B):
        img[b, cy[b] - hw[b, 0]:cy[b] + hw[b, 0], cx[b] - hw[b, 1]:cx[b] + hw[b, 1]] = 1
    return img


# ---- colored dSprites (colored_dSprites/rp_color.py:363-516, utils_rp_color.py:24-139, utils_pxy.py:48-57,92-110) -----------------
def affine_color_regularzier(real_code, trans_code):
    aff = affine_regularzier(real_code, trans_code)
    rel = (trans_code[:, 4:7] * 0.5 + 1) / (real_code[:, 4:7] * 0.5 + 1)
    return torch.cat((aff, (rel - 1) / 0.5), dim=1).float()


class ColoredOracle:
    """3-channel variant: code = 4 affine + 3 RGB gains (c*.5+1); sprites are coloured with U(.5,1) gains (first np draw of the
    iteration); the frozen Encoder_pxy emits 6 codes (p,x,y + 3 gains c*.1+1) and the aligned image is divided by those gains;
    all Adams use lr 2e-4 (rp_color.py:42,274-280)."""

    def __init__(self, seed=0, pxy=None, lrs=(2e-4, 2e-4)):
        self.E, self.D, self.G = init_state(seed, ch=3, code=7, pxy_out=6)
        self.P = pxy if pxy is not None else make_encoder_pxy(ch=3, pxy_out=6)
        self.opt_D = torch.optim.Adam(trainable(self.D), lr=lrs[0], betas=(0.5, 0.999))
        self.opt_info = torch.optim.Adam(trainable(self.G) + trainable(self.E), lr=lrs[1], betas=(0.5, 0.999))

    def train_step(self, img_u8, gains, code1, labels1, code2, labels2):
        E, D, G, P = self.E, self.D, self.G, self.P
        img = (img_u8.unsqueeze(1).repeat(1, 3, 1, 1) * gains.double()[:, :, None, None]).float()
        B = img.shape[0]
        valid, fake = torch.ones(B, 1), torch.zeros(B, 1)
        with torch.no_grad():
            pcode = encoder_pxy_forward(P, img)
            align = warp(img, torch.inverse(get_matrix_pxy_align(pcode))[:, 0:2]) / (pcode[:, 3:] * 0.1 + 1)[:, :, None, None]
        col = lambda c: (c[:, 4:] * 0.5 + 1)[:, :, None, None]
        trans = warp(align, get_matrix(code1[:, :4])[:, 0:2]) * col(code1)
        gen = generator_forward(G, torch.cat((F.one_hot(labels1, NCLS).float(), code1), dim=1))
        d_real = discriminator_forward(D, trans)
        d_fake = discriminator_forward(D, gen.detach())
        d_loss = (F.binary_cross_entropy(d_fake, fake) + F.binary_cross_entropy(d_real, valid)) / 2
        self.opt_D.zero_grad()
        d_loss.backward()
        self.opt_D.step()
        onehot2 = F.one_hot(labels2, NCLS).float()
        gen = generator_forward(G, torch.cat((onehot2, code2), dim=1))
        rec_cat, rec_cont = encoder_forward(E, gen)
        g_loss = F.binary_cross_entropy(discriminator_forward(D, gen), valid)
        info_loss = mutual_info_loss(rec_cat, onehot2) + F.mse_loss(rec_cont, code2)
        trans2 = warp(align, get_matrix(code2[:, :4])[:, 0:2]) * col(code2)
        align_cat, align_cont = encoder_forward(E, align)
        trans_cat, trans_cont = encoder_forward(E, trans2)
        affine_loss = F.mse_loss(affine_color_regularzier(align_cont, trans_cont), code2)
        relative_cat_loss = mutual_info_loss(trans_cat, align_cat.detach())
        total = info_loss + affine_loss + relative_cat_loss + g_loss
        self.opt_info.zero_grad()
        for v in trainable(D):
            v.grad = None
        total.backward()
        self.opt_info.step()
        return {k: float(v.detach()) for k, v in dict(d_loss=d_loss, g_loss=g_loss, info_loss=info_loss, affine_loss=affine_loss,
                                                     relative_cat_loss=relative_cat_loss).items()}


def draw_colored_inputs(rng: np.random.RandomState, B: int):
    """rp_color.py:372-378 draws the colour gains first, then the dSprites sequence with code_dim 7."""
    gains = torch.tensor(rng.uniform(0.5, 1, [B, 3, 1, 1]).reshape(B, 3), dtype=torch.float64)
    return (gains,) + draw_step_inputs(rng, B, code_dim=7)


# ---- stage-1 trainer (dSprites/pxy.py): fits Encoder_pxy, whose checkpoint the stage-2 loop above loads -------------------------
def get_matrix_pxy(code3):
    """A = diag(p,p,1) @ Trans(x,y); p = 1 + .1 c0, x,y = .1 c1, .1 c2  (utils_pxy.py:24-34,49-66)."""
    B = code3.shape[0]
    p, x, y = code3[:, 0] * 0.1 + 1, code3[:, 1] * 0.1, code3[:, 2] * 0.1
    one, zero = torch.ones(B), torch.zeros(B)
    mk = lambda *e: torch.stack(e, 1).view(B, 3, 3)
    return mk(p, zero, zero, zero, p, zero, zero, zero, one) @ mk(one, zero, x, zero, one, y, zero, zero, one)


def affine_regularzier_pxy(real_code, trans_code):
    """utils_pxy.py:107-126: relative matrix -> (p, x, y) -> latent units."""
    rel = get_matrix_pxy(trans_code) @ torch.inverse(get_matrix_pxy(real_code))
    p = (rel[:, 0, 0] + rel[:, 1, 1]) / 2
    return torch.stack(((p - 1) / 0.1, rel[:, 0, 2] / p / 0.1, rel[:, 1, 2] / p / 0.1), dim=1).float()


class PxyOracle:
    """dSprites/pxy.py:156-191: E(img), E(warp(img, A(code))) -> affine_regularzier_pxy -> MSE vs the drawn code; Adam(lr 2e-4,
    betas (.5,.999)) on Encoder_pxy (:127).  Neither the image nor A carries a gradient, so the backward ends in the first conv."""

    def __init__(self, seed=0, lr=2e-4):
        torch.manual_seed(seed)                        # pxy.py:123 constructs Encoder_pxy first
        P = _containers(ch=CH, pxy_out=3)[0]
        self.P = _to_dict(P)
        self.opt = torch.optim.Adam(trainable(self.P), lr=lr, betas=(0.5, 0.999))

    def train_step(self, img_u8, code):
        img = img_u8.unsqueeze(1).float()
        real_code = encoder_pxy_forward(self.P, img)
        trans_img = warp(img, get_matrix_pxy(code)[:, 0:2])
        trans_code = encoder_pxy_forward(self.P, trans_img)
        loss = F.mse_loss(affine_regularzier_pxy(real_code, trans_code), code)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return {"affine_loss": float(loss.detach())}


def draw_pxy_inputs(rng: np.random.RandomState, B: int):
    """the loop's one numpy draw (pxy.py:166)"""
    return torch.from_numpy(rng.uniform(-1, 1, (B, 3))).float()


# ---- colored stage-1 trainer (colored_dSprites/pxy_color.py) -------------------------------------------------------------------
def warp_zeros(img, theta):
    """transformation_2D of pxy_color.py:86-92: grid_sample(padding_mode='zeros')"""
    grid = F.affine_grid(theta, list(img.shape), align_corners=False)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def affine_regularzier_pxy_color(real_code, trans_code):
    """colored_dSprites/utils_pxy.py:150-176: (p, x, y) as affine_regularzier_pxy + colour ratio (1+.1 t)/(1+.1 r) -> latent units."""
    aff = affine_regularzier_pxy(real_code[:, :3], trans_code[:, :3])
    col = ((trans_code[:, 3:] * 0.1 + 1) / (real_code[:, 3:] * 0.1 + 1) - 1) / 0.1
    return torch.cat((aff, col), dim=1).float()


class PxyColorOracle:
    """colored_dSprites/pxy_color.py:160-216: sprites x U(.5,1) colour gains; code 6-d; trans = warp_zeros(img, A(code[:3])) x
    (1 + .1 code[3:]); MSE(affine_regularzier_pxy(E(img), E(trans)), code); Adam(lr 2e-4, betas (.5,.999)) on Encoder_pxy(3 ch, 6 out)."""

    def __init__(self, seed=0, lr=2e-4):
        torch.manual_seed(seed)
        P = _containers(ch=3, pxy_out=6)[0]
        self.P = _to_dict(P)
        self.opt = torch.optim.Adam(trainable(self.P), lr=lr, betas=(0.5, 0.999))

    def train_step(self, img_u8, gains, code):
        img = (img_u8.unsqueeze(1).repeat(1, 3, 1, 1) * gains.view(-1, 3, 1, 1)).float()       # :166-177 (float64 product, then .float())
        real_code = encoder_pxy_forward(self.P, img)
        trans = warp_zeros(img, get_matrix_pxy(code[:, :3])[:, 0:2]) * (code[:, 3:] * 0.1 + 1).view(-1, 3, 1, 1)
        trans_code = encoder_pxy_forward(self.P, trans)
        loss = F.mse_loss(affine_regularzier_pxy_color(real_code, trans_code), code)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return {"affine_loss": float(loss.detach())}


def draw_pxy_color_inputs(rng: np.random.RandomState, B: int):
    """pxy_color.py:170-186: colour gains first, then the 6-d code"""
    gains = torch.tensor(rng.uniform(0.5, 1, [B, 3, 1, 1]).reshape(B, 3), dtype=torch.float64)
    code = torch.from_numpy(rng.uniform(-1, 1, (B, 6))).float()
    return gains, code
