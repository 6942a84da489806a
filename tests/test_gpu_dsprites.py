"""dSprites hot path on the MI355X vs the CPU oracle (oracle/dsprites_oracle.py, pinned to the reference by
tests/golden/dsprites_b8_s3.npz)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import dsprites_oracle as do

pytestmark = pytest.mark.gpu
DEV = "cuda"
eg = None
PRE_BN_BIAS = ("conv_block.0.bias", "conv_block.3.bias", "conv_block.6.bias")
NAMES = ("d_loss", "g_loss", "info_loss", "affine_loss", "relative_cat_loss")


def setup_module(module):
    global eg
    eg = importlib.import_module("ead-gan_amd")
    torch.set_num_threads(16)


def rel_err(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(seed, dtype, lrs=(2e-4, 1e-4), pxy_seed=321):
    pxy = do.make_encoder_pxy(pxy_seed)
    orc = do.DspritesOracle(seed=seed, pxy=pxy, lrs=lrs)
    ds = eg.dsprites
    P, G, D, E = ds.Encoder_pxy(dtype=dtype).to(DEV), ds.Generator(dtype=dtype).to(DEV), ds.Discriminator(dtype=dtype).to(DEV), ds.Encoder(dtype=dtype).to(DEV)
    for m, ref in ((P, pxy), (G, orc.G), (D, orc.D), (E, orc.E)):
        assert list(m.state_dict().keys()) == list(ref.keys())
        m.load_state_dict({k: v.detach() for k, v in ref.items()})
    return orc, P, G, D, E


def test_affine_utils_and_encoder_pxy():
    ds = eg.dsprites
    orc, P, G, D, E = build(0, "f32")
    g = torch.Generator().manual_seed(2)
    code = torch.rand(16, 4, generator=g) * 2 - 1
    np.testing.assert_allclose(ds.get_matrix(code.to(DEV)).cpu().numpy(), do.get_matrix(code).numpy(), rtol=2e-6, atol=2e-7)
    c3 = torch.rand(16, 3, generator=g) * 2 - 1
    np.testing.assert_allclose(ds.get_matrix_pxy_align(c3.to(DEV)).cpu().numpy(), do.get_matrix_pxy_align(c3).numpy(), rtol=2e-6, atol=2e-7)
    rc = (torch.rand(16, 4, generator=g) * 2 - 1).requires_grad_(True)
    tc = (torch.rand(16, 4, generator=g) * 2 - 1).requires_grad_(True)
    w = torch.randn(16, 4, generator=g)
    want = do.affine_regularzier(rc, tc)
    (want * w).sum().backward()
    rcd, tcd = rc.detach().to(DEV).requires_grad_(True), tc.detach().to(DEV).requires_grad_(True)
    got = ds.affine_regularzier(rcd, tcd)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-4, atol=2e-5)
    (got * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(rcd.grad.cpu().numpy(), rc.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(tcd.grad.cpu().numpy(), tc.grad.numpy(), rtol=2e-3, atol=2e-4)
    img = do.synthetic_sprites(8).unsqueeze(1).float()
    assert rel_err(P(img.to(DEV)), do.encoder_pxy_forward(orc.P, img)) < 2e-5


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 3e-2)])
def test_generator_forward_backward(dtype, tol):
    B = 8
    orc, P, G, D, E = build(1, dtype)
    g = torch.Generator().manual_seed(5)
    code = torch.rand(B, 4, generator=g) * 2 - 1
    onehot = F.one_hot(torch.randint(0, 3, (B,), generator=g), 3).float()
    want = do.generator_forward(orc.G, torch.cat((onehot, code), dim=1))
    dimg = torch.randn(want.shape, generator=g) * 1e-2
    want.backward(dimg)
    ge = G.engine(B)
    got = ge.forward(onehot.to(DEV), code.to(DEV))
    assert got.shape == (B, 1, 64, 64) and rel_err(got, want) < tol
    grad = torch.zeros_like(G.arena.grad)
    ge.backward(dimg.to(DEV), grad)
    for k in dict(G.named_parameters()):
        if k in PRE_BN_BIAS:
            continue
        off, n = G.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.G[k].grad) < tol * 20, k


def test_discriminator_and_encoder_tapes_fp32():
    B = 8
    orc, P, G, D, E = build(2, "f32")
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(B, 1, 64, 64, generator=g) for _ in range(3)]
    leaves = [im.clone().requires_grad_(True) for im in imgs]
    # discriminator: two tapes batched
    dd = [torch.randn(B, 1, generator=g) for _ in range(2)]
    sum((do.discriminator_logit(orc.D, leaves[t]) * dd[t]).sum() for t in range(2)).backward()
    de = D.engine(B)
    out = de.forward([imgs[0].to(DEV), imgs[1].to(DEV)])["fc2"]
    grad = torch.zeros_like(D.arena.grad)
    dimg = de.backward(0, 2, {"fc2": torch.cat(dd).to(DEV).contiguous()}, grad, need_wgrad=True, need_dimg=True)
    assert rel_err(dimg, leaves[0].grad) < 3e-4
    for k in dict(D.named_parameters()):
        off, n = D.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.D[k].grad) < 3e-4, k
    # encoder: three tapes batched
    leaves = [im.clone().requires_grad_(True) for im in imgs]
    dc = [torch.randn(B, 3, generator=g) for _ in range(3)]
    dn = [torch.randn(B, 4, generator=g) for _ in range(3)]
    tot, ref = 0, []
    for t in range(3):
        cat, cont = do.encoder_logits(orc.E, leaves[t])
        ref.append((cat, cont))
        tot = tot + (cat * dc[t]).sum() + (cont * dn[t]).sum()
    tot.backward()
    ee = E.engine(B)
    outs = ee.forward([im.to(DEV) for im in imgs])
    for t in range(3):
        assert rel_err(outs["cat_layer.0"][t * B:(t + 1) * B], ref[t][0]) < 2e-5
        assert rel_err(outs["cont_layer.0"][t * B:(t + 1) * B], ref[t][1]) < 2e-5
    grad = torch.zeros_like(E.arena.grad)
    dimg = ee.backward(0, 3, {"cat_layer.0": torch.cat(dc).to(DEV).contiguous(), "cont_layer.0": torch.cat(dn).to(DEV).contiguous()}, grad,
                       need_wgrad=True, need_dimg=True)
    assert rel_err(dimg, leaves[0].grad) < 3e-4
    for k in dict(E.named_parameters()):
        off, n = E.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.E[k].grad) < 3e-4, k
    for k in ("fc1.0.weight_u", "fc2.0.weight_v", "cat_layer.0.weight_u", "conv_block.6.weight_v"):
        assert rel_err(E.state_dict()[k], orc.E[k]) < 1e-4, k


def run_steps(dtype, B, steps, seed=0, lrs=(2e-4, 1e-4)):
    orc, P, G, D, E = build(seed, dtype, lrs)
    tr = eg.dsprites.DspritesTrainer(P, G, D, E, B, dtype=dtype, lrs=lrs)
    rng = np.random.RandomState(seed)
    sprites = do.synthetic_sprites(B * steps, seed=99).view(steps, B, 64, 64)
    got, want = [], []
    for i in range(steps):
        inp = do.draw_step_inputs(rng, B)
        got.append(tr.train_step(sprites[i].to(DEV), *[t.to(DEV) for t in inp]))
        want.append(orc.train_step(sprites[i], *inp))
    return orc, G, D, E, tr, got, want


def test_train_step_fp32_and_golden():
    gold = np.load(os.path.join(GOLDEN, "dsprites_b8_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    orc, G, D, E, tr, got, want = run_steps("f32", B, steps, seed=seed)
    for k in NAMES:
        t0 = 2e-5 if k == "d_loss" else 3e-4              # everything but d_loss sits behind the D update of the same iteration
        assert abs(got[0][k] - want[0][k]) < t0, (k, got[0][k], want[0][k])
        assert abs(got[0][k] - gold[k][0]) < t0, (k, got[0][k], gold[k][0])
    for i in (1, 2):
        for k in NAMES:
            assert abs(got[i][k] - gold[k][i]) < 5e-2, (i, k, got[i][k], gold[k][i])


def test_train_step_gradients_lr0_fp32():
    orc, G, D, E, tr, got, want = run_steps("f32", 8, 1, seed=3, lrs=(0.0, 0.0))
    for k in NAMES:
        assert abs(got[0][k] - want[0][k]) < 2e-5, (k, got[0][k], want[0][k])
    for mod, ref in ((G, orc.G), (E, orc.E)):
        for k, p in mod.named_parameters():
            if k in PRE_BN_BIAS:
                continue
            assert rel_err(p.grad, ref[k].grad) < 2e-2, k


def test_train_step_bf16_tracks_oracle():
    orc, G, D, E, tr, got, want = run_steps("bf16", 16, 2)
    for i in range(2):
        for k in NAMES:
            assert abs(got[i][k] - want[i][k]) < 5e-2 * max(1.0, abs(want[i][k])), (i, k, got[i][k], want[i][k])


def test_graph_replay_equals_eager():
    """eager two-chain step == its hipGraph replay == the serial (one stream) body, bit for bit: losses and parameters after 3 steps"""
    B = 8
    res = []
    for capture, overlap in ((False, True), (True, True), (False, False)):
        orc, P, G, D, E = build(7, "f32")
        tr = eg.dsprites.DspritesTrainer(P, G, D, E, B, dtype="f32", overlap=overlap)
        assert tr.overlap == overlap
        rng = np.random.RandomState(1)
        sprites = do.synthetic_sprites(B, seed=5).to(DEV)
        out = []
        for i in range(3):
            tr.load_inputs(sprites, *[t.to(DEV) for t in do.draw_step_inputs(rng, B)])
            if capture and i == 1:
                tr.capture()
            out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        res.append((torch.stack(out).cpu(), G.arena.flat.clone().cpu(), D.arena.flat.clone().cpu(), E.arena.flat.clone().cpu(),
                    G.state_dict()["conv_block.1.running_mean"].clone().cpu()))
    for r in res[1:]:
        for a, b in zip(r, res[0]):
            assert torch.equal(a, b)


def test_teacher_forced_loss_curve_100_steps_fp32():
    """100 consecutive iterations of the loop body (dSprites/rp.py:363-482), each started from the oracle's state (parameters, buffers, Adam moments and
    step counts): every loss within 1e-3 of the oracle's (tests/teacher_forced.py; profiles/scripts/teacher_forced_curve.py runs 1000)."""
    import teacher_forced
    dev, names = teacher_forced.curve("dsprites", 100)
    assert dev.max() < 1e-3, (names, dev.max(axis=0), np.argmax(dev, axis=0))
