"""MNIST hot path on the MI355X vs the CPU oracle (oracle/mnist_oracle.py, pinned to the reference by
tests/golden/mnist_b8_s3.npz and mnist_affine.npz)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import mnist_oracle as mo

pytestmark = pytest.mark.gpu
DEV = "cuda"
eg = None
PRE_BN_BIAS = ("l1.0.bias", "conv_blocks.2.bias", "conv_blocks.6.bias")      # layers feeding a BatchNorm: zero gradient up to rounding


def setup_module(module):
    global eg
    eg = importlib.import_module("ead-gan_amd")
    torch.set_num_threads(16)


def rel_err(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(seed, dtype, lrs=None, mlp_seed=123):
    mlp = mo.make_approximator(mlp_seed)
    orc = mo.MnistOracle(seed=seed, mlp=mlp, lrs=lrs)
    eg.mnist.load_approximator(mlp)
    G, D, E = eg.mnist.Generator(dtype=dtype).to(DEV), eg.mnist.Discriminator(dtype=dtype).to(DEV), eg.mnist.Encoder(dtype=dtype).to(DEV)
    for m, ref in ((G, orc.G), (D, orc.D), (E, orc.E)):
        assert list(m.state_dict().keys()) == list(ref.keys())
        m.load_state_dict({k: v.detach() for k, v in ref.items()})
    return orc, G, D, E


def test_affine_utils_match_reference_golden():
    gold = np.load(os.path.join(GOLDEN, "mnist_affine.npz"))
    eg.mnist.load_approximator(mo.make_approximator(int(gold["mlp_seed"])))
    A = eg.mnist.get_matrix(torch.tensor(gold["code"]).to(DEV))
    np.testing.assert_allclose(A.cpu().numpy(), gold["A"], rtol=2e-6, atol=2e-7)
    rc = torch.tensor(gold["real_code"], device=DEV, requires_grad=True)
    tc = torch.tensor(gold["trans_code"], device=DEV, requires_grad=True)
    pred = eg.mnist.affine_regularizer(rc, tc)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), gold["pred"], rtol=2e-4, atol=5e-5)
    (pred * torch.tensor(gold["w"], device=DEV)).sum().backward()
    np.testing.assert_allclose(rc.grad.cpu().numpy(), gold["d_real"], rtol=2e-3, atol=3e-4)
    np.testing.assert_allclose(tc.grad.cpu().numpy(), gold["d_trans"], rtol=2e-3, atol=3e-4)


def test_sumpool_and_gather_add():
    ops = eg.ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 8, 8, 16, generator=g)
    for dtype in (0, 1):
        xd = x.to(DEV).to(ops.torch_dtype(dtype))
        y = torch.empty(3, 4, 4, 16, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.sumpool2x2(dtype, xd, y, 3, 4, 4, 16)
        want = xd.float().cpu().view(3, 4, 2, 4, 2, 16).sum(dim=(2, 4))
        torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2 if dtype else 1e-6, atol=2e-2 if dtype else 1e-6)
    src = torch.randn(8192, generator=g)
    out = torch.ones(8192, device=DEV)
    ops.gather_add(out, src.to(DEV), 8192, 64, 1, 128)
    f = torch.arange(8192)
    torch.testing.assert_close(out.cpu() - 1, src[(f // 64) + (f % 64) * 128])


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 3e-2)])
def test_generator_forward_backward(dtype, tol):
    B = 8
    orc, G, D, E = build(1, dtype)
    rng = np.random.RandomState(5)
    z, code, labels = mo.draw_step_inputs(rng, B)
    onehot = F.one_hot(labels, 10).float()
    taps = []
    want = mo.generator_forward(orc.G, z, onehot, code, taps)
    dimg = torch.randn(want.shape, generator=torch.Generator().manual_seed(3)) * 1e-2
    want.backward(dimg)
    ge = G.engine(B)
    got = ge.forward(z.to(DEV), onehot.to(DEV), code.to(DEV))
    assert got.shape == (B, 1, 32, 32)
    assert rel_err(got, want) < tol
    grad = torch.zeros_like(G.arena.grad)
    ge.backward(dimg.to(DEV), grad)
    # a BatchNorm output within rounding of 0 can take the other LeakyReLU branch than in torch (summation order); one such unit
    # moves the upstream gradients by ~1e-3 in fp32 -> the tight bound holds when the masks agree, an allowance per flip otherwise
    flips = sum(int(((a.permute(0, 3, 1, 2).float().cpu() > 0) != (t > 0)).sum()) for a, t in zip((ge.a1, ge.a2), taps)) if dtype == "f32" else 0
    assert flips <= 3, flips
    for k in dict(G.named_parameters()):
        if k in PRE_BN_BIAS:
            continue
        off, n = G.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.G[k].grad) < tol * 20 + 4e-3 * flips, (k, flips)
    for k in ("conv_blocks.0.running_mean", "conv_blocks.3.running_var", "conv_blocks.7.running_mean"):
        assert rel_err(G.state_dict()[k], orc.G[k]) < max(tol, 1e-4), k


def test_discriminator_two_tapes_fp32():
    B = 8
    orc, G, D, E = build(2, "f32")
    de = D.engine(B)
    imgs = [mo.synthetic_real(B, seed=s) for s in (1, 2)]
    g = torch.Generator().manual_seed(0)
    douts = [torch.randn(B, 1, generator=g) for _ in range(2)]
    leaves = [im.clone().requires_grad_(True) for im in imgs]
    total = sum((mo.discriminator_forward(orc.D, leaves[t]) * douts[t]).sum() for t in range(2))
    total.backward()
    out = de.forward([im.to(DEV) for im in imgs])["adv_layer.0"]
    assert out.shape == (2 * B, 1)
    grad = torch.zeros_like(D.arena.grad)
    dimg = de.backward(0, 2, {"adv_layer.0": torch.cat(douts).to(DEV).contiguous()}, grad, need_wgrad=True, need_dimg=True)
    assert rel_err(dimg, leaves[0].grad) < 2e-4
    for k in dict(D.named_parameters()):
        off, n = D.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.D[k].grad) < 2e-4, k
    for k in ("conv_blocks.0.weight_u", "conv_blocks.6.weight_v", "adv_layer.0.weight_u", "adv_layer.0.weight_v"):
        assert rel_err(D.state_dict()[k], orc.D[k]) < 1e-4, k


def test_encoder_three_tapes_fp32():
    B = 8
    orc, G, D, E = build(3, "f32")
    ee = E.engine(B)
    imgs = [mo.synthetic_real(B, seed=s) for s in (4, 5, 6)]
    g = torch.Generator().manual_seed(0)
    dcat = [torch.randn(B, 10, generator=g) for _ in range(3)]
    dlat = [torch.randn(B, 7, generator=g) for _ in range(3)]
    leaves = [im.clone().requires_grad_(True) for im in imgs]
    total, ref_out = 0, []
    for t in range(3):
        x = leaves[t]
        for n, i in enumerate(mo.E_IDX):
            x = F.leaky_relu(F.conv2d(x, mo.spectral_weight(orc.E, f"conv_blocks.{i}."), orc.E[f"conv_blocks.{i}.bias"], 2, 1), 0.2)
            if n > 0:
                x = mo.batchnorm_train(x, orc.E, f"conv_blocks.{i + 2}.", eps=0.8)
        x = x.view(B, -1)
        logits = F.linear(x, mo.spectral_weight(orc.E, "aux_layer.0."), orc.E["aux_layer.0.bias"])
        lat = F.linear(x, mo.spectral_weight(orc.E, "latent_layer.0."), orc.E["latent_layer.0.bias"])
        mo.spectral_weight(orc.E, "noise_layer.0.")              # the noise head's power iteration still runs
        ref_out.append((logits, lat))
        total = total + (logits * dcat[t]).sum() + (lat * dlat[t]).sum()
    total.backward()
    outs = ee.forward([im.to(DEV) for im in imgs])
    for t in range(3):
        assert rel_err(outs["aux_layer.0"][t * B:(t + 1) * B], ref_out[t][0]) < 2e-5
        assert rel_err(outs["latent_layer.0"][t * B:(t + 1) * B], ref_out[t][1]) < 2e-5
    grad = torch.zeros_like(E.arena.grad)
    dimg = ee.backward(0, 3, {"aux_layer.0": torch.cat(dcat).to(DEV).contiguous(), "latent_layer.0": torch.cat(dlat).to(DEV).contiguous()}, grad,
                       need_wgrad=True, need_dimg=True)
    assert rel_err(dimg, leaves[0].grad) < 5e-4
    for k in dict(E.named_parameters()):
        if k.startswith("noise_layer"):
            assert float(grad[E.arena.slices[k][0]:E.arena.slices[k][0] + E.arena.slices[k][1]].abs().max()) == 0.0
            continue
        off, n = E.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.E[k].grad) < 5e-4, k
    for k in ("conv_blocks.4.running_mean", "conv_blocks.10.running_var", "noise_layer.0.weight_u", "aux_layer.0.weight_v"):
        assert rel_err(E.state_dict()[k], orc.E[k]) < 1e-4, k
    assert int(E.state_dict()["conv_blocks.4.num_batches_tracked"]) == 3


def run_steps(dtype, B, steps, seed=0, lrs=None):
    orc, G, D, E = build(seed, dtype, lrs)
    kw = {"lrs": lrs} if lrs else {}
    tr = eg.mnist.MnistTrainer(G, D, E, B, dtype=dtype, **kw)
    rng = np.random.RandomState(seed)
    real = mo.synthetic_real(B * steps, seed=4321).view(steps, B, 1, 32, 32)
    got, want = [], []
    for i in range(steps):
        z, code, labels = mo.draw_step_inputs(rng, B)
        got.append(tr.train_step(real[i].to(DEV), z.to(DEV), code.to(DEV), labels.to(DEV)))
        want.append(orc.train_step(real[i], z, code, labels))
    return orc, G, D, E, tr, got, want


def test_train_step_fp32_and_golden():
    gold = np.load(os.path.join(GOLDEN, "mnist_b8_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    orc, G, D, E, tr, got, want = run_steps("f32", B, steps, seed=seed)
    for k, t0 in (("g_loss", 2e-5), ("d_loss", 2e-5), ("info_loss", 3e-4)):
        assert abs(got[0][k] - want[0][k]) < t0, (k, got[0][k], want[0][k])
        assert abs(got[0][k] - gold[k][0]) < t0, (k, got[0][k], gold[k][0])
    for i in (1, 2):                                   # free-running: chaotic after Adam, loose
        for k in ("g_loss", "d_loss", "info_loss"):
            assert abs(got[i][k] - gold[k][i]) < 5e-2, (i, k, got[i][k], gold[k][i])


def test_train_step_gradients_lr0_fp32():
    orc, G, D, E, tr, got, want = run_steps("f32", 8, 1, seed=3, lrs=(0.0, 0.0, 0.0))
    for k in ("g_loss", "d_loss", "info_loss"):
        assert abs(got[0][k] - want[0][k]) < 2e-5, (k, got[0][k], want[0][k])
    for mod, ref in ((G, orc.G), (E, orc.E)):
        for k, p in mod.named_parameters():
            if k in PRE_BN_BIAS or k.startswith("noise_layer") or ref[k].grad is None:
                continue
            assert rel_err(p.grad, ref[k].grad) < 2e-2, k


def test_train_step_bf16_tracks_oracle():
    orc, G, D, E, tr, got, want = run_steps("bf16", 16, 2)
    for i in range(2):
        for k in ("g_loss", "d_loss", "info_loss"):
            assert abs(got[i][k] - want[i][k]) < 5e-2 * max(1.0, abs(want[i][k])), (i, k, got[i][k], want[i][k])


def test_graph_replay_equals_eager():
    B = 8
    res = []
    for capture in (False, True):
        orc, G, D, E = build(7, "f32")
        tr = eg.mnist.MnistTrainer(G, D, E, B, dtype="f32")
        rng = np.random.RandomState(1)
        real = mo.synthetic_real(B, seed=5).to(DEV)
        out = []
        for i in range(3):
            z, code, labels = mo.draw_step_inputs(rng, B)
            tr.load_inputs(real, z.to(DEV), code.to(DEV), labels.to(DEV))
            if capture and i == 1:
                tr.capture()
            out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        res.append(torch.stack(out).cpu())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_side_lanes_are_bit_identical_to_one_stream(dtype):
    """the generator's weight-gradient chains on side lanes (overlap=True; off by default for this trainer, where it measured slower) vs everything on one stream: same kernels, same
    order inside every chain -> identical losses and parameters, eager and replayed"""
    B = 16
    res = []
    for overlap in (False, True):
        orc, G, D, E = build(5, dtype)
        tr = eg.mnist.MnistTrainer(G, D, E, B, dtype=dtype, overlap=overlap)
        assert (tr.side is not None) == overlap
        rng = np.random.RandomState(2)
        real = mo.synthetic_real(B, seed=6).to(DEV)
        out = []
        for i in range(4):
            z, code, labels = mo.draw_step_inputs(rng, B)
            tr.load_inputs(real, z.to(DEV), code.to(DEV), labels.to(DEV))
            if i == 2:
                tr.capture()
                assert tr.graph is not None
            out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        res.append((torch.stack(out).cpu(), G.arena.flat.clone().cpu(), E.arena.flat.clone().cpu()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def test_teacher_forced_loss_curve_100_steps_fp32():
    """100 consecutive iterations of the loop body (MNIST/EAD-GAN_rpqmnxy.py:338-446), each started from the oracle's state (parameters, buffers, Adam moments and
    step counts): every loss within 1e-3 of the oracle's (tests/teacher_forced.py; profiles/scripts/teacher_forced_curve.py runs 1000)."""
    import teacher_forced
    dev, names = teacher_forced.curve("mnist", 100)
    assert dev.max() < 1e-3, (names, dev.max(axis=0), np.argmax(dev, axis=0))
