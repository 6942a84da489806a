"""CelebA hot path on the MI355X vs the CPU oracle (oracle/celeba_oracle.py, pinned to the reference by
tests/golden/celeba_b4_s3.npz): network forward/backward, the fused train step, hipGraph replay."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from oracle import celeba_oracle as co

pytestmark = pytest.mark.gpu
DEV = "cuda"
eg = None


def setup_module(module):
    global eg
    eg = importlib.import_module("ead-gan_amd")
    torch.set_num_threads(16)


PRE_BN_BIAS = ("conv_blocks.1.bias", "conv_blocks.4.bias", "conv_blocks.7.bias")   # zero gradient up to rounding noise


def build_pair(seed, dtype, lrs=(1e-3, 2e-4, 2e-4)):
    orc = co.CelebAOracle(seed=seed, lrs=lrs)
    G = eg.celeba.Generator(dtype=dtype).to(DEV)
    D = eg.celeba.Discriminator(dtype=dtype).to(DEV)
    assert list(G.state_dict().keys()) == list(orc.G.keys())
    assert list(D.state_dict().keys()) == list(orc.D.keys())
    G.load_state_dict({k: v.detach() for k, v in orc.G.items()})
    D.load_state_dict({k: v.detach() for k, v in orc.D.items()})
    return orc, G, D


def rel_err(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_state_dict_is_reference_compatible():
    orc, G, D = build_pair(0, "f32")
    for k, v in orc.D.items():
        assert D.state_dict()[k].shape == v.shape, k
    for k, v in orc.G.items():
        assert G.state_dict()[k].shape == v.shape, k


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 3e-2)])
def test_generator_forward_backward(dtype, tol):
    B = 4
    orc, G, D = build_pair(1, dtype)
    rng = np.random.RandomState(5)
    z, code, labels = co.draw_step_inputs(rng, B)
    onehot = F.one_hot(labels, 10).float()
    want = co.generator_forward(orc.G, z, onehot, code)
    got = G(z.to(DEV), onehot.to(DEV), code.to(DEV))
    assert got.shape == (B, 3, 64, 64)
    assert rel_err(got, want) < tol
    w = torch.randn(want.shape, generator=torch.Generator().manual_seed(3))
    (want * w).sum().backward()
    (got * w.to(DEV)).sum().backward()
    for k, p in G.named_parameters():
        ref = orc.G[k].grad
        if k in PRE_BN_BIAS:
            continue
        assert rel_err(p.grad, ref) < tol * 20, k
    # BatchNorm running statistics follow the reference (momentum 0.1, unbiased variance)
    for k in ("conv_blocks.2.running_mean", "conv_blocks.5.running_var", "conv_blocks.8.running_mean"):
        assert rel_err(G.state_dict()[k], orc.G[k]) < max(tol, 1e-4), k
    assert int(G.state_dict()["conv_blocks.2.num_batches_tracked"]) == 1


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 3e-2)])
def test_discriminator_forward_backward(dtype, tol):
    B = 4
    orc, G, D = build_pair(2, dtype)
    img = co.synthetic_real(B, seed=9).requires_grad_(True)
    imgd = img.detach().to(DEV).requires_grad_(True)
    cat, cont, val = co.discriminator_forward(orc.D, img)
    gcat, gcont, gval = D(imgd)
    assert rel_err(gcat, cat) < tol and rel_err(gcont, cont) < tol * 5 and rel_err(gval, val) < tol
    # spectral-norm buffers advanced by exactly one power iteration
    for i in range(4):
        assert rel_err(D.state_dict()[f"main.{2 * i}.weight_u"], orc.D[f"main.{2 * i}.weight_u"]) < 1e-4
        assert rel_err(D.state_dict()[f"main.{2 * i}.weight_v"], orc.D[f"main.{2 * i}.weight_v"]) < 1e-4
    g = torch.Generator().manual_seed(4)
    w1, w2, w3 = torch.randn(cat.shape, generator=g), torch.randn(cont.shape, generator=g), torch.randn(val.shape, generator=g)
    ((cat * w1).sum() + (cont * w2).sum() + (val * w3).sum()).backward()
    ((gcat * w1.to(DEV)).sum() + (gcont * w2.to(DEV)).sum() + (gval * w3.to(DEV)).sum()).backward()
    assert rel_err(imgd.grad, img.grad) < tol * 20
    for k, p in D.named_parameters():
        assert rel_err(p.grad, orc.D[k].grad) < tol * 20, k


def run_steps(dtype, B, steps, seed=0, lrs=(1e-3, 2e-4, 2e-4)):
    orc, G, D = build_pair(seed, dtype, lrs)
    tr = eg.celeba.CelebATrainer(G, D, B, dtype=dtype, lr_g=lrs[0], lr_d=lrs[1], lr_info=lrs[2])
    rng = np.random.RandomState(seed)
    real = co.synthetic_real(B * steps, seed=1234).view(steps, B, 3, 64, 64)
    got, want = [], []
    for i in range(steps):
        z, code, labels = co.draw_step_inputs(rng, B)
        got.append(tr.train_step(real[i].to(DEV), z.to(DEV), code.to(DEV), labels.to(DEV)))
        want.append(orc.train_step(real[i], z, code, labels))
    return orc, G, D, tr, got, want


def test_train_step_fp32_parity():
    """fp32 tolerance: step 0 losses 2e-5 abs; info-step gradients 1e-3 relative (L2); later steps looser because
    Adam's first updates amplify rounding noise (see tests/test_oracle_golden.py)."""
    orc, G, D, tr, got, want = run_steps("f32", 8, 2)
    for k, t0 in (("g_loss", 2e-5), ("d_loss", 2e-5), ("info_loss", 2e-4)):    # info_loss is evaluated after two Adam updates
        assert abs(got[0][k] - want[0][k]) < t0, (k, got[0][k], want[0][k])
        assert abs(got[1][k] - want[1][k]) < 3e-2, (k, got[1][k], want[1][k])   # free-running: chaotic after Adam


def test_discriminator_three_forwards_then_backward_fp32():
    """The info step runs three D forwards (each with its own power iteration, sigma, u, v) before one backward:
    every tape must be back-propagated with ITS sigma/u/v.  Random O(1) images and head gradients keep the
    LeakyReLU pre-activations away from 0, so the comparison is tight."""
    B = 8
    orc, G, D = build_pair(3, "f32")
    de = D.engine(B)
    imgs = [co.synthetic_real(B, seed=s) for s in (1, 2, 3)]
    g = torch.Generator().manual_seed(0)
    douts = [torch.randn(B, 19, generator=g) for _ in range(3)]
    leaves = [im.clone().requires_grad_(True) for im in imgs]
    total = 0
    acts = [[] for _ in range(4)]
    for t in range(3):
        x = leaves[t]
        for i in range(4):
            x = F.leaky_relu(F.conv2d(x, co.spectral_weight(orc.D, f"main.{2 * i}."), orc.D[f"main.{2 * i}.bias"], 2, 1), 0.1)
            acts[i].append(x.detach())
        total = total + (F.conv2d(x, orc.D["main.8.weight"], orc.D["main.8.bias"]).squeeze() * douts[t]).sum()
    total.backward()
    # (a) the three forwards batched into single launches (what the trainer does), one batched backward
    out = de.forward([im.to(DEV) for im in imgs])
    assert out.shape == (3 * B, 19)
    grad = torch.zeros_like(D.arena.grad)
    dimg = de.backward(0, 3, torch.cat(douts).to(DEV).contiguous(), grad, need_wgrad=True, need_dimg=True)
    # A LeakyReLU unit whose pre-activation is within fp32 rounding of 0 can land on the other side of 0 than in torch
    # (different summation order); ONE such unit among the 4.7 M moves every upstream gradient by ~1.5e-3 (measured), so
    # the tight bound is asserted when the activation signs agree and a per-flip allowance is added otherwise.
    count_flips = lambda e: sum(int(((e.a[i].permute(0, 3, 1, 2).float().cpu() > 0) != (torch.cat(acts[i]) > 0)).sum()) for i in range(4))
    flips = count_flips(de)
    assert flips <= 3, flips
    gtol = 2e-4 + 4e-3 * flips
    assert rel_err(dimg, leaves[0].grad) < gtol
    for k in dict(D.named_parameters()):
        off, n = D.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.D[k].grad) < gtol, (k, flips)
    for i in range(4):
        assert rel_err(D.state_dict()[f"main.{2 * i}.weight_u"], orc.D[f"main.{2 * i}.weight_u"]) < 1e-4
    # (b) same weights, tapes run one at a time (eager drop-in path) -> per-tape image gradients
    orc2, G2, D2 = build_pair(3, "f32")
    de2 = D2.engine(B)
    for t in range(3):
        de2.forward([imgs[t].to(DEV)], t)
    flips2 = count_flips(de2)
    assert flips2 <= 3, flips2
    grad2 = torch.zeros_like(D2.arena.grad)
    for t in (2, 1, 0):
        dimg = de2.backward(t, 1, douts[t].to(DEV).contiguous(), grad2, need_wgrad=True, need_dimg=True)
        assert rel_err(dimg, leaves[t].grad) < 2e-4 + 4e-3 * flips2, t
    assert rel_err(grad2, grad) < 1e-5 + 4e-3 * (flips + flips2)


def test_generator_backward_from_given_image_gradient_fp32():
    B = 8
    orc, G, D = build_pair(3, "f32")
    rng = np.random.RandomState(3)
    z, code, labels = co.draw_step_inputs(rng, B)
    onehot = F.one_hot(labels, 10).float()
    taps = []
    want = co.generator_forward(orc.G, z, onehot, code, taps)
    dimg = torch.randn(want.shape, generator=torch.Generator().manual_seed(1)) * 1e-3
    want.backward(dimg)
    ge = G.engine(B)
    got = ge.forward(z.to(DEV), onehot.to(DEV), code.to(DEV))
    assert rel_err(got, want) < 2e-5
    grad = torch.zeros_like(G.arena.grad)
    ge.backward(dimg.to(DEV), grad)
    # a BatchNorm output within fp32 rounding of 0 may take the other ReLU branch than in torch (summation order); one such
    # unit among 1.8 M moves the upstream gradients by ~2e-3 (measured) -> tight bound when the masks agree, allowance per flip
    flips = sum(int(((ge.a[i].permute(0, 3, 1, 2).float().cpu() > 0) != (taps[i] > 0)).sum()) for i in range(3))
    assert flips <= 3, flips
    for k in dict(G.named_parameters()):
        if k in PRE_BN_BIAS:
            continue
        off, n = G.arena.slices[k]
        assert rel_err(grad[off:off + n], orc.G[k].grad) < 1e-4 + 4e-3 * flips, (k, flips)


def test_train_step_gradients_fp32():
    """All three sub-steps with the learning rates set to 0 on both sides (no Adam sign amplification): losses agree
    to 2e-5 and the gradients the info step leaves behind to 2e-2 in relative L2.  The gradient bound is set by
    LeakyReLU sign flips, not by arithmetic: at initialisation the generated images are ~0, D's pre-activations
    cluster around the biases and a handful of the ~2M units sit within fp32 rounding of 0; each flipped unit moves
    the gradient norm by ~1/sqrt(N) (measured 3e-3; with O(1) inputs the same kernels agree to 1e-6, see the two tests
    above).  BN running statistics and spectral-norm u/v still advance (2 G forwards, 6 D forwards)."""
    orc, G, D, tr, got, want = run_steps("f32", 8, 1, seed=3, lrs=(0.0, 0.0, 0.0))
    for k in ("g_loss", "d_loss", "info_loss"):
        assert abs(got[0][k] - want[0][k]) < 2e-5, (k, got[0][k], want[0][k])
    for mod, ref in ((G, orc.G), (D, orc.D)):
        for k, p in mod.named_parameters():
            if k in PRE_BN_BIAS:
                continue
            assert rel_err(p.grad, ref[k].grad) < 2e-2, k
        for k, v in mod.state_dict().items():
            if k.endswith(("running_mean", "running_var", "weight_u", "weight_v")):
                assert rel_err(v, ref[k]) < 1e-4, k


def test_teacher_forced_second_step_fp32():
    """Free-running trajectories separate chaotically (Adam's +-lr first updates amplify rounding noise), so the second
    iteration is checked teacher-forced: parameters, buffers and all three Adams' moments/step counts are copied from
    the oracle after its first iteration, then both sides run iteration 2 on the same inputs.  g/d losses (functions of
    the synchronised state) must agree to 2e-5, info_loss (behind this iteration's two Adam updates) to 3e-4."""
    B = 8
    orc, G, D = build_pair(5, "f32")
    tr = eg.celeba.CelebATrainer(G, D, B, dtype="f32")
    rng = np.random.RandomState(5)
    real = co.synthetic_real(2 * B, seed=77).view(2, B, 3, 64, 64)
    z, code, labels = co.draw_step_inputs(rng, B)
    orc.train_step(real[0], z, code, labels)
    G.load_state_dict({k: v.detach() for k, v in orc.G.items()})
    D.load_state_dict({k: v.detach() for k, v in orc.D.items()})
    tr.import_adam_state(orc.opt_G, orc.opt_D, orc.opt_info)
    z, code, labels = co.draw_step_inputs(rng, B)
    got = tr.train_step(real[1].to(DEV), z.to(DEV), code.to(DEV), labels.to(DEV))
    want = orc.train_step(real[1], z, code, labels)
    assert tr.steps.tolist() == [2, 2, 2]
    for k, t in (("g_loss", 2e-5), ("d_loss", 2e-5), ("info_loss", 3e-4)):
        assert abs(got[k] - want[k]) < t, (k, got[k], want[k])


def teacher_forced_curve(steps, B=4, seed=11, progress=None):
    """The form in which BASELINE.json's "loss curves within 1e-3 of reference for 1k steps" is claimed: the
    oracle free-runs the reference loop; before EVERY iteration the HIP side is reset to the oracle's state -- parameters, BatchNorm
    running statistics, spectral-norm u / v, and the moments and step counts of all three Adams -- then both run that iteration on
    the same inputs.  Returns the per-iteration |loss difference| for g / d / info, [steps, 3].  (Free-running, two fp32
    implementations of this GAN separate chaotically after a few Adam updates: reference vs oracle on one machine does.)"""
    orc, G, D = build_pair(seed, "f32")
    tr = eg.celeba.CelebATrainer(G, D, B, dtype="f32")
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    dev = np.zeros((steps, 3))
    for i in range(steps):
        if i:
            G.load_state_dict({k: v.detach() for k, v in orc.G.items()})
            D.load_state_dict({k: v.detach() for k, v in orc.D.items()})
            tr.import_adam_state(orc.opt_G, orc.opt_D, orc.opt_info)
        real = torch.rand((B, 3, 64, 64), generator=g) * 2 - 1
        z, code, labels = co.draw_step_inputs(rng, B)
        got = tr.train_step(real.to(DEV), z.to(DEV), code.to(DEV), labels.to(DEV))
        want = orc.train_step(real, z, code, labels)
        dev[i] = [abs(got[k] - want[k]) for k in ("g_loss", "d_loss", "info_loss")]
        if progress is not None:
            progress(i, dev)
    return dev


def test_teacher_forced_loss_curve_100_steps_fp32():
    """100 consecutive iterations, each from the oracle's state: every g / d / info loss within 1e-3 of the oracle's
    (profiles/scripts/teacher_forced_curve.py runs the same function for 1000 iterations; its table is committed under profiles/;
    the MNIST / dSprites / colored-dSprites loops have the same test in their files)."""
    dev = teacher_forced_curve(100)
    assert dev.max() < 1e-3, (dev.max(axis=0), np.argmax(dev, axis=0))
    assert np.median(dev[:, :2]) < 5e-5          # g / d losses are functions of the synchronised state alone


def test_train_step_bf16_tracks_oracle():
    """bf16 MFMA inputs, fp32 accumulate/master weights: losses within 3e-2 of the fp32 oracle over 3 steps."""
    orc, G, D, tr, got, want = run_steps("bf16", 8, 3)
    for i in range(3):
        for k in ("g_loss", "d_loss", "info_loss"):
            assert abs(got[i][k] - want[i][k]) < 3e-2 * max(1.0, abs(want[i][k])), (i, k, got[i][k], want[i][k])


def test_matches_reference_golden_losses():
    gold = np.load(os.path.join(GOLDEN, "celeba_b4_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    orc, G, D, tr, got, want = run_steps("f32", B, steps, seed=seed)
    for i, tol in enumerate((2e-5, 2e-2, 5e-2)):
        for k in ("g_loss", "d_loss", "info_loss"):
            t = 2e-4 if (i == 0 and k == "info_loss") else tol        # step-0 info_loss already sits behind two Adam updates
            assert abs(got[i][k] - gold[k][i]) < t, (i, k, got[i][k], gold[k][i])


def test_graph_replay_equals_eager():
    """the captured hipGraph must reproduce the eager launch sequence bit for bit (deterministic kernels)."""
    B = 4
    losses = []
    for capture in (False, True):
        orc, G, D = build_pair(7, "bf16")
        tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16")
        rng = np.random.RandomState(1)
        real = co.synthetic_real(B, seed=5).to(DEV)
        out = []
        for i in range(3):
            z, code, labels = co.draw_step_inputs(rng, B)
            tr.load_inputs(real, z.to(DEV), code.to(DEV), labels.to(DEV))
            if capture and i == 1:
                tr.capture()          # step 0 ran eagerly: kernels are loaded, capture records without executing
            out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        losses.append(torch.stack(out).cpu())
    assert torch.equal(losses[0], losses[1]), (losses[0], losses[1])


def test_iteration_as_four_graphs_on_two_streams_equals_one_graph(monkeypatch):
    """EG_MULTI_GRAPH=1 (engine.MultiGraph, an experiment switch: DESIGN.md): the iteration cut into four hipGraphs replayed on two streams
    gives the bits of the one-graph replay"""
    B = 4
    losses = []
    for multi in (False, True):
        monkeypatch.setattr(eg.celeba, "MULTI_GRAPH", multi)
        orc, G, D = build_pair(7, "bf16")
        tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16")
        rng = np.random.RandomState(1)
        real = co.synthetic_real(B, seed=5).to(DEV)
        out = []
        for i in range(4):
            z, code, labels = co.draw_step_inputs(rng, B)
            tr.load_inputs(real, z.to(DEV), code.to(DEV), labels.to(DEV))
            if i == 1:
                tr.capture()
                assert hasattr(tr.graph, "segments") == multi
            out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        losses.append(torch.stack(out).cpu())
    assert torch.equal(losses[0], losses[1]), (losses[0], losses[1])


def test_workspace_growth_keeps_captured_graphs_valid():
    """Workspaces are shared per device and grow when a larger engine is built.  A hipGraph captured before has the old addresses
    baked in: the outgrown buffers must stay alive, and the side lanes of the new trainer must get scratch of the new size.
    Trainer A (B=4) is captured, trainer B (B=16) is built and stepped (eager, with side lanes), then A's graph is replayed:
    bit-identical to a second A that never saw B."""
    def run_a(with_b):
        orc, G, D = build_pair(9, "bf16")
        tr = eg.celeba.CelebATrainer(G, D, 4, dtype="bf16")
        rng = np.random.RandomState(2)
        real = co.synthetic_real(4, seed=6).to(DEV)
        z, code, labels = co.draw_step_inputs(rng, 4)
        tr.load_inputs(real, z.to(DEV), code.to(DEV), labels.to(DEV))
        tr.step_resident()
        tr.capture()
        out = [tr.step_resident().clone()]
        if with_b:
            orc2, G2, D2 = build_pair(10, "bf16")
            trb = eg.celeba.CelebATrainer(G2, D2, 16, dtype="bf16")
            zb, cb, lb = co.draw_step_inputs(rng, 16)
            lossb = trb.train_step(co.synthetic_real(16, seed=8).to(DEV), zb.to(DEV), cb.to(DEV), lb.to(DEV))
            assert all(np.isfinite(v) for v in lossb.values())
        out.append(tr.step_resident().clone())
        out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        return torch.stack(out).cpu()
    assert torch.equal(run_a(False), run_a(True))


def test_side_stream_overlap_is_bit_identical(monkeypatch):
    """Weight-gradient chains and re-packing on the side streams (eager and captured) vs everything on one stream: the same
    kernels on the same operands in the same per-chain order -> identical losses and identical parameters after 3 steps.
    The side lanes' chip-share hint (engine.LANE_WGS_TARGET: their weight-gradient GEMMs aim for 128 workgroups, i.e. other K-split
    counts and another summation order than the single-stream schedule) is switched off for the bit-for-bit comparison; with the hint
    the schedules agree to fp32 summation-order rounding of bf16 gradients."""
    B = 8

    def run(overlap, capture):
        orc, G, D = build_pair(11, "bf16")
        tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16", overlap=overlap)
        rng = np.random.RandomState(2)
        real = co.synthetic_real(B, seed=6).to(DEV)
        out = []
        for i in range(3):
            z, code, labels = co.draw_step_inputs(rng, B)
            tr.load_inputs(real, z.to(DEV), code.to(DEV), labels.to(DEV))
            if capture and i == 1:
                tr.capture()
            out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        return torch.stack(out).cpu(), G.arena.flat.clone().cpu(), D.arena.flat.clone().cpu()

    hinted = run(True, True)
    monkeypatch.setattr(eg.engine, "LANE_WGS_TARGET", 0)
    res = [run(overlap, capture) for overlap, capture in ((False, False), (True, False), (True, True))]
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]), (r[0], res[0][0])
        assert torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
    assert torch.allclose(hinted[0], res[0][0], rtol=2e-3, atol=2e-3), (hinted[0], res[0][0])
    # Adam's first steps move every weight by about lr whatever the gradient's size: a weight whose gradient is rounding noise may go
    # the other way (2e-4 per step against weights of scale 2e-2) -> parameters agree to a fraction of a percent, not to rounding
    for a, b in ((hinted[1], res[0][1]), (hinted[2], res[0][2])):
        assert float((a - b).norm() / b.norm()) < 2e-2


@pytest.mark.parametrize("overlap", [False, True])
def test_fused_adam_and_repacking_is_bit_identical(overlap, monkeypatch):
    """optimizer.step() of a convolution weight + refresh of its packed panels as one launch per layer (ops.adam_pack_conv /
    adam_pack_rows) against the Adam launch over the arena followed by the re-packing launches: the same element update, the same
    panel bytes -> identical losses, masters, moments and panels after 3 steps (celebA/EAD-GAN_celebA.py:344,365,400)."""
    B = 8

    def run(fuse):
        monkeypatch.setattr(eg.celeba, "FUSE_ADAM", fuse)
        orc, G, D = build_pair(12, "bf16")
        tr = eg.celeba.CelebATrainer(G, D, B, dtype="bf16", overlap=overlap)
        rng = np.random.RandomState(3)
        real = co.synthetic_real(B, seed=7).to(DEV)
        out = []
        for i in range(3):
            z, code, labels = co.draw_step_inputs(rng, B)
            tr.load_inputs(real, z.to(DEV), code.to(DEV), labels.to(DEV))
            out.append(tr.step_resident().clone())
        torch.cuda.synchronize()
        panels = [t.clone() for r in tr.ge.mid + tr.de.mid for t in (r.wp_fwd, r.wp_bwd)] + [tr.ge.l0.wp_fwd.clone(), tr.de.head.wp_fwd.clone()]
        state = [G.arena.flat, D.arena.flat, tr.mG, tr.vG, tr.mD, tr.vD, tr.miG, tr.viG, tr.miD, tr.viD, tr.steps]
        return torch.stack(out).cpu(), [t.clone() for t in state], panels

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]), (a[0], b[0])
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert torch.equal(x, y)


def test_generator_eval_mode_uses_running_statistics():
    """generate_image.py / gen_imgs.py of the reference sample from G.eval(): BatchNorm with the running statistics, nothing
    updated.  Two training forwards move the running statistics, then eval forwards (fp32, bf16) are compared with torch."""
    B = 8
    orc, G, D = build_pair(5, "f32")
    rng = np.random.RandomState(4)
    for _ in range(2):
        z, code, labels = co.draw_step_inputs(rng, B)
        G(z.to(DEV), F.one_hot(labels, 10).float().to(DEV), code.to(DEV))
    z, code, labels = co.draw_step_inputs(rng, B)
    onehot = F.one_hot(labels, 10).float()
    sd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
    x = torch.cat((z, onehot, code), -1).view(B, -1, 1, 1)
    x = F.conv_transpose2d(x, sd["conv_blocks.0.weight"], sd["conv_blocks.0.bias"], 1, 0)
    for idx in (1, 4, 7):
        x = F.conv_transpose2d(x, sd[f"conv_blocks.{idx}.weight"], sd[f"conv_blocks.{idx}.bias"], 2, 1)
        x = F.relu(F.batch_norm(x, sd[f"conv_blocks.{idx + 1}.running_mean"], sd[f"conv_blocks.{idx + 1}.running_var"],
                                sd[f"conv_blocks.{idx + 1}.weight"], sd[f"conv_blocks.{idx + 1}.bias"], False, 0.1, 1e-5))
    want = torch.tanh(F.conv_transpose2d(x, sd["conv_blocks.10.weight"], sd["conv_blocks.10.bias"], 2, 1))
    G.eval()
    got = G(z.to(DEV), onehot.to(DEV), code.to(DEV))
    assert not got.requires_grad
    assert rel_err(got, want) < 2e-5
    for k, v in G.state_dict().items():           # eval touched neither running statistics nor num_batches_tracked
        assert torch.equal(v.cpu(), sd[k]), k
    G.set_compute_dtype("bf16")
    assert rel_err(G(z.to(DEV), onehot.to(DEV), code.to(DEV)), want) < 3e-2


def test_long_free_run_stays_on_the_reference_regime():
    """120 free-running fp32 iterations (B=4) on the inputs of tests/golden/celeba_curve_b4_s120.npz, a loss curve recorded from the
    reference script itself.  Per-step agreement is impossible (the CPU oracle and the reference -- same torch, same machine -- already
    differ by up to 1.4 in the 20-step windowed mean of g_loss and by 1.6 per step: GAN dynamics at B=4, lr 1e-3 amplify rounding
    noise), so this pins what survives the chaos: the first iterations, the info loss (windowed, it moves slowly) and the regime
    of the adversarial losses (finite, same order of magnitude): optimizer step counters, BatchNorm running statistics and
    spectral-norm vectors keep evolving correctly over many graph replays."""
    gold = np.load(os.path.join(GOLDEN, "celeba_curve_b4_s120.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    orc, G, D = build_pair(seed, "f32")
    tr = eg.celeba.CelebATrainer(G, D, B, dtype="f32")
    rng = np.random.RandomState(seed)
    real = co.synthetic_real(B * steps, seed=int(gold["real_seed"])).view(steps, B, 3, 64, 64).to(DEV)
    rec = []
    for i in range(steps):
        z, code, labels = co.draw_step_inputs(rng, B)
        tr.load_inputs(real[i], z.to(DEV), code.to(DEV), labels.to(DEV))
        if i == 2:
            tr.capture()
        rec.append(tr.step_resident().clone())
    got = torch.stack(rec).cpu().numpy()            # [steps, 4] = g, d, info
    assert np.isfinite(got).all()
    for col, k, t0 in ((0, "g_loss", 2e-5), (1, "d_loss", 2e-5), (2, "info_loss", 2e-4)):
        assert abs(got[0, col] - gold[k][0]) < t0, (k, got[0, col], gold[k][0])
        assert abs(got[1, col] - gold[k][1]) < 3e-2, (k, got[1, col], gold[k][1])
    w = 20
    ma = lambda a: np.convolve(a, np.ones(w) / w, "valid")
    assert np.abs(ma(got[:, 2]) - ma(gold["info_loss"])).max() < 0.15
    for col, k in ((0, "g_loss"), (1, "d_loss")):
        m = ma(got[:, col])
        assert m.min() > 0.25 * ma(gold[k]).min() and m.max() < 4.0 * ma(gold[k]).max(), (k, m.min(), m.max())


def test_train_step_fp16_tracks_oracle():
    """third compute dtype (EG_F16): CelebA step in IEEE half operands, same bound as the bf16 mode"""
    orc, G, D, tr, got, want = run_steps("f16", 8, 2)
    for i in range(2):
        for k in ("g_loss", "d_loss", "info_loss"):
            assert abs(got[i][k] - want[i][k]) < 3e-2 * max(1.0, abs(want[i][k])), (i, k, got[i][k], want[i][k])


@pytest.mark.parametrize("B", [1, 5, 37])
def test_ragged_batch_sizes_through_the_drop_in_modules(B):
    """The reference's DataLoader hands out a short last batch (no drop_last, celebA.py:194-206) and generate_image-style calls use
    arbitrary batch sizes: the drop-in modules build an engine per batch size; odd sizes leave every GEMM with a ragged last row
    tile (M = 16 B ... 1024 B).  B = 1 is only exercised on G: the reference's D ``.squeeze()`` (:136) breaks at B = 1."""
    orc, G, D = build_pair(9, "f32")
    rng = np.random.RandomState(B)
    z, code, labels = co.draw_step_inputs(rng, B)
    onehot = F.one_hot(labels, 10).float()
    want = co.generator_forward(orc.G, z, onehot, code)
    got = G(z.to(DEV), onehot.to(DEV), code.to(DEV))
    assert got.shape == (B, 3, 64, 64) and rel_err(got, want) < 2e-5
    if B == 1:
        return
    img = co.synthetic_real(B, seed=B)
    cat, cont, val = D(img.to(DEV))
    wc, wo, wv = co.discriminator_forward(orc.D, img)
    assert rel_err(cat, wc) < 2e-5 and rel_err(cont, wo) < 2e-5 and rel_err(val, wv) < 2e-5
    (cat.sum() + (cont * cont).sum() + val.sum()).backward()
    (wc.sum() + (wo * wo).sum() + wv.sum()).backward()
    for k, p in D.named_parameters():
        assert rel_err(p.grad, orc.D[k].grad) < 2e-2, k          # flip-bounded (see the three-tape test); typically 1e-6


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-6), ("bf16", 2e-2)])
def test_image_side_layers_both_formulations_agree(dtype, tol, monkeypatch):
    """The 128 -> 3 transposed convolutions (G's last layer forward, D's first-layer backward-to-image) as one GEMM + col2im gather (the
    default) and as the 4-phase implicit GEMM (EG_IMG_GEMM=0, kept as an A/B switch) are the same function: fp32 up to summation order,
    bf16 up to one extra rounding of the per-tap partial sums."""
    B = 4
    orc, G, D = build_pair(2, dtype)
    rng = np.random.RandomState(6)
    z, code, labels = co.draw_step_inputs(rng, B)
    onehot = F.one_hot(labels, 10).float()
    img = torch.rand(B, 3, 64, 64, generator=torch.Generator().manual_seed(3)) * 2 - 1
    outs = {}
    G.eval()
    D.eval()                                             # frozen spectral-norm vectors: a training-mode forward would move sigma between the two runs
    for flag in (True, False):
        monkeypatch.setattr(eg.celeba, "IMG_GEMM", flag)
        with torch.no_grad():
            gen = G(z.to(DEV), onehot.to(DEV), code.to(DEV)).float().cpu().clone()
        x = img.to(DEV).requires_grad_(True)
        cat, cont, val = D(x)
        (val.sum() + cont.sum()).backward()
        outs[flag] = (gen, x.grad.float().cpu().clone())
    assert rel_err(outs[True][0], outs[False][0]) < tol
    assert rel_err(outs[True][1], outs[False][1]) < tol
    assert float(outs[False][1].abs().max()) > 0
