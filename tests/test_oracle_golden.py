"""Oracle (CPU restatement) vs golden vectors produced by the reference itself (oracle/make_golden.py)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import celeba_oracle as co
from oracle import mnist_oracle as mo


def check_probes(prefix, tensors, gold, rtol, atol, noise_floor=0.0):
    """tensors: name -> tensor; compares head/sum/abs fingerprints.  Scale-aware: tolerances are relative
    to the mean |value| of the tensor so that near-zero entries do not dominate."""
    for k, v in tensors.items():
        t = v.detach().double().flatten().cpu()
        ab = float(gold[f"{prefix}/{k}/abs"])
        scale = ab / max(t.numel(), 1)
        if scale < noise_floor:      # e.g. gradients of conv biases in front of a BatchNorm: pure rounding noise
            continue
        tol = rtol * scale + atol
        np.testing.assert_allclose(t[:8].numpy(), gold[f"{prefix}/{k}/head"], rtol=0, atol=8 * tol, err_msg=f"{prefix}/{k}")
        assert abs(t.abs().sum().item() - ab) <= rtol * ab + atol * t.numel() ** 0.5, f"{prefix}/{k} abs"
        assert abs(t.sum().item() - float(gold[f"{prefix}/{k}/sum"])) <= rtol * ab + atol * t.numel() ** 0.5, f"{prefix}/{k} sum"


def test_celeba_step_matches_reference():
    gold = np.load(os.path.join(GOLDEN, "celeba_b4_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    torch.set_num_threads(8)
    orc = co.CelebAOracle(seed=seed)
    rng = np.random.RandomState(seed)
    real = co.synthetic_real(B * steps, seed=int(gold["real_seed"])).view(steps, B, 3, 64, 64)
    for i in range(steps):
        z, code, labels = co.draw_step_inputs(rng, B)
        out = orc.train_step(real[i], z, code, labels)
        # step 0 is a pure function of the seeded init; later steps pass through Adam's first-step
        # sign amplification of rounding noise, hence the looser bound.
        tol = (2e-6, 3e-4, 5e-3)[i]
        for k in ("g_loss", "d_loss", "info_loss"):
            assert abs(out[k] - gold[k][i]) < tol, (i, k, out[k], gold[k][i])
        if i == 0:
            # gradients left by the info step: smooth quantities -> tight
            check_probes("gG1", {k: v.grad for k, v in orc.G.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("gD1", {k: v.grad for k, v in orc.D.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            # post-Adam state: entries with ~0 gradient move by +-lr on rounding noise -> atol 2.5e-3 on heads
            check_probes("G1", orc.G, gold, 2e-3, 3e-4)
            check_probes("D1", orc.D, gold, 2e-3, 3e-4)


def test_celeba_affine_functions_match_reference():
    gold = np.load(os.path.join(GOLDEN, "celeba_affine.npz"))
    code = torch.tensor(gold["code"])
    A = co.get_matrix(code[:, :5])
    np.testing.assert_allclose(A.numpy(), gold["A"], rtol=1e-6, atol=1e-7)
    img = co.synthetic_real(4, seed=int(gold["img_seed"]))
    warped = co.warp(img, A[:4, 0:2])
    np.testing.assert_allclose(warped.numpy(), gold["warped"], rtol=1e-5, atol=1e-6)
    rc = torch.tensor(gold["real_code"], requires_grad=True)
    tc = torch.tensor(gold["trans_code"], requires_grad=True)
    pred = co.affine_regularzier(rc, tc)
    np.testing.assert_allclose(pred.detach().numpy(), gold["pred"], rtol=1e-4, atol=1e-5)
    (pred * torch.tensor(gold["w"])).sum().backward()
    np.testing.assert_allclose(rc.grad.numpy(), gold["d_real"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(tc.grad.numpy(), gold["d_trans"], rtol=1e-3, atol=1e-4)


def test_mnist_step_matches_reference():
    gold = np.load(os.path.join(GOLDEN, "mnist_b8_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    torch.set_num_threads(8)
    orc = mo.MnistOracle(seed=seed, mlp=mo.make_approximator(int(gold["mlp_seed"])))
    rng = np.random.RandomState(seed)
    real = mo.synthetic_real(B * steps, seed=int(gold["real_seed"])).view(steps, B, 1, 32, 32)
    for i in range(steps):
        z, code, labels = mo.draw_step_inputs(rng, B)
        out = orc.train_step(real[i], z, code, labels)
        tol = (2e-6, 1e-3, 1e-2)[i]
        for k in ("g_loss", "d_loss", "info_loss"):
            assert abs(out[k] - gold[k][i]) < tol, (i, k, out[k], gold[k][i])
        if i == 0:
            check_probes("gG1", {k: v.grad for k, v in orc.G.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("gE1", {k: v.grad for k, v in orc.E.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("G1", orc.G, gold, 2e-3, 3e-4)
            check_probes("D1", orc.D, gold, 2e-3, 3e-4)
            check_probes("E1", orc.E, gold, 2e-3, 3e-4)


def test_mnist_affine_functions_match_reference():
    gold = np.load(os.path.join(GOLDEN, "mnist_affine.npz"))
    mlp = mo.make_approximator(int(gold["mlp_seed"]))
    A = mo.get_matrix(torch.tensor(gold["code"]))
    np.testing.assert_allclose(A.numpy(), gold["A"], rtol=1e-6, atol=1e-7)
    rc = torch.tensor(gold["real_code"], requires_grad=True)
    tc = torch.tensor(gold["trans_code"], requires_grad=True)
    pred = mo.affine_regularizer(mlp, rc, tc)
    np.testing.assert_allclose(pred.detach().numpy(), gold["pred"], rtol=1e-4, atol=1e-5)
    (pred * torch.tensor(gold["w"])).sum().backward()
    np.testing.assert_allclose(rc.grad.numpy(), gold["d_real"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(tc.grad.numpy(), gold["d_trans"], rtol=1e-3, atol=1e-4)


def test_dsprites_step_matches_reference():
    from oracle import dsprites_oracle as do
    gold = np.load(os.path.join(GOLDEN, "dsprites_b8_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    torch.set_num_threads(8)
    orc = do.DspritesOracle(seed=seed, pxy=do.make_encoder_pxy(int(gold["pxy_seed"])))
    rng = np.random.RandomState(seed)
    sprites = do.synthetic_sprites(B * steps, seed=int(gold["sprite_seed"])).view(steps, B, 64, 64)
    names = ("d_loss", "g_loss", "info_loss", "affine_loss", "relative_cat_loss")
    for i in range(steps):
        out = orc.train_step(sprites[i], *do.draw_step_inputs(rng, B))
        tol = (3e-6, 1e-3, 1e-2)[i]
        for k in names:
            assert abs(out[k] - gold[k][i]) < tol, (i, k, out[k], gold[k][i])
        if i == 0:
            check_probes("gG1", {k: v.grad for k, v in orc.G.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("gE1", {k: v.grad for k, v in orc.E.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("G1", orc.G, gold, 2e-3, 3e-4)
            check_probes("D1", orc.D, gold, 2e-3, 3e-4)
            check_probes("E1", orc.E, gold, 2e-3, 3e-4)


def test_colored_dsprites_step_matches_reference():
    from oracle import dsprites_oracle as do
    gold = np.load(os.path.join(GOLDEN, "colored_b8_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    torch.set_num_threads(8)
    orc = do.ColoredOracle(seed=seed, pxy=do.make_encoder_pxy(int(gold["pxy_seed"]), ch=3, pxy_out=6))
    rng = np.random.RandomState(seed)
    sprites = do.synthetic_sprites(B * steps, seed=int(gold["sprite_seed"])).view(steps, B, 64, 64)
    for i in range(steps):
        out = orc.train_step(sprites[i], *do.draw_colored_inputs(rng, B))
        tol = (3e-6, 1e-3, 1e-2)[i]
        assert abs(out["d_loss"] - gold["d_loss"][i]) < tol and abs(out["g_loss"] - gold["g_loss"][i]) < tol
        assert abs(out["info_loss"] - (gold["cat_loss"][i] + gold["cont_loss"][i])) < tol
        assert abs(out["affine_loss"] - gold["affine_color_loss"][i]) < tol
        assert abs(out["relative_cat_loss"] - gold["relative_cat_loss"][i]) < tol
        if i == 0:
            check_probes("gG1", {k: v.grad for k, v in orc.G.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("gE1", {k: v.grad for k, v in orc.E.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)


def test_pxy_stage1_step_matches_reference():
    """dSprites/pxy.py (stage-1 trainer of Encoder_pxy): oracle vs the loop body run through the harness"""
    from oracle import dsprites_oracle as do
    gold = np.load(os.path.join(GOLDEN, "pxy_b8_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    torch.set_num_threads(8)
    orc = do.PxyOracle(seed=seed)
    rng = np.random.RandomState(seed)
    sprites = do.synthetic_sprites(B * steps, seed=int(gold["sprite_seed"])).view(steps, B, 64, 64)
    for i in range(steps):
        out = orc.train_step(sprites[i], do.draw_pxy_inputs(rng, B))
        assert abs(out["affine_loss"] - gold["affine_loss"][i]) < (2e-6, 3e-4, 2e-3)[i], (i, out, gold["affine_loss"][i])
        if i == 0:
            check_probes("gP1", {k: v.grad for k, v in orc.P.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("P1", orc.P, gold, 2e-3, 3e-4)


def test_pxy_color_stage1_step_matches_reference():
    """colored_dSprites/pxy_color.py (stage-1 trainer of the colored Encoder_pxy): oracle vs the loop body run through the harness"""
    from oracle import dsprites_oracle as do
    gold = np.load(os.path.join(GOLDEN, "pxy_color_b8_s3.npz"))
    B, steps, seed = int(gold["B"]), int(gold["steps"]), int(gold["seed"])
    torch.set_num_threads(8)
    orc = do.PxyColorOracle(seed=seed)
    rng = np.random.RandomState(seed)
    sprites = do.synthetic_sprites(B * steps, seed=int(gold["sprite_seed"])).view(steps, B, 64, 64)
    for i in range(steps):
        out = orc.train_step(sprites[i], *do.draw_pxy_color_inputs(rng, B))
        assert abs(out["affine_loss"] - gold["affine_loss"][i]) < (2e-6, 3e-4, 2e-3)[i], (i, out, gold["affine_loss"][i])
        if i == 0:
            check_probes("gP1", {k: v.grad for k, v in orc.P.items() if getattr(v, "grad", None) is not None}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("P1", orc.P, gold, 2e-3, 3e-4)


def test_approximator_fit_matches_reference():
    """MNIST/approximate_rpqmnxy.py (fit of the affine-inverse MLP): oracle vs the script's __main__ loop run through the harness"""
    gold = np.load(os.path.join(GOLDEN, "approximator_fit_s5.npz"))
    steps, seed, B = int(gold["steps"]), int(gold["seed"]), int(gold["B"])
    torch.set_num_threads(8)
    orc = mo.ApproximatorOracle(seed=seed)
    rng = np.random.RandomState(seed)
    for i in range(steps):
        out = orc.train_step(mo.draw_approximator_inputs(rng, B))
        assert abs(out["affine_loss"] - gold["affine_loss"][i]) < (2e-6, 1e-4, 3e-4, 1e-3, 2e-3)[i], (i, out, gold["affine_loss"][i])
        if i == 0:
            check_probes("gM1", {k: v.grad for k, v in orc.mlp.items()}, gold, 1e-2, 1e-9, noise_floor=1e-8)
            check_probes("M1", {k: v.detach() for k, v in orc.mlp.items()}, gold, 2e-3, 3e-4)
