"""Fit of the affine-inverse MLP (MNIST/approximate_rpqmnxy.py:109-153) on the MI355X vs the CPU oracle
(oracle/mnist_oracle.ApproximatorOracle, pinned to the reference script by tests/golden/approximator_fit_s5.npz)."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import mnist_oracle as mo

pytestmark = pytest.mark.gpu
DEV = "cuda"
eg = None


def setup_module(module):
    global eg
    eg = importlib.import_module("ead-gan_amd")
    torch.set_num_threads(16)


def rel_err(a, b):
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(seed, dtype, lr=2e-4, B=128):
    orc = mo.ApproximatorOracle(seed=seed, lr=lr)
    M = eg.mnist.Affine_classifier().to(DEV)
    assert list(M.state_dict().keys()) == list(orc.mlp.keys())
    M.load_state_dict({k: v.detach() for k, v in orc.mlp.items()})
    return orc, M, eg.mnist.ApproximatorTrainer(M, B, dtype=dtype, lr=lr)


def test_affine_para_kernel():
    code = (torch.rand(37, 7) * 2 - 1)
    para = torch.empty(37, 7, device=DEV)
    eg.ops.affine_para_rpqmnxy(code.to(DEV), 7, 37, para)
    assert torch.allclose(para.cpu(), mo.latent_to_affine_para(code), atol=1e-7, rtol=1e-6)


def test_fit_f32_follows_oracle_and_golden():
    gold = np.load(os.path.join(GOLDEN, "approximator_fit_s5.npz"))
    steps, seed, B = int(gold["steps"]), int(gold["seed"]), int(gold["B"])
    orc, M, tr = build(seed, "f32", B=B)
    rng = np.random.RandomState(seed)
    for i in range(steps):
        code = mo.draw_approximator_inputs(rng, B)
        o = orc.train_step(code)
        h = tr.train_step(code.to(DEV))
        assert abs(h["affine_loss"] - o["affine_loss"]) < (2e-6, 1e-4, 3e-4, 1e-3, 2e-3)[i], (i, h, o)
        assert abs(h["affine_loss"] - gold["affine_loss"][i]) < (4e-6, 2e-4, 6e-4, 2e-3, 4e-3)[i], (i, h, gold["affine_loss"][i])


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 3e-2), ("f16", 5e-3)])
def test_gradients_lr0(dtype, tol):
    """lr = 0: parameters stay put, the arena gradient is the reference's .grad after loss.backward()"""
    orc, M, tr = build(3, dtype, lr=0.0)
    code = mo.draw_approximator_inputs(np.random.RandomState(5), 128)
    o = orc.train_step(code)
    h = tr.train_step(code.to(DEV))
    assert abs(h["affine_loss"] - o["affine_loss"]) < tol * max(1.0, o["affine_loss"])
    for k, p in orc.mlp.items():
        assert rel_err(tr.arena.grad_of(k), p.grad) < tol * 5, (k, rel_err(tr.arena.grad_of(k), p.grad))


def test_graph_replay_is_bit_identical_and_ragged_batch():
    code = [mo.draw_approximator_inputs(np.random.RandomState(9 + i), 37).to(DEV) for i in range(4)]
    outs = []
    for graph in (False, True):
        _, M, tr = build(1, "f32", B=37)
        if graph:
            tr.capture()
            assert tr.graph is not None
        losses = [tr.train_step(c)["affine_loss"] for c in code]
        outs.append((losses, tr.arena.flat.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])


def test_fit_converges_and_installs():
    """300 iterations drive the loss down by > 10x, and the fitted weights feed the regulariser kernel (load_approximator)"""
    _, M, tr = build(0, "bf16")
    tr.capture()
    g = torch.Generator(device="cpu").manual_seed(0)
    first = last = None
    for i in range(300):
        last = tr.train_step((torch.rand(128, 7, generator=g) * 2 - 1).to(DEV))["affine_loss"]
        first = last if first is None else first
    assert last < first / 10, (first, last)
    blob = tr.install()
    assert blob.numel() == eg.ops.mlp_rpqmnxy_floats()
    sd = M.state_dict()
    assert torch.equal(blob[:256 * 6].view(256, 6), sd["fc_block.0.weight"])
