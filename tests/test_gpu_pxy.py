"""Stage-1 trainer of Encoder_pxy (dSprites/pxy.py:156-191) on the MI355X vs the CPU oracle (oracle/dsprites_oracle.PxyOracle,
pinned to the reference by tests/golden/pxy_b8_s3.npz)."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import dsprites_oracle as do

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


def build(seed, dtype, lr=2e-4):
    orc = do.PxyOracle(seed=seed, lr=lr)
    P = eg.dsprites.Encoder_pxy(dtype=dtype).to(DEV)
    assert list(P.state_dict().keys()) == list(orc.P.keys())
    P.load_state_dict({k: v.detach() for k, v in orc.P.items()})
    return orc, P


def run(dtype, B, steps, seed=0, lr=2e-4, capture_at=None):
    orc, P = build(seed, dtype, lr)
    tr = eg.dsprites.PxyTrainer(P, B, dtype=dtype, lr=lr)
    rng = np.random.RandomState(seed)
    sprites = do.synthetic_sprites(B * steps, seed=98).view(steps, B, 64, 64)
    got, want = [], []
    for i in range(steps):
        code = do.draw_pxy_inputs(rng, B)
        if capture_at is not None and i == capture_at:
            tr.capture()
        got.append(tr.train_step(sprites[i].to(DEV), code.to(DEV))["affine_loss"])
        want.append(orc.train_step(sprites[i], code)["affine_loss"])
    return orc, P, tr, got, want


def test_loss_kernel_and_theta_match_the_oracle_functions():
    ops = eg.ops
    g = torch.Generator().manual_seed(4)
    B = 32
    rc = (torch.rand(B, 3, generator=g) * 2 - 1).requires_grad_(True)
    tc = (torch.rand(B, 3, generator=g) * 2 - 1).requires_grad_(True)
    code = torch.rand(B, 3, generator=g) * 2 - 1
    loss = torch.nn.functional.mse_loss(do.affine_regularzier_pxy(rc, tc), code)
    loss.backward()
    L = torch.zeros(1, device=DEV)
    dr, dt_ = torch.empty(B, 3, device=DEV), torch.empty(B, 3, device=DEV)
    ops.loss_affine_pxy(rc.detach().to(DEV), tc.detach().to(DEV), 3, 0, B, code.to(DEV), 3, 1.0, L, dr, dt_)
    assert abs(float(L) - float(loss)) < 1e-6
    torch.testing.assert_close(dr.cpu(), rc.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(dt_.cpu(), tc.grad, rtol=1e-5, atol=1e-7)
    theta = torch.empty(B, 2, 3, device=DEV)
    ops.theta_pxy(code.to(DEV), 3, B, theta)
    torch.testing.assert_close(theta.cpu(), do.get_matrix_pxy(code)[:, 0:2], rtol=1e-6, atol=1e-7)


def test_train_step_fp32_oracle_and_reference_golden():
    gold = np.load(os.path.join(GOLDEN, "pxy_b8_s3.npz"))
    orc, P, tr, got, want = run("f32", int(gold["B"]), int(gold["steps"]), seed=int(gold["seed"]))
    assert abs(got[0] - want[0]) < 2e-5 and abs(got[0] - gold["affine_loss"][0]) < 2e-5
    for i in (1, 2):                          # free-running after Adam: rounding noise on ~0 gradients flips +-lr updates
        assert abs(got[i] - want[i]) < 2e-2 and abs(got[i] - gold["affine_loss"][i]) < 2e-2


def test_gradients_with_lr0_fp32():
    orc, P, tr, got, want = run("f32", 8, 1, seed=1, lr=0.0)
    assert abs(got[0] - want[0]) < 2e-5
    for k, v in orc.P.items():
        if getattr(v, "grad", None) is None:
            continue
        off, n = tr.arena.slices[k]
        # bound set by LeakyReLU units within rounding of 0 (see tests/test_gpu_celeba.py), typical agreement is 1e-6
        assert rel_err(tr.arena.grad[off:off + n], v.grad) < 2e-2, k


def test_bf16_tracks_oracle_and_graph_replay_equals_eager():
    orc, P, tr, got, want = run("bf16", 16, 3)
    for i in range(3):
        assert abs(got[i] - want[i]) < 5e-2 * max(1.0, abs(want[i])), (i, got[i], want[i])
    _, _, _, eager, _ = run("bf16", 16, 4)
    _, _, _, graph, _ = run("bf16", 16, 4, capture_at=1)
    assert eager == graph


def test_checkpoint_feeds_the_stage2_encoder():
    """the trained state_dict is what dSprites/rp.py loads into its frozen Encoder_pxy (:258-259): same keys, loadable, same outputs"""
    orc, P, tr, got, want = run("f32", 8, 2)
    sd = {k: v.detach().clone() for k, v in P.state_dict().items()}
    Q = eg.dsprites.Encoder_pxy(dtype="f32").to(DEV)
    Q.load_state_dict(sd)
    img = do.synthetic_sprites(8, seed=3).unsqueeze(1).float()
    want_codes = do.encoder_pxy_forward({k: v.cpu() for k, v in sd.items()}, img)
    assert rel_err(Q(img.to(DEV)), want_codes) < 2e-5


# ---- colored variant (colored_dSprites/pxy_color.py:160-216) --------------------------------------------------------------------
def run_color(dtype, B, steps, seed=0, lr=2e-4):
    orc = do.PxyColorOracle(seed=seed, lr=lr)
    P = eg.colored.Encoder_pxy(dtype=dtype).to(DEV)
    assert list(P.state_dict().keys()) == list(orc.P.keys())
    P.load_state_dict({k: v.detach() for k, v in orc.P.items()})
    tr = eg.colored.PxyColorTrainer(P, B, dtype=dtype, lr=lr)
    rng = np.random.RandomState(seed)
    sprites = do.synthetic_sprites(B * steps, seed=97).view(steps, B, 64, 64)
    got, want = [], []
    for i in range(steps):
        gains, code = do.draw_pxy_color_inputs(rng, B)
        got.append(tr.train_step(sprites[i].to(DEV), gains.float().to(DEV), code.to(DEV))["affine_loss"])
        want.append(orc.train_step(sprites[i], gains, code)["affine_loss"])
    return orc, P, tr, got, want


def test_zero_padding_warp_and_colour_loss_kernels():
    ops = eg.ops
    g = torch.Generator().manual_seed(6)
    B = 16
    img = torch.rand(B, 3, 64, 64, generator=g)
    code = torch.rand(B, 6, generator=g) * 2 - 1
    code[:4, :3] *= 8.0                                    # large zoom / shift: samples far outside the image
    theta = do.get_matrix_pxy(code[:, :3])[:, 0:2].contiguous()
    out = torch.empty(B, 3, 64, 64, device=DEV)
    ops.warp_affine_zeros(img.to(DEV), theta.to(DEV), out, B, 3, 64, 64)
    torch.testing.assert_close(out.cpu(), do.warp_zeros(img, theta), rtol=1e-5, atol=2e-5)     # sample coordinates up to ~10 image widths away
    rc = (torch.rand(B, 6, generator=g) * 2 - 1).requires_grad_(True)
    tc = (torch.rand(B, 6, generator=g) * 2 - 1).requires_grad_(True)
    loss = torch.nn.functional.mse_loss(do.affine_regularzier_pxy_color(rc, tc), code)
    loss.backward()
    L = torch.zeros(1, device=DEV)
    dr, dt_ = torch.empty(B, 6, device=DEV), torch.empty(B, 6, device=DEV)
    ops.loss_affine_pxy(rc.detach().to(DEV), tc.detach().to(DEV), 6, 0, B, code.to(DEV), 6, 1.0, L, dr, dt_, ncol=3)
    assert abs(float(L) - float(loss.detach())) < 1e-5
    torch.testing.assert_close(dr.cpu(), rc.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(dt_.cpu(), tc.grad, rtol=1e-5, atol=1e-7)


def test_color_train_step_fp32_oracle_reference_golden_and_bf16():
    gold = np.load(os.path.join(GOLDEN, "pxy_color_b8_s3.npz"))
    orc, P, tr, got, want = run_color("f32", int(gold["B"]), int(gold["steps"]), seed=int(gold["seed"]))
    assert abs(got[0] - want[0]) < 2e-5 and abs(got[0] - gold["affine_loss"][0]) < 2e-5
    for i in (1, 2):
        assert abs(got[i] - want[i]) < 2e-2 and abs(got[i] - gold["affine_loss"][i]) < 2e-2
    orc, P, tr, got, want = run_color("f32", 8, 1, seed=2, lr=0.0)
    for k, v in orc.P.items():
        if getattr(v, "grad", None) is not None:
            off, n = tr.arena.slices[k]
            assert rel_err(tr.arena.grad[off:off + n], v.grad) < 2e-2, k
    orc, P, tr, got, want = run_color("bf16", 16, 3)
    for i in range(3):
        assert abs(got[i] - want[i]) < 5e-2 * max(1.0, abs(want[i])), (i, got[i], want[i])
