"""eg_dense_small_fwd_slices + eg_head_fused (include/eadgan_hip.h): the discriminator's 19-channel head (celebA/EAD-GAN_celebA.py:110-122),
the sub-step's losses (:334-345 G adversarial, :353-366 D real / fake, :375-401 the info step's MSE + CE + affine consistency) and the head's
input gradient in TWO launches, against the four or five stand-alone launches they replace (dense head + slice combine, loss kernels, dense
backward) -- bit for bit: head output, loss gradients, input gradient, and the batch losses where the stand-alone reduction owns one sample
per thread -- and against torch in fp64."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

eg = None
ops = None
DEV = "cuda"
N, CD, NC = 19, 8, 10
SLOPE = 0.2


def setup_module(module):
    global eg, ops
    eg = importlib.import_module("ead-gan_amd")
    ops = eg.ops


def _inputs(B, T, K, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    tdt = ops.torch_dtype(dtype)
    x = torch.randn(T * B, K, generator=g)
    x = torch.where(x > 0, x, x * SLOPE).to(DEV).to(tdt)                    # a leaky-ReLU output (its own gradient mask)
    wp = (torch.randn(N, K, generator=g) * 0.02).to(DEV).to(tdt)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    sigma = (torch.rand(T, generator=g) + 0.5).to(DEV)
    code = (torch.rand(B, CD, generator=g) * 2 - 1).to(DEV)
    labels = torch.randint(0, NC, (B,), generator=g).to(DEV)
    return x, wp, bias, sigma, code, labels


def _fused(dtype, x, wp, bias, sigma, B, T, K, loss, **kw):
    tdt = ops.torch_dtype(dtype)
    y = torch.empty(T * B, N, device=DEV)
    dout = torch.full((T * B, N), float("nan"), device=DEV)
    dx = torch.empty(T * B, K, device=DEV, dtype=tdt)
    terms = torch.empty(3 * B, device=DEV)
    counter = torch.zeros(1, device=DEV, dtype=torch.int32)
    ws = torch.empty(16 * T * B * N, device=DEV)
    ns = ops.dense_small_fwd_slices(dtype, x, wp, T * B, K, K, N, ws)
    ops.head_fused(dtype, x, wp, bias, ws, ns, y, dout, dx, sigma, B, T, K, K, N, loss, terms, counter, eg.ops.ACT_LRELU, SLOPE, **kw)
    torch.cuda.synchronize()
    assert int(counter.item()) == 0                                        # the arrival counter is left zero for the next launch
    return y, dout, dx


def _check_common(dtype, x, wp, bias, sigma, B, T, K, y, dout, dx):
    # the head output against fp64
    ref = x.double() @ wp.double().t() + bias.double()
    assert torch.allclose(y.double(), ref, rtol=0, atol=2e-4 * float(ref.abs().max()))
    # ... and against the sliced stand-alone launch
    y2 = torch.empty_like(y)
    ws = torch.empty(16 * T * B * N, device=DEV)
    ops.dense_small_fwd(dtype, x, wp, bias, y2, T * B, K, K, N, ws)
    assert torch.equal(y, y2)
    # the input gradient from the fused launch's own loss gradient: the stand-alone kernel gives the same bits
    dx2 = torch.empty_like(dx)
    ops.dense_small_bwd(dtype, dout, wp, x, dx2, T * B, K, K, N, eg.ops.ACT_LRELU, SLOPE, sigma, B)
    torch.cuda.synchronize()
    assert torch.equal(dx.view(torch.int16) if dx.element_size() == 2 else dx, dx2.view(torch.int16) if dx2.element_size() == 2 else dx2)
    # ... and torch agrees within the storage type's rounding
    t_dx = (dout.double() @ wp.double()) * torch.where(x > 0, 1.0, SLOPE).double() / sigma.double().repeat_interleave(B)[:, None]
    ok = t_dx.abs() < 6e4                                                   # (a random code can put the affine term's gradient beyond fp16's range)
    zero = torch.zeros((), device=DEV, dtype=torch.float64)
    err = torch.where(ok, (dx.double() - t_dx).abs(), zero).max() / torch.where(ok, t_dx.abs(), zero).max()
    assert err < (1e-5 if dtype == 0 else 1e-2), err


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("B,T", [(128, 1), (128, 2), (48, 2), (300, 1)])
def test_adversarial_head_matches_the_separate_launches(B, T, dtype):
    K = 2048 if dtype == 0 else 16384
    x, wp, bias, sigma, _, _ = _inputs(B, T, K, dtype, 11 + B + T)
    targets, scales = ((1.0,), (1.0,)) if T == 1 else ((1.0, 0.0), (0.5, 0.5))
    loss = torch.full((1,), 0.25, device=DEV)
    y, dout, dx = _fused(dtype, x, wp, bias, sigma, B, T, K, loss, targets=targets, scales=scales)
    _check_common(dtype, x, wp, bias, sigma, B, T, K, y, dout, dx)
    loss2 = torch.full((1,), 0.25, device=DEV)
    dout2 = torch.full_like(dout, float("nan"))
    for t in range(T):
        ops.loss_bce_sigmoid(y[t * B:(t + 1) * B], N, 0, B, targets[t], scales[t], loss2, dout2[t * B:(t + 1) * B])
    torch.cuda.synchronize()
    assert torch.equal(dout, dout2)
    if B <= 256:                                                            # one sample per thread of the stand-alone reduction: the same sum
        assert torch.equal(loss, loss2)
    else:
        assert torch.allclose(loss, loss2, rtol=1e-6)
    p = torch.sigmoid(y[:, 0].double())
    tl = sum(scales[t] * torch.nn.functional.binary_cross_entropy(p[t * B:(t + 1) * B], torch.full((B,), targets[t], device=DEV, dtype=torch.float64))
             for t in range(T))
    assert abs(float(loss) - 0.25 - float(tl)) < 1e-5 * max(1.0, abs(float(tl)))


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("B", [128, 40, 272])
def test_info_head_matches_the_separate_launches(B, dtype):
    T, K = 3, (2048 if dtype == 0 else 16384)
    x, wp, bias, sigma, code, labels = _inputs(B, T, K, dtype, 5 + B)
    lcat, lcon, laff = 1.0, 0.1, 0.5
    loss = torch.zeros(1, device=DEV)
    y, dout, dx = _fused(dtype, x, wp, bias, sigma, B, T, K, loss, info=(1, CD, NC, code, labels, lcat, lcon, laff))
    _check_common(dtype, x, wp, bias, sigma, B, T, K, y, dout, dx)
    loss2 = torch.zeros(1, device=DEV)
    dout2 = torch.full_like(dout, float("nan"))
    ops.loss_info_rpqxy(y[:B], y[B:2 * B], y[2 * B:], N, 1, CD, NC, B, code, CD, labels, lcat, lcon, laff, loss2, dout2[:B], dout2[B:2 * B], dout2[2 * B:])
    torch.cuda.synchronize()
    assert torch.equal(dout, dout2)
    if B <= 128:
        assert torch.equal(loss, loss2)
    else:
        assert torch.allclose(loss, loss2, rtol=1e-6)


def test_fused_head_refuses_other_heads():
    assert not ops.head_fused_ok(1, 4, 16384, 19)
    assert not ops.head_fused_ok(1, 2, 16384, 12)
    assert not ops.head_fused_ok(1, 2, 16380, 19)
    assert ops.head_fused_ok(0, 3, 16380, 19)
