"""Per-kernel parity: hand-written HIP kernels (through the C ABI) vs torch-CPU fp32 on the same seeded inputs.
fp32 kernels use the exact-fp32 MFMA -> tight tolerances; bf16 kernels round inputs to bf16 (fp32 accumulate)
-> tolerances scale with sqrt(K) * 2^-8."""
import importlib
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

eg = None
ops = None


def setup_module(module):
    global eg, ops
    eg = importlib.import_module("ead-gan_amd")
    ops = eg.ops


DEV = "cuda"
DTYPES = [0, 1, 2]      # EG_F32, EG_BF16, EG_F16


def tol(dtype, K):
    return (2e-5, 2e-5) if dtype == 0 else (3e-2, 2e-2 * math.sqrt(max(K, 1)) / 8)


def nhwc(x, dtype):
    """NCHW fp32 CPU -> NHWC device tensor in compute dtype"""
    return x.permute(0, 2, 3, 1).contiguous().to(DEV).to(ops.torch_dtype(dtype))


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rq(x, dtype):
    """round a CPU fp32 tensor through the compute dtype (what the kernel sees)"""
    return x if dtype == 0 else x.to(torch.bfloat16 if dtype == 1 else torch.float16).float()


CONV_CASES = [
    # B, H, Cin, Cout, k, s, p, up
    (2, 16, 32, 48, 4, 2, 1, 0),
    (3, 8, 64, 160, 4, 2, 1, 0),
    (2, 8, 16, 24, 3, 1, 1, 1),
    (2, 16, 16, 32, 3, 2, 1, 0),
    (5, 1, 24, 40, 1, 1, 0, 0),
    (4, 4, 128, 19 + 5, 4, 1, 0, 0),
    (1, 32, 8, 8, 4, 2, 1, 0),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(case, dtype):
    B, H, Cin, Cout, k, s, p, up = case
    g = torch.Generator().manual_seed(1)
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    c = ops.make_conv(B, H, H, Cin, Cout, k, s, p, up)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    want = F.leaky_relu(F.conv2d(xin, rq(w, dtype), b, s, p), 0.2)
    OH = want.shape[-1]
    y = torch.empty(B, OH, OH, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
    bias = b.to(DEV)
    ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, y, ops.epilogue(bias=bias, act=ops.ACT_LRELU, slope=0.2))
    torch.cuda.synchronize()
    rt, at = tol(dtype, Cin * k * k)
    torch.testing.assert_close(nchw(y), want, rtol=rt, atol=at)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES[:4] + [(2, 16, 3, 32, 4, 2, 1, 0)])
def test_conv_bwd_data(case, dtype):
    """dX = conv_transpose2d(dY, W) incl. sigma scaling, activation-gradient mask and NCHW fp32 output."""
    B, H, Cin, Cout, k, s, p, up = case
    g = torch.Generator().manual_seed(2)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    OH = ((H << up) + 2 * p - k) // s + 1
    dy = rq(torch.randn(B, Cout, OH, OH, generator=g), dtype)
    c = ops.make_conv(B, H, H, Cin, Cout, k, s, p, up)
    wp = torch.empty(ops.pack_bwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_bwd(c, dtype, w.to(DEV), wp)
    XH = H << up
    want = F.conv_transpose2d(dy, rq(w, dtype), None, s, p, output_padding=XH - ((OH - 1) * s - 2 * p + k))
    sigma = torch.tensor([1.7], device=DEV)
    rt, at = tol(dtype, Cout * k * k / (s * s))
    if Cin % ops.vec(dtype) == 0:
        a = rq(torch.randn(B, Cin, XH, XH, generator=g), dtype)
        dx = torch.empty(B, XH, XH, Cin, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_bwd_data(c, dtype, nhwc(dy, dtype), wp, dx,
                          ops.epilogue(sigma=sigma, mask=nhwc(a, dtype), mask_act=ops.ACT_LRELU, mask_slope=0.1))
        torch.cuda.synchronize()
        torch.testing.assert_close(nchw(dx), want / 1.7 * torch.where(a > 0, 1.0, 0.1), rtol=rt, atol=at)
    out = torch.empty(B, Cin, XH, XH, device=DEV, dtype=torch.float32)
    b = torch.randn(Cin, generator=g)
    ops.conv_bwd_data(c, dtype, nhwc(dy, dtype), wp, out, ops.epilogue(bias=b.to(DEV), act=ops.ACT_TANH, out_mode=ops.OUT_NCHW_F32))
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu(), torch.tanh(want + b[None, :, None, None]), rtol=rt, atol=at)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES[:5] + [(4, 32, 16, 32, 4, 2, 1, 0), (8, 16, 128, 256, 4, 2, 1, 0)])
def test_conv_wgrad(case, dtype):
    B, H, Cin, Cout, k, s, p, up = case
    g = torch.Generator().manual_seed(3)
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    y = F.conv2d(xin, w, None, s, p)
    dy = rq(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    c = ops.make_conv(B, H, H, Cin, Cout, k, s, p, up)
    slab = torch.empty(ops.conv_wgrad_ws_bytes(c, dtype) // 4, device=DEV)
    ns = ops.conv_wgrad(c, dtype, nhwc(x, dtype), nhwc(dy, dtype), slab)
    grad = torch.ones(Cout, Cin, k, k, device=DEV)
    ops.wgrad_reduce(slab, ns, Cout, Cout, Cin, k * k, grad, accumulate=True)
    torch.cuda.synchronize()
    rt, at = tol(dtype, B * y.shape[-1] ** 2)
    torch.testing.assert_close(grad.cpu() - 1.0, w.grad, rtol=rt, atol=at * 4)
    # spectral-norm reduction: grad += G/sigma - <G,W>/sigma^2 u v^T
    u = F.normalize(torch.randn(Cout, generator=g), dim=0)
    v = F.normalize(torch.randn(Cin * k * k, generator=g), dim=0)
    sig = torch.tensor([1.3])
    gtmp = torch.empty(Cout * Cin * k * k, device=DEV)
    part = torch.empty(4096, device=DEV)
    grad2 = torch.zeros(Cout, Cin, k, k, device=DEV)
    ops.wgrad_reduce_sn(c, slab, ns, w.detach().to(DEV), sig.to(DEV), u.to(DEV), v.to(DEV), gtmp, part, grad2)
    torch.cuda.synchronize()
    G = w.grad
    want = G / 1.3 - (G * w.detach()).sum() / 1.3 ** 2 * torch.outer(u, v).view_as(G)
    torch.testing.assert_close(grad2.cpu(), want, rtol=rt * 2, atol=at * 8)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 8, 16, 24), (2, 16, 128, 64), (4, 4, 128, 128), (2, 2, 8, 8)])
def test_upsample_conv3_as_transposed_conv4(case, dtype):
    """nn.Upsample(scale_factor=2) + Conv2d(Cin -> Cout, 3, 1, 1) (MNIST/EAD-GAN_rpqmnxy.py:81-82, 85-86) through eg_up3_expand + the
    transposed 4x4 / stride-2 / pad-1 convolution: forward, input gradient (low resolution, no sum-pool) and weight gradient
    (eg_up3_contract) against torch's upsample + conv2d and its autograd"""
    B, H, Cin, Cout = case
    g = torch.Generator().manual_seed(11)
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype).requires_grad_(True)
    w3 = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(Cout, generator=g)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w3, b, 1, 1)
    dy = rq(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    tdt = ops.torch_dtype(dtype)
    w4 = torch.empty(Cin, Cout, 4, 4, device=DEV)
    ops.up3_expand(w3.detach().to(DEV), w4, Cout, Cin)
    torch.cuda.synchronize()
    # the expansion against its definition, and the forward in torch through the expanded weight (exact identity up to rounding)
    torch.testing.assert_close(F.conv_transpose2d(x.detach(), w4.cpu(), b, 2, 1), y.detach(), rtol=1e-5, atol=1e-5)
    c = ops.make_conv(B, 2 * H, 2 * H, Cout, Cin, 4, 2, 1)           # conv view: input = the 2H x 2H output side
    wpf = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=tdt)
    wpb = torch.empty(ops.pack_bwd_elems(c, dtype), device=DEV, dtype=tdt)
    ops.pack_conv(c, dtype, w4, wpf, wpb)
    got = torch.empty(B, 2 * H, 2 * H, Cout, device=DEV, dtype=tdt)
    ops.conv_bwd_data(c, dtype, nhwc(x.detach(), dtype), wpb, got, ops.epilogue(bias=b.to(DEV)))
    dx = torch.empty(B, H, H, Cin, device=DEV, dtype=tdt)
    ops.conv_fwd(c, dtype, nhwc(dy, dtype), wpf, dx, None)
    slab = torch.empty(ops.conv_wgrad_ws_bytes(c, dtype) // 4, device=DEV)
    ns = ops.conv_wgrad(c, dtype, nhwc(dy, dtype), nhwc(x.detach(), dtype), slab)
    dw4 = torch.empty(Cin, Cout, 4, 4, device=DEV)
    ops.wgrad_reduce(slab, ns, Cin, Cin, Cout, 16, dw4, accumulate=False)
    dw3 = torch.ones(Cout, Cin, 3, 3, device=DEV)
    ops.up3_contract(dw4, dw3, Cout, Cin)                            # accumulates
    torch.cuda.synchronize()
    # 16-bit: the packed taps are the ROUNDED tap sums (one rounding of w0 + w1 instead of two products): within the 16-bit bound
    rt, at = tol(dtype, Cin * 9)
    torch.testing.assert_close(nchw(got), y.detach(), rtol=rt, atol=at)
    torch.testing.assert_close(nchw(dx), x.grad, rtol=rt, atol=tol(dtype, Cout * 9)[1])
    rt, at = tol(dtype, B * 4 * H * H)
    torch.testing.assert_close(dw3.cpu() - 1.0, w3.grad, rtol=rt, atol=at * 16)      # (sums of up to four 4x4 entries of ~2000-term sums each)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H", [(3, 32), (300, 32), (2, 8), (1, 16)])
def test_single_output_channel_weight_gradient_reads_the_activation_once(B, H, dtype):
    """eg_wgrad_c1 (weight gradient of Conv2d(64, 1, 3, 1, 1), MNIST/EAD-GAN_rpqmnxy.py:88) against torch's autograd on the operands as the
    kernel sees them (x and dy rounded through the compute dtype) and against the per-tap GEMM path it replaces; more units than workgroups
    at B = 300 (several units per workgroup)"""
    g = torch.Generator().manual_seed(21)
    x = rq(torch.randn(B, 64, H, H, generator=g), dtype)
    w = (torch.randn(1, 64, 3, 3, generator=g) * 0.1).requires_grad_(True)
    dy32 = torch.randn(B, 1, H, H, generator=g)
    dy = rq(dy32, dtype)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    assert ops.wgrad_c1_ok(dtype, 64, H, H, 1, 3, 1, 1)
    slab = torch.empty(ops.wgrad_c1_splits(B, H) * 9 * 64, device=DEV)
    xd = nhwc(x, dtype)
    ns = ops.wgrad_c1(dtype, xd, dy32.to(DEV), slab, B, H, H, 64)
    assert ns == ops.wgrad_c1_splits(B, H)
    grad = torch.ones(1, 64, 3, 3, device=DEV)
    ops.wgrad_reduce(slab, ns, 1, 1, 64, 9, grad)
    # the GEMM path: output gradient padded to 8 channels, per-tap kernel
    c = ops.make_conv(B, H, H, 64, 8, 3, 1, 1)
    p8 = torch.empty(B * H * H, 8, device=DEV, dtype=ops.torch_dtype(dtype))
    ops.cast_pad(dtype, dy32.to(DEV), p8, B * H * H, 1, 8)
    slab2 = torch.empty(ops.conv_wgrad_ws_bytes(c, dtype) // 4, device=DEV)
    ns2 = ops.conv_wgrad(c, dtype, xd, p8, slab2)
    grad2 = torch.ones(1, 64, 3, 3, device=DEV)
    ops.wgrad_reduce(slab2, ns2, 8, 1, 64, 9, grad2)
    torch.cuda.synchronize()
    rt, at = tol(dtype, B * H * H)
    at = max(at * 4, 2e-5 * math.sqrt(B * H * H))      # entries are sums of B*H*H products of O(1): rounding of the running sum grows with it
    torch.testing.assert_close(grad.cpu() - 1.0, w.grad, rtol=rt, atol=at)
    torch.testing.assert_close(grad.cpu(), grad2.cpu(), rtol=rt, atol=at)
    assert not ops.wgrad_c1_ok(dtype, 32, H, H, 1, 3, 1, 1) and not ops.wgrad_c1_ok(dtype, 64, H, H, 3, 3, 1, 1) and not ops.wgrad_c1_ok(dtype, 64, H, H, 1, 4, 2, 1)


# B, H (input), Cin, Cout: 4x4 / stride-2 / pad-1 layers with channel counts in multiples of 128 -> the parity-class weight-gradient
# kernel (igemm_tn8.hip).  Output lattices 16x16 (bands of 4 rows per K step), 8x8 (one image per step), 4x4 (four images per step),
# 32x32 (2 rows per step), 2x2 (16 images per step); several splits over m; both channel tilings > 1
TN8_CASES = [(8, 32, 128, 256), (12, 16, 256, 128), (16, 8, 128, 128), (4, 64, 128, 128), (64, 4, 256, 256), (5 * 4, 16, 128, 128),
             # 64-channel tiles (the dSprites networks' layers; 192 = 3 tiles of 64)
             (8, 64, 64, 64), (16, 32, 64, 64), (16, 16, 64, 192), (32, 8, 64, 64), (24, 8, 192, 64),
             # 32-channel tiles (first trunk layers of the dSprites networks: 32 -> 32 at 32x32 inputs, 32 -> 64 at 16x16; P tile = four DMA pieces)
             (8, 32, 32, 32), (16, 16, 32, 64), (12, 64, 32, 32), (16, 8, 96, 32), (32, 16, 32, 32)]


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("case", TN8_CASES)
def test_conv_wgrad_parity_class_kernel(case, dtype):
    """igemm_tn8: every tap of every parity class lands in the right slab row with the right border handling -- compared with the
    autograd weight gradient of the same 16-bit-rounded operands, per tap."""
    B, H, Cin, Cout = case
    g = torch.Generator().manual_seed(31)
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 1)
    dy = rq(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
    import ctypes
    # H = 4 (2x2 outputs: 16 images x 3x3 patch = 144 pixels per K step) is over the kernel's 136 patch slots: the per-tap kernel runs it
    assert eg._lib.lib().query("eg_conv_wgrad_variant", ctypes.byref(c), dtype) == (1 if H == 4 else 2)
    nbytes = ops.conv_wgrad_ws_bytes(c, dtype)
    slab = torch.full((nbytes // 4,), float("nan"), device=DEV)
    ns = ops.conv_wgrad(c, dtype, nhwc(x, dtype), nhwc(dy, dtype), slab)
    assert nbytes == ns * Cout * 16 * Cin * 4
    grad = torch.zeros(Cout, Cin, 4, 4, device=DEV)
    ops.wgrad_reduce(slab, ns, Cout, Cout, Cin, 16, grad, accumulate=True)
    torch.cuda.synchronize()
    rt, at = tol(dtype, B * y.shape[-1] ** 2)
    got, want = grad.cpu(), w.grad
    for t in range(16):
        torch.testing.assert_close(got[:, :, t // 4, t % 4], want[:, :, t // 4, t % 4], rtol=rt, atol=at * 4, msg=lambda m, t=t: f"tap {t}: {m}")
    # deterministic: no atomics, fixed reduction order
    slab2 = torch.zeros_like(slab)
    ops.conv_wgrad(c, dtype, nhwc(x, dtype), nhwc(dy, dtype), slab2)
    torch.cuda.synchronize()
    assert torch.equal(slab, slab2)
    # chip-share hint of the side lanes (eg_conv_wgrad_target): fewer, longer K splits, the same gradient
    slab3 = torch.full_like(slab, float("nan"))
    ns3 = ops.conv_wgrad(c, dtype, nhwc(x, dtype), nhwc(dy, dtype), slab3, 16)
    assert 1 <= ns3 <= ns
    grad3 = torch.zeros_like(grad)
    ops.wgrad_reduce(slab3, ns3, Cout, Cout, Cin, 16, grad3, accumulate=True)
    torch.cuda.synchronize()
    torch.testing.assert_close(grad3.cpu(), got, rtol=1e-4, atol=at)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 8, 64, 1e-5, 2), (3, 16, 24, 0.8, 1), (4, 4, 128, 1e-5, 0)])
def test_batchnorm(shape, dtype):
    B, H, C, eps, act = shape
    g = torch.Generator().manual_seed(4)
    x = rq(torch.randn(B, C, H, H, generator=g) * 2 + 0.5, dtype).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C, eps)
    with torch.no_grad():
        bn.weight.normal_(1, 0.2, generator=g)
        bn.bias.normal_(0, 0.2, generator=g)
    fact = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu}[act]
    y = fact(bn(x))
    da = rq(torch.randn(y.shape, generator=g), dtype)
    y.backward(da)
    M = B * H * H
    tdt = ops.torch_dtype(dtype)
    ws = torch.empty(max(ops.bn_ws_floats(M, C), 1), device=DEV)
    gam, bet = bn.weight.detach().to(DEV), bn.bias.detach().to(DEV)
    rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    sm, si = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    xd = nhwc(x.detach(), dtype)
    yd = torch.empty_like(xd)
    ops.bn_fwd_train(dtype, xd, yd, M, C, gam, bet, eps, 0.1, rm, rv, nbt, sm, si, ws, act, 0.2)
    torch.cuda.synchronize()
    rt, at = (1e-5, 2e-5) if dtype == 0 else (2e-2, 2e-2)
    torch.testing.assert_close(nchw(yd), y.detach(), rtol=rt, atol=at)
    torch.testing.assert_close(rm.cpu(), bn.running_mean, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-5)
    assert int(nbt) == 1
    dz = torch.empty_like(xd)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    sums = torch.empty(2 * C, device=DEV)
    ops.bn_bwd(dtype, xd, nhwc(da, dtype), dz, M, C, gam, bet, sm, si, act, 0.2, dg, db, sums, ws)
    torch.cuda.synchronize()
    torch.testing.assert_close(nchw(dz), x.grad, rtol=rt * 5, atol=at * 2)
    torch.testing.assert_close(dg.cpu(), bn.weight.grad, rtol=rt * 5, atol=at * 10)
    torch.testing.assert_close(db.cpu(), bn.bias.grad, rtol=rt * 5, atol=at * 10)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 3, 64, 128, 4, 2, 1), (2, 1, 32, 16, 3, 2, 1), (2, 1, 64, 32, 4, 2, 1), (2, 1, 32, 64, 3, 1, 1)])
def test_conv_img(case, dtype):
    B, CI, H, N, k, s, p = case
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, CI, H, H, generator=g)
    w = (torch.randn(N, CI, k, k, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(N, generator=g)
    want = F.leaky_relu(F.conv2d(img, w, None, s, p) / 1.5 + b[None, :, None, None], 0.1)
    OH = want.shape[-1]
    out = torch.empty(B, OH, OH, N, device=DEV, dtype=ops.torch_dtype(dtype))
    sig = torch.tensor([1.5], device=DEV)
    ops.conv_img_fwd(dtype, img.to(DEV), w.detach().to(DEV), out, B, CI, H, H, N, k, s, p,
                     ops.epilogue(bias=b.to(DEV), sigma=sig, act=ops.ACT_LRELU, slope=0.1))
    torch.cuda.synchronize()
    rt, at = (1e-5, 1e-5) if dtype == 0 else (1e-2, 1e-2)
    torch.testing.assert_close(nchw(out), want.detach(), rtol=rt, atol=at)
    # weight gradient
    dz = rq(torch.randn(B, N, OH, OH, generator=g), dtype)
    F.conv2d(img, w, None, s, p).backward(dz)
    slab = torch.empty(ops.conv_img_wgrad_ws_bytes(B, CI, N, k) // 4, device=DEV)
    ops.conv_img_wgrad(dtype, nhwc(dz, dtype), img.to(DEV), slab, B, CI, H, H, N, k, s, p)
    grad = torch.zeros(N, CI, k, k, device=DEV)
    ops.flat_reduce(slab, B, grad.numel(), grad)
    torch.cuda.synchronize()
    torch.testing.assert_close(grad.cpu(), w.grad, rtol=1e-4, atol=2e-4 * math.sqrt(B * OH * OH))


@pytest.mark.parametrize("dtype", DTYPES)
def test_dense_small(dtype):
    B, C, T, N = 6, 64, 16, 19
    K = C * T
    g = torch.Generator().manual_seed(6)
    x = rq(torch.randn(B, C, 4, 4, generator=g), dtype).requires_grad_(True)        # NCHW reference
    w = (torch.randn(N, C, 4, 4, generator=g) * 0.05).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    a = F.leaky_relu(x, 0.1)
    dy = torch.randn(B, N, generator=g)
    y = F.conv2d(a, rq(w.detach(), dtype), b.detach()).squeeze()        # forward / input gradient see the rounded weights
    y.backward(dy)
    F.conv2d(rq(a.detach(), dtype), w, b).squeeze().backward(dy)           # weight / bias gradient
    c = ops.make_conv(B, 4, 4, C, N, 4, 1, 0)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.detach().to(DEV), wp)
    Kpad = ops.round_up(K, ops.bk(dtype))
    ad = nhwc(rq(a.detach(), dtype), dtype)
    yd = torch.empty(B, N, device=DEV)
    ops.dense_small_fwd(dtype, ad, wp, b.detach().to(DEV), yd, B, K, Kpad, N)
    torch.cuda.synchronize()
    rt, at = (1e-5, 2e-5) if dtype == 0 else (2e-2, 5e-2)
    torch.testing.assert_close(yd.cpu(), y.detach(), rtol=rt, atol=at)
    dx = torch.empty_like(ad)
    ops.dense_small_bwd(dtype, dy.to(DEV), wp, ad, dx, B, K, Kpad, N, ops.ACT_LRELU, 0.1)
    gw, gb = torch.zeros(N, C, 4, 4, device=DEV), torch.zeros(N, device=DEV)
    ops.dense_small_wgrad(dtype, dy.to(DEV), ad, gw, gb, B, K, N, C, T)
    torch.cuda.synchronize()
    torch.testing.assert_close(nchw(dx), x.grad, rtol=rt, atol=at)
    torch.testing.assert_close(gw.cpu(), w.grad, rtol=rt, atol=at)
    torch.testing.assert_close(gb.cpu(), b.grad, rtol=1e-5, atol=1e-5)


def test_spectral_norm_power_iteration():
    g = torch.Generator().manual_seed(7)
    for R, Kd in [(128, 48), (256, 2048), (1024, 8192), (16, 9)]:
        conv = torch.nn.Linear(Kd, R, bias=False)
        m = torch.nn.utils.spectral_norm(conv)
        w, u, v = m.weight_orig.detach().clone(), m.weight_u.clone(), m.weight_v.clone()
        ud, vd, sd = u.to(DEV), v.to(DEV), torch.empty(1, device=DEV)
        us, vs = torch.empty_like(ud), torch.empty_like(vd)
        ws = torch.empty(ops.sn_ws_floats(R, Kd), device=DEV)
        for it in range(3):
            m.train()
            _ = m(torch.zeros(1, Kd))         # one power iteration (training-mode forward)
            ops.sn_power_iter(w.to(DEV), R, Kd, ud, vd, sd, us, vs, ws, True, 1e-12)
            torch.cuda.synchronize()
            sigma_ref = torch.dot(m.weight_u, torch.mv(w, m.weight_v))
            torch.testing.assert_close(ud.cpu(), m.weight_u, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(vd.cpu(), m.weight_v, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(us.cpu(), m.weight_u, rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(sd.cpu()[0], sigma_ref, rtol=1e-5, atol=1e-6)


def test_adam_matches_torch():
    g = torch.Generator().manual_seed(8)
    n = 10007
    p = torch.randn(n, generator=g).requires_grad_(True)
    opt = torch.optim.Adam([p], lr=2e-4, betas=(0.5, 0.999))
    pd, m, v = p.detach().clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for it in range(5):
        gr = torch.randn(n, generator=g) * (10.0 ** (it - 3))
        p.grad = gr.clone()
        opt.step()
        ops.adam_step(pd, gr.to(DEV), m, v, n, 2e-4, 0.5, 0.999, 1e-8, step, True)
    torch.cuda.synchronize()
    assert int(step) == 5
    torch.testing.assert_close(pd.cpu(), p.detach(), rtol=1e-6, atol=1e-7)


def test_adam_bucket_slices_equal_the_whole_arena_update():
    """The trainers update bucket by bucket (eg_adam_step_zero on slices whose ends are not 16-byte aligned, float4 middle + scalar
    ends, gradient cleared in the same pass): bit-identical to one eg_adam_step over the arena; only the first slice ticks the counter."""
    g = torch.Generator().manual_seed(9)
    n = 10007
    p0 = torch.randn(n, generator=g).to(DEV)
    gr = torch.randn(n, generator=g).to(DEV)
    m0, v0 = torch.rand(n, generator=g).to(DEV) * 1e-2, torch.rand(n, generator=g).to(DEV) * 1e-4
    pa, ma, va, sa = p0.clone(), m0.clone(), v0.clone(), torch.full((1,), 3, dtype=torch.int32, device=DEV)
    ops.adam_step(pa, gr, ma, va, n, 2e-4, 0.5, 0.999, 1e-8, sa, True)
    pb, mb, vb, gb, sb = p0.clone(), m0.clone(), v0.clone(), gr.clone(), torch.full((1,), 3, dtype=torch.int32, device=DEV)
    cuts = [0, 3, 4, 6147, 6150, 9001, n]
    for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        ops.adam_step_zero(pb[lo:hi], gb[lo:hi], mb[lo:hi], vb[lo:hi], hi - lo, 2e-4, 0.5, 0.999, 1e-8, sb, k == 0, k % 2 == 0)
    torch.cuda.synchronize()
    assert int(sa) == 4 and int(sb) == 4
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        assert torch.equal(gb[lo:hi], torch.zeros(hi - lo, device=DEV) if k % 2 == 0 else gr[lo:hi]), k


def test_info_losses_in_one_launch_equal_the_three_launches():
    """eg_loss_info_rpqxy (the CelebA info step's MSE + CE + affine-consistency losses, celebA/EAD-GAN_celebA.py:390-396, as one launch)
    against eg_loss_mse, eg_loss_ce_softmaxed and eg_loss_affine_rpqxy launched one after the other: the same loss and the same three
    gradient blocks, bit for bit (the bodies are shared, non-inlined device functions)."""
    B, cd, nc = 37, 8, 10
    g = torch.Generator().manual_seed(12)
    out = torch.randn(3 * B, 19, generator=g).to(DEV)
    code = (torch.rand(B, cd, generator=g) * 2 - 1).to(DEV)
    labels = torch.randint(0, nc, (B,), generator=g).to(DEV)
    o_gen, o_trans, o_real = out[:B], out[B:2 * B], out[2 * B:]

    def three():
        L, d = torch.zeros(4, device=DEV), torch.full((3 * B, 19), 7.0, device=DEV)
        ops.loss_mse(o_gen, 19, 1, cd, B, code, cd, 0.0, 0.7, L[2:3], d[:B])
        ops.loss_ce_softmaxed(o_gen, 19, cd + 1, nc, B, labels, 1.3, L[2:3], d[:B])
        ops.loss_affine_rpqxy(o_real, o_trans, 19, 1, B, code, cd, 0.9, L[2:3], d[2 * B:], d[B:2 * B])
        return L, d

    def one():
        L, d = torch.zeros(4, device=DEV), torch.full((3 * B, 19), 7.0, device=DEV)
        ops.loss_info_rpqxy(o_gen, o_trans, o_real, 19, 1, cd, nc, B, code, cd, labels, 1.3, 0.7, 0.9, L[2:3], d[:B], d[B:2 * B], d[2 * B:])
        return L, d
    (La, da), (Lb, db) = three(), one()
    torch.cuda.synchronize()
    assert torch.equal(La, Lb) and torch.equal(da, db) and float(La[2]) > 0


def test_warp_and_theta_match_golden():
    import os
    from conftest import GOLDEN
    from oracle import celeba_oracle as co
    gold = np.load(os.path.join(GOLDEN, "celeba_affine.npz"))
    code = torch.tensor(gold["code"]).to(DEV)
    A = eg.celeba.get_matrix(code[:, :5].contiguous())
    np.testing.assert_allclose(A.cpu().numpy(), gold["A"], rtol=2e-6, atol=2e-7)
    img = co.synthetic_real(4, seed=int(gold["img_seed"])).to(DEV)
    warped = eg.celeba.transformation_2D()(img, A[:4, 0:2])
    np.testing.assert_allclose(warped.cpu().numpy(), gold["warped"], rtol=1e-4, atol=2e-5)


def test_affine_regularizer_matches_golden():
    import os
    from conftest import GOLDEN
    gold = np.load(os.path.join(GOLDEN, "celeba_affine.npz"))
    rc = torch.tensor(gold["real_code"], device=DEV, requires_grad=True)
    tc = torch.tensor(gold["trans_code"], device=DEV, requires_grad=True)
    pred = eg.celeba.affine_regularzier(rc, tc)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), gold["pred"], rtol=2e-4, atol=2e-5)
    (pred * torch.tensor(gold["w"], device=DEV)).sum().backward()
    np.testing.assert_allclose(rc.grad.cpu().numpy()[:, :5], gold["d_real"][:, :5], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(tc.grad.cpu().numpy()[:, :5], gold["d_trans"][:, :5], rtol=2e-3, atol=2e-4)
    assert float(rc.grad[:, 5:].abs().max()) == 0.0


def test_loss_heads():
    g = torch.Generator().manual_seed(9)
    B = 37
    o = (torch.randn(B, 19, generator=g) * 2).requires_grad_(True)
    code = torch.rand(B, 8, generator=g) * 2 - 1
    labels = torch.randint(0, 10, (B,), generator=g)
    od = o.detach().to(DEV)
    loss = torch.zeros(4, device=DEV)
    dout = torch.full((B, 19), 7.0, device=DEV)
    # BCE(sigmoid(o0), 1) * 0.5
    l = 0.5 * F.binary_cross_entropy(torch.sigmoid(o[:, 0]), torch.ones(B))
    (gr,) = torch.autograd.grad(l, o)
    ops.loss_bce_sigmoid(od, 19, 0, B, 1.0, 0.5, loss[0:1], dout)
    torch.cuda.synchronize()
    torch.testing.assert_close(loss[0].cpu(), l.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dout.cpu(), gr, rtol=1e-4, atol=1e-7)
    # info loss: CE(softmax(o[9:19])) + MSE(o[1:9], code)
    l = F.cross_entropy(F.softmax(o[:, 9:19], dim=1), labels) + F.mse_loss(o[:, 1:9], code)
    (gr,) = torch.autograd.grad(l, o)
    ops.loss_mse(od, 19, 1, 8, B, code.to(DEV), 8, 0.0, 1.0, loss[1:2], dout)
    ops.loss_ce_softmaxed(od, 19, 9, 10, B, labels.to(DEV), 1.0, loss[1:2], dout)
    torch.cuda.synchronize()
    torch.testing.assert_close(loss[1].cpu(), l.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dout.cpu(), gr, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("dtype", DTYPES)
def test_image_side_layers_via_im2col(dtype):
    """first D conv / last G ConvTranspose gradients as 1x1-conv GEMMs over im2col patches (MFMA path)."""
    B, CI, H, N, k, s, p = 3, 3, 64, 128, 4, 2, 1
    g = torch.Generator().manual_seed(11)
    img = torch.randn(B, CI, H, H, generator=g)
    w = (torch.randn(N, CI, k, k, generator=g) * 0.2).requires_grad_(True)
    b = torch.randn(N, generator=g)
    OH = H // 2
    Kp = CI * k * k
    tdt = ops.torch_dtype(dtype)
    patches = torch.empty(B * OH * OH, Kp, device=DEV, dtype=tdt)
    ops.im2col_img(dtype, img.to(DEV), patches, B, CI, H, H, k, s, p, Kp)
    want_p = F.unfold(img, k, padding=p, stride=s).transpose(1, 2).reshape(B * OH * OH, Kp)
    torch.testing.assert_close(patches.float().cpu(), rq(want_p, dtype), rtol=0, atol=0)
    c = ops.make_conv(B, OH, OH, Kp, N, 1, 1, 0)
    Kpad = ops.round_up(Kp, ops.bk(dtype))
    wp = torch.empty(N * Kpad, device=DEV, dtype=tdt)
    ops.pack_strided(dtype, w.detach().to(DEV), wp, N, Kp, Kpad, 1, Kp, 0, 1)
    y = torch.empty(B, OH, OH, N, device=DEV, dtype=tdt)
    ops.conv_fwd(c, dtype, patches, wp, y, ops.epilogue(bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.1))
    want = F.leaky_relu(F.conv2d(rq(img, dtype), rq(w.detach(), dtype), b, s, p), 0.1)
    rt, at = tol(dtype, Kp)
    torch.testing.assert_close(nchw(y), want, rtol=rt, atol=at)
    dz = rq(torch.randn(B, N, OH, OH, generator=g), dtype)
    F.conv2d(rq(img, dtype), w, None, s, p).backward(dz)
    slab = torch.empty(ops.conv_wgrad_ws_bytes(c, dtype) // 4, device=DEV)
    ns = ops.conv_wgrad(c, dtype, patches, nhwc(dz, dtype), slab)
    grad = torch.zeros(N, CI, k, k, device=DEV)
    ops.wgrad_reduce(slab, ns, N, N, Kp, 1, grad)
    torch.cuda.synchronize()
    rt, at = tol(dtype, B * OH * OH)
    torch.testing.assert_close(grad.cpu(), w.grad, rtol=rt * 5, atol=at * math.sqrt(B * OH * OH) / 4)   # 3072-term fp32 sums


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,B,Hin,k,s,p", [(3, 3, 32, 4, 2, 1), (1, 2, 16, 4, 2, 1), (3, 2, 7, 3, 1, 1), (1, 5, 5, 4, 1, 0)])
def test_transposed_conv_as_gemm_plus_col2im(dtype, C, B, Hin, k, s, p):
    """ConvTranspose2d(128 -> C) as one GEMM over the input pixels (N = k*k*C columns, eg_conv_fwd on a 1x1 geometry) + eg_col2im_img:
    the gather alone is exact against F.fold; the pair against F.conv_transpose2d (+ bias, tanh)."""
    CI = 128
    g = torch.Generator().manual_seed(21)
    tdt = ops.torch_dtype(dtype)
    OH = (Hin - 1) * s - 2 * p + k
    ncol = k * k * C
    # (1) the gather: out = fold(cols); cols[m][t*C + c]  <->  fold input [B, C*k*k (c-major), L]
    cols = rq(torch.randn(B * Hin * Hin, ncol, generator=g), dtype)
    out = torch.empty(B, C, OH, OH, device=DEV)
    ops.col2im_img(dtype, cols.to(DEV).to(tdt), B, C, Hin, Hin, k, s, p, None, ops.ACT_NONE, 0.0, out)
    fin = cols.view(B, Hin * Hin, k * k, C).permute(0, 3, 2, 1).reshape(B, C * k * k, Hin * Hin)
    want = F.fold(fin, (OH, OH), k, padding=p, stride=s)
    torch.testing.assert_close(out.cpu(), want, rtol=1e-6, atol=1e-6)
    # (2) the layer (the GEMM engine wants power-of-two lattices and N a multiple of the vector width)
    if Hin & (Hin - 1) or ncol % 8:
        return
    x = rq(torch.randn(B, CI, Hin, Hin, generator=g), dtype)
    w = rq(torch.randn(CI, C, k, k, generator=g) * 0.1, dtype)
    bias = torch.randn(C, generator=g)
    c = ops.make_conv(B, Hin, Hin, CI, ncol, 1, 1, 0)
    Kpad = ops.round_up(CI, ops.bk(dtype))
    wp = torch.empty(ncol * Kpad, device=DEV, dtype=tdt)
    ops.pack_strided(dtype, w.to(DEV), wp, ncol, CI, Kpad, C, 1, k * k, k * k * C)        # wp[t*C + c][ci] = w[ci][c][t]
    colsd = torch.empty(B * Hin * Hin, ncol, device=DEV, dtype=tdt)
    ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, colsd, None)
    ops.col2im_img(dtype, colsd, B, C, Hin, Hin, k, s, p, bias.to(DEV), ops.ACT_TANH, 0.0, out)
    want = torch.tanh(F.conv_transpose2d(x, w, bias, s, p))
    rt, at = tol(dtype, CI * k * k // (s * s))
    torch.testing.assert_close(out.cpu(), want, rtol=rt, atol=at * 2)        # + one rounding of the per-tap partial sums to the compute dtype


def test_tanh_backward_with_bias_gradient_and_cast_pad():
    g = torch.Generator().manual_seed(12)
    B, C, HW = 5, 3, 4096
    a = torch.tanh(torch.randn(B, C, HW, generator=g))
    gr = torch.randn(B, C, HW, generator=g)
    out = torch.empty(B, C, HW, device=DEV)
    part = torch.empty(B * C, device=DEV)
    gb = torch.ones(C, device=DEV)
    ops.act_grad_mul_bias_nchw(gr.to(DEV), a.to(DEV), out, B, C, HW, ops.ACT_TANH, 0.0, part, gb)
    want = gr * (1 - a * a)
    torch.testing.assert_close(out.cpu(), want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gb.cpu() - 1, want.sum(dim=(0, 2)), rtol=1e-4, atol=1e-3)
    src = torch.randn(7, 19, generator=g)
    for dtype in DTYPES:
        dst = torch.full((7, 32), 3.0, device=DEV).to(ops.torch_dtype(dtype))
        ops.cast_pad(dtype, src.to(DEV), dst, 7, 19, 32)
        torch.testing.assert_close(dst.float().cpu()[:, :19], rq(src, dtype), rtol=0, atol=0)
        assert float(dst.float().abs()[:, 19:].max()) == 0.0


def test_spectral_norm_multi_layer_launch_matches_single(monkeypatch):
    g = torch.Generator().manual_seed(13)
    shapes = [(128, 48), (256, 2048), (512, 4096), (16, 9)]
    ws_single = torch.empty(max(ops.sn_ws_floats(r, k) for r, k in shapes), device=DEV)
    ent, ref = [], []
    for R, Kd in shapes:
        w = torch.randn(R, Kd, generator=g).to(DEV)
        u = F.normalize(torch.randn(R, generator=g), dim=0).to(DEV)
        v = F.normalize(torch.randn(Kd, generator=g), dim=0).to(DEV)
        sg, us, vs = torch.empty(1, device=DEV), torch.empty(R, device=DEV), torch.empty(Kd, device=DEV)
        u1, v1, s1 = u.clone(), v.clone(), torch.empty(1, device=DEV)
        ops.sn_power_iter(w, R, Kd, u1, v1, s1, None, None, ws_single, True, 1e-12)
        ent.append((w, u, v, sg, us, vs))
        ref.append((u1, v1, s1))
    arr = ops.sn_layers(ent)
    ws = torch.empty(ops.sn_multi_ws_floats(arr), device=DEV)
    ops.sn_power_iter_multi(arr, ws, True, 1e-12)
    torch.cuda.synchronize()
    for (w, u, v, sg, us, vs), (u1, v1, s1) in zip(ent, ref):
        torch.testing.assert_close(u, u1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(v, v1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(us, u1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(vs, v1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sg, s1, rtol=1e-5, atol=1e-6)
    # the two-launch form (each stage finished by the layer's last workgroup to arrive): the four-launch form's bits, twice in a row
    # (the arrival counters must come back to zero)
    state = [(u.clone(), v.clone()) for (w, u, v, sg, us, vs) in ent]
    ops.sn_power_iter_multi(arr, ws, True, 1e-12)
    four = [(u.clone(), v.clone(), sg.clone(), us.clone(), vs.clone()) for (w, u, v, sg, us, vs) in ent]
    counters = torch.zeros(16, device=DEV, dtype=torch.int32)
    monkeypatch.setattr(ops, "SN_TWO_LAUNCHES", True)       # (an experiment switch, default off: EG_SN2)
    for rep in range(2):
        for (w, u, v, sg, us, vs), (u0, v0) in zip(ent, state):
            u.copy_(u0); v.copy_(v0); sg.fill_(-1); us.fill_(-1); vs.fill_(-1)
        ops.sn_power_iter_multi(arr, ws, True, 1e-12, counters)
        torch.cuda.synchronize()
        assert int(counters.abs().sum()) == 0
        for (w, u, v, sg, us, vs), (u4, v4, s4, us4, vs4) in zip(ent, four):
            assert torch.equal(u, u4) and torch.equal(v, v4) and torch.equal(sg, s4) and torch.equal(us, us4) and torch.equal(vs, vs4)


# eg_epilogue.nt_variant values (include/eadgan_hip.h EG_NT_*) -> label eg_igemm_nt_tile reports for the 128-column cases below
NT_REG, NT_BUF128, NT_PERS, NT_S8, NT_S8P = 1, 2, 3, 4, 5
NT_PATCHED = (NT_S8P,)
NT_VARIANTS = [(NT_BUF128, 128131), (NT_PERS, 128135), (NT_S8, 256147), (NT_S8P, 256149), (NT_REG, 128128)]


def nt_tile(c, dtype, bwd, variant, splitk):
    import ctypes
    return eg._lib.lib().query("eg_igemm_nt_tile", ctypes.byref(c), dtype, bwd, variant, splitk)


def same(got, ref, variant, dtype, K):
    """Variants that accumulate K in the reference order must match bit for bit; the input-patch variant accumulates class by class
    (fp32 sums in another order, then one rounding to the output type): a few output ulps."""
    if variant not in NT_PATCHED:
        return torch.equal(got, ref)
    g, r = got.float(), ref.float()
    scale = r.abs().max().item() + 1e-6
    tol_ = (1e-5 if dtype == 0 else 1.6e-2) * scale
    return (g - r).abs().max().item() <= tol_


@pytest.mark.parametrize("dtype", DTYPES)
def test_igemm_variants_match_register_staged_kernel(dtype):
    """Every LDS-DMA variant of the NT kernel, forced per call through eg_epilogue.nt_variant (the library keeps no tuning state):
    128x128 buffer-descriptor kernel, persistent pipeline, and the 8-wave 256x128 kernel of igemm_nt8s.hip in both modes -- bit-exact vs
    the register-staged kernel where the K order is the same (a few output ulps for the input-patch mode) and within tolerance of
    torch, for a padded stride-2 forward conv, a 4-phase backward-data with per-tape 1/sigma and an activation mask, and a
    256-column forward."""
    lib = eg._lib.lib()
    g = torch.Generator().manual_seed(21)
    # forward: B=64, 64x64x64 -> 32x32x128  (M = 65536: 512 tiles of 128 rows, 256 of 256)
    B, H, Cin, Cout = 64, 64, 64, 128
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    xd = nhwc(x, dtype)
    outs = []
    for variant, label in NT_VARIANTS:
        assert nt_tile(c, dtype, 0, variant, 1) == label
        y = torch.zeros(B, 32, 32, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_fwd(c, dtype, xd, wp, y, ops.epilogue(bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2, nt_variant=variant, nt_splitk=1))
        torch.cuda.synchronize()
        outs.append(y)
    assert all(same(o, outs[-1], v, dtype, Cin * 16) for o, (v, _) in zip(outs[:-1], NT_VARIANTS))
    want = F.leaky_relu(F.conv2d(x, rq(w, dtype), b, 2, 1), 0.2)
    rt, at = tol(dtype, Cin * 16)
    for o in (outs[0], outs[2], outs[3]):
        torch.testing.assert_close(nchw(o), want, rtol=rt, atol=at)
    # the planner on its own: 256 tiles of 256 x 128 -> the 8-wave kernel (16-bit: wherever it can run from 48 tiles up; fp32: a full round
    # with a K loop of at least 32 K tiles, here 32 elements per K tile)
    assert nt_tile(c, dtype, 0, 0, 0) == 256147
    # backward-data: dY [16,32,32,64] -> dX [16,64,64,128], 4 phases of M = 16384, two tapes
    B, H, Cin, Cout = 16, 64, 128, 64
    dy = rq(torch.randn(B, Cout, 32, 32, generator=g), dtype)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.05
    a = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
    wp = torch.empty(ops.pack_bwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_bwd(c, dtype, w.to(DEV), wp)
    sig = torch.tensor([1.3, 0.7], device=DEV)
    outs = []
    for variant, _ in NT_VARIANTS:
        dx = torch.zeros(B, H, H, Cin, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_bwd_data(c, dtype, nhwc(dy, dtype), wp, dx,
                          ops.epilogue(sigma=sig, sigma_rows=8 * 32 * 32, mask=nhwc(a, dtype), mask_act=ops.ACT_LRELU, mask_slope=0.1,
                                       nt_variant=variant, nt_splitk=1))
        torch.cuda.synchronize()
        outs.append(dx)
    assert all(same(o, outs[-1], v, dtype, Cout * 4) for o, (v, _) in zip(outs[:-1], NT_VARIANTS))
    want = F.conv_transpose2d(dy, rq(w, dtype), None, 2, 1) * torch.where(a > 0, 1.0, 0.1)
    want[:8] /= 1.3
    want[8:] /= 0.7
    rt, at = tol(dtype, Cout * 4)
    for o in (outs[0], outs[2], outs[3]):
        torch.testing.assert_close(nchw(o), want, rtol=rt, atol=at)
    # 256 output channels: two N tiles per M tile (adjacent workgroups of one XCD share the gathered rows)
    B, H, Cin, Cout = 64, 32, 64, 256
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    outs = []
    v256 = ((NT_S8, 256147), (NT_S8P, 256149), (NT_BUF128, 128131), (NT_REG, 128128))
    for variant, label in v256:
        assert nt_tile(c, dtype, 0, variant, 1) == label
        y = torch.zeros(B, 16, 16, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, y, ops.epilogue(bias=b.to(DEV), act=ops.ACT_RELU, nt_variant=variant, nt_splitk=1))
        torch.cuda.synchronize()
        outs.append(y)
    assert all(same(o, outs[-1], v, dtype, Cin * 16) for o, (v, _) in zip(outs[:-1], v256))
    rt, at = tol(dtype, Cin * 16)
    for o in (outs[0], outs[1]):
        torch.testing.assert_close(nchw(o), F.relu(F.conv2d(x, rq(w, dtype), b, 2, 1)), rtol=rt, atol=at)
    # a variant that cannot run the problem is an error, not a silent substitution (64 output channels: no 128-column tile)
    with pytest.raises(RuntimeError):
        c = ops.make_conv(4, 8, 8, 64, 64, 3, 1, 1)
        ops.conv_fwd(c, dtype, torch.zeros(4, 8, 8, 64, device=DEV, dtype=ops.torch_dtype(dtype)),
                     torch.zeros(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype)),
                     torch.zeros(4, 8, 8, 64, device=DEV, dtype=ops.torch_dtype(dtype)), ops.epilogue(nt_variant=NT_S8))


@pytest.mark.parametrize("variant,Cout", [(NT_BUF128, 128), (NT_PERS, 128), (NT_PERS, 256), (NT_S8, 128), (NT_S8, 256), (NT_S8P, 128), (NT_S8P, 256)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_igemm_dma_variants_ragged_rows_and_k_padding(dtype, variant, Cout):
    """Buffer-descriptor LDS-DMA NT kernels (128x128, persistent, 256x128 in both modes) on a launch whose last row tile is almost empty
    (M = 1025*64) and whose 3x3 filter walks 9 taps with image borders on every side (9 K tiles: odd, so the rings end mid-cycle):
    bit-exact vs the register-staged kernel."""
    g = torch.Generator().manual_seed(22)
    B, H, Cin = 1025, 8, 64
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    c = ops.make_conv(B, H, H, Cin, Cout, 3, 1, 1)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    xd = nhwc(x, dtype)
    outs = []
    for v in (variant, NT_REG):
        y = torch.zeros(B, H, H, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_fwd(c, dtype, xd, wp, y, ops.epilogue(nt_variant=v, nt_splitk=1))
        torch.cuda.synchronize()
        outs.append(y)
    assert same(outs[0], outs[1], variant, dtype, Cin * 9)
    want = F.conv2d(x[:64], rq(w, dtype), None, 1, 1)
    rt, at = tol(dtype, Cin * 9)
    torch.testing.assert_close(nchw(outs[0][:64]), want, rtol=rt, atol=at)
    torch.testing.assert_close(nchw(outs[0][-64:]), F.conv2d(x[-64:], rq(w, dtype), None, 1, 1), rtol=rt, atol=at)


@pytest.mark.parametrize("variant,nk_taps", [(NT_S8, 1), (NT_S8, 2), (NT_S8, 3), (NT_S8, 4), (NT_S8, 5), (NT_S8, 6), (NT_S8, 7), (NT_S8P, 1), (NT_S8P, 2),
                                             (NT_S8P, 3), (NT_S8P, 4), (NT_S8P, 5)])
def test_igemm_nt8s_short_k_loops(variant, nk_taps):
    """K loops of one to seven K tiles (1x1 conv over 64 .. 448 channels in bf16): the ring's prologue, main-loop and tail counts."""
    dtype = 1
    g = torch.Generator().manual_seed(26)
    B, H, Cin, Cout = 16, 8, 64 * nk_taps, 256
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
    c = ops.make_conv(B, H, H, Cin, Cout, 1, 1, 0)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    outs = []
    for v in (variant, NT_REG):
        y = torch.zeros(B, H, H, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, y, ops.epilogue(nt_variant=v, nt_splitk=1))
        torch.cuda.synchronize()
        outs.append(y)
    assert same(outs[0], outs[1], variant, dtype, Cin)


@pytest.mark.parametrize("dtype", DTYPES)
def test_igemm_splitk_small_m_deep_k(dtype):
    """Few-row / deep-K launches (the last Discriminator conv: celebA/EAD-GAN_celebA.py:118) split K across workgroups into fp32
    partial tiles summed in split order before the fused epilogue (by a second launch in the 128-row kernel, by the last-arriving workgroup
    inside the launch in the 8-wave kernel -- its arrival counters at the scratch's tail are zero again afterwards): forward (bias + LeakyReLU) and 4-phase
    backward-data (1/sigma per tape + activation-gradient mask) against torch, and repeatable bit for bit -- planner's choice (128-row
    tiles at this M), and the 8-wave kernel forced with 4 to 16 splits in both modes."""
    lib = eg._lib.lib()
    ws = torch.zeros(16 << 20, device=DEV, dtype=torch.float32)
    g = torch.Generator().manual_seed(23)
    B, H, Cin, Cout = 24, 8, 256, 256
    c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
    assert ops.conv_splitk_ws_bytes(c, dtype, 0) > 0 and ops.conv_splitk_ws_bytes(c, dtype, 1) > 0
    assert nt_tile(c, dtype, 0, 0, 0) == 128132
    assert nt_tile(c, dtype, 0, NT_S8, 4) == 256148 and nt_tile(c, dtype, 1, NT_S8P, 4) == 256150
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.03
    b = torch.randn(Cout, generator=g)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    dy = rq(torch.randn(B, Cout, 4, 4, generator=g), dtype)
    a = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    wpb = torch.empty(ops.pack_bwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_bwd(c, dtype, w.to(DEV), wpb)
    sig = torch.tensor([1.3, 0.7], device=DEV)
    for variant, splitk in ((0, 0), (NT_S8, 4), (NT_S8, 8), (NT_S8P, 4), (NT_S8P, 8), (NT_S8P, 16)):
        ys = []
        for _ in range(2):
            y = torch.zeros(B, 4, 4, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
            ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, y,
                         ops.epilogue(bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2, nt_variant=variant, nt_splitk=splitk, splitk_ws=ws))
            torch.cuda.synchronize()
            ys.append(y)
        assert torch.equal(ys[0], ys[1])
        want = F.leaky_relu(F.conv2d(x, rq(w, dtype), b, 2, 1), 0.2)
        rt, at = tol(dtype, Cin * 16)
        torch.testing.assert_close(nchw(ys[0]), want, rtol=rt, atol=at)
        # backward-data of the same layer, two tapes of 12 images
        dx = torch.zeros(B, H, H, Cin, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_bwd_data(c, dtype, nhwc(dy, dtype), wpb, dx,
                          ops.epilogue(sigma=sig, sigma_rows=12 * 16, mask=nhwc(a, dtype), mask_act=ops.ACT_LRELU, mask_slope=0.1,
                                       nt_variant=variant, nt_splitk=splitk, splitk_ws=ws))
        torch.cuda.synchronize()
        want = F.conv_transpose2d(dy, rq(w, dtype), None, 2, 1) * torch.where(a > 0, 1.0, 0.1)
        want[:12] /= 1.3
        want[12:] /= 0.7
        rt, at = tol(dtype, Cout * 4)
        torch.testing.assert_close(nchw(dx), want, rtol=rt, atol=at)
        assert int(ws[-1024:].view(torch.int32).abs().sum()) == 0          # the last 4 KiB: arrival counters, left at zero by every launch


def test_igemm_splitk_inkernel_reduction_repeatable_under_load():
    """The 8-wave kernel's in-launch K-split reduction hands partial tiles from workgroup to workgroup (sc1 stores / loads + an arrival
    counter): 60 launches of a production shape (CelebA 256 -> 512 at B = 128, 4 splits) with other kernels interleaved must all give
    the first launch's bytes, and leave the counters at zero.  (profiles/scripts/splitk_race_screen.py is the long version.)"""
    dtype = 1
    ws = torch.zeros(24 << 20, device=DEV, dtype=torch.float32)
    g = torch.Generator(device=DEV).manual_seed(5)
    B, H, Cin, Cout = 128, 16, 256, 512
    c = ops.make_conv(B, H, H, Cin, Cout, 4, 2, 1)
    assert nt_tile(c, dtype, 0, NT_S8, 4) == 256148
    w = (torch.rand(Cout, Cin, 4, 4, device=DEV, generator=g) - 0.5) * 0.1
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w, wp)
    x = (torch.rand(B, H, H, Cin, device=DEV, generator=g) * 2 - 1).to(ops.torch_dtype(dtype))
    run = lambda out: ops.conv_fwd(c, dtype, x, wp, out, ops.epilogue(act=ops.ACT_LRELU, slope=0.1, nt_variant=NT_S8, nt_splitk=4, splitk_ws=ws))
    ref = torch.empty(B, H // 2, H // 2, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
    run(ref)
    filler = torch.randn(1024, 1024, device=DEV)
    bad = torch.zeros(1, device=DEV, dtype=torch.int64)
    for r in range(60):
        out = torch.full_like(ref, 3.0)
        if r % 3 == 1:
            filler @ filler
        run(out)
        bad += (out != ref).any().to(torch.int64)
    torch.cuda.synchronize()
    assert int(bad) == 0 and int(ws[-1024:].view(torch.int32).abs().sum()) == 0
    # and the sum itself: against the unsplit launch to 16-bit rounding of the output
    plain = torch.empty_like(ref)
    ops.conv_fwd(c, dtype, x, wp, plain, ops.epilogue(act=ops.ACT_LRELU, slope=0.1, nt_variant=NT_S8, nt_splitk=1, splitk_ws=None))
    torch.cuda.synchronize()
    torch.testing.assert_close(ref.float(), plain.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("variant", [0, NT_BUF128, NT_S8, NT_S8P])
@pytest.mark.parametrize("dtype", DTYPES)
def test_igemm_splitk_range_starting_inside_a_tap(dtype, variant):
    """3x3 conv whose K split boundaries fall inside filter taps (9 taps x 256 channels, 4 splits of 9 (bf16) / 18 (fp32) K steps
    against 4 / 8 steps per tap): the per-split start state (tap, channel offset) must be reconstructed exactly."""
    ws = torch.zeros(16 << 20, device=DEV, dtype=torch.float32)
    g = torch.Generator().manual_seed(24)
    B, H, Cin, Cout = 8, 8, 256, 256
    c = ops.make_conv(B, H, H, Cin, Cout, 3, 1, 1)
    assert ops.conv_splitk_ws_bytes(c, dtype, 0) > 0
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    y = torch.zeros(B, H, H, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
    ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, y, ops.epilogue(nt_variant=variant, nt_splitk=4 if variant else 0, splitk_ws=ws))
    torch.cuda.synchronize()
    rt, at = tol(dtype, Cin * 9)
    torch.testing.assert_close(nchw(y), F.conv2d(x, rq(w, dtype), None, 1, 1), rtol=rt, atol=at)


@pytest.mark.parametrize("dtype", DTYPES)
def test_igemm_shallow_single_tap_runs_on_persistent_pipeline(dtype):
    """The image-side layers run as 1x1 convolutions over 48-channel patch rows (K = 48: one K step in bf16, two in fp32, the row
    is not a whole number of K tiles).  Such launches go to the persistent pipeline by the planner's own choice; bit-exact vs the
    register-staged kernel, several tiles per workgroup, ragged last tile."""
    lib = eg._lib.lib()
    g = torch.Generator().manual_seed(25)
    B, H, Cin, Cout = 131, 32, 48, 128          # M = 134144 -> 1048 tiles over 512 workgroups
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    sig = torch.tensor([0.9], device=DEV)
    c = ops.make_conv(B, H, H, Cin, Cout, 1, 1, 0)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=ops.torch_dtype(dtype))
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    outs = []
    assert nt_tile(c, dtype, 0, 0, 0) == 128135
    for v in (0, NT_REG):
        y = torch.zeros(B, H, H, Cout, device=DEV, dtype=ops.torch_dtype(dtype))
        ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, y, ops.epilogue(bias=b.to(DEV), sigma=sig, act=ops.ACT_LRELU, slope=0.1, nt_variant=v))
        torch.cuda.synchronize()
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    rt, at = tol(dtype, Cin)
    torch.testing.assert_close(nchw(outs[0][:16]), F.leaky_relu(F.conv2d(x[:16], rq(w, dtype) / 0.9, b), 0.1), rtol=rt, atol=at)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,stride,pad,Cin,Cout", [(4, 2, 1, 32, 64), (4, 2, 1, 128, 48), (3, 1, 1, 64, 32), (4, 1, 0, 64, 16), (3, 1, 1, 32, 16)])
def test_pack_conv_tiled_equals_per_element_pack(dtype, k, stride, pad, Cin, Cout):
    """eg_pack_conv (LDS-tiled, both panels in one pass; falls back where K needs padding) writes exactly the bytes of
    eg_pack_fwd + eg_pack_bwd."""
    g = torch.Generator().manual_seed(31)
    w = torch.randn(Cout, Cin, k, k, generator=g).to(DEV)
    c = ops.make_conv(2, 16, 16, Cin, Cout, k, stride, pad)
    tdt = ops.torch_dtype(dtype)
    f0 = torch.full((ops.pack_fwd_elems(c, dtype),), 7.0, device=DEV, dtype=tdt)
    b0 = torch.full((ops.pack_bwd_elems(c, dtype),), 7.0, device=DEV, dtype=tdt)
    f1, b1 = f0.clone(), b0.clone()
    ops.pack_fwd(c, dtype, w, f0)
    ops.pack_bwd(c, dtype, w, b0)
    ops.pack_conv(c, dtype, w, f1, b1)
    torch.cuda.synchronize()
    assert torch.equal(f0, f1) and torch.equal(b0, b1)
    f2 = torch.full_like(f0, 7.0)
    ops.pack_conv(c, dtype, w, f2, None)
    torch.cuda.synchronize()
    assert torch.equal(f0, f2)


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_packs_in_one_launch_give_the_same_panels(dtype):
    """ops.PackBatch (eg_pack_record_begin / _end / eg_pack_multi): the weight packs of several layers as ONE launch -- tile kernel (4x4
    stride-2 layer, both panels), per-element gather (ragged channel count), the two strided packs, an fp32 job beside 16-bit ones --
    against the same packs launched one by one: identical panel bytes; the second run reuses the uploaded job table."""
    tdt = ops.torch_dtype(dtype)
    g = torch.Generator().manual_seed(3)
    layers = [ops.make_conv(8, 16, 16, 64, 32, 4, 2, 1), ops.make_conv(8, 8, 8, 24, 40, 3, 1, 1), ops.make_conv(8, 8, 8, 32, 64, 4, 2, 1)]
    masters = [torch.randn(c.Cout, c.Cin, c.k, c.k, generator=g).to(DEV) for c in layers]
    w2 = torch.randn(1024, 128, generator=g).to(DEV)
    bias = torch.randn(1024, generator=g).to(DEV)

    def buffers():
        bufs = []
        for c in layers:
            bufs.append((torch.full((ops.pack_fwd_elems(c, dtype),), 7, device=DEV, dtype=tdt),
                         torch.full((ops.pack_bwd_elems(c, dtype),), 7, device=DEV, dtype=tdt) if c.stride <= 2 and c.k == 4 else None))
        kp = ops.round_up(1024, ops.bk(dtype))
        return bufs, torch.full((1024 * 128,), 7, device=DEV, dtype=tdt), torch.full((128 * kp,), 7, device=DEV, dtype=tdt), torch.full((1024,), 7.0, device=DEV)

    def packs(bufs, f2, b2, bperm):
        for c, w, (pf, pb) in zip(layers, masters, bufs):
            ops.pack_conv(c, dtype, w, pf, pb)
        ops.pack_strided(dtype, w2, f2, 1024, 128, 128, 64, 128, 16 * 128, 1)
        ops.pack_strided2(dtype, w2, b2, 128, 1024, ops.round_up(1024, ops.bk(dtype)), 1, 1, 0, 64, 128, 16 * 128)
        ops.pack_strided(0, bias, bperm, 1024, 1, 1, 64, 1, 16, 0)

    ref = buffers()
    packs(*ref)
    got = buffers()
    batch = ops.PackBatch()
    batch.run(lambda: packs(*got))
    torch.cuda.synchronize()
    assert batch.n >= 6 and batch.dev is not None

    def flat(b):
        out = []
        for pf, pb in b[0]:
            out += [pf] + ([pb] if pb is not None else [])
        return out + list(b[1:])
    for a, b in zip(flat(ref), flat(got)):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
    table = batch.dev
    for t in flat(got):
        t.fill_(3)
    batch.run(lambda: packs(*got))
    torch.cuda.synchronize()
    assert batch.dev is table                                # same pointers, same geometry: no new upload
    for a, b in zip(flat(ref), flat(got)):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(128, 8, 64, 64, 4, 2, 1), (96, 8, 64, 32, 4, 2, 1), (40, 8, 32, 64, 4, 2, 1)])
def test_few_row_deep_k_launch_splits_k_inside_the_register_staged_kernel(case, dtype):
    """The small networks' last trunk layer (64 -> 64, 8x8 -> 4x4: 16-48 tiles of 128 rows, 16 K steps) with split-K scratch lent and the
    per-call hint nt_splitk = 8 (an experiment, default off: DESIGN.md 6.0): the K loop is cut into slices reduced inside the launch by the
    last workgroup to arrive -- against torch and against the unsplit launch
    (no scratch), forward and backward-data (the four sub-pixel phases, with fused column statistics on 16-bit types)."""
    B, H, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(5)
    x = rq(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    c = ops.make_conv(B, H, H, Cin, Cout, k, s, p)
    tdt = ops.torch_dtype(dtype)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=tdt)
    ops.pack_fwd(c, dtype, w.to(DEV), wp)
    want = F.leaky_relu(F.conv2d(x, rq(w, dtype), b, s, p), 0.2)
    OH = want.shape[-1]
    ws = torch.zeros(eg.engine.SPLITK_WS_BYTES // 4, device=DEV)
    ys = {}
    for name, scratch in (("plain", None), ("split", ws)):
        ys[name] = torch.empty(B, OH, OH, Cout, device=DEV, dtype=tdt)
        ops.conv_fwd(c, dtype, nhwc(x, dtype), wp, ys[name], ops.epilogue(bias=b.to(DEV), act=ops.ACT_LRELU, slope=0.2, splitk_ws=scratch, nt_splitk=8 if scratch is not None else 0))
    torch.cuda.synchronize()
    assert float(ws[-1024:].abs().sum()) == 0.0                               # the arrival counters are left at zero
    rt, at = tol(dtype, Cin * k * k)
    torch.testing.assert_close(nchw(ys["split"]), want, rtol=rt, atol=at)
    torch.testing.assert_close(ys["split"].float(), ys["plain"].float(), rtol=rt, atol=at)
    assert (ys["split"] != ys["plain"]).float().mean() < (1e-6 if dtype == 0 else 2e-2) or dtype == 0      # another summation order: rounding-level differences only
    # backward-data of the same layer (conv view of a ConvTranspose: 4 phases x 2 x 2 taps) with BatchNorm moments from the epilogue
    cb = ops.make_conv(B, 2 * OH, 2 * OH, Cout, Cin, 4, 2, 1)
    dy = rq(torch.randn(B, Cin, OH, OH, generator=g), dtype)
    wb = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.05
    wpb = torch.empty(ops.pack_bwd_elems(cb, dtype), device=DEV, dtype=tdt)
    ops.pack_bwd(cb, dtype, wb.to(DEV), wpb)
    want_b = F.conv_transpose2d(dy, rq(wb, dtype), None, 2, 1)
    zs = {}
    for name, scratch in (("plain", None), ("split", ws)):
        zs[name] = torch.empty(B, 2 * OH, 2 * OH, Cout, device=DEV, dtype=tdt)
        hint = 8 if scratch is not None else 0
        ep = ops.epilogue(splitk_ws=scratch, nt_splitk=hint)
        nrb = ops.conv_stat_blocks(cb, dtype, True, ep) if dtype != 0 else 0
        if nrb:
            stat = torch.full((2 * Cout * nrb,), float("nan"), device=DEV)
            ep = ops.epilogue(splitk_ws=scratch, nt_splitk=hint, stat_mode=ops.STAT_MOMENTS, stat_out=stat)
        ops.conv_bwd_data(cb, dtype, nhwc(dy, dtype), wpb, zs[name], ep)
        torch.cuda.synchronize()
        if nrb:
            M = B * 4 * OH * OH
            mean = stat[:Cout * nrb].reshape(Cout, nrb).mean(1)
            torch.testing.assert_close(mean, zs[name].float().reshape(M, Cout).mean(0), rtol=1e-3, atol=1e-4)
    rt, at = tol(dtype, Cin * 4)
    torch.testing.assert_close(nchw(zs["split"]), want_b, rtol=rt, atol=at)
    torch.testing.assert_close(zs["split"].float(), zs["plain"].float(), rtol=rt, atol=at)
