"""Column statistics fused into the convolution epilogue (eg_epilogue.stat_mode, include/eadgan_hip.h) against the stand-alone reduction
kernels they replace and against torch: BatchNorm batch statistics (celebA/EAD-GAN_celebA.py:79,83,87), the two sums of the BatchNorm
backward, and the bias gradient + spectral-norm coefficient of a spectrally normalised layer (:110-122).  The convolution outputs
themselves must be bit-identical with and without the statistics (EG_STAT_BN_BWD stores dy = da * relu'(bn(z)) instead of da)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

eg = None
ops = None
DEV = "cuda"


def setup_module(module):
    global eg, ops
    eg = importlib.import_module("ead-gan_amd")
    ops = eg.ops


def _rand(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV).to(ops.torch_dtype(dtype))


def _splitk_ws():
    ws = torch.zeros(eg.engine.SPLITK_WS_BYTES // 4, device=DEV)
    return ws


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("B,splitk", [(64, 0), (64, 1), (128, 0), (512, 0)])      # (512: >= 1024 row blocks -> a workgroup per channel finishes the sums)
@pytest.mark.parametrize("Ci,Co", [(256, 128), (64, 64), (64, 32)])
def test_batchnorm_forward_statistics_from_the_transposed_convolution(B, splitk, dtype, Ci, Co):
    """ConvTranspose2d(Ci -> Co, 4, 2, 1) + BatchNorm2d + ReLU: statistics from the 4-phase backward-data launch (with and without K
    splits): 256 -> 128 on the 8-wave kernels, 64 -> 64 / 32 (the dSprites generators) on the register-staged kernel"""
    H = 8                                                  # ConvT: [B,8,8,Ci] -> [B,16,16,Co]
    c = ops.make_conv(B, 2 * H, 2 * H, Co, Ci, 4, 2, 1)    # conv view
    tdt = ops.torch_dtype(dtype)
    x = _rand((B, H, H, Ci), dtype, 1)
    w = _rand((Ci, Co, 4, 4), 0, 2, 0.05)                  # conv view master [Cout_cv = Ci][Cin_cv = Co][4][4]
    bias = _rand((Co,), 0, 3) + 0.5
    wp = torch.empty(ops.pack_bwd_elems(c, dtype), device=DEV, dtype=tdt)
    ops.pack_bwd(c, dtype, w, wp)
    ws = _splitk_ws()
    M = B * 4 * H * H
    z0 = torch.empty(B, 2 * H, 2 * H, Co, device=DEV, dtype=tdt)
    z1 = torch.empty_like(z0)
    ep0 = ops.epilogue(bias=bias, splitk_ws=ws, nt_splitk=splitk)
    nrb = ops.conv_stat_blocks(c, dtype, True, ep0)
    assert nrb in (4 * (B * H * H) // 256, 4 * (B * H * H) // 128), nrb       # row blocks of 256 (igemm_nt8s) or 128 (igemm_nt8h) lattice rows
    stat = torch.full((2 * Co * nrb,), float("nan"), device=DEV)
    ops.conv_bwd_data(c, dtype, x, wp, z0, ep0)
    ops.conv_bwd_data(c, dtype, x, wp, z1, ops.epilogue(bias=bias, splitk_ws=ws, nt_splitk=splitk, stat_mode=ops.STAT_MOMENTS, stat_out=stat))
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    assert torch.isfinite(stat).all()
    gamma, beta = _rand((Co,), 0, 4) * 0.1 + 1.0, _rand((Co,), 0, 5) * 0.1
    out, rm, rv, nbt, mean, istd = {}, {}, {}, {}, {}, {}
    small = torch.empty(ops.bn_ws_floats(M, Co), device=DEV)
    for k in ("plain", "fused"):
        out[k] = torch.empty_like(z0)
        rm[k], rv[k] = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        nbt[k] = torch.zeros(1, device=DEV, dtype=torch.int64)
        mean[k], istd[k] = torch.empty(Co, device=DEV), torch.empty(Co, device=DEV)
    ops.bn_fwd_train(dtype, z0, out["plain"], M, Co, gamma, beta, 1e-5, 0.1, rm["plain"], rv["plain"], nbt["plain"], mean["plain"], istd["plain"], small, ops.ACT_RELU)
    ops.bn_fwd_train_fused(dtype, z0, out["fused"], M, Co, stat, nrb, M // nrb, gamma, beta, 1e-5, 0.1, rm["fused"], rv["fused"], nbt["fused"], mean["fused"],
                           istd["fused"], small, ops.ACT_RELU)
    torch.cuda.synchronize()
    # both are fp64 combinations of exact fp32 block moments of the same stored values: they agree far below the 16-bit output step
    torch.testing.assert_close(mean["fused"], mean["plain"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(istd["fused"], istd["plain"], rtol=1e-5, atol=0)
    torch.testing.assert_close(rm["fused"], rm["plain"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(rv["fused"], rv["plain"], rtol=1e-5, atol=0)
    assert int(nbt["fused"]) == 1
    assert (out["fused"] != out["plain"]).float().mean() < 1e-3           # a 16-bit output flips only where it sits on a rounding boundary
    # and against torch on the stored tensor
    zf = z0.float().reshape(M, Co)
    torch.testing.assert_close(mean["fused"], zf.mean(0), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(istd["fused"], (zf.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=0)


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("Ci,Co", [(128, 256), (64, 64), (32, 32)])
def test_batchnorm_backward_sums_from_the_producing_convolution(dtype, Ci, Co):
    """d(activation) of a BatchNorm + ReLU layer produced by a forward convolution launch: dy = da * relu'(bn(z)) stored, the two sums fused"""
    B, H = 128, 16                                         # conv: [B,16,16,Ci] -> [B,8,8,Co]
    c = ops.make_conv(B, H, H, Ci, Co, 4, 2, 1)
    tdt = ops.torch_dtype(dtype)
    x = _rand((B, H, H, Ci), dtype, 11)
    w = _rand((Co, Ci, 4, 4), 0, 12, 0.05)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=tdt)
    ops.pack_fwd(c, dtype, w, wp)
    ws = _splitk_ws()
    OH = H // 2
    M = B * OH * OH
    z = _rand((B, OH, OH, Co), dtype, 13)                  # the BatchNorm input of the layer whose activation gradient is formed
    gamma, beta = _rand((Co,), 0, 14) * 0.1 + 1.0, _rand((Co,), 0, 15) * 0.3
    zf = z.float().reshape(M, Co)
    mean = zf.mean(0).contiguous()
    istd = (zf.var(0, unbiased=False) + 1e-5).rsqrt().contiguous()
    da = torch.empty(B, OH, OH, Co, device=DEV, dtype=tdt)
    dy = torch.empty_like(da)
    ep0 = ops.epilogue(splitk_ws=ws)
    nrb = ops.conv_stat_blocks(c, dtype, False, ep0)
    assert nrb in (M // 256, M // 128)
    stat = torch.full((2 * Co * nrb,), float("nan"), device=DEV)
    ops.conv_fwd(c, dtype, x, wp, da, ep0)
    ops.conv_fwd(c, dtype, x, wp, dy, ops.epilogue(splitk_ws=ws, stat_mode=ops.STAT_BN_BWD, stat_out=stat, stat_aux=z, stat_p=(mean, istd, gamma, beta),
                                                   stat_act=ops.ACT_RELU))
    torch.cuda.synchronize()
    pre = zf * (gamma * istd) + (beta - mean * gamma * istd)
    on = (pre > 0).reshape(da.shape)
    near = (pre.abs() < 1e-5).reshape(da.shape)             # units within rounding of the ReLU kink may be decided either way
    want_dy = torch.where(on, da, torch.zeros_like(da))
    assert torch.equal(dy[~near], want_dy[~near])
    small = torch.empty(ops.bn_ws_floats(M, Co), device=DEV)
    sums = {k: torch.empty(2 * Co, device=DEV) for k in ("plain", "fused")}
    dz = {k: torch.empty_like(da) for k in ("plain", "fused")}
    dg = {k: torch.zeros(Co, device=DEV) for k in ("plain", "fused")}
    db = {k: torch.zeros(Co, device=DEV) for k in ("plain", "fused")}
    ops.bn_bwd(dtype, z, da, dz["plain"], M, Co, gamma, beta, mean, istd, ops.ACT_RELU, 0.0, dg["plain"], db["plain"], sums["plain"], small)
    ops.bn_bwd_fused(dtype, z, dy, dz["fused"], M, Co, stat, nrb, gamma, beta, mean, istd, dg["fused"], db["fused"], sums["fused"], small)
    torch.cuda.synchronize()
    scale = float(sums["plain"].abs().max())
    torch.testing.assert_close(sums["fused"], sums["plain"], rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(dg["fused"], dg["plain"], rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(db["fused"], db["plain"], rtol=1e-4, atol=1e-5 * scale)
    diff = (dz["fused"].float() - dz["plain"].float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(dz["plain"].float().abs().max())     # one 16-bit step of the largest element at most
    assert (diff > 0).float().mean() < 2e-2
    # torch: BatchNorm backward of the masked gradient
    dyf = want_dy.float().reshape(M, Co)
    xh = (zf - mean) * istd
    want = gamma * istd * (dyf - dyf.mean(0) - xh * (dyf * xh).mean(0))
    err = float((dz["fused"].float().reshape(M, Co) - want).norm() / want.norm())
    assert err < 8e-3, err


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("T,B", [(1, 64), (2, 64), (3, 64), (3, 256)])             # (3 x 256 images: 1536 row blocks -> a workgroup per sum)
@pytest.mark.parametrize("Ci,Co", [(128, 256), (64, 64), (32, 64)])
def test_spectral_norm_bias_gradient_and_coefficient_from_the_backward_data_launch(T, B, dtype, Ci, Co):
    """dzs of a spectrally normalised LeakyReLU layer (T tapes batched along M) from conv_bwd_data with the fused mask: per-tape column sums
    and <dzs, z - bias> from the epilogue == eg_bias_grad_sn on the stored tensor"""
    H = 16                                                 # layer below: [T*B,16,16,Ci]; this launch: dY [T*B,8,8,Co] -> dX [T*B,16,16,Ci]
    c = ops.make_conv(T * B, H, H, Ci, Co, 4, 2, 1)
    tdt = ops.torch_dtype(dtype)
    dyy = _rand((T * B, H // 2, H // 2, Co), dtype, 21)
    w = _rand((Co, Ci, 4, 4), 0, 22, 0.05)
    wp = torch.empty(ops.pack_bwd_elems(c, dtype), device=DEV, dtype=tdt)
    ops.pack_bwd(c, dtype, w, wp)
    a = _rand((T * B, H, H, Ci), dtype, 23)                # activation output of the layer below (the mask)
    bias = _rand((Ci,), 0, 24) * 0.2
    sigma = (torch.arange(T, device=DEV, dtype=torch.float32) * 0.3 + 1.3).contiguous()
    rows_src = B * (H // 2) ** 2                            # lattice rows of one tape in this launch
    ws = _splitk_ws()
    kw = dict(sigma=sigma, sigma_rows=rows_src, mask=a, mask_act=ops.ACT_LRELU, mask_slope=0.1, splitk_ws=ws)
    d0 = torch.empty(T * B, H, H, Ci, device=DEV, dtype=tdt)
    d1 = torch.empty_like(d0)
    ep0 = ops.epilogue(**kw)
    nrb = ops.conv_stat_blocks(c, dtype, True, ep0)
    assert nrb in (4 * T * rows_src // 256, 4 * T * rows_src // 128)
    tiles_m = nrb // 4
    stat = torch.full((Ci * nrb + nrb * max(Ci // 128, 1),), float("nan"), device=DEV)
    ops.conv_bwd_data(c, dtype, dyy, wp, d0, ep0)
    ops.conv_bwd_data(c, dtype, dyy, wp, d1, ops.epilogue(stat_mode=ops.STAT_SN_BIAS, stat_out=stat, stat_p=(bias,), stat_slope=0.1, **kw))
    torch.cuda.synchronize()
    assert torch.equal(d0, d1)
    assert torch.isfinite(stat).all()
    rows = T * B * H * H
    small = torch.empty(ops.bias_grad_sn_ws_floats(rows, Ci, B * H * H), device=DEV)
    gb = {k: torch.zeros(Ci, device=DEV) for k in ("plain", "fused")}
    coef = {k: torch.zeros(4, device=DEV) for k in ("plain", "fused")}
    ops.bias_grad_sn(dtype, d0, a, bias, rows, Ci, B * H * H, sigma, 0.1, small, gb["plain"], coef["plain"])
    ops.bias_grad_sn_fused(stat, nrb, Ci, tiles_m, tiles_m // T, T, sigma, gb["fused"], coef["fused"])
    torch.cuda.synchronize()
    torch.testing.assert_close(gb["fused"], gb["plain"], rtol=1e-4, atol=1e-5 * float(gb["plain"].abs().max()))
    torch.testing.assert_close(coef["fused"][:T], coef["plain"][:T], rtol=2e-4, atol=1e-5 * float(coef["plain"].abs().max()))
    # torch on the stored tensor
    df = d0.float().reshape(T, B * H * H, Ci)
    af = a.float().reshape(T, B * H * H, Ci)
    zpre = torch.where(af > 0, af, af / 0.1) - bias
    want_gb = (df.sum(1) * sigma[:, None]).sum(0)
    want_coef = (df * zpre).sum((1, 2))
    torch.testing.assert_close(gb["fused"], want_gb, rtol=1e-3, atol=1e-4 * float(want_gb.abs().max()))
    torch.testing.assert_close(coef["fused"][:T], want_coef, rtol=1e-3, atol=1e-4 * float(want_coef.abs().max()))


def test_statistics_are_refused_where_the_launch_cannot_fuse_them():
    """a small problem runs on another kernel: conv_stat_blocks answers 0 and a launch that asks for statistics anyway fails loudly"""
    c = ops.make_conv(2, 16, 16, 128, 256, 4, 2, 1)
    assert ops.conv_stat_blocks(c, 1, False, ops.epilogue(splitk_ws=None)) == 0
    x = _rand((2, 16, 16, 128), 1, 1)
    wp = torch.zeros(ops.pack_fwd_elems(c, 1), device=DEV, dtype=torch.bfloat16)
    y = torch.empty(2, 8, 8, 256, device=DEV, dtype=torch.bfloat16)
    stat = torch.zeros(4096, device=DEV)
    with pytest.raises(RuntimeError, match="cannot fuse column statistics"):
        ops.conv_fwd(c, 1, x, wp, y, ops.epilogue(splitk_ws=None, stat_mode=ops.STAT_MOMENTS, stat_out=stat))
