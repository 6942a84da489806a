"""Image-side convolution without patch rows (eg_conv_img_mfma, csrc/img_conv.hip) against the path it replaces on the step's critical chain,
eg_im2col_img + eg_conv_fwd over the patch rows: the first Discriminator layer (celebA/EAD-GAN_celebA.py:110, spectral norm per tape, bias,
LeakyReLU(0.1)) and the input gradient of the Generator's ConvTranspose2d(128 -> 3) + Tanh (:90-91).  Same MFMA operands in the same slots,
same epilogue arithmetic: bit-identical."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
eg = None
ops = None


def setup_module(module):
    global eg, ops
    eg = importlib.import_module("ead-gan_amd")
    ops = eg.ops


def _patch_path(dtype, imgs, wp, B, C, S, ep_kw):
    T = len(imgs)
    tdt = ops.torch_dtype(dtype)
    npix = B * (S // 2) ** 2
    patches = torch.empty(T * npix, 64, device=DEV, dtype=tdt)
    for t, im in enumerate(imgs):
        ops.im2col_img(dtype, im, patches[t * npix:(t + 1) * npix], B, C, S, S, 4, 2, 1, 64)
    c = ops.make_conv(T * B, S // 2, S // 2, 64, 128, 1, 1, 0)
    out = torch.empty(T * B, S // 2, S // 2, 128, device=DEV, dtype=tdt)
    ops.conv_fwd(c, dtype, patches, wp, out, ops.epilogue(**ep_kw))
    return out


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("T,B,C", [(1, 8, 3), (3, 5, 3), (2, 4, 1), (1, 128, 3)])
def test_first_discriminator_layer_without_patch_rows(T, B, C, dtype):
    S = 64
    g = torch.Generator().manual_seed(5)
    imgs = [(torch.rand(B, C, S, S, generator=g) * 2 - 1).to(DEV) for _ in range(T)]
    w = (torch.randn(128, C, 4, 4, generator=g) * 0.1).to(DEV)
    bias = torch.randn(128, generator=g).to(DEV) * 0.1
    sigma = (torch.arange(T, dtype=torch.float32) * 0.4 + 1.1).to(DEV)
    tdt = ops.torch_dtype(dtype)
    wp = torch.empty(128 * 64, device=DEV, dtype=tdt)
    ops.pack_strided(dtype, w, wp, 128, C * 16, 64, 1, C * 16, 0, 1)
    kw = dict(bias=bias, sigma=sigma, sigma_rows=B * (S // 2) ** 2, act=ops.ACT_LRELU, slope=0.1)
    want = _patch_path(dtype, imgs, wp, B, C, S, kw)
    assert ops.conv_img_mfma_ok(dtype, C, S, S, 128, 4, 2, 1)
    got = torch.empty_like(want)
    ops.conv_img_mfma(dtype, imgs, wp, got, B, C, S, S, ops.epilogue(**kw))
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # and against torch on the rounded operands
    rq = lambda x: x.to(tdt).float()
    for t in range(T):
        ref = F.leaky_relu(F.conv2d(rq(imgs[t]), rq(w), None, 2, 1) / sigma[t] + bias[None, :, None, None], 0.1).permute(0, 2, 3, 1)
        torch.testing.assert_close(got[t * B:(t + 1) * B].float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("T,B,C,N", [(1, 8, 1, 32), (3, 4, 3, 32), (2, 16, 3, 64), (1, 128, 1, 32), (2, 6, 4, 64)])
def test_first_trunk_layer_of_the_dsprites_networks_without_patch_rows(T, B, C, N, dtype):
    """Conv2d(C -> 32, 4, 2, 1) + LeakyReLU of the dSprites / colored dSprites trunks (dSprites/rp.py:95-97; 64 channels: the same kernel's other
    instantiation) straight from the fp32 images: bit-identical to eg_im2col_img + the K = 64 GEMM over patch rows"""
    S = 64
    g = torch.Generator().manual_seed(7)
    imgs = [(torch.rand(B, C, S, S, generator=g) * 2 - 1).to(DEV) for _ in range(T)]
    w = (torch.randn(N, C, 4, 4, generator=g) * 0.1).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV) * 0.1
    sigma = (torch.arange(T, dtype=torch.float32) * 0.3 + 0.9).to(DEV)
    tdt = ops.torch_dtype(dtype)
    wp = torch.empty(N * 64, device=DEV, dtype=tdt)
    ops.pack_strided(dtype, w, wp, N, C * 16, 64, 1, C * 16, 0, 1)
    kw = dict(bias=bias, sigma=sigma, sigma_rows=B * (S // 2) ** 2, act=ops.ACT_LRELU, slope=0.2)
    npix = B * (S // 2) ** 2
    patches = torch.empty(T * npix, 64, device=DEV, dtype=tdt)
    for t, im in enumerate(imgs):
        ops.im2col_img(dtype, im, patches[t * npix:(t + 1) * npix], B, C, S, S, 4, 2, 1, 64)
    want = torch.empty(T * B, S // 2, S // 2, N, device=DEV, dtype=tdt)
    ops.conv_fwd(ops.make_conv(T * B, S // 2, S // 2, 64, N, 1, 1, 0), dtype, patches, wp, want, ops.epilogue(**kw))
    assert ops.conv_img_mfma_ok(dtype, C, S, S, N, 4, 2, 1)
    got = torch.full_like(want, 7.0)
    ops.conv_img_mfma(dtype, imgs, wp, got, B, C, S, S, ops.epilogue(**kw), N=N)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    rq = lambda x: x.to(tdt).float()
    for t in range(T):
        ref = F.leaky_relu(F.conv2d(rq(imgs[t]), rq(w), None, 2, 1) / sigma[t] + bias[None, :, None, None], 0.2).permute(0, 2, 3, 1)
        torch.testing.assert_close(got[t * B:(t + 1) * B].float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [1, 2])
def test_generator_last_layer_input_gradient_without_patch_rows(dtype):
    """d(loss)/d(a) of ConvTranspose2d(128 -> 3, 4, 2, 1) + Tanh == Conv2d(3 -> 128, 4, 2, 1) of dimg * (1 - img^2) with the layer's weights"""
    B, C, S = 16, 3, 64
    g = torch.Generator().manual_seed(6)
    dimg = torch.randn(B, C, S, S, generator=g).to(DEV)
    img = torch.tanh(torch.randn(B, C, S, S, generator=g)).to(DEV)
    w = (torch.randn(128, C, 4, 4, generator=g) * 0.1).to(DEV)      # ConvTranspose2d master [in = 128][out = C][4][4]
    tdt = ops.torch_dtype(dtype)
    wp = torch.empty(128 * 64, device=DEV, dtype=tdt)
    ops.pack_strided(dtype, w, wp, 128, C * 16, 64, 1, C * 16, 0, 1)
    dz = torch.empty_like(dimg)
    part = torch.empty(B * C, device=DEV)
    gb = torch.zeros(C, device=DEV)
    ops.act_grad_mul_bias_nchw(dimg, img, dz, B, C, S * S, ops.ACT_TANH, 0.0, part, gb)
    want = _patch_path(dtype, [dz], wp, B, C, S, {})
    got = torch.empty_like(want)
    ops.conv_img_mfma(dtype, [dimg], wp, got, B, C, S, S, None, gates=[img], gate_act=ops.ACT_TANH)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("T,B,C,N", [(1, 4, 1, 32), (3, 5, 3, 32), (2, 160, 3, 32), (1, 6, 3, 64), (2, 3, 4, 64), (3, 4, 3, 128), (1, 40, 3, 128)])
def test_image_side_weight_gradient_without_patch_rows(T, B, C, N, dtype):
    """eg_wgrad_img against eg_im2col_img + the per-tap GEMM over the patch rows (the path it replaces) and against torch's autograd on the
    operands as the kernel sees them (images and output gradients rounded through the compute dtype); (2, 160, .., 32) and (1, 40, .., 128): more
    tiles than workgroups; N = 128: the first Discriminator layer of the CelebA script"""
    S = 64
    g = torch.Generator().manual_seed(9)
    tdt = ops.torch_dtype(dtype)
    rq = lambda x: x.to(tdt).float()
    imgs = [(torch.rand(B, C, S, S, generator=g) * 2 - 1) for _ in range(T)]
    dy = rq(torch.randn(T * B, N, S // 2, S // 2, generator=g))
    w = torch.zeros(N, C, 4, 4, requires_grad=True)
    F.conv2d(rq(torch.cat(imgs)), w, None, 2, 1).backward(dy)
    P = dy.permute(0, 2, 3, 1).contiguous().to(DEV).to(tdt)
    dimgs = [im.to(DEV) for im in imgs]
    assert ops.wgrad_img_ok(dtype, C, S, S, N, 4, 2, 1)
    slab = torch.full((ops.wgrad_img_splits(T * B, N) * N * 16 * C,), float("nan"), device=DEV)
    ns = ops.wgrad_img(dtype, dimgs, P, slab, B, C, S, S, N)
    assert ns == ops.wgrad_img_splits(T * B, N)
    got = torch.zeros(N, C, 4, 4, device=DEV)
    ops.wgrad_reduce(slab, ns, N, N, 16 * C, 1, got, accumulate=True)
    # the path it replaces
    kp = 16 * C
    npix = B * (S // 2) ** 2
    patches = torch.empty(T * npix, kp, device=DEV, dtype=tdt)
    for t in range(T):
        ops.im2col_img(dtype, dimgs[t], patches[t * npix:(t + 1) * npix], B, C, S, S, 4, 2, 1, kp)
    c = ops.make_conv(T * B, S // 2, S // 2, kp, N, 1, 1, 0)
    slab2 = torch.empty(ops.conv_wgrad_ws_bytes(c, dtype) // 4, device=DEV)
    ns2 = ops.conv_wgrad(c, dtype, patches, P, slab2)
    ref = torch.zeros(N, C, 4, 4, device=DEV)
    ops.wgrad_reduce(slab2, ns2, N, N, kp, 1, ref, accumulate=True)
    torch.cuda.synchronize()
    scale = (T * B * 1024) ** 0.5
    torch.testing.assert_close(got.cpu(), w.grad, rtol=2e-3, atol=2e-4 * scale)
    torch.testing.assert_close(got.cpu(), ref.cpu(), rtol=1e-4, atol=2e-5 * scale)      # same 16-bit products, fp32 sums in another order
    # deterministic
    slab3 = torch.zeros_like(slab)
    ops.wgrad_img(dtype, dimgs, P, slab3, B, C, S, S, N)
    torch.cuda.synchronize()
    assert torch.equal(slab, slab3)
    assert not ops.wgrad_img_ok(0, C, S, S, N, 4, 2, 1) and not ops.wgrad_img_ok(dtype, C, 32, 32, N, 4, 2, 1) and not ops.wgrad_img_ok(dtype, C, S, S, 48, 4, 2, 1)


def test_other_shapes_are_refused():
    assert not ops.conv_img_mfma_ok(0, 3, 64, 64, 128, 4, 2, 1)          # fp32
    assert not ops.conv_img_mfma_ok(1, 3, 32, 32, 128, 4, 2, 1)          # 32-pixel rows
    assert not ops.conv_img_mfma_ok(1, 3, 64, 64, 48, 4, 2, 1)           # 48 output channels (32, 64 and 128 run)
    assert not ops.conv_img_mfma_ok(1, 3, 64, 64, 32, 3, 2, 1)           # 3x3 filters


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("K", [128, 64])
@pytest.mark.parametrize("B,C,act", [(4, 3, 3), (128, 3, 3), (6, 3, 0), (5, 1, 3), (7, 1, 4)])
def test_transposed_image_convolution_in_one_launch(B, C, act, dtype, K):
    """ConvTranspose2d(K -> C, 4, 2, 1) (+ bias + Tanh: the CelebA Generator's last layer, K = 128; + Sigmoid (act 4): the dSprites generators',
    K = 64; plain: the backward-to-image of the first Discriminator layer) as ONE launch with the GEMM's columns in LDS == eg_conv_fwd (N = 16 C
    columns, stored as dtype T) + eg_col2im_img, bit for bit"""
    Hin = 32
    g = torch.Generator().manual_seed(8)
    tdt = ops.torch_dtype(dtype)
    a = torch.randn(B, Hin, Hin, K, generator=g).to(DEV).to(tdt)
    w = (torch.randn(K, C, 4, 4, generator=g) * 0.05).to(DEV)          # ConvTranspose2d master [in = K][out = C][4][4]
    bias = (torch.randn(C, generator=g) * 0.1).to(DEV) if act else None
    kp = 16 * C
    c = ops.make_conv(B, Hin, Hin, K, kp, 1, 1, 0)
    wp = torch.empty(ops.pack_fwd_elems(c, dtype), device=DEV, dtype=tdt)
    ops.pack_strided(dtype, w, wp, kp, K, K, C, 1, 16, kp)              # wp[t*C + c][ci] = W[ci][c][t]
    cols = torch.empty(B * Hin * Hin, kp, device=DEV, dtype=tdt)
    ops.conv_fwd(c, dtype, a, wp, cols, None)
    want = torch.empty(B, C, 2 * Hin, 2 * Hin, device=DEV)
    ops.col2im_img(dtype, cols, B, C, Hin, Hin, 4, 2, 1, bias, act, 0.0, want)
    assert ops.convt_img_mfma_ok(dtype, C, Hin, Hin, K, 4, 2, 1) and not ops.convt_img_mfma_ok(dtype, C, Hin, Hin, 96, 4, 2, 1)
    got = torch.full_like(want, float("nan"))
    ops.convt_img_mfma(dtype, a, wp, bias, got, B, C, Hin, Hin, act, 0.0, K=K)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = F.conv_transpose2d(a.float().permute(0, 3, 1, 2), w.to(tdt).float(), bias, 2, 1)
    ref = torch.tanh(ref) if act == 3 else (torch.sigmoid(ref) if act == 4 else ref)
    torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("N", [128, 64])
def test_input_gradient_with_batchnorm_backward_sums(dtype, N):
    """the same launch also forms dy = da * relu'(bn(z)) and the two sums of the BatchNorm backward of the layer below (EG_STAT_BN_BWD):
    dy / dz / dgamma / dbeta against the stand-alone kernels on the plain launch's output (N = 64: the dSprites generators)"""
    B, C, S = 16, 3, 64
    g = torch.Generator().manual_seed(9)
    tdt = ops.torch_dtype(dtype)
    dimg = torch.randn(B, C, S, S, generator=g).to(DEV)
    img = torch.tanh(torch.randn(B, C, S, S, generator=g)).to(DEV)
    w = (torch.randn(N, C, 4, 4, generator=g) * 0.1).to(DEV)
    wp = torch.empty(N * 64, device=DEV, dtype=tdt)
    ops.pack_strided(dtype, w, wp, N, C * 16, 64, 1, C * 16, 0, 1)
    M = B * (S // 2) ** 2
    z = torch.randn(B, S // 2, S // 2, N, generator=g).to(DEV).to(tdt)
    gamma, beta = (torch.randn(N, generator=g) * 0.1 + 1).to(DEV), (torch.randn(N, generator=g) * 0.3).to(DEV)
    zf = z.float().reshape(M, N)
    mean, istd = zf.mean(0).contiguous(), (zf.var(0, unbiased=False) + 1e-5).rsqrt().contiguous()
    da = torch.empty(B, S // 2, S // 2, N, device=DEV, dtype=tdt)
    dy = torch.empty_like(da)
    ops.conv_img_mfma(dtype, [dimg], wp, da, B, C, S, S, None, gates=[img], gate_act=ops.ACT_TANH, N=N)
    nrb = ops.conv_img_mfma_stat_blocks(B, S, S)
    stat = torch.full((2 * N * nrb,), float("nan"), device=DEV)
    ops.conv_img_mfma(dtype, [dimg], wp, dy, B, C, S, S, ops.epilogue(stat_mode=ops.STAT_BN_BWD, stat_out=stat, stat_aux=z, stat_p=(mean, istd, gamma, beta),
                                                                    stat_act=ops.ACT_RELU), gates=[img], gate_act=ops.ACT_TANH, N=N)
    torch.cuda.synchronize()
    assert torch.isfinite(stat).all()
    pre = zf * (gamma * istd) + (beta - mean * gamma * istd)
    near = (pre.abs() < 1e-5).reshape(da.shape)
    assert torch.equal(dy[~near], torch.where((pre > 0).reshape(da.shape), da, torch.zeros_like(da))[~near])
    small = torch.empty(ops.bn_ws_floats(M, N), device=DEV)
    res = {}
    for k in ("plain", "fused"):
        res[k] = dict(dz=torch.empty_like(da), dg=torch.zeros(N, device=DEV), db=torch.zeros(N, device=DEV), sums=torch.empty(2 * N, device=DEV))
    ops.bn_bwd(dtype, z, da, res["plain"]["dz"], M, N, gamma, beta, mean, istd, ops.ACT_RELU, 0.0, res["plain"]["dg"], res["plain"]["db"], res["plain"]["sums"], small)
    ops.bn_bwd_fused(dtype, z, dy, res["fused"]["dz"], M, N, stat, nrb, gamma, beta, mean, istd, res["fused"]["dg"], res["fused"]["db"], res["fused"]["sums"], small)
    torch.cuda.synchronize()
    scale = float(res["plain"]["sums"].abs().max())
    for k in ("sums", "dg", "db"):
        torch.testing.assert_close(res["fused"][k], res["plain"][k], rtol=1e-4, atol=1e-5 * scale)
    diff = (res["fused"]["dz"].float() - res["plain"]["dz"].float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(res["plain"]["dz"].float().abs().max()) and (diff > 0).float().mean() < 2e-2
