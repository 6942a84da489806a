// Image-side layers of the CelebA networks on the MFMA units WITHOUT patch rows in HBM (16-bit types, gfx950):
//
//   conv_img_mfma : Conv2d(C <= 4 -> 128, 4, 2, 1) over fp32 NCHW images -> 16-bit NHWC activations
//                   * the first Discriminator layer, forward (celebA/EAD-GAN_celebA.py:110; spectral norm: 1/sigma per tape, bias, LeakyReLU)
//                   * the input gradient of the Generator's last ConvTranspose2d (:90-91), whose backward is this very convolution of
//                     d(img) * tanh'(img) with the same weights
//
// Before, these ran as eg_im2col_img (fp32 image -> 16-bit patch rows [B*OH*OW][64] in HBM) + a K = 64 GEMM reading them back: 50 MB of
// traffic per 128 images for 6 MB of input and 34 MB of output, 38 us (one tape) to 95 us (three) on the step's critical chain.  Here a
// workgroup stages the six fp32 image rows that two output rows need in LDS (as 16-bit values), every lane builds its two K = 32
// fragments from them (fragment chunk = one channel, two filter rows, four columns: 2 x 8 bytes of LDS), the 128 x 64 weight panel
// lives in registers, 16 MFMAs per 16 pixels, and the tile leaves through LDS as whole 256-byte pixel rows.
// Same operands in the same MFMA slots as the patch-row GEMM (K order = master weight order (c, ky, kx), zero padded to 64; k halves in
// order) and the same epilogue arithmetic -> bit-identical outputs (tests/test_gpu_img_conv.py).
#include <stdlib.h>

#include "eg_common.h"
#include "igemm_nt.h"

#ifndef EG_IMG_CONV_OCC
#define EG_IMG_CONV_OCC 3          // workgroups (of 4 waves) per CU the register allocation aims for
#endif

struct ImgMfmaParams {
    const float* img[4];      // per tape: [B][C][H][W] fp32
    const float* gate[4];     // per tape or null: input = img * act'(gate) (gate = the activation OUTPUT the image gradient passes through)
    const void* wp;           // [128][64] panel, dtype T: wp[n][c*16 + ky*4 + kx], columns C*16 .. 63 zero
    void* out;                // [ntapes*B][H/2][W/2][128] dtype T
    const float* bias;        // [128] or null
    const float* sigma;       // [ntapes] or null: out = acc / sigma[tape] + bias
    int B, C, H, W;
    int act;
    float slope;
    int gate_act;
    float gate_slope;
    // EG_STAT_BN_BWD (STAT instantiation): the output is d(activation) of a BatchNorm + activation layer; z = its BatchNorm input (layout of
    // out), per-channel mean / invstd / gamma / beta; dy = out * act'(bn(z)) is stored instead and (sum dy, sum dy * xhat) of every tile of 64
    // pixels go to stat_out[(which * 128 + n) * ntiles + tile]
    const void* stat_z;
    const float* stat_p[4];
    float* stat_out;
    int stat_act;
    float stat_slope;
};

// N = 128 (CelebA), 64 or 32 (the first trunk layer of the dSprites networks, dSprites/rp.py:95-97, rp_color.py:95-97): output channels
// workgroups per CU: the narrow instantiations hold a 4 / 2-tile weight panel in registers instead of 8 tiles -- more workgroups in flight
constexpr int img_conv_occ(int N) { return N == 128 ? EG_IMG_CONV_OCC : (N == 64 ? 4 : 6); }
template <typename T, bool STAT, int N = 128>
__global__ __launch_bounds__(256, STAT ? 2 : img_conv_occ(N)) void conv_img_mfma_kernel(const ImgMfmaParams p, int ntiles) {
    static_assert(N == 128 || N == 64 || N == 32, "column tiles of 16, whole 64-byte chunks per pixel row");
    static_assert(!STAT || N == 128 || N == 64, "the statistics epilogue writes 2 N sums from 256 threads");
    constexpr int NJ = N / 16;                            // column tiles of 16 channels
    constexpr int XS = 72, OS = N + 8;                    // LDS row pitches in elements: 66 used columns (x = -1 .. 64); 16-byte aligned pixel rows
    __shared__ __attribute__((aligned(16))) unsigned short s_in[4][6][XS];
    __shared__ __attribute__((aligned(16))) unsigned short s_out[4][16][OS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int OH = p.H >> 1, OW = p.W >> 1;
    const int gxn = OW >> 5, tpi = (OH >> 1) * gxn;       // 32-column groups per row; tiles (2 output rows x 32 columns) per image
    const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);

    // weight fragments, loaded once per workgroup: column tile j, k half s -> panel rows j*16 + frow, columns s*32 + fq*8 .. +7
    uint4 bf[2][NJ];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[s][j] = *reinterpret_cast<const uint4*>(wp + (size_t)(j * 16 + frow) * 64 + s * 32 + fq * 8);
    __shared__ __attribute__((aligned(16))) float s_bias[N];
    __shared__ __attribute__((aligned(16))) float s_red[STAT ? 4 * 2 * N : 4];      // per wave: the two sums of its 16 pixels, per channel
    if (tid < N) s_bias[tid] = p.bias ? p.bias[tid] : 0.f;      // (published by the loop's first barrier)
    // BatchNorm backward operands per channel in LDS: mean, invstd, P = gamma * invstd, Q = beta - mean * P (published by the first barrier)
    __shared__ __attribute__((aligned(16))) float s_par[STAT ? 4 * N : 4];
    if constexpr (STAT) {
        if (tid < N) {
            const float mu = p.stat_p[0][tid], is = p.stat_p[1][tid], g = p.stat_p[2][tid], bb = p.stat_p[3][tid];
            const float P = __fmul_rn(g, is);
            s_par[tid] = mu; s_par[N + tid] = is; s_par[2 * N + tid] = P; s_par[3 * N + tid] = __fsub_rn(bb, __fmul_rn(mu, P));
        }
    }
    const EgGradFast sgf = eg_grad_fast(p.stat_act, p.stat_slope);
    const EgActFast af = eg_act_fast(p.act, p.slope);

    // staging slots of this thread: elements e = tid and tid + 256 of the 4 x 6 x 16 float4 grid (channel, image row, 4-column group)
    struct Slot { float4 v; float lo, hi; };
    auto fetch = [&](int tile, int e, Slot& sl) {
        sl.v = make_float4(0.f, 0.f, 0.f, 0.f); sl.lo = 0.f; sl.hi = 0.f;
        if (e >= 4 * 6 * 16 || tile >= ntiles) return;
        const int ib = tile / tpi, rem_t = tile - ib * tpi;
        const int tape = ib / p.B, b = ib - tape * p.B;
        const int gx = rem_t % gxn, oy0 = (rem_t / gxn) * 2;
        const int c = e / 96, rem = e - c * 96, r = rem >> 4, x4 = rem & 15;
        const int iy = 2 * oy0 - 1 + r, x0 = gx * 64;
        if (c >= p.C || iy < 0 || iy >= p.H) return;
        const size_t o = (((size_t)b * p.C + c) * p.H + iy) * p.W + x0 + x4 * 4;
        const float* __restrict__ img = p.img[tape];
        const float* __restrict__ gate = p.gate[tape];
        sl.v = *reinterpret_cast<const float4*>(img + o);
        if (x4 == 0 && x0 > 0) sl.lo = img[o - 1];
        if (x4 == 15 && x0 + 64 < p.W) sl.hi = img[o + 4];
        if (gate) {
            const float4 g = *reinterpret_cast<const float4*>(gate + o);
            sl.v.x *= eg_act_grad_from_out(g.x, p.gate_act, p.gate_slope); sl.v.y *= eg_act_grad_from_out(g.y, p.gate_act, p.gate_slope);
            sl.v.z *= eg_act_grad_from_out(g.z, p.gate_act, p.gate_slope); sl.v.w *= eg_act_grad_from_out(g.w, p.gate_act, p.gate_slope);
            if (x4 == 0 && x0 > 0) sl.lo *= eg_act_grad_from_out(gate[o - 1], p.gate_act, p.gate_slope);
            if (x4 == 15 && x0 + 64 < p.W) sl.hi *= eg_act_grad_from_out(gate[o + 4], p.gate_act, p.gate_slope);
        }
    };
    auto stage = [&](int e, const Slot& sl) {           // 16-bit values into LDS; image column x sits at x + 1
        if (e >= 4 * 6 * 16) return;
        const int c = e / 96, rem = e - c * 96, r = rem >> 4, x4 = rem & 15;
        T h[4];
        Elt<T>::st(h + 0, sl.v.x); Elt<T>::st(h + 1, sl.v.y); Elt<T>::st(h + 2, sl.v.z); Elt<T>::st(h + 3, sl.v.w);
        unsigned short* row = &s_in[c][r][0];
        const unsigned short* hb = reinterpret_cast<const unsigned short*>(h);
        row[1 + x4 * 4] = hb[0];
        *reinterpret_cast<unsigned*>(row + 2 + x4 * 4) = (unsigned)hb[1] | ((unsigned)hb[2] << 16);
        row[4 + x4 * 4] = hb[3];
        if (x4 == 0) { T t; Elt<T>::st(&t, sl.lo); row[0] = *reinterpret_cast<const unsigned short*>(&t); }
        if (x4 == 15) { T t; Elt<T>::st(&t, sl.hi); row[65] = *reinterpret_cast<const unsigned short*>(&t); }
    };

    // this wave inside a tile: output row (wave >> 1), columns 16 * (wave & 1) + frow
    const int ry = wave >> 1, oxl = (wave & 1) * 16 + frow;
    // K chunk fq of k half 0: channel fq >> 1, filter rows 2*(fq & 1) and +1, four columns; k half 1: channel 2 + (fq >> 1)
    const int r0 = 2 * ry + (fq & 1) * 2;
    auto ld8 = [&](int c, int r) {
        const unsigned* q = reinterpret_cast<const unsigned*>(&s_in[c][r][2 * oxl]);     // image column 2*ox - 1 + kx sits at 2*ox + kx
        return make_uint2(q[0], q[1]);
    };
    T* __restrict__ out = reinterpret_cast<T*>(p.out);

    Slot s0, s1;
    int tile = blockIdx.x;
    fetch(tile, tid, s0); fetch(tile, tid + 256, s1);
    for (; tile < ntiles; tile += gridDim.x) {
        stage(tid, s0); stage(tid + 256, s1);
        __syncthreads();                                 // the tile's image rows are in LDS (and the previous tile's output rows have been read)
        fetch(tile + gridDim.x, tid, s0); fetch(tile + gridDim.x, tid + 256, s1);      // next tile's loads fly under this tile's work
        const uint2 a00 = ld8(fq >> 1, r0), a01 = ld8(fq >> 1, r0 + 1);
        const uint2 a10 = ld8(2 + (fq >> 1), r0), a11 = ld8(2 + (fq >> 1), r0 + 1);
        const uint4 a0 = make_uint4(a00.x, a00.y, a01.x, a01.y), a1 = make_uint4(a10.x, a10.y, a11.x, a11.y);
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) mfma_step<T>(a0, bf[0][j], acc[j]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) mfma_step<T>(a1, bf[1][j], acc[j]);
        // epilogue: acc[j][r] = C[pixel frow][channel j*16 + fq*4 + r]; the arithmetic of the NT kernels' epilogue
        const int ib = tile / tpi, rem_t = tile - ib * tpi;
        const int tape = ib / p.B;
        const int gx = rem_t % gxn, oy0 = (rem_t / gxn) * 2;
        const float inv_sigma = p.sigma ? 1.f / p.sigma[tape] : 1.f;
        const size_t pix0 = ((size_t)ib * OH + oy0 + ry) * OW + gx * 32 + (wave & 1) * 16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            T h[4];
            const float4 bv = *reinterpret_cast<const float4*>(&s_bias[j * 16 + fq * 4]);
            const float bj[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = __fmul_rn(acc[j][r], inv_sigma);
                if (p.bias) x = __fadd_rn(x, bj[r]);
                x = af.special ? eg_act(x, p.act, p.slope) : eg_act_apply(x, af);
                Elt<T>::st(h + r, x);
            }
            if constexpr (STAT) {
                // dy = da * act'(bn(z)) (the forward's activation input z * P + Q), sums over this wave's 16 pixels by DPP row reductions
                const uint2 zv = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(p.stat_z) + (pix0 + frow) * N + j * 16 + fq * 4);
                const T* ze = reinterpret_cast<const T*>(&zv);
                float s1[4], s2[4];
                const int nb = j * 16 + fq * 4;
                const float4 vmu = *reinterpret_cast<const float4*>(&s_par[nb]), vis = *reinterpret_cast<const float4*>(&s_par[N + nb]);
                const float4 vp = *reinterpret_cast<const float4*>(&s_par[2 * N + nb]), vq = *reinterpret_cast<const float4*>(&s_par[3 * N + nb]);
                const float k_mu[4] = {vmu.x, vmu.y, vmu.z, vmu.w}, k_is[4] = {vis.x, vis.y, vis.z, vis.w};
                const float k_p[4] = {vp.x, vp.y, vp.z, vp.w}, k_q[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = Elt<T>::ld(ze + r);
                    const float xh = (z - k_mu[r]) * k_is[r];
                    const float dy = Elt<T>::ld(h + r) * eg_grad_apply(fmaf(z, k_p[r], k_q[r]), sgf);
                    Elt<T>::st(h + r, dy);
                    s1[r] = dy;
                    s2[r] = dy * xh;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {             // sum over frow = the 16 lanes of a DPP row (fixed order)
                    s1[r] += eg_dpp<0xB1>(s1[r]); s1[r] += eg_dpp<0x4E>(s1[r]); s1[r] += eg_dpp<0x141>(s1[r]); s1[r] += eg_dpp<0x140>(s1[r]);
                    s2[r] += eg_dpp<0xB1>(s2[r]); s2[r] += eg_dpp<0x4E>(s2[r]); s2[r] += eg_dpp<0x141>(s2[r]); s2[r] += eg_dpp<0x140>(s2[r]);
                }
                if (frow == 0) {
                    *reinterpret_cast<float4*>(&s_red[(wave * 2 + 0) * N + j * 16 + fq * 4]) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                    *reinterpret_cast<float4*>(&s_red[(wave * 2 + 1) * N + j * 16 + fq * 4]) = make_float4(s2[0], s2[1], s2[2], s2[3]);
                }
            }
            *reinterpret_cast<uint2*>(&s_out[wave][frow][j * 16 + fq * 4]) = *reinterpret_cast<const uint2*>(h);
        }
        __syncthreads();                                 // output rows complete; every wave is done reading the image rows
        if constexpr (STAT) {
            if (tid < 2 * N) {
                const int which = tid / N, n = tid - which * N;
                const float t = ((s_red[(0 * 2 + which) * N + n] + s_red[(1 * 2 + which) * N + n]) + s_red[(2 * 2 + which) * N + n]) + s_red[(3 * 2 + which) * N + n];
                p.stat_out[((size_t)which * N + n) * ntiles + tile] = t;
            }
        }
        constexpr int CPR = N / 8, RPI = 64 / CPR;       // 16-byte chunks per pixel row; pixel rows one store instruction covers
#pragma unroll
        for (int it = 0; it < 16 / RPI; ++it) {
            const int row = it * RPI + lane / CPR, chunk = lane % CPR;
            const uint4 v = *reinterpret_cast<const uint4*>(&s_out[wave][row][chunk * 8]);
            *reinterpret_cast<uint4*>(out + (pix0 + row) * N + chunk * 8) = v;
        }
    }
}

/* row blocks (tiles of 64 pixels) of the EG_STAT_BN_BWD sums an eg_conv_img_mfma launch writes */
extern "C" int eg_conv_img_mfma_stat_blocks(int B, int H, int W, int ntapes) { return (H / 4) * (W / 64) * B * ntapes; }

extern "C" int eg_conv_img_mfma_ok(int dtype, int C, int H, int W, int N, int k, int stride, int pad) {
    return dtype != EG_F32 && C >= 1 && C <= 4 && (N == 128 || N == 64 || N == 32) && k == 4 && stride == 2 && pad == 1 && H >= 4 && (H % 4) == 0 && (W % 64) == 0;
}

extern "C" int eg_conv_img_mfma_n(int dtype, const float* img0, const float* img1, const float* img2, const float* gate0, const float* gate1,
                                  const float* gate2, int ntapes, const void* wp, void* out, int B, int C, int H, int W, int N, const eg_epilogue* ep,
                                  int gate_act, float gate_slope, eg_stream_t s) {
    EG_REQUIRE(img0 && wp && out && ntapes >= 1 && ntapes <= 3 && B > 0, "eg_conv_img_mfma: bad argument");
    EG_REQUIRE(eg_conv_img_mfma_ok(dtype, C, H, W, N, 4, 2, 1), "eg_conv_img_mfma: 16-bit types, C <= 4, N = 32 / 64 / 128, H %% 4 == 0, W %% 64 == 0 only (use eg_im2col_img + eg_conv_fwd)");
    EG_REQUIRE((ntapes < 2 || img1) && (ntapes < 3 || img2), "eg_conv_img_mfma: one image pointer per tape");
    EG_REQUIRE(!ep || (!ep->mask && ep->out_mode == EG_OUT_NHWC && ep->bias_mod == 0 && (ep->stat_mode == EG_STAT_NONE || ep->stat_mode == EG_STAT_BN_BWD)),
               "eg_conv_img_mfma: unsupported epilogue field");
    const bool stat = ep && ep->stat_mode == EG_STAT_BN_BWD;
    EG_REQUIRE(!stat || N == 128 || N == 64, "eg_conv_img_mfma: EG_STAT_BN_BWD with 128 or 64 output channels only");
    EG_REQUIRE(!stat || (ep->stat_out && ep->stat_aux && ep->stat_p0 && ep->stat_p1 && ep->stat_p2 && ep->stat_p3), "eg_conv_img_mfma: EG_STAT_BN_BWD needs stat_out, stat_aux (z) and stat_p0..p3");
    ImgMfmaParams p;
    memset(&p, 0, sizeof(p));
    p.img[0] = img0; p.img[1] = img1; p.img[2] = img2;
    p.gate[0] = gate0; p.gate[1] = gate1; p.gate[2] = gate2;
    p.wp = wp; p.out = out; p.B = B; p.C = C; p.H = H; p.W = W;
    p.bias = ep ? ep->bias : nullptr;
    p.sigma = ep ? ep->sigma : nullptr;
    p.act = ep ? ep->act : EG_ACT_NONE;
    p.slope = ep ? ep->slope : 0.f;
    p.gate_act = gate_act; p.gate_slope = gate_slope;
    if (stat) {
        p.stat_z = ep->stat_aux; p.stat_out = ep->stat_out; p.stat_act = ep->stat_act; p.stat_slope = ep->stat_slope;
        p.stat_p[0] = ep->stat_p0; p.stat_p[1] = ep->stat_p1; p.stat_p[2] = ep->stat_p2; p.stat_p[3] = ep->stat_p3;
    }
    const int ntiles = (H / 4) * (W / 64) * B * ntapes;  // 2 output rows x 32 columns each
    static const int wgs = [] { const char* e = getenv("EG_IMG_CONV_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256 * EG_IMG_CONV_OCC; }();
    const int wgs_n = N == 128 ? wgs : 256 * img_conv_occ(N);
    const dim3 grid(ntiles < wgs_n ? ntiles : wgs_n);     // persistent: a few workgroups per CU walk the tiles with the weight panel in registers
    if (stat) {
        const dim3 gs(ntiles < 512 ? ntiles : 512);      // two workgroups per CU: the sums take the registers
        if (N == 64) {
            if (dtype == EG_F16) hipLaunchKernelGGL((conv_img_mfma_kernel<f16_t, true, 64>), gs, dim3(256), 0, (hipStream_t)s, p, ntiles);
            else hipLaunchKernelGGL((conv_img_mfma_kernel<bf16_t, true, 64>), gs, dim3(256), 0, (hipStream_t)s, p, ntiles);
        } else if (dtype == EG_F16) hipLaunchKernelGGL((conv_img_mfma_kernel<f16_t, true>), gs, dim3(256), 0, (hipStream_t)s, p, ntiles);
        else hipLaunchKernelGGL((conv_img_mfma_kernel<bf16_t, true>), gs, dim3(256), 0, (hipStream_t)s, p, ntiles);
    } else if (N == 128) {
        if (dtype == EG_F16) hipLaunchKernelGGL((conv_img_mfma_kernel<f16_t, false>), grid, dim3(256), 0, (hipStream_t)s, p, ntiles);
        else hipLaunchKernelGGL((conv_img_mfma_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)s, p, ntiles);
    } else if (N == 64) {
        if (dtype == EG_F16) hipLaunchKernelGGL((conv_img_mfma_kernel<f16_t, false, 64>), grid, dim3(256), 0, (hipStream_t)s, p, ntiles);
        else hipLaunchKernelGGL((conv_img_mfma_kernel<bf16_t, false, 64>), grid, dim3(256), 0, (hipStream_t)s, p, ntiles);
    } else {
        if (dtype == EG_F16) hipLaunchKernelGGL((conv_img_mfma_kernel<f16_t, false, 32>), grid, dim3(256), 0, (hipStream_t)s, p, ntiles);
        else hipLaunchKernelGGL((conv_img_mfma_kernel<bf16_t, false, 32>), grid, dim3(256), 0, (hipStream_t)s, p, ntiles);
    }
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_conv_img_mfma(int dtype, const float* img0, const float* img1, const float* img2, const float* gate0, const float* gate1,
                                const float* gate2, int ntapes, const void* wp, void* out, int B, int C, int H, int W, const eg_epilogue* ep,
                                int gate_act, float gate_slope, eg_stream_t s) {
    return eg_conv_img_mfma_n(dtype, img0, img1, img2, gate0, gate1, gate2, ntapes, wp, out, B, C, H, W, 128, ep, gate_act, gate_slope, s);
}

// ------------------------------------------------------------------------------------------------
//   convt_img_mfma : ConvTranspose2d(128 -> C <= 3, 4, 2, 1) from 16-bit NHWC activations to an fp32 NCHW image
//                    * the Generator's last layer + Tanh (celebA/EAD-GAN_celebA.py:90-91)
//                    * the backward-to-image of the first Discriminator layer Conv2d(C -> 128, 4, 2, 1) (:110)
// Before: ONE GEMM over the input pixels with N = 16 taps x C columns (cols[m][t*C + c], stored to HBM as dtype T) + eg_col2im_img
// gathering four taps per output pixel: 29-50 us on the critical chain.  Here a workgroup takes 16 input rows of one image plus one halo
// row on either side, runs that GEMM on the MFMA units (the 48 x 128 panel in registers, activations straight from global memory into
// fragments), keeps the columns in LDS (rounded to T, as the stored ones were) and gathers its 32 output rows from there.
// Same MFMA operands in the same order, same rounding of the columns, same tap order in the gather: bit-identical to GEMM + col2im.
// ------------------------------------------------------------------------------------------------
struct ImgTParams {
    const void* a;            // [B][Hin][Win][128] dtype T
    const void* wp;           // [16*C rows (n = t*C + c)][128] dtype T
    const float* bias;        // [C] or null
    float* out;               // [B][C][2*Hin][2*Win] fp32
    int B, C, Hin, Win;
    int act;
    float slope;
};

// K = 128 (CelebA) or 64 (the dSprites generators' last layer, dSprites/rp.py:139-140): input channels
template <typename T, int R, int K = 128>
__global__ __launch_bounds__(256) void convt_img_mfma_kernel(const ImgTParams p) {
    constexpr int KS = K / 32;                           // MFMA K steps
    constexpr int WMAX = 32, NP = 56;           // R input rows per workgroup (+ 2 halo rows); LDS pitch of a pixel's 48 columns
    __shared__ __attribute__((aligned(16))) unsigned short s_cols[(R + 2) * WMAX * NP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int nblk = p.Hin / R, b = blockIdx.x / nblk, r0 = (blockIdx.x % nblk) * R;
    const int ncol = 16 * p.C;                           // <= 48
    const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp);
    const T* __restrict__ a = reinterpret_cast<const T*>(p.a) + (size_t)b * p.Hin * p.Win * K;

    // weight fragments: column tile j (n = j*16 + frow), k step s
    uint4 bf[KS][3];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = j * 16 + frow;
            bf[s][j] = n < ncol ? *reinterpret_cast<const uint4*>(wp + (size_t)n * K + s * 32 + fq * 8) : make_uint4(0, 0, 0, 0);
        }
    // GEMM over the (R + 2) x Win pixels of this block, 16 pixels per wave and pass; rows outside the image are skipped (never gathered)
    const int npix = (R + 2) * p.Win, ngrp = npix / 16;
    auto fetch = [&](int g, uint4 (&af)[KS]) {            // fragments of pass g: pixel g*16 + frow, k chunks s*32 + fq*8
        const int pl = g * 16 + frow;
        const int iy = r0 - 1 + pl / p.Win, ix = pl % p.Win;
        const bool ok = g < ngrp && iy >= 0 && iy < p.Hin;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            af[s] = ok ? *reinterpret_cast<const uint4*>(a + ((size_t)iy * p.Win + ix) * K + s * 32 + fq * 8) : make_uint4(0, 0, 0, 0);
    };
    auto compute = [&](int g, const uint4 (&af)[KS]) {
        const int pl = g * 16 + frow;                    // local pixel: row pl / Win (0 = halo row r0 - 1), column pl % Win
        f32x4 acc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 3; ++j) mfma_step<T>(af[s], bf[s][j], acc[j]);
        // acc[j][r] = cols[pixel frow][n = j*16 + fq*4 + r], rounded to T like the stored columns were
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) Elt<T>::st(h + r, __fmul_rn(acc[j][r], 1.f));
            *reinterpret_cast<uint2*>(&s_cols[(size_t)pl * NP + j * 16 + fq * 4]) = *reinterpret_cast<const uint2*>(h);
        }
    };
    // two passes in flight: the loads of pass g + 4 are issued before pass g is multiplied (a wave alone would wait out every load)
    uint4 fa[KS], fb[KS];
    fetch(wave, fa);
    for (int g = wave; g < ngrp; g += 8) {
        fetch(g + 4, fb);
        compute(g, fa);
        if (g + 4 < ngrp) {
            fetch(g + 8, fa);
            compute(g + 4, fb);
        }
    }
    __syncthreads();
    // gather: output rows 2*r0 .. 2*(r0 + R) - 1, all columns; taps in eg_col2im_img's order (kh ascending, kw ascending)
    const int OH = 2 * p.Hin, OW = 2 * p.Win;
    const T* __restrict__ cols = reinterpret_cast<const T*>(s_cols);
    for (int o = tid; o < 2 * R * OW; o += 256) {
        const int ox = o % OW, oy = 2 * r0 + o / OW;
        float acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = (p.bias && c < p.C) ? p.bias[c] : 0.f;
        for (int kh = (oy + 1) & 1; kh < 4; kh += 2) {
            const int iy = (oy + 1 - kh) >> 1;
            if (iy < 0 || iy >= p.Hin) continue;
            for (int kw = (ox + 1) & 1; kw < 4; kw += 2) {
                const int ix = (ox + 1 - kw) >> 1;
                if (ix < 0 || ix >= p.Win) continue;
                const T* src = cols + (size_t)((iy - (r0 - 1)) * p.Win + ix) * NP + (kh * 4 + kw) * p.C;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < p.C) acc[c] += Elt<T>::ld(src + c);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < p.C) p.out[(((size_t)b * p.C + c) * OH + oy) * OW + ox] = eg_act(acc[c], p.act, p.slope);
    }
}

extern "C" int eg_convt_img_mfma_ok(int dtype, int C, int Hin, int Win, int K, int k, int stride, int pad) {
    return dtype != EG_F32 && C >= 1 && C <= 3 && (K == 128 || K == 64) && k == 4 && stride == 2 && pad == 1 && Hin >= 16 && (Hin % 16) == 0 && (Win == 16 || Win == 32);
}

extern "C" int eg_convt_img_mfma_k(int dtype, const void* a, const void* wp, const float* bias, float* out, int B, int C, int Hin, int Win, int K, int act,
                                   float slope, eg_stream_t s) {
    EG_REQUIRE(a && wp && out && B > 0, "eg_convt_img_mfma: bad argument");
    EG_REQUIRE(eg_convt_img_mfma_ok(dtype, C, Hin, Win, K, 4, 2, 1), "eg_convt_img_mfma: 16-bit types, C <= 3, K = 64 / 128, Hin %% 16 == 0, Win 16 or 32 only (use eg_conv_fwd + eg_col2im_img)");
    ImgTParams p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.wp = wp; p.bias = bias; p.out = out; p.B = B; p.C = C; p.Hin = Hin; p.Win = Win; p.act = act; p.slope = slope;
    hipStream_t st = (hipStream_t)s;
    if (K == 64) {                                      // (8 input rows per workgroup)
        const dim3 grid(B * (Hin / 8));
        if (dtype == EG_F16) hipLaunchKernelGGL((convt_img_mfma_kernel<f16_t, 8, 64>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((convt_img_mfma_kernel<bf16_t, 8, 64>), grid, dim3(256), 0, st, p);
        EG_LAUNCH_CHECK();
        return 0;
    }
    // input rows per workgroup: 8 (+ 2 halo rows: 25 % of the GEMM done twice, 36 KiB of LDS, several workgroups per CU) or 16
    static const int rows = [] { const char* e = getenv("EG_CONVT_IMG_ROWS"); return e && atoi(e) == 16 ? 16 : 8; }();
    if (rows == 16) {
        const dim3 grid(B * (Hin / 16));
        if (dtype == EG_F16) hipLaunchKernelGGL((convt_img_mfma_kernel<f16_t, 16>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((convt_img_mfma_kernel<bf16_t, 16>), grid, dim3(256), 0, st, p);
    } else {
        const dim3 grid(B * (Hin / 8));
        if (dtype == EG_F16) hipLaunchKernelGGL((convt_img_mfma_kernel<f16_t, 8>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((convt_img_mfma_kernel<bf16_t, 8>), grid, dim3(256), 0, st, p);
    }
    EG_LAUNCH_CHECK();
    return 0;
}
extern "C" int eg_convt_img_mfma(int dtype, const void* a, const void* wp, const float* bias, float* out, int B, int C, int Hin, int Win, int act,
                                 float slope, eg_stream_t s) {
    return eg_convt_img_mfma_k(dtype, a, wp, bias, out, B, C, Hin, Win, 128, act, slope, s);
}

// ------------------------------------------------------------------------------------------------
//   wgrad_img : weight gradient of the image-side 4x4 / stride-2 / pad-1 layers of the dSprites networks WITHOUT patch rows in HBM
//               S[n][c*16 + ky*4 + kx] = sum_{b,oy,ox} P[(b, oy, ox)][n] * img[b][c][2 oy - 1 + ky][2 ox - 1 + kx]
//               * Conv2d(C -> 32, 4, 2, 1), the first trunk layer (dSprites/rp.py:95-97, 165-167): P = d(loss)/d(pre-activation), N = 32
//               * ConvTranspose2d(64 -> C, 4, 2, 1), the generator's last layer (:139-140): img = d(loss)/d(image pre-activation), P = the
//                 layer's input activation, N = 64 (conv view: same sum)
// Before: eg_im2col_img wrote the patch rows [B*1024][16 C] to HBM (67 MB per tape at B = 512, C = 3; 39 us on the main chain) and the per-tap
// TN GEMM read them back beside P.  Here a workgroup takes 4 output rows x 32 columns of one image, stages the 10 fp32 image rows they
// touch in LDS as 16-bit values, expands them to patch rows IN LDS, copies the 128 rows of P beside them and multiplies with both operands read
// transposed (ds_read_b64_tr_b16), one K = 32 step (= one output row) per wave; workgroups walk the tiles and write one fp32 slab each in the
// per-tap kernel's layout (slab[split][n][16 C]) for the same slab reductions.  64 x 64 images only.
// ------------------------------------------------------------------------------------------------
struct ImgWgradParams {
    const float* img[4];      // per tape: [B][C][64][64] fp32
    const void* P;            // [ntapes * B * 1024][N] dtype T (tape-major rows)
    float* slab;              // [gridDim.x][N][16 * C]
    int B, C, ntiles;         // tiles of 128 output pixels over all tapes
};

template <typename T, int N>
__global__ __launch_bounds__(256) void wgrad_img_kernel(const ImgWgradParams p) {
    constexpr int H = 64, W = 64, OW = 32, XS = 72;      // image size; LDS pitch of an image row (x = -1 .. 64 at columns 0 .. 65)
    constexpr int PP = N + 8, PX = 72;                   // LDS pitches (elements) of a P row and of a patch row (64 columns used)
    constexpr int NI = N / 16;
    __shared__ __attribute__((aligned(16))) unsigned short s_in[4][10][XS];
    __shared__ __attribute__((aligned(16))) unsigned short s_px[128 * PP + 128 * PX];
    unsigned short* s_p = s_px;
    unsigned short* s_x = s_px + 128 * PP;
    // the four waves' sums of one column tile at a time: over the operand tiles, behind the loop's last barrier (N = 128: 32 KiB over 52)
    static_assert(4 * N * 16 * sizeof(float) <= (128 * PP + 128 * PX) * sizeof(unsigned short), "the reduction scratch lies over the operand tiles");
    float (*s_red)[N][16] = reinterpret_cast<float (*)[N][16]>(s_px);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pc = li & 3;
    const int kc = p.C * 16, nj = p.C;                   // patch columns, column tiles of 16
    const T* __restrict__ P = reinterpret_cast<const T*>(p.P);
    for (int i = tid; i < 4 * 10 * XS; i += 256) (&s_in[0][0][0])[i] = 0;       // halo columns and unused channels stay zero
    for (int i = tid; i < 128 * PX; i += 256) s_x[i] = 0;                       // columns 16 C .. 63 stay zero
    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    auto tr2 = [&](const unsigned short* base, int pitch, int row, int blk) {   // 8 K-consecutive elements of column blk*16 + li (rows row .. row + 7)
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + (row + q) * pitch + blk * 16 + pc * 4));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + (row + 4 + q) * pitch + blk * 16 + pc * 4));
        return make_uint4(((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                          ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16));
    };
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int ib = tile >> 3, oy0 = (tile & 7) * 4;  // image over all tapes; first of the tile's four output rows
        const int tape = ib / p.B, b = ib - tape * p.B;
        const float* __restrict__ img = p.img[tape] + (size_t)b * p.C * H * W;
        __syncthreads();                                 // the previous tile's fragments have been read
        // image rows 2 oy0 - 1 .. 2 oy0 + 8 of every channel, fp32 -> T; rows outside the image are zero
        for (int e = tid; e < p.C * 10 * 16; e += 256) {
            const int c = e / 160, rem = e - c * 160, r = rem >> 4, x4 = rem & 15;
            const int iy = 2 * oy0 - 1 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H) v = *reinterpret_cast<const float4*>(img + ((size_t)c * H + iy) * W + x4 * 4);
            T h[4];
            Elt<T>::st(h + 0, v.x); Elt<T>::st(h + 1, v.y); Elt<T>::st(h + 2, v.z); Elt<T>::st(h + 3, v.w);
            const unsigned short* hb = reinterpret_cast<const unsigned short*>(h);
            unsigned short* row = &s_in[c][r][1 + x4 * 4];
            row[0] = hb[0]; row[1] = hb[1]; row[2] = hb[2]; row[3] = hb[3];
        }
        // the tile's 128 rows of P: contiguous in memory (4 output rows x 32 columns of one image)
        const T* __restrict__ Pt = P + ((size_t)ib * 1024 + (size_t)oy0 * OW) * N;
        for (int e = tid; e < 128 * (N / 8); e += 256) {
            const int row = e / (N / 8), ch = e - row * (N / 8);
            *reinterpret_cast<uint4*>(&s_p[row * PP + ch * 8]) = *reinterpret_cast<const uint4*>(Pt + (size_t)row * N + ch * 8);
        }
        __syncthreads();
        // patch rows: pixel (ly, ox), column c*16 + ky*4 + kx <- image row 2 ly + ky of the staged ten, column 2 ox + kx
        for (int e = tid; e < 128 * kc; e += 256) {
            const int pix = e / kc, k = e - pix * kc;
            const int c = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
            s_x[pix * PX + k] = s_in[c][2 * (pix >> 5) + ky][2 * (pix & 31) + kx];
        }
        __syncthreads();
        // this wave's K step: the 32 pixels of output row oy0 + wave
        const int kb = wave * 32 + 8 * g;
        uint4 af[NI], bfr[4];
#pragma unroll
        for (int i = 0; i < NI; ++i) af[i] = tr2(s_p, PP, kb, i);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nj) bfr[j] = tr2(s_x, PX, kb, j);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nj) {
                    if constexpr (std::is_same<T, f16_t>::value)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, af[i]), __builtin_bit_cast(f16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                }
    }
    // acc[i][j][r] = S[n = i*16 + g*4 + r][k = j*16 + li] of this wave's pixels: sum the four waves (fixed order), one column tile at a time
    float* slab = p.slab + (size_t)blockIdx.x * N * kc;
    for (int j = 0; j < nj; ++j) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (jj == j) v = acc[i][jj][r];
                s_red[wave][i * 16 + g * 4 + r][li] = v;
            }
        __syncthreads();
        for (int e = tid; e < N * 16; e += 256) {
            const int n = e >> 4, k = e & 15;
            slab[(size_t)n * kc + j * 16 + k] = ((s_red[0][n][k] + s_red[1][n][k]) + s_red[2][n][k]) + s_red[3][n][k];
        }
    }
}

extern "C" int eg_wgrad_img_ok(int dtype, int C, int H, int W, int N, int k, int stride, int pad) {
    return dtype != EG_F32 && C >= 1 && C <= 4 && (N == 32 || N == 64 || N == 128) && k == 4 && stride == 2 && pad == 1 && H == 64 && W == 64;
}
/* workgroups (= slabs of N x 16 C floats) an eg_wgrad_img launch over `images` images (all tapes) with N output channels uses */
extern "C" int eg_wgrad_img_splits_n(int images, int N) {
    const long long tiles = (long long)images * 8, cap = N >= 128 ? 256 : 1024;      // (wide slabs: fewer of them to reduce)
    return (int)(tiles < cap ? tiles : cap);
}
extern "C" int eg_wgrad_img_splits(int images) { return eg_wgrad_img_splits_n(images, 32); }
extern "C" int eg_wgrad_img(int dtype, const float* img0, const float* img1, const float* img2, int ntapes, const void* P, float* slab, int B, int C,
                            int H, int W, int N, int* nsplit_out, eg_stream_t s) {
    EG_REQUIRE(img0 && P && slab && nsplit_out && ntapes >= 1 && ntapes <= 3 && B > 0, "eg_wgrad_img: bad argument");
    EG_REQUIRE((ntapes < 2 || img1) && (ntapes < 3 || img2), "eg_wgrad_img: one image pointer per tape");
    EG_REQUIRE(eg_wgrad_img_ok(dtype, C, H, W, N, 4, 2, 1), "eg_wgrad_img: 16-bit types, C <= 4, N = 32 / 64 / 128, 64 x 64 images, 4x4 / stride 2 / pad 1 only (use eg_im2col_img + eg_conv_wgrad)");
    ImgWgradParams p;
    memset(&p, 0, sizeof(p));
    p.img[0] = img0; p.img[1] = img1; p.img[2] = img2;
    p.P = P; p.slab = slab; p.B = B; p.C = C; p.ntiles = ntapes * B * 8;
    const int grid = eg_wgrad_img_splits_n(ntapes * B, N);
    hipStream_t st = (hipStream_t)s;
    if (N == 128) {
        if (dtype == EG_F16) hipLaunchKernelGGL((wgrad_img_kernel<f16_t, 128>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wgrad_img_kernel<bf16_t, 128>), dim3(grid), dim3(256), 0, st, p);
    } else if (N == 32) {
        if (dtype == EG_F16) hipLaunchKernelGGL((wgrad_img_kernel<f16_t, 32>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wgrad_img_kernel<bf16_t, 32>), dim3(grid), dim3(256), 0, st, p);
    } else {
        if (dtype == EG_F16) hipLaunchKernelGGL((wgrad_img_kernel<f16_t, 64>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((wgrad_img_kernel<bf16_t, 64>), dim3(grid), dim3(256), 0, st, p);
    }
    *nsplit_out = grid;
    EG_LAUNCH_CHECK();
    return 0;
}
