// Skinny ends of the networks, where the GEMM view has K or N of 1..48 and the work is HBM/LDS bound:
//   * image-side convolution (NCHW fp32 image with 1..4 channels  ->  NHWC dtype-T features) and its
//     weight gradient:  first Discriminator/Encoder conv (celebA/EAD-GAN_celebA.py:110, dSprites/rp.py:95,
//     MNIST/EAD-GAN_rpqmnxy.py:107,143) and the input-gradient of the Generator's last ConvTranspose (:90);
//   * small-N dense heads (celebA/EAD-GAN_celebA.py:122 1024x4x4 -> 19; MNIST/dSprites Linear heads).
// These use VALU FMAs with LDS-staged operands; MFMA tiles would idle >80 % of their lanes here.
#include "eg_common.h"

// ------------------------------------------------------------------------------------------------
// image-side convolution forward
// ------------------------------------------------------------------------------------------------
struct ImgConvP {
    const float* img;    // [B][CI][H][W]
    const float* w;      // master [N][CI][k][k] fp32
    void* out;           // [B][OH][OW][N] dtype T
    const float* bias;
    const float* sigma;
    const void* mask;    // same layout/dtype as out (activation output) or null
    int B, CI, H, W, N, k, stride, pad, OH, OW;
    int act; float slope; int mask_act; float mask_slope;
    int rows_per_block;
};

template <typename T, int CPT>
__global__ __launch_bounds__(256) void conv_img_fwd_kernel(const ImgConvP p) {
    extern __shared__ __attribute__((aligned(16))) float sm_f[];
    const int NT = p.CI * p.k * p.k;
    float* wl = sm_f;                       // [NT][N]
    float* il = sm_f + NT * p.N;            // [CI][k][WP] one strip of input rows
    const int WP = p.W + 2 * p.pad;
    const int tid = threadIdx.x;
    for (int i = tid; i < NT * p.N; i += 256) {
        const int n = i / NT, t = i % NT;
        wl[t * p.N + n] = p.w[i];
    }
    const int groups = 256 / p.OW;
    const int ox = tid % p.OW, cg = tid / p.OW;
    const int nb = cg * CPT;
    const int b = blockIdx.y;
    const float inv_sigma = p.sigma ? 1.f / p.sigma[0] : 1.f;
    const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    for (int ry = 0; ry < p.rows_per_block; ++ry) {
        const int oy = blockIdx.x * p.rows_per_block + ry;
        if (oy >= p.OH) break;
        __syncthreads();
        for (int i = tid; i < p.CI * p.k * WP; i += 256) {
            const int xx = i % WP, r = (i / WP) % p.k, ci = i / (WP * p.k);
            const int iy = oy * p.stride - p.pad + r, ix = xx - p.pad;
            float v = 0.f;
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = p.img[(((size_t)b * p.CI + ci) * p.H + iy) * p.W + ix];
            il[i] = v;
        }
        __syncthreads();
        if (cg < groups && nb < p.N) {
            float acc[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
            for (int ci = 0; ci < p.CI; ++ci)
                for (int kh = 0; kh < p.k; ++kh)
                    for (int kw = 0; kw < p.k; ++kw) {
                        const float xv = il[(ci * p.k + kh) * WP + ox * p.stride + kw];
                        const float* wr = wl + ((ci * p.k + kh) * p.k + kw) * p.N + nb;
#pragma unroll
                        for (int j = 0; j < CPT; ++j) acc[j] += xv * wr[j];
                    }
            const size_t o = (((size_t)b * p.OH + oy) * p.OW + ox) * p.N + nb;
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                float v = acc[j] * inv_sigma;
                if (p.bias) v += p.bias[nb + j];
                v = eg_act(v, p.act, p.slope);
                if (mask) v *= eg_act_grad_from_out(Elt<T>::ld(mask + o + j), p.mask_act, p.mask_slope);
                Elt<T>::st(out + o + j, v);
            }
        }
    }
}

/* image conv forward: out[B,OH,OW,N] = act(conv(img, w)/sigma + bias) [* act'(mask)] */
extern "C" int eg_conv_img_fwd(int dtype, const float* img, const float* w_master, void* out, int B, int CI, int H, int W, int N, int k,
                               int stride, int pad, const eg_epilogue* ep, eg_stream_t s) {
    EG_REQUIRE(img && w_master && out && CI >= 1 && CI <= 4, "eg_conv_img_fwd: bad argument");
    ImgConvP p;
    memset(&p, 0, sizeof(p));
    p.img = img; p.w = w_master; p.out = out;
    p.B = B; p.CI = CI; p.H = H; p.W = W; p.N = N; p.k = k; p.stride = stride; p.pad = pad;
    p.OH = (H + 2 * pad - k) / stride + 1; p.OW = (W + 2 * pad - k) / stride + 1;
    EG_REQUIRE(p.OW <= 256 && 256 % p.OW == 0, "eg_conv_img_fwd: OW must divide 256");
    if (ep) { p.bias = ep->bias; p.sigma = ep->sigma; p.act = ep->act; p.slope = ep->slope; p.mask = ep->mask; p.mask_act = ep->mask_act; p.mask_slope = ep->mask_slope; }
    const int groups = 256 / p.OW;
    EG_REQUIRE(N % groups == 0 || N < groups, "eg_conv_img_fwd: N=%d not divisible by %d thread groups", N, groups);
    const int cpt = N >= groups ? N / groups : 1;
    p.rows_per_block = p.OH >= 4 ? 4 : p.OH;
    dim3 grid(cdiv(p.OH, p.rows_per_block), B);
    const size_t lds = (size_t)(CI * k * k * N + CI * k * (W + 2 * pad)) * sizeof(float);
    hipStream_t st = (hipStream_t)s;
#define EG_CASE(T_, C_) hipLaunchKernelGGL((conv_img_fwd_kernel<T_, C_>), grid, dim3(256), lds, st, p)
#define EG_DISPATCH(C_) do { if (dtype == EG_F32) EG_CASE(float, C_); else if (dtype == EG_F16) EG_CASE(f16_t, C_); else EG_CASE(bf16_t, C_); } while (0)
    switch (cpt) {
        case 1: EG_DISPATCH(1); break;
        case 2: EG_DISPATCH(2); break;
        case 4: EG_DISPATCH(4); break;
        case 8: EG_DISPATCH(8); break;
        case 16: EG_DISPATCH(16); break;
        default: EG_FAIL(-1, "eg_conv_img_fwd: unsupported channels-per-thread %d", cpt);
    }
#undef EG_DISPATCH
#undef EG_CASE
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// image-side weight gradient: slab[b][n][ci][kh][kw] = sum_{oy,ox} dz[b,oy,ox,n] * img[b,ci,oy*s-p+kh,ox*s-p+kw]
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXT>
__global__ __launch_bounds__(256) void conv_img_wgrad_kernel(const T* __restrict__ dz, const float* __restrict__ img, float* __restrict__ slab,
                                                             int CI, int H, int W, int N, int k, int stride, int pad, int OH, int OW) {
    extern __shared__ __attribute__((aligned(16))) float il[];   // [CI][H+2p][W+2p]
    const int HP = H + 2 * pad, WP = W + 2 * pad;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < CI * HP * WP; i += 256) {
        const int xx = i % WP, yy = (i / WP) % HP, ci = i / (WP * HP);
        const int iy = yy - pad, ix = xx - pad;
        il[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[(((size_t)b * CI + ci) * H + iy) * W + ix] : 0.f;
    }
    __syncthreads();
    const int NT = CI * k * k;
    const int G = 256 / N;                 // tap groups (N <= 256, N divides 256)
    const int n = tid % N, tg = tid / N;
    int toff[MAXT];
    float acc[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        const int t = tg + j * G;
        acc[j] = 0.f;
        if (t < NT) {
            const int kw = t % k, kh = (t / k) % k, ci = t / (k * k);
            toff[j] = (ci * HP + kh) * WP + kw;
        } else
            toff[j] = -1;
    }
    if (tg < G) {
        const T* dzb = dz + (size_t)b * OH * OW * N + n;
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                const float g = Elt<T>::ld(dzb + (size_t)(oy * OW + ox) * N);
                const int base = oy * stride * WP + ox * stride;
#pragma unroll
                for (int j = 0; j < MAXT; ++j)
                    if (toff[j] >= 0) acc[j] += g * il[base + toff[j]];
            }
        float* o = slab + ((size_t)b * N + n) * NT;
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            const int t = tg + j * G;
            if (t < NT) o[t] = acc[j];
        }
    }
}

extern "C" size_t eg_conv_img_wgrad_ws_bytes(int B, int CI, int N, int k) { return (size_t)B * N * CI * k * k * sizeof(float); }

extern "C" int eg_conv_img_wgrad(int dtype, const void* dz, const float* img, float* slab, int B, int CI, int H, int W, int N, int k,
                                 int stride, int pad, eg_stream_t s) {
    EG_REQUIRE(dz && img && slab && N <= 256 && 256 % N == 0, "eg_conv_img_wgrad: N must divide 256");
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    const int NT = CI * k * k, G = 256 / N;
    const int maxt = cdiv(NT, G);
    const size_t lds = (size_t)CI * (H + 2 * pad) * (W + 2 * pad) * sizeof(float);
    EG_REQUIRE(lds <= 64 * 1024, "eg_conv_img_wgrad: image does not fit LDS");
    hipStream_t st = (hipStream_t)s;
#define EG_CASE(T_, M_) hipLaunchKernelGGL((conv_img_wgrad_kernel<T_, M_>), dim3(B), dim3(256), lds, st, (const T_*)dz, img, slab, CI, H, W, N, k, stride, pad, OH, OW)
#define EG_DISPATCH(M_) do { if (dtype == EG_F32) EG_CASE(float, M_); else if (dtype == EG_F16) EG_CASE(f16_t, M_); else EG_CASE(bf16_t, M_); } while (0)
    if (maxt <= 1) EG_DISPATCH(1);
    else if (maxt <= 2) EG_DISPATCH(2);
    else if (maxt <= 3) EG_DISPATCH(3);
    else if (maxt <= 6) EG_DISPATCH(6);
    else if (maxt <= 12) EG_DISPATCH(12);
    else if (maxt <= 24) EG_DISPATCH(24);
    else EG_FAIL(-1, "eg_conv_img_wgrad: too many taps per thread (%d)", maxt);
#undef EG_DISPATCH
#undef EG_CASE
    EG_LAUNCH_CHECK();
    return 0;
}

// flat slab reduction (master order already): out (+)= sum_z slab[z][i]; SN variant writes gtmp + <G,W> partials
template <bool SN>
__global__ void flat_reduce_kernel(const float* __restrict__ slab, int nslab, size_t total, float* __restrict__ out, int accumulate,
                                   const float* __restrict__ w_orig, float* __restrict__ partials) {
    __shared__ float sm[16];
    float dot = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int z = 0; z < nslab; ++z) a += slab[(size_t)z * total + i];
        if (SN) { out[i] = a; dot += a * w_orig[i]; }
        else out[i] = accumulate ? out[i] + a : a;
    }
    if (SN) {
        const float tot = block_sum(dot, sm);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

__global__ void sn_grad_apply_kernel2(const float* __restrict__ gtmp, const float* __restrict__ partials, int npart,
                                      const float* __restrict__ sigma, const float* __restrict__ u, const float* __restrict__ v,
                                      long long total, int Kdim, float* __restrict__ grad) {
    __shared__ float sm[16];
    float d = 0.f;
    for (int i = threadIdx.x; i < npart; i += blockDim.x) d += partials[i];
    const float dot = block_sum(d, sm);
    const float sg = sigma[0];
    const float inv = 1.f / sg, coef = dot / (sg * sg);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / Kdim), kk = (int)(i % Kdim);
        grad[i] += gtmp[i] * inv - coef * u[n] * v[kk];
    }
}

// out = g * act'(a) on NCHW fp32 tensors, plus partial[b*C+c] = sum_hw out  (bias gradient of the layer that produced a)
__global__ void act_grad_mul_rowsum_kernel(const float* __restrict__ g, const float* __restrict__ a, float* __restrict__ out, int HW, int act,
                                           float slope, float* __restrict__ partial) {
    __shared__ float sm[16];
    const size_t base = (size_t)blockIdx.x * HW;
    float acc = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float v = g[base + i] * eg_act_grad_from_out(a[base + i], act, slope);
        out[base + i] = v;
        acc += v;
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// gb[c] += sum_b partial[b][c]: 256 threads = Cp (C rounded up to a power of two) channels x 256/Cp batch lanes, combined through LDS in lane
// order (deterministic).  One thread per channel walking all B partials was a chain of B dependent loads (40 us at B = 512).
__global__ __launch_bounds__(256) void rowsum_final_kernel(const float* __restrict__ partial, int B, int C, float* __restrict__ gb) {
    __shared__ float cs[256];
    int Cp = 1;
    while (Cp < C) Cp <<= 1;
    const int L = 256 / Cp, c = threadIdx.x % Cp, l = threadIdx.x / Cp;
    float a = 0.f;
    if (c < C)
        for (int b = l; b < B; b += L) a += partial[b * C + c];
    cs[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int q = 0; q < L; ++q) t += cs[q * Cp + threadIdx.x];
        gb[threadIdx.x] += t;
    }
}
extern "C" int eg_act_grad_mul_bias_nchw(const float* g, const float* a, float* out, int B, int C, int HW, int act, float slope, float* partial,
                                         float* gb, eg_stream_t s) {
    EG_REQUIRE(g && a && out && partial && gb && C <= 64, "eg_act_grad_mul_bias_nchw: bad argument");
    hipLaunchKernelGGL(act_grad_mul_rowsum_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)s, g, a, out, HW, act, slope, partial);
    hipLaunchKernelGGL(rowsum_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, partial, B, C, gb);
    EG_LAUNCH_CHECK();
    return 0;
}

// per-channel sum of an NCHW fp32 tensor with few channels: gb[c] += sum_{b,hw} x[b][c][hw]
__global__ void bias_grad_nchw_kernel(const float* __restrict__ x, int B, int C, int HW, float* __restrict__ gb) {
    __shared__ float sm[16];
    const int c = blockIdx.x;
    float a = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* p = x + ((size_t)b * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) a += p[i];
    }
    const float tot = block_sum(a, sm);
    if (threadIdx.x == 0) gb[c] += tot;
}
extern "C" int eg_bias_grad_nchw(const float* x, int B, int C, int HW, float* gb, eg_stream_t s) {
    EG_REQUIRE(x && gb, "eg_bias_grad_nchw: null pointer");
    hipLaunchKernelGGL(bias_grad_nchw_kernel, dim3(C), dim3(1024), 0, (hipStream_t)s, x, B, C, HW, gb);
    EG_LAUNCH_CHECK();
    return 0;
}

/* grad (+)= sum_z slab[z]  over `total` floats in master order */
extern "C" int eg_flat_reduce(const float* slab, int nslab, size_t total, float* grad, int accumulate, eg_stream_t s) {
    EG_REQUIRE(slab && grad && nslab > 0, "eg_flat_reduce: bad argument");
    const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(flat_reduce_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)s, slab, nslab, total, grad, accumulate, (const float*)nullptr, (float*)nullptr);
    EG_LAUNCH_CHECK();
    return 0;
}

/* spectral-norm variant: rows = Cout, Kdim = elements per row; partials >= 1024 floats */
extern "C" int eg_flat_reduce_sn(const float* slab, int nslab, int rows, int Kdim, const float* w_orig, const float* sigma, const float* u,
                                 const float* v, float* gtmp, float* partials, float* grad, eg_stream_t s) {
    EG_REQUIRE(slab && w_orig && sigma && u && v && gtmp && partials && grad && nslab > 0, "eg_flat_reduce_sn: bad argument");
    const size_t total = (size_t)rows * Kdim;
    const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
    hipLaunchKernelGGL(flat_reduce_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)s, slab, nslab, total, gtmp, 0, w_orig, partials);
    hipLaunchKernelGGL(sn_grad_apply_kernel2, dim3(blocks), dim3(256), 0, (hipStream_t)s, gtmp, partials, blocks, sigma, u, v, (long long)total, Kdim, grad);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// im2col of a 1..4 channel NCHW fp32 image into K-contiguous patch rows [B*OH*OW][Kp] (dtype T), patch index
// t = (ci*k + kh)*k + kw == the master weight order, so the first/last image-side layers run on the MFMA kernels
// as 1x1 convolutions with K = CI*k*k (48 for CelebA) instead of scalar FMAs.
// ------------------------------------------------------------------------------------------------
// KK: filter size as a compile-time constant (4: CelebA / dSprites, 3: MNIST; 0: run-time k).  With a run-time k the ~24 integer divisions
// per thread made this copy ALU-bound (17.7 us for 23 MB); constants turn them into shifts and multiplies.
template <typename T, int KK>
__global__ void im2col_img_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int CI, int H, int W, int k_rt, int stride, int pad,
                                  int OH, int OW, int K, int Kp) {
    constexpr int VEC = Elt<T>::VEC;
    const int k = KK ? KK : k_rt;
    const unsigned cpr = Kp / VEC;
    const unsigned total = (unsigned)B * OH * OW * cpr;                 // < 2^31 (checked by the launcher)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned j = i % cpr, m = i / cpr;
        const unsigned ox = m % OW, oyb = m / OW;
        const unsigned oy = oyb % OH, b = oyb / OH;
        const int iy0 = (int)oy * stride - pad, ix0 = (int)ox * stride - pad;
        const float* base = img + (size_t)b * CI * H * W;
        uint4 v;
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int t = (int)j * VEC + q;
            float f = 0.f;
            if (t < K) {
                const int kw = t % k, kh = (t / k) % k, ci = t / (k * k);
                const int iy = iy0 + kh, ix = ix0 + kw;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) f = base[((size_t)ci * H + iy) * W + ix];
            }
            Elt<T>::st(e + q, f);
        }
        *reinterpret_cast<uint4*>(out + (size_t)i * VEC) = v;
    }
}

template <typename T>
static void launch_im2col_img(const float* img, T* out, int B, int CI, int H, int W, int k, int stride, int pad, int OH, int OW, int Kp, int blocks,
                              hipStream_t st) {
    if (k == 4) hipLaunchKernelGGL((im2col_img_kernel<T, 4>), dim3(blocks), dim3(256), 0, st, img, out, B, CI, H, W, k, stride, pad, OH, OW, CI * k * k, Kp);
    else if (k == 3) hipLaunchKernelGGL((im2col_img_kernel<T, 3>), dim3(blocks), dim3(256), 0, st, img, out, B, CI, H, W, k, stride, pad, OH, OW, CI * k * k, Kp);
    else hipLaunchKernelGGL((im2col_img_kernel<T, 0>), dim3(blocks), dim3(256), 0, st, img, out, B, CI, H, W, k, stride, pad, OH, OW, CI * k * k, Kp);
}

extern "C" int eg_im2col_img(int dtype, const float* img, void* out, int B, int CI, int H, int W, int k, int stride, int pad, int Kp,
                             eg_stream_t s) {
    EG_REQUIRE(img && out && B > 0 && CI > 0 && k > 0 && stride > 0 && Kp >= CI * k * k && Kp % (dtype == EG_F32 ? 4 : 8) == 0, "eg_im2col_img: bad argument");
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    EG_REQUIRE(OH > 0 && OW > 0, "eg_im2col_img: empty output");
    const size_t total = (size_t)B * OH * OW * (Kp / (dtype == EG_F32 ? 4 : 8));
    EG_REQUIRE(total < 0x7fffffffull, "eg_im2col_img: more than 2^31 output vectors");
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    if (dtype == EG_F32) launch_im2col_img<float>(img, (float*)out, B, CI, H, W, k, stride, pad, OH, OW, Kp, blocks, (hipStream_t)s);
    else if (dtype == EG_F16) launch_im2col_img<f16_t>(img, (f16_t*)out, B, CI, H, W, k, stride, pad, OH, OW, Kp, blocks, (hipStream_t)s);
    else launch_im2col_img<bf16_t>(img, (bf16_t*)out, B, CI, H, W, k, stride, pad, OH, OW, Kp, blocks, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

// dst[r][0:n] = cast(src[r][0:n]), dst[r][n:npad] = 0   (fp32 head gradients -> MFMA operand)
template <typename T>
__global__ void cast_pad_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int n, int npad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * npad) return;
    const int r = i / npad, c = i % npad;
    Elt<T>::st(dst + i, c < n ? src[(size_t)r * n + c] : 0.f);
}
extern "C" int eg_cast_pad(int dtype, const float* src, void* dst, int rows, int n, int npad, eg_stream_t s) {
    EG_REQUIRE(src && dst && npad >= n, "eg_cast_pad: bad argument");
    if (dtype == EG_F32) hipLaunchKernelGGL(cast_pad_kernel<float>, dim3(cdiv(rows * npad, 256)), dim3(256), 0, (hipStream_t)s, src, (float*)dst, rows, n, npad);
    else if (dtype == EG_F16) hipLaunchKernelGGL(cast_pad_kernel<f16_t>, dim3(cdiv(rows * npad, 256)), dim3(256), 0, (hipStream_t)s, src, (f16_t*)dst, rows, n, npad);
    else hipLaunchKernelGGL(cast_pad_kernel<bf16_t>, dim3(cdiv(rows * npad, 256)), dim3(256), 0, (hipStream_t)s, src, (bf16_t*)dst, rows, n, npad);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// small-N dense head:  y[b][n] = sum_k x[b][k] * Wp[n][k] (+ bias[n]);  x dtype T [B][K], Wp dtype T [N][Kpad]
// ------------------------------------------------------------------------------------------------
#define EG_DS_ROWS 4   // batch rows per workgroup: the weight panel (N x K, e.g. 19 x 16384) is streamed once per 4 rows
template <typename T>
__global__ __launch_bounds__(256) void dense_small_fwd_kernel(const T* __restrict__ x, const T* __restrict__ wp, const float* __restrict__ bias,
                                                              float* __restrict__ y, int B, int K, int Kpad, int N, const float* __restrict__ sigma,
                                                              int sigma_rows, float* __restrict__ partials, int kper) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int R = EG_DS_ROWS;
    __shared__ float part[4][R][64];
    const int b0 = blockIdx.x * R, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blockIdx.y = K slice (kper elements): with `partials` the slices' sums go to partials[slice][b][n] and dense_small_combine_kernel
    // adds them in slice order -- a head over B*16384 inputs has only B/4 row groups, far too few workgroups without the K split
    const int kbeg = blockIdx.y * kper, kend = min(K, kbeg + kper);
    for (int n0 = 0; n0 < N; n0 += 8) {          // 8 outputs x R rows per sweep over x
        float a[R][8];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 8; ++q) a[r][q] = 0.f;
        for (int k0 = kbeg + threadIdx.x * VEC; k0 < kend; k0 += 256 * VEC) {
            // every request of this step (R rows of x, 8 weight rows) is issued before the first value is used; the weight row index is
            // clamped instead of branching around the load.  Written the obvious way the compiler waited for each 16-byte load in turn
            // (vmcnt(0) after every row) and the kernel was a chain of ~12 dependent round trips per workgroup.
            uint4 xv[R], wv[8];
#pragma unroll
            for (int r = 0; r < R; ++r) xv[r] = *reinterpret_cast<const uint4*>(x + (size_t)min(b0 + r, B - 1) * K + k0);
#pragma unroll
            for (int q = 0; q < 8; ++q) wv[q] = *reinterpret_cast<const uint4*>(wp + (size_t)min(n0 + q, N - 1) * Kpad + k0);
            float xf[R][VEC];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const T* xe = reinterpret_cast<const T*>(&xv[r]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) xf[r][j] = Elt<T>::ld(xe + j);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const T* we = reinterpret_cast<const T*>(&wv[q]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float wf = Elt<T>::ld(we + j);
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r][q] += xf[r][j] * wf;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float w = wave_sum(a[r][q]);
                if (lane == 0) part[wave][r][n0 + q < 64 ? n0 + q : 63] = w;
            }
    }
    __syncthreads();
    const int r = threadIdx.x >> 6, n = threadIdx.x & 63;
    if (r < R && n < N && b0 + r < B) {
        const int b = b0 + r;
        const float tot = part[0][r][n] + part[1][r][n] + part[2][r][n] + part[3][r][n];
        if (partials) {
            partials[((size_t)blockIdx.y * B + b) * N + n] = tot;
        } else {
            const float inv = sigma ? 1.f / sigma[sigma_rows ? b / sigma_rows : 0] : 1.f;
            y[(size_t)b * N + n] = tot * inv + (bias ? bias[n] : 0.f);
        }
    }
}

__global__ void dense_small_combine_kernel(const float* __restrict__ partials, int nslice, const float* __restrict__ bias, float* __restrict__ y, int B,
                                           int N, const float* __restrict__ sigma, int sigma_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i - b * N;
    float tot = 0.f;
    for (int z = 0; z < nslice; ++z) tot += partials[(size_t)z * B * N + i];
    const float inv = sigma ? 1.f / sigma[sigma_rows ? b / sigma_rows : 0] : 1.f;
    y[i] = tot * inv + (bias ? bias[n] : 0.f);
}

template <typename T>
static void launch_dense_small_fwd(const T* x, const T* wp, const float* bias, float* y, int B, int K, int Kpad, int N, const float* sigma, int sigma_rows,
                                   float* ws, size_t ws_floats, hipStream_t st) {
    constexpr int VEC = Elt<T>::VEC;
    const int groups = cdiv(B, EG_DS_ROWS);
    int ns = 1;
    if (ws) {
        while (ns < 16 && groups * ns < 256 && K / (ns * 2) >= 256 * VEC) ns *= 2;
        while (ns > 1 && (size_t)ns * B * N > ws_floats) ns /= 2;
    }
    if (ns == 1) {
        hipLaunchKernelGGL(dense_small_fwd_kernel<T>, dim3(groups), dim3(256), 0, st, x, wp, bias, y, B, K, Kpad, N, sigma, sigma_rows, (float*)nullptr, K);
        return;
    }
    const int kper = cdiv(cdiv(K, VEC), ns) * VEC;
    hipLaunchKernelGGL(dense_small_fwd_kernel<T>, dim3(groups, ns), dim3(256), 0, st, x, wp, bias, y, B, K, Kpad, N, sigma, sigma_rows, ws, kper);
    hipLaunchKernelGGL(dense_small_combine_kernel, dim3(cdiv(B * N, 256)), dim3(256), 0, st, ws, ns, bias, y, B, N, sigma, sigma_rows);
}

template <typename T>
static int launch_dense_small_slices(const T* x, const T* wp, int B, int K, int Kpad, int N, float* ws, size_t ws_floats, hipStream_t st) {
    constexpr int VEC = Elt<T>::VEC;
    const int groups = cdiv(B, EG_DS_ROWS);
    int ns = 1;                                         // the slice count of launch_dense_small_fwd: the same sums
    while (ns < 16 && groups * ns < 256 && K / (ns * 2) >= 256 * VEC) ns *= 2;
    while (ns > 1 && (size_t)ns * B * N > ws_floats) ns /= 2;
    const int kper = cdiv(cdiv(K, VEC), ns) * VEC;
    hipLaunchKernelGGL(dense_small_fwd_kernel<T>, dim3(groups, ns), dim3(256), 0, st, x, wp, (const float*)nullptr, (float*)nullptr, B, K, Kpad, N,
                       (const float*)nullptr, 0, ws, ns == 1 ? K : kper);
    return ns;
}

extern "C" int eg_dense_small_fwd_slices(int dtype, const void* x, const void* wp, int B, int K, int Kpad, int N, float* partials, size_t ws_floats,
                                         int* nslice_out, eg_stream_t s) {
    EG_REQUIRE(x && wp && partials && nslice_out && N <= 64 && K % (dtype == EG_F32 ? 4 : 8) == 0 && ws_floats >= (size_t)B * N,
               "eg_dense_small_fwd_slices: bad argument (N<=64)");
    if (dtype == EG_F32) *nslice_out = launch_dense_small_slices<float>((const float*)x, (const float*)wp, B, K, Kpad, N, partials, ws_floats, (hipStream_t)s);
    else if (dtype == EG_F16) *nslice_out = launch_dense_small_slices<f16_t>((const f16_t*)x, (const f16_t*)wp, B, K, Kpad, N, partials, ws_floats, (hipStream_t)s);
    else *nslice_out = launch_dense_small_slices<bf16_t>((const bf16_t*)x, (const bf16_t*)wp, B, K, Kpad, N, partials, ws_floats, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

// dx[b][k] = (sum_n dy[b][n] * Wp[n][k]) * act'(mask[b][k]);  EG_DS_ROWS batch rows per workgroup share every weight vector (one row per
// workgroup re-read the whole N x K panel from L2 per row: 240 MB for 384 rows of the 19 x 16384 head)
template <typename T>
__global__ __launch_bounds__(256) void dense_small_bwd_kernel(const float* __restrict__ dy, const T* __restrict__ wp, const T* __restrict__ mask,
                                                              T* __restrict__ dx, int B, int K, int Kpad, int N, int mask_act, float mask_slope,
                                                              const float* __restrict__ sigma, int sigma_rows) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int R = EG_DS_ROWS;
    __shared__ float dl[R][64];
    const int b0 = blockIdx.x * R;
    for (int i = threadIdx.x; i < R * 64; i += 256) {
        const int r = i >> 6, n = i & 63;
        dl[r][n] = (n < N && b0 + r < B) ? dy[(size_t)(b0 + r) * N + n] : 0.f;
    }
    __syncthreads();
    float post[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int b = min(b0 + r, B - 1);
        post[r] = sigma ? 1.f / sigma[sigma_rows ? b / sigma_rows : 0] : 1.f;
    }
    for (int k0 = (blockIdx.y * 256 + threadIdx.x) * VEC; k0 < K; k0 += gridDim.y * 256 * VEC) {
        float a[R][VEC];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[r][j] = 0.f;
        uint4 mv[R];
        if (mask) {
#pragma unroll
            for (int r = 0; r < R; ++r) mv[r] = *reinterpret_cast<const uint4*>(mask + (size_t)min(b0 + r, B - 1) * K + k0);
        }
        for (int n0 = 0; n0 < N; n0 += 8) {            // 8 weight rows in flight at a time; summation order n = 0..N-1 as before
            uint4 wv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) wv[q] = *reinterpret_cast<const uint4*>(wp + (size_t)min(n0 + q, N - 1) * Kpad + k0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (n0 + q < N) {
                    const T* we = reinterpret_cast<const T*>(&wv[q]);
                    float wf[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) wf[j] = Elt<T>::ld(we + j);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float g = dl[r][n0 + q];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) a[r][j] += g * wf[j];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (b0 + r >= B) break;
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
            if (mask) {
                const T* me = reinterpret_cast<const T*>(&mv[r]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) a[r][j] *= eg_act_grad_from_out(Elt<T>::ld(me + j), mask_act, mask_slope);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) Elt<T>::st(oe + j, a[r][j] * post[r]);
            *reinterpret_cast<uint4*>(dx + (size_t)(b0 + r) * K + k0) = ov;
        }
    }
}

// gw_master[(n*Cin + ci)*T + t] += sum_b dy[b][n] * x[b][k],  k = t*Cin + ci
template <typename T>
__global__ __launch_bounds__(256) void dense_small_wgrad_kernel(const float* __restrict__ dy, const T* __restrict__ x, float* __restrict__ gw,
                                                                int B, int K, int N, int Cin, int Ttaps) {
    extern __shared__ float dl[];   // [B][N]
    for (int i = threadIdx.x; i < B * N; i += 256) dl[i] = dy[i];
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float acc[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) acc[n] = 0.f;
    for (int b = 0; b < B; ++b) {
        const float xv = Elt<T>::ld(x + (size_t)b * K + k);
#pragma unroll
        for (int n = 0; n < 32; ++n)
            if (n < N) acc[n] += xv * dl[b * N + n];
    }
    const int t = k / Cin, ci = k % Cin;
#pragma unroll
    for (int n = 0; n < 32; ++n)
        if (n < N) gw[((size_t)n * Cin + ci) * Ttaps + t] += acc[n];
}

extern "C" int eg_dense_small_fwd(int dtype, const void* x, const void* wp, const float* bias, float* y, int B, int K, int Kpad, int N,
                                  float* ws, size_t ws_floats, eg_stream_t s) {
    EG_REQUIRE(x && wp && y && N <= 64 && K % (dtype == EG_F32 ? 4 : 8) == 0, "eg_dense_small_fwd: bad argument (N<=64)");
    if (dtype == EG_F32) launch_dense_small_fwd<float>((const float*)x, (const float*)wp, bias, y, B, K, Kpad, N, nullptr, 0, ws, ws_floats, (hipStream_t)s);
    else if (dtype == EG_F16) launch_dense_small_fwd<f16_t>((const f16_t*)x, (const f16_t*)wp, bias, y, B, K, Kpad, N, nullptr, 0, ws, ws_floats, (hipStream_t)s);
    else launch_dense_small_fwd<bf16_t>((const bf16_t*)x, (const bf16_t*)wp, bias, y, B, K, Kpad, N, nullptr, 0, ws, ws_floats, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_dense_small_fwd_sn(int dtype, const void* x, const void* wp, const float* bias, float* y, int B, int K, int Kpad, int N,
                                     const float* sigma, int sigma_rows, float* ws, size_t ws_floats, eg_stream_t s) {
    EG_REQUIRE(x && wp && y && sigma && N <= 64 && K % (dtype == EG_F32 ? 4 : 8) == 0, "eg_dense_small_fwd_sn: bad argument (N<=64)");
    if (dtype == EG_F32) launch_dense_small_fwd<float>((const float*)x, (const float*)wp, bias, y, B, K, Kpad, N, sigma, sigma_rows, ws, ws_floats, (hipStream_t)s);
    else if (dtype == EG_F16) launch_dense_small_fwd<f16_t>((const f16_t*)x, (const f16_t*)wp, bias, y, B, K, Kpad, N, sigma, sigma_rows, ws, ws_floats, (hipStream_t)s);
    else launch_dense_small_fwd<bf16_t>((const bf16_t*)x, (const bf16_t*)wp, bias, y, B, K, Kpad, N, sigma, sigma_rows, ws, ws_floats, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

// single block: dys = dy / sigma[tape] (cast, written at column col0 of a [rows][npad] buffer), coef[tape] = sum dys*(y-bias),
// gb[n] += sum_rows dy[row][n].  Head tensors are tiny (rows <= a few thousand, N <= 64).
template <typename T>
__global__ void head_prep_sn_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ y, int ldyy, const float* __restrict__ bias,
                                    int rows, int N, const float* __restrict__ sigma, int rows_per_tape, T* __restrict__ dys, int npad, int col0,
                                    float* __restrict__ gb, float* __restrict__ coef, float* __restrict__ dys32, int ld32) {
    __shared__ float sm[16];
    const int ntapes = rows / rows_per_tape;
    for (int t = 0; t < ntapes; ++t) {
        const float inv = sigma ? 1.f / sigma[t] : 1.f;
        float dot = 0.f;
        for (int i = threadIdx.x; i < rows_per_tape * N; i += blockDim.x) {
            const int r = t * rows_per_tape + i / N, n = i % N;
            const float gs = dy[(size_t)r * ldy + n] * inv;
            Elt<T>::st(dys + (size_t)r * npad + col0 + n, gs);
            if (dys32) dys32[(size_t)r * ld32 + col0 + n] = gs;
            dot += gs * (y[(size_t)r * ldyy + n] - bias[n]);
        }
        const float tot = block_sum(dot, sm);
        if (threadIdx.x == 0 && coef) coef[t] = tot;
    }
    if (gb) {
        // column sums with all 256 threads: Np = N rounded up to a power of two columns x 256/Np row lanes, combined through LDS in lane
        // order (deterministic).  N threads walking all rows one dependent load at a time took 60 us for 1536 rows.
        __shared__ float cs[256];
        int Np = 1;
        while (Np < N) Np <<= 1;
        const int L = 256 / Np, n = threadIdx.x % Np, l = threadIdx.x / Np;
        float a = 0.f;
        if (n < N)
            for (int r = l; r < rows; r += L) a += dy[(size_t)r * ldy + n];
        cs[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < N) {
            float t = 0.f;
            for (int q = 0; q < L; ++q) t += cs[q * Np + threadIdx.x];
            gb[threadIdx.x] += t;
        }
    }
}

extern "C" int eg_head_prep_sn(int dtype, const float* dy, int ldy, const float* y, int ldyy, const float* bias, int rows, int N,
                               const float* sigma, int rows_per_tape, void* dys, int npad, int col0, float* gb, float* coef, float* dys32, int ld32,
                               eg_stream_t s) {
    EG_REQUIRE(dy && y && bias && dys && rows_per_tape > 0 && rows % rows_per_tape == 0 && N <= 64 && col0 + N <= npad, "eg_head_prep_sn: bad argument");
    if (dtype == EG_F32) hipLaunchKernelGGL(head_prep_sn_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)s, dy, ldy, y, ldyy, bias, rows, N, sigma, rows_per_tape, (float*)dys, npad, col0, gb, coef, dys32, ld32);
    else if (dtype == EG_F16) hipLaunchKernelGGL(head_prep_sn_kernel<f16_t>, dim3(1), dim3(256), 0, (hipStream_t)s, dy, ldy, y, ldyy, bias, rows, N, sigma, rows_per_tape, (f16_t*)dys, npad, col0, gb, coef, dys32, ld32);
    else hipLaunchKernelGGL(head_prep_sn_kernel<bf16_t>, dim3(1), dim3(256), 0, (hipStream_t)s, dy, ldy, y, ldyy, bias, rows, N, sigma, rows_per_tape, (bf16_t*)dys, npad, col0, gb, coef, dys32, ld32);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_dense_small_bwd(int dtype, const float* dy, const void* wp, const void* mask, void* dx, int B, int K, int Kpad, int N,
                                  int mask_act, float mask_slope, const float* sigma, int sigma_rows, eg_stream_t s) {
    EG_REQUIRE(dy && wp && dx && B > 0 && N > 0 && N <= 64 && K > 0 && K % (dtype == EG_F32 ? 4 : 8) == 0, "eg_dense_small_bwd: bad argument");
    const int vec = dtype == EG_F32 ? 4 : 8;
    const int groups = cdiv(B, EG_DS_ROWS), kblocks = cdiv(K, 256 * vec);
    int gy = 1;                                       // enough workgroups for 256 CUs: split the K sweep (every output has one owner)
    while (gy * 2 <= kblocks && groups * gy < 512) gy *= 2;
    const dim3 grid(groups, gy);
    if (dtype == EG_F32) hipLaunchKernelGGL(dense_small_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, dy, (const float*)wp, (const float*)mask, (float*)dx, B, K, Kpad, N, mask_act, mask_slope, sigma, sigma_rows);
    else if (dtype == EG_F16) hipLaunchKernelGGL(dense_small_bwd_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)s, dy, (const f16_t*)wp, (const f16_t*)mask, (f16_t*)dx, B, K, Kpad, N, mask_act, mask_slope, sigma, sigma_rows);
    else hipLaunchKernelGGL(dense_small_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, dy, (const bf16_t*)wp, (const bf16_t*)mask, (bf16_t*)dx, B, K, Kpad, N, mask_act, mask_slope, sigma, sigma_rows);
    EG_LAUNCH_CHECK();
    return 0;
}

/* gw (master [N][Cin][taps], accumulate) and gb[n] += sum_b dy[b][n] */
__global__ __launch_bounds__(512) void dense_small_bgrad_kernel(const float* dy, float* gb, int B, int N) {
    // 8 row groups x 64 columns: independent loads in flight instead of one dependent chain per column; fixed-order combine
    __shared__ float part[8][64];
    const int n = threadIdx.x & 63, g = threadIdx.x >> 6;
    float a = 0.f;
    if (n < N)
        for (int b = g; b < B; b += 8) a += dy[(size_t)b * N + n];
    part[g][n] = a;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][n];
        gb[n] += t;
    }
}

extern "C" int eg_dense_small_bgrad(const float* dy, float* gb, int B, int N, eg_stream_t s) {
    EG_REQUIRE(dy && gb && N <= 64, "eg_dense_small_bgrad: bad argument");
    hipLaunchKernelGGL(dense_small_bgrad_kernel, dim3(1), dim3(512), 0, (hipStream_t)s, dy, gb, B, N);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_dense_small_wgrad(int dtype, const float* dy, const void* x, float* gw, float* gb, int B, int K, int N, int Cin, int taps,
                                    eg_stream_t s) {
    EG_REQUIRE(dy && x && gw && N <= 32 && (size_t)B * N * 4 <= 64 * 1024 && Cin * taps == K, "eg_dense_small_wgrad: bad argument (N<=32)");
    const size_t lds = (size_t)B * N * sizeof(float);
    if (dtype == EG_F32) hipLaunchKernelGGL(dense_small_wgrad_kernel<float>, dim3(cdiv(K, 256)), dim3(256), lds, (hipStream_t)s, dy, (const float*)x, gw, B, K, N, Cin, taps);
    else if (dtype == EG_F16) hipLaunchKernelGGL(dense_small_wgrad_kernel<f16_t>, dim3(cdiv(K, 256)), dim3(256), lds, (hipStream_t)s, dy, (const f16_t*)x, gw, B, K, N, Cin, taps);
    else hipLaunchKernelGGL(dense_small_wgrad_kernel<bf16_t>, dim3(cdiv(K, 256)), dim3(256), lds, (hipStream_t)s, dy, (const bf16_t*)x, gw, B, K, N, Cin, taps);
    if (gb) hipLaunchKernelGGL(dense_small_bgrad_kernel, dim3(1), dim3(512), 0, (hipStream_t)s, dy, gb, B, N);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// col2im of a transposed convolution with few output channels (ConvTranspose2d(128 -> 3, 4, 2, 1): the generator's last layer,
// celebA/EAD-GAN_celebA.py:90-91, and the backward-to-image of the discriminator's first Conv2d(3 -> 128, 4, 2, 1), :110).
// The layer runs as ONE GEMM over the input pixels with N = k*k*C columns, cols[m][t*C + c] = sum_ci x[m][ci] W[ci][c][t] (every
// activation read once; as 4 phases x 4 taps of an implicit GEMM each one was fetched 16 times through L2), then this gather:
//   out[b][c][oy][ox] = act(bias[c] + sum over taps (kh,kw) with oy = iy*stride - pad + kh, ox = ix*stride - pad + kw of cols[(b,iy,ix)][t*C + c])
// One thread per output pixel; every cols element is read exactly once; out is fp32 NCHW (the image / image gradient).
// ------------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ __launch_bounds__(256) void col2im_img_kernel(const T* __restrict__ cols, int B, int Hin, int Win, int k, int stride, int pad, int OH, int OW,
                                                         const float* __restrict__ bias, int act, float slope, float* __restrict__ out) {
    const unsigned total = (unsigned)B * OH * OW;
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned ox = i % OW, r = i / OW;
    const unsigned oy = r % OH, b = r / OH;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = bias ? bias[c] : 0.f;
    const int ncol = k * k * C;
    // taps of one axis that reach output coordinate o: kh = (o + pad) % stride + j * stride, input coordinate (o + pad - kh) / stride
    for (int kh = (int)(oy + pad) % stride; kh < k; kh += stride) {
        const int iy = ((int)oy + pad - kh) / stride;
        if (iy < 0 || iy >= Hin) continue;
        for (int kw = (int)(ox + pad) % stride; kw < k; kw += stride) {
            const int ix = ((int)ox + pad - kw) / stride;
            if (ix < 0 || ix >= Win) continue;
            const T* src = cols + ((size_t)(b * Hin + iy) * Win + ix) * ncol + (kh * k + kw) * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += Elt<T>::ld(src + c);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[(((size_t)b * C + c) * OH + oy) * OW + ox] = eg_act(acc[c], act, slope);
}

template <typename T>
static int launch_col2im_img(const T* cols, int B, int C, int Hin, int Win, int k, int stride, int pad, int OH, int OW, const float* bias, int act,
                             float slope, float* out, hipStream_t st) {
    const unsigned blocks = (unsigned)(((size_t)B * OH * OW + 255) / 256);
    if (C == 1) hipLaunchKernelGGL((col2im_img_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, cols, B, Hin, Win, k, stride, pad, OH, OW, bias, act, slope, out);
    else if (C == 3) hipLaunchKernelGGL((col2im_img_kernel<T, 3>), dim3(blocks), dim3(256), 0, st, cols, B, Hin, Win, k, stride, pad, OH, OW, bias, act, slope, out);
    else return -1;
    return 0;
}

extern "C" int eg_col2im_img(int dtype, const void* cols, int B, int C, int Hin, int Win, int k, int stride, int pad, const float* bias, int act,
                             float slope, float* out, eg_stream_t s) {
    EG_REQUIRE(cols && out && B > 0 && Hin > 0 && Win > 0 && k > 0 && stride > 0 && pad >= 0, "eg_col2im_img: bad argument");
    EG_REQUIRE(C == 1 || C == 3, "eg_col2im_img: image channels must be 1 or 3");
    EG_REQUIRE(dtype == EG_F32 || dtype == EG_BF16 || dtype == EG_F16, "dtype must be EG_F32, EG_BF16 or EG_F16");
    const int OH = (Hin - 1) * stride - 2 * pad + k, OW = (Win - 1) * stride - 2 * pad + k;
    EG_REQUIRE(OH > 0 && OW > 0 && (size_t)B * OH * OW < 0x7fffffffull, "eg_col2im_img: bad output size");
    int rc;
    if (dtype == EG_F32) rc = launch_col2im_img<float>((const float*)cols, B, C, Hin, Win, k, stride, pad, OH, OW, bias, act, slope, out, (hipStream_t)s);
    else if (dtype == EG_F16) rc = launch_col2im_img<f16_t>((const f16_t*)cols, B, C, Hin, Win, k, stride, pad, OH, OW, bias, act, slope, out, (hipStream_t)s);
    else rc = launch_col2im_img<bf16_t>((const bf16_t*)cols, B, C, Hin, Win, k, stride, pad, OH, OW, bias, act, slope, out, (hipStream_t)s);
    EG_REQUIRE(rc == 0, "eg_col2im_img: unsupported channel count");
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 / stride 1 / pad 1 convolution onto ONE output channel from 64 input channels -- the MNIST generator's last
// layer, Conv2d(64, 1, 3, 1, 1) (MNIST/EAD-GAN_rpqmnxy.py:88):   S[t][c] = sum_{b,y,x} dy[b][y][x] * X[b][y + ty - 1][x + tx - 1][c]
// As a GEMM this is an N = 1 (padded to 8) x K = 9 * 64 output over M = B * H * W rows: the per-tap kernel streamed the 67 MB activation nine
// times for 0.3 GFLOP (152 us at B = 256).  Here lane = input channel: a wave walks image rows, every pixel's 64 channels are ONE coalesced
// load, and the nine products go to nine per-lane accumulators with dy broadcast from a zero-haloed LDS copy of the tile's gradient
// rows -- the activation is read once.  dy passes through T like the padded copy the GEMM path multiplied (same operands, fp32 sums).
// Slab layout of the per-tap kernel with N = 1: slab[split][t][c]; eg_wgrad_reduce(slab, nsplit, 1, 1, 64, 9, grad) finishes it.
// ------------------------------------------------------------------------------------------------
#define EG_C1_ROWS 8            // image rows per work unit (two per wave)
template <typename T>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const T* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int B, int H, int W,
                                                       int units) {
    constexpr int C = 64, RB = EG_C1_ROWS;
    extern __shared__ float s_c1[];                      // [RB + 2][W + 2] gradient rows with a zero halo, then [4][9][C] wave sums
    float* s_dy = s_c1;
    float* s_acc = s_c1 + (RB + 2) * (W + 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ub = H / RB;                               // units per image
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int b = u / ub, y0 = (u - b * ub) * RB;
        __syncthreads();                                 // the previous unit's rows have been read
        for (int i = tid; i < (RB + 2) * (W + 2); i += 256) {
            const int r = i / (W + 2), cx = i - r * (W + 2);
            const int y = y0 - 1 + r, xx = cx - 1;
            float v = 0.f;
            if (y >= 0 && y < H && xx >= 0 && xx < W) {
                T h;
                Elt<T>::st(&h, dy[((size_t)b * H + y) * W + xx]);
                v = Elt<T>::ld(&h);
            }
            s_dy[i] = v;
        }
        __syncthreads();
        // input pixel (iy, ix) meets output pixel (iy - ty + 1, ix - tx + 1) under tap (ty, tx): s_dy row (iy - y0) + 2 - ty, column ix + 2 - tx
        for (int rr = 0; rr < RB / 4; ++rr) {
            const int ly = wave * (RB / 4) + rr, iy = y0 + ly;
            const T* __restrict__ xr = x + (((size_t)b * H + iy) * W) * C + lane;
            for (int ix0 = 0; ix0 < W; ix0 += 8) {
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = Elt<T>::ld(xr + (size_t)(ix0 + j) * C);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ix = ix0 + j;
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = fmaf(xv[j], s_dy[(ly + 2 - ty) * (W + 2) + ix + 2 - tx], acc[ty * 3 + tx]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) s_acc[(wave * 9 + t) * C + lane] = acc[t];
    __syncthreads();
    for (int i = tid; i < 9 * C; i += 256)
        slab[(size_t)blockIdx.x * 9 * C + i] = ((s_acc[i] + s_acc[9 * C + i]) + s_acc[2 * 9 * C + i]) + s_acc[3 * 9 * C + i];
}

extern "C" int eg_wgrad_c1_ok(int dtype, int C, int H, int W, int Cout, int k, int stride, int pad) {
    (void)dtype;
    return C == 64 && Cout == 1 && k == 3 && stride == 1 && pad == 1 && H >= EG_C1_ROWS && (H % EG_C1_ROWS) == 0 && W >= 8 && (W % 8) == 0 && W <= 128;
}
/* workgroups (= slabs) eg_wgrad_c1 launches for B images: the caller's slab holds that many x 9 x 64 floats */
extern "C" int eg_wgrad_c1_splits(int B, int H) {
    const long long units = (long long)B * (H / EG_C1_ROWS);
    return (int)(units < 1024 ? units : 1024);
}
extern "C" int eg_wgrad_c1(int dtype, const void* x, const float* dy, float* slab, int B, int H, int W, int C, int* nsplit_out, eg_stream_t s) {
    EG_REQUIRE(x && dy && slab && nsplit_out && B > 0, "eg_wgrad_c1: bad argument");
    EG_REQUIRE(eg_wgrad_c1_ok(dtype, C, H, W, 1, 3, 1, 1), "eg_wgrad_c1: 64 input channels, one output channel, 3x3 / stride 1 / pad 1, H %% 8 == 0, W %% 8 == 0 only");
    EG_REQUIRE(dtype == EG_F32 || dtype == EG_BF16 || dtype == EG_F16, "dtype must be EG_F32, EG_BF16 or EG_F16");
    const int units = B * (H / EG_C1_ROWS), grid = eg_wgrad_c1_splits(B, H);
    const size_t lds = ((size_t)(EG_C1_ROWS + 2) * (W + 2) + 4 * 9 * 64) * sizeof(float);
    hipStream_t st = (hipStream_t)s;
    if (dtype == EG_F32) hipLaunchKernelGGL(wgrad_c1_kernel<float>, dim3(grid), dim3(256), lds, st, (const float*)x, dy, slab, B, H, W, units);
    else if (dtype == EG_F16) hipLaunchKernelGGL(wgrad_c1_kernel<f16_t>, dim3(grid), dim3(256), lds, st, (const f16_t*)x, dy, slab, B, H, W, units);
    else hipLaunchKernelGGL(wgrad_c1_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, (const bf16_t*)x, dy, slab, B, H, W, units);
    *nsplit_out = grid;
    EG_LAUNCH_CHECK();
    return 0;
}
