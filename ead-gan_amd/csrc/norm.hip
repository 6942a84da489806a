// BatchNorm2d (training mode) over NHWC activations viewed as [M = B*H*W][C].
// replaces nn.BatchNorm2d forward/backward: celebA/EAD-GAN_celebA.py:79,83,87; MNIST/EAD-GAN_rpqmnxy.py:80,83,87,145
// (eps is a parameter: the MNIST nets pass 0.8); dSprites/rp.py:130-138.
// Statistics are deterministic: per-row-block (count, mean, M2) partials combined with Chan's formula in fp64.
#include "eg_common.h"

#define BN_RPB 256   // rows per block in the partial kernels

// rows per workgroup of the partial kernels: 256, halved (down to 16) until the launch has >= 512 workgroups
static int bn_rpb(int M, int gx) {
    int rpb = BN_RPB;
    while (rpb > 16 && (long long)gx * cdiv(M, rpb) < 512) rpb >>= 1;
    return rpb;
}
static int bn_gx(int C, int dtype) {
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    return cdiv(cpr, cpr < 256 ? cpr : 256);
}

// block = 256 threads = (C/VEC chunk columns) x (row lanes); 16-byte loads; ONE pass: sums of (x - pivot) and (x - pivot)^2 with the
// block's first row as pivot (no cancellation for row blocks of <= 256 rows), LDS combine of the row lanes in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const T* __restrict__ x, int M, int C, int rpb, float* __restrict__ partial) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ float sm[2][256 * VEC];
    const int cpr = C / VEC;
    const int ccols = cpr < 256 ? cpr : 256;
    const int lanes = 256 / ccols;
    const int cj = threadIdx.x % ccols, rl = threadIdx.x / ccols;
    const int chunk = blockIdx.x * ccols + cj;
    const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    float s[VEC], q[VEC], pv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[j] = 0.f; q[j] = 0.f; pv[j] = 0.f; }
    if (chunk < cpr && rl < lanes) {        // (256 % ccols threads idle when ccols does not divide 256)
        const uint4 p0 = *reinterpret_cast<const uint4*>(x + (size_t)r0 * C + (size_t)chunk * VEC);
        const T* pe = reinterpret_cast<const T*>(&p0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) pv[j] = Elt<T>::ld(pe + j);
        for (int r = r0 + rl; r < r1; r += lanes) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)r * C + (size_t)chunk * VEC);
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float d = Elt<T>::ld(e + j) - pv[j];
                s[j] += d;
                q[j] += d * d;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sm[0][threadIdx.x * VEC + j] = s[j]; sm[1][threadIdx.x * VEC + j] = q[j]; }
    __syncthreads();
    if (rl == 0 && chunk < cpr) {
        const float cnt = (float)(r1 - r0);
        float* o = partial + (size_t)blockIdx.y * 3 * C + (size_t)chunk * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float ts = 0.f, tq = 0.f;
            for (int l = 0; l < lanes; ++l) { ts += sm[0][(l * ccols + cj) * VEC + j]; tq += sm[1][(l * ccols + cj) * VEC + j]; }
            o[j] = cnt;
            o[C + j] = pv[j] + ts / cnt;
            o[2 * C + j] = fmaxf(tq - ts * ts / cnt, 0.f);
        }
    }
}

// one wave per channel: lanes combine row-block partials (Chan) in fp64, then a shuffle tree merges the lanes
__global__ void bn_stats_final_kernel(const float* __restrict__ partial, int nrb, int C, int M, float eps, float momentum,
                                      float* running_mean, float* running_var, long long* nbt, float* save_mean, float* save_invstd,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ coef) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int r = lane; r < nrb; r += 64) {
        const float* o = partial + (size_t)r * 3 * C;
        const double nb = o[c], mb = o[C + c], qb = o[2 * C + c];
        const double tot = n + nb, delta = mb - mean;
        mean += delta * nb / tot;
        m2 += qb + delta * delta * n * nb / tot;
        n = tot;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double n2 = __shfl_xor(n, o, 64), mean2 = __shfl_xor(mean, o, 64), m22 = __shfl_xor(m2, o, 64);
        const double tot = n + n2;
        if (tot > 0.0) {
            const double delta = mean2 - mean;
            const double nm = mean + delta * n2 / tot;
            m2 = m2 + m22 + delta * delta * n * n2 / tot;
            mean = nm;
            n = tot;
        }
    }
    if (lane != 0) return;
    const double var = m2 / (double)M;
    save_mean[c] = (float)mean;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    save_invstd[c] = istd;
    coef[c] = gamma[c] * istd;
    coef[C + c] = beta[c] - (float)mean * gamma[c] * istd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unb = M > 1 ? m2 / (double)(M - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// y = act(x * A[c] + B[c]);  each thread keeps its 16-byte channel chunk fixed (grid stride is a multiple of the row length)
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, size_t nrows, int C, const float* __restrict__ coef, int act, float slope) {
    constexpr int VEC = Elt<T>::VEC;
    const int cpr = C / VEC;
    const int chunk = (blockIdx.x * blockDim.x + threadIdx.x) % cpr;
    const size_t row0 = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) / cpr;
    const size_t rstride = (size_t)gridDim.x * blockDim.x / cpr;
    float A[VEC], Bc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { A[j] = coef[chunk * VEC + j]; Bc[j] = coef[C + chunk * VEC + j]; }
    for (size_t r = row0; r < nrows; r += rstride) {
        const size_t o = r * C + (size_t)chunk * VEC;
        uint4 v = *reinterpret_cast<const uint4*>(x + o);
        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) Elt<T>::st(e + j, eg_act(Elt<T>::ld(e + j) * A[j] + Bc[j], act, slope));
        *reinterpret_cast<uint4*>(y + o) = v;
    }
}

static inline int bn_apply_blocks(size_t nrows, int cpr) {
    // total threads must be a multiple of cpr so that a thread's chunk column never changes across its grid-stride rows
    size_t want = (nrows * cpr + 255) / 256;
    if (want > 2048) want = 2048;
    int a = cpr, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }      // a = gcd(cpr, 256)
    const size_t mult = (size_t)cpr / a;                   // blocks must be a multiple of this
    want = (want + mult - 1) / mult * mult;
    return (int)want;
}

// bound for every dtype: bn_rpb stops halving once there are 512 workgroups -> at most max(M / 256, 1024) row blocks
extern "C" size_t eg_bn_ws_floats(int M, int C) { return (size_t)(cdiv(M, BN_RPB) > 1024 ? cdiv(M, BN_RPB) : 1024) * 3 * C + 5 * (size_t)C; }

extern "C" int eg_bn_fwd_train(int dtype, const void* x, void* y, int M, int C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                               float* save_mean, float* save_invstd, float* ws, int act, float slope, eg_stream_t s) {
    EG_REQUIRE(x && y && gamma && beta && save_mean && save_invstd && ws && M > 0 && C > 0, "eg_bn_fwd_train: bad argument");
    EG_REQUIRE(C % (dtype == EG_F32 ? 4 : 8) == 0, "eg_bn_fwd_train: C must be a multiple of the 16-byte vector width");
    const int gx = bn_gx(C, dtype), rpb = bn_rpb(M, gx);
    const int nrb = cdiv(M, rpb);
    dim3 g1(gx, nrb);
    hipStream_t st = (hipStream_t)s;
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_stats_partial_kernel<float>, g1, dim3(256), 0, st, (const float*)x, M, C, rpb, ws);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_stats_partial_kernel<f16_t>, g1, dim3(256), 0, st, (const f16_t*)x, M, C, rpb, ws);
    else hipLaunchKernelGGL(bn_stats_partial_kernel<bf16_t>, g1, dim3(256), 0, st, (const bf16_t*)x, M, C, rpb, ws);
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    float* coef = ws + (size_t)nrb * 3 * C;           // 2*C floats behind the partials
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, ws, nrb, C, M, eps, momentum, running_mean, running_var,
                       num_batches_tracked, save_mean, save_invstd, gamma, beta, coef);
    const int blocks = bn_apply_blocks((size_t)M, cpr);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, (size_t)M, C, coef, act, slope);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, (size_t)M, C, coef, act, slope);
    else hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (size_t)M, C, coef, act, slope);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- statistics from the producing convolution's epilogue (eg_epilogue.stat_mode = EG_STAT_MOMENTS) -----------------------------
// stat = [2][C][nrb]: (mean, M2) of nrb row blocks of `cnt` rows each.  One wave per channel, contiguous reads.  Equal counts make
// Chan's combination a plain two-pass form without divisions in the loops: mean = avg(mean_b), M2 = sum M2_b + cnt * sum (mean_b - mean)^2;
// fp64, lanes in a fixed order.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// WIDE: one 256-thread workgroup per channel instead of one wave (launches with >= 1024 row blocks: the small networks at B = 128 .. 512 have
// 1024-12288 of them and 32-64 channels -- 16 waves walking 4096 partials each took 18-25 us on the main chain; few row blocks keep the wave
// form and its bits)
__device__ __forceinline__ double block4_sum_f64(double v, double* sh) {
    v = wave_sum_f64(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double t = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return t;
}
template <bool WIDE>
__global__ void bn_stats_final_eq_kernel(const float* __restrict__ stat, int nrb, int C, int cnt, int M, float eps, float momentum,
                                         float* running_mean, float* running_var, long long* nbt, float* save_mean, float* save_invstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ coef) {
    __shared__ double sh[4];
    const int c = WIDE ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = WIDE ? threadIdx.x : threadIdx.x & 63, step = WIDE ? 256 : 64;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
    if (c >= C) return;
    const float* __restrict__ mb = stat + (size_t)c * nrb;
    const float* __restrict__ qb = stat + ((size_t)C + c) * nrb;
    double sm = 0.0;
    for (int r = lane; r < nrb; r += step) sm += (double)mb[r];
    const double mean = (WIDE ? block4_sum_f64(sm, sh) : wave_sum_f64(sm)) / (double)nrb;
    double m2 = 0.0;
    for (int r = lane; r < nrb; r += step) {
        const double d = (double)mb[r] - mean;
        m2 += (double)qb[r] + (double)cnt * d * d;
    }
    m2 = WIDE ? block4_sum_f64(m2, sh) : wave_sum_f64(m2);
    if (lane != 0) return;
    const double var = m2 / (double)M;
    save_mean[c] = (float)mean;
    const float istd = (float)(1.0 / sqrt(var + (double)eps));
    save_invstd[c] = istd;
    coef[c] = gamma[c] * istd;
    coef[C + c] = beta[c] - (float)mean * gamma[c] * istd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unb = M > 1 ? m2 / (double)(M - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
#define EG_FINAL_WIDE_NRB 1024

extern "C" int eg_bn_fwd_train_fused(int dtype, const void* x, void* y, int M, int C, const float* stat, int nrb, int rows_per_block,
                                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                     long long* num_batches_tracked, float* save_mean, float* save_invstd, float* ws, int act, float slope,
                                     eg_stream_t s) {
    EG_REQUIRE(x && y && stat && gamma && beta && save_mean && save_invstd && ws && M > 0 && C > 0 && nrb > 0, "eg_bn_fwd_train_fused: bad argument");
    EG_REQUIRE((long long)nrb * rows_per_block == M, "eg_bn_fwd_train_fused: nrb * rows_per_block must be M (whole row blocks of equal size)");
    EG_REQUIRE(C % (dtype == EG_F32 ? 4 : 8) == 0, "eg_bn_fwd_train_fused: C must be a multiple of the 16-byte vector width");
    hipStream_t st = (hipStream_t)s;
    float* coef = ws;                                   // 2*C floats
    if (nrb >= EG_FINAL_WIDE_NRB)
        hipLaunchKernelGGL(bn_stats_final_eq_kernel<true>, dim3(C), dim3(256), 0, st, stat, nrb, C, rows_per_block, M, eps, momentum, running_mean,
                           running_var, num_batches_tracked, save_mean, save_invstd, gamma, beta, coef);
    else
        hipLaunchKernelGGL(bn_stats_final_eq_kernel<false>, dim3(cdiv(C, 4)), dim3(256), 0, st, stat, nrb, C, rows_per_block, M, eps, momentum, running_mean,
                           running_var, num_batches_tracked, save_mean, save_invstd, gamma, beta, coef);
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    const int blocks = bn_apply_blocks((size_t)M, cpr);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, (size_t)M, C, coef, act, slope);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, (size_t)M, C, coef, act, slope);
    else hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (size_t)M, C, coef, act, slope);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- synchronised BatchNorm (data parallel: statistics over the GLOBAL batch) ---------------------------------------
// forward:  eg_bn_stats_local -> (n, mean, M2) per channel of this rank's rows; the host layer gathers the 3*C floats of every rank
//           (slot-wise all-reduce); eg_bn_fwd_from_stats combines them with Chan's formula (same kernel as the single-rank path, the
//           "row blocks" now being ranks) and normalises the local rows.
// backward: eg_bn_bwd_sums_local -> local sum(dy), sum(dy*xhat) (added to dbeta / dgamma: parameter gradients stay per-rank, the
//           gradient all-reduce averages them); the host all-reduces the 2*C sums; eg_bn_bwd_from_sums applies dz with M_global.
__global__ void bn_stats_local_kernel(const float* __restrict__ partial, int nrb, int C, float* __restrict__ stats) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int r = lane; r < nrb; r += 64) {
        const float* o = partial + (size_t)r * 3 * C;
        const double nb = o[c], mb = o[C + c], qb = o[2 * C + c];
        const double tot = n + nb, delta = mb - mean;
        mean += delta * nb / tot;
        m2 += qb + delta * delta * n * nb / tot;
        n = tot;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double n2 = __shfl_xor(n, o, 64), mean2 = __shfl_xor(mean, o, 64), m22 = __shfl_xor(m2, o, 64);
        const double tot = n + n2;
        if (tot > 0.0) {
            const double delta = mean2 - mean;
            const double nm = mean + delta * n2 / tot;
            m2 = m2 + m22 + delta * delta * n * n2 / tot;
            mean = nm;
            n = tot;
        }
    }
    if (lane != 0) return;
    stats[c] = (float)n;
    stats[C + c] = (float)mean;
    stats[2 * C + c] = (float)m2;
}

template <typename T>
static void launch_bn_stats_partial(const void* x, int M, int C, int dtype, float* ws, int* nrb_out, hipStream_t st) {
    const int gx = bn_gx(C, dtype), rpb = bn_rpb(M, gx);
    *nrb_out = cdiv(M, rpb);
    hipLaunchKernelGGL(bn_stats_partial_kernel<T>, dim3(gx, *nrb_out), dim3(256), 0, st, (const T*)x, M, C, rpb, ws);
}

extern "C" int eg_bn_stats_local(int dtype, const void* x, int M, int C, float* ws, float* stats, eg_stream_t s) {
    EG_REQUIRE(x && ws && stats && M > 0 && C > 0 && C % (dtype == EG_F32 ? 4 : 8) == 0, "eg_bn_stats_local: bad argument");
    int nrb = 0;
    hipStream_t st = (hipStream_t)s;
    if (dtype == EG_F32) launch_bn_stats_partial<float>(x, M, C, dtype, ws, &nrb, st);
    else if (dtype == EG_F16) launch_bn_stats_partial<f16_t>(x, M, C, dtype, ws, &nrb, st);
    else launch_bn_stats_partial<bf16_t>(x, M, C, dtype, ws, &nrb, st);
    hipLaunchKernelGGL(bn_stats_local_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, ws, nrb, C, stats);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_bn_fwd_from_stats(int dtype, const void* x, void* y, int M_local, int C, const float* stats_all, int nranks, int M_global,
                                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float* save_mean, float* save_invstd, float* ws, int act, float slope,
                                    eg_stream_t s) {
    EG_REQUIRE(x && y && stats_all && gamma && beta && save_mean && save_invstd && ws && M_local > 0 && M_global >= M_local && nranks > 0,
               "eg_bn_fwd_from_stats: bad argument");
    EG_REQUIRE(C % (dtype == EG_F32 ? 4 : 8) == 0, "eg_bn_fwd_from_stats: C must be a multiple of the 16-byte vector width");
    hipStream_t st = (hipStream_t)s;
    float* coef = ws;                                   // 2*C floats
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, stats_all, nranks, C, M_global, eps, momentum, running_mean, running_var,
                       num_batches_tracked, save_mean, save_invstd, gamma, beta, coef);
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    const int blocks = bn_apply_blocks((size_t)M_local, cpr);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, (size_t)M_local, C, coef, act, slope);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, (size_t)M_local, C, coef, act, slope);
    else hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (size_t)M_local, C, coef, act, slope);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- eval-mode forward (running statistics; generate_image.py / gen_imgs.py of the reference put the generator in .eval()) ------
__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                    const float* __restrict__ rv, float eps, int C, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float a = gamma[c] / sqrtf(rv[c] + eps);       // torch: (x - mean) / sqrt(var + eps) * weight + bias
    coef[c] = a;
    coef[C + c] = beta[c] - rm[c] * a;
}

extern "C" int eg_bn_fwd_eval(int dtype, const void* x, void* y, int M, int C, const float* gamma, const float* beta, float eps,
                              const float* running_mean, const float* running_var, float* ws, int act, float slope, eg_stream_t s) {
    EG_REQUIRE(x && y && gamma && beta && running_mean && running_var && ws && M > 0 && C > 0, "eg_bn_fwd_eval: bad argument");
    EG_REQUIRE(C % (dtype == EG_F32 ? 4 : 8) == 0, "eg_bn_fwd_eval: C must be a multiple of the 16-byte vector width");
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, gamma, beta, running_mean, running_var, eps, C, ws);
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    const int blocks = bn_apply_blocks((size_t)M, cpr);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, (size_t)M, C, ws, act, slope);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, (size_t)M, C, ws, act, slope);
    else hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (size_t)M, C, ws, act, slope);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- backward -----------------------------------------------------------------------------------
__device__ __forceinline__ float pre_act_grad(float y, int act, float slope) {
    switch (act) {
        case EG_ACT_LRELU: return y > 0.f ? 1.f : slope;
        case EG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        default: return 1.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ z, const T* __restrict__ da, int M, int C, int rpb,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, int act, float slope,
                                                             float* __restrict__ partial) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ float sm[2][256 * VEC];
    const int cpr = C / VEC;
    const int ccols = cpr < 256 ? cpr : 256;
    const int lanes = 256 / ccols;
    const int cj = threadIdx.x % ccols, rl = threadIdx.x / ccols;
    const int chunk = blockIdx.x * ccols + cj;
    const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (chunk < cpr && rl < lanes) {
        float mu[VEC], is[VEC], g[VEC], bb[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = chunk * VEC + j;
            mu[j] = mean[c]; is[j] = invstd[c]; g[j] = gamma[c]; bb[j] = beta[c];
        }
        for (int r = r0 + rl; r < r1; r += lanes) {
            const size_t o = (size_t)r * C + (size_t)chunk * VEC;
            const uint4 vz = *reinterpret_cast<const uint4*>(z + o);
            const uint4 vd = *reinterpret_cast<const uint4*>(da + o);
            const T* ez = reinterpret_cast<const T*>(&vz);
            const T* ed = reinterpret_cast<const T*>(&vd);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (Elt<T>::ld(ez + j) - mu[j]) * is[j];
                const float dy = Elt<T>::ld(ed + j) * pre_act_grad(xh * g[j] + bb[j], act, slope);
                s1[j] += dy;
                s2[j] += dy * xh;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sm[0][threadIdx.x * VEC + j] = s1[j]; sm[1][threadIdx.x * VEC + j] = s2[j]; }
    __syncthreads();
    if (rl == 0 && chunk < cpr) {
        float* o = partial + (size_t)blockIdx.y * 2 * C + (size_t)chunk * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t1 = 0.f, t2 = 0.f;
            for (int l = 0; l < lanes; ++l) { t1 += sm[0][(l * ccols + cj) * VEC + j]; t2 += sm[1][(l * ccols + cj) * VEC + j]; }
            o[j] = t1;
            o[C + j] = t2;
        }
    }
}

// dz = A*dy - Bz*z - Cc  with  A = g*is, Bz = g*is*is*s2/M, Cc = g*is*(s1/M - mean*is*s2/M);  mask: z*P + Q > 0
__global__ void bn_bwd_final_kernel(const float* __restrict__ partial, int nrb, int C, float* sums, float* dgamma, float* dbeta,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, int M, float* __restrict__ coef) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int r = lane; r < nrb; r += 64) {
        s1 += partial[(size_t)r * 2 * C + c];
        s2 += partial[(size_t)r * 2 * C + C + c];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane != 0) return;
    sums[c] = s1;
    sums[C + c] = s2;
    if (dbeta) dbeta[c] += s1;
    if (dgamma) dgamma[c] += s2;
    const float g = gamma[c], is = invstd[c], mu = mean[c], invM = 1.f / (float)M;
    coef[c] = g * is;
    coef[C + c] = g * is * is * s2 * invM;
    coef[2 * C + c] = g * is * (s1 * invM - mu * is * s2 * invM);
    coef[3 * C + c] = g * is;
    coef[4 * C + c] = beta[c] - mu * g * is;
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ z, const T* __restrict__ da, T* __restrict__ dz, size_t nrows, int C,
                                    const float* __restrict__ coef, int act, float slope, int post_act, float post_slope,
                                    const float* __restrict__ post_sigma) {
    constexpr int VEC = Elt<T>::VEC;
    const int cpr = C / VEC;
    const int chunk = (blockIdx.x * blockDim.x + threadIdx.x) % cpr;
    const size_t row0 = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) / cpr;
    const size_t rstride = (size_t)gridDim.x * blockDim.x / cpr;
    float A[VEC], Bz[VEC], Cc[VEC], P[VEC], Q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = chunk * VEC + j;
        A[j] = coef[c]; Bz[j] = coef[C + c]; Cc[j] = coef[2 * C + c]; P[j] = coef[3 * C + c]; Q[j] = coef[4 * C + c];
    }
    const float post = post_sigma ? 1.f / post_sigma[0] : 1.f;
    for (size_t r = row0; r < nrows; r += rstride) {
        const size_t o = r * C + (size_t)chunk * VEC;
        const uint4 vz = *reinterpret_cast<const uint4*>(z + o);
        uint4 vd = *reinterpret_cast<const uint4*>(da + o);
        const T* ez = reinterpret_cast<const T*>(&vz);
        T* ed = reinterpret_cast<T*>(&vd);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float zz = Elt<T>::ld(ez + j);
            const float dy = Elt<T>::ld(ed + j) * pre_act_grad(zz * P[j] + Q[j], act, slope);
            Elt<T>::st(ed + j, (A[j] * dy - Bz[j] * zz - Cc[j]) * eg_act_grad_from_out(zz, post_act, post_slope) * post);
        }
        *reinterpret_cast<uint4*>(dz + o) = vd;
    }
}

// ws: >= eg_bn_ws_floats(M,C) floats; sums: 2*C floats
static int bn_bwd_impl(int dtype, const void* z, const void* da, void* dz, int M, int C, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, int act, float slope, float* dgamma, float* dbeta, float* sums, float* ws,
                       int post_act, float post_slope, const float* post_sigma, hipStream_t st) {
    const int gx = bn_gx(C, dtype), rpb = bn_rpb(M, gx);
    const int nrb = cdiv(M, rpb);
    dim3 g1(gx, nrb);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_bwd_partial_kernel<float>, g1, dim3(256), 0, st, (const float*)z, (const float*)da, M, C, rpb, gamma, beta, save_mean, save_invstd, act, slope, ws);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_bwd_partial_kernel<f16_t>, g1, dim3(256), 0, st, (const f16_t*)z, (const f16_t*)da, M, C, rpb, gamma, beta, save_mean, save_invstd, act, slope, ws);
    else hipLaunchKernelGGL(bn_bwd_partial_kernel<bf16_t>, g1, dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)da, M, C, rpb, gamma, beta, save_mean, save_invstd, act, slope, ws);
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    float* coef = ws + (size_t)nrb * 3 * C;
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, ws, nrb, C, sums, dgamma, dbeta, gamma, beta, save_mean, save_invstd, M, coef);
    const int blocks = bn_apply_blocks((size_t)M, cpr);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)z, (const float*)da, (float*)dz, (size_t)M, C, coef, act, slope, post_act, post_slope, post_sigma);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_bwd_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)z, (const f16_t*)da, (f16_t*)dz, (size_t)M, C, coef, act, slope, post_act, post_slope, post_sigma);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)da, (bf16_t*)dz, (size_t)M, C, coef, act, slope, post_act, post_slope, post_sigma);
    return 0;
}

// synchronised backward, stage 1: this rank's sums (also added to dbeta / dgamma), no dz yet
extern "C" int eg_bn_bwd_sums_local(int dtype, const void* z, const void* da, int M, int C, const float* gamma, const float* beta,
                                    const float* save_mean, const float* save_invstd, int act, float slope, float* dgamma, float* dbeta,
                                    float* sums, float* ws, eg_stream_t s) {
    EG_REQUIRE(z && da && gamma && beta && save_mean && save_invstd && sums && ws && M > 0 && C > 0, "eg_bn_bwd_sums_local: bad argument");
    hipStream_t st = (hipStream_t)s;
    const int gx = bn_gx(C, dtype), rpb = bn_rpb(M, gx);
    const int nrb = cdiv(M, rpb);
    dim3 g1(gx, nrb);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_bwd_partial_kernel<float>, g1, dim3(256), 0, st, (const float*)z, (const float*)da, M, C, rpb, gamma, beta, save_mean, save_invstd, act, slope, ws);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_bwd_partial_kernel<f16_t>, g1, dim3(256), 0, st, (const f16_t*)z, (const f16_t*)da, M, C, rpb, gamma, beta, save_mean, save_invstd, act, slope, ws);
    else hipLaunchKernelGGL(bn_bwd_partial_kernel<bf16_t>, g1, dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)da, M, C, rpb, gamma, beta, save_mean, save_invstd, act, slope, ws);
    float* coef = ws + (size_t)nrb * 3 * C;             // written but unused here (local M): stage 2 recomputes it from the global sums
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, ws, nrb, C, sums, dgamma, dbeta, gamma, beta, save_mean, save_invstd, M, coef);
    EG_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_bwd_coef_kernel(const float* __restrict__ sums, int C, int M, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s1 = sums[c], s2 = sums[C + c];
    const float g = gamma[c], is = invstd[c], mu = mean[c], invM = 1.f / (float)M;
    coef[c] = g * is;
    coef[C + c] = g * is * is * s2 * invM;
    coef[2 * C + c] = g * is * (s1 * invM - mu * is * s2 * invM);
    coef[3 * C + c] = g * is;
    coef[4 * C + c] = beta[c] - mu * g * is;
}

// stage 2: dz of the local rows from the GLOBAL sums (after the host's all-reduce) and the global row count
extern "C" int eg_bn_bwd_from_sums(int dtype, const void* z, const void* da, void* dz, int M_local, int C, const float* sums_global, int M_global,
                                   const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int act, float slope,
                                   float* ws, eg_stream_t s) {
    EG_REQUIRE(z && da && dz && sums_global && gamma && beta && save_mean && save_invstd && ws && M_local > 0 && M_global >= M_local,
               "eg_bn_bwd_from_sums: bad argument");
    hipStream_t st = (hipStream_t)s;
    float* coef = ws;                                   // 5*C floats
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, sums_global, C, M_global, gamma, beta, save_mean, save_invstd, coef);
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    const int blocks = bn_apply_blocks((size_t)M_local, cpr);
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)z, (const float*)da, (float*)dz, (size_t)M_local, C, coef, act, slope, EG_ACT_NONE, 0.f, (const float*)nullptr);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_bwd_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)z, (const f16_t*)da, (f16_t*)dz, (size_t)M_local, C, coef, act, slope, EG_ACT_NONE, 0.f, (const float*)nullptr);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)da, (bf16_t*)dz, (size_t)M_local, C, coef, act, slope, EG_ACT_NONE, 0.f, (const float*)nullptr);
    EG_LAUNCH_CHECK();
    return 0;
}

// the two sums from the epilogue of the convolution that produced dy (EG_STAT_BN_BWD), stat = [2][C][nrb]: one wave per channel
template <bool WIDE>       // (WIDE: a workgroup per channel, as bn_stats_final_eq_kernel)
__global__ void bn_bwd_final_t_kernel(const float* __restrict__ stat, int nrb, int C, float* sums, float* dgamma, float* dbeta,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                      const float* __restrict__ invstd, int M, float* __restrict__ coef) {
    __shared__ float sh[2][4];
    const int c = WIDE ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = WIDE ? threadIdx.x : threadIdx.x & 63, step = WIDE ? 256 : 64;
    if (c >= C) return;
    const float* __restrict__ a = stat + (size_t)c * nrb;
    const float* __restrict__ b = stat + ((size_t)C + c) * nrb;
    float s1 = 0.f, s2 = 0.f;
    for (int r = lane; r < nrb; r += step) { s1 += a[r]; s2 += b[r]; }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (WIDE) {
        if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
        __syncthreads();
        s1 = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
        s2 = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
    }
    if (lane != 0) return;
    sums[c] = s1;
    sums[C + c] = s2;
    if (dbeta) dbeta[c] += s1;
    if (dgamma) dgamma[c] += s2;
    const float g = gamma[c], is = invstd[c], mu = mean[c], invM = 1.f / (float)M;
    coef[c] = g * is;
    coef[C + c] = g * is * is * s2 * invM;
    coef[2 * C + c] = g * is * (s1 * invM - mu * is * s2 * invM);
    coef[3 * C + c] = g * is;
    coef[4 * C + c] = beta[c] - mu * g * is;
}

extern "C" int eg_bn_bwd_fused(int dtype, const void* z, const void* dy, void* dz, int M, int C, const float* stat, int nrb,
                               const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                               float* sums, float* ws, eg_stream_t s) {
    EG_REQUIRE(z && dy && dz && stat && gamma && beta && save_mean && save_invstd && sums && ws && M > 0 && C > 0 && nrb > 0, "eg_bn_bwd_fused: bad argument");
    EG_REQUIRE(C % (dtype == EG_F32 ? 4 : 8) == 0, "eg_bn_bwd_fused: C must be a multiple of the 16-byte vector width");
    hipStream_t st = (hipStream_t)s;
    float* coef = ws;                                   // 5*C floats
    if (nrb >= EG_FINAL_WIDE_NRB)
        hipLaunchKernelGGL(bn_bwd_final_t_kernel<true>, dim3(C), dim3(256), 0, st, stat, nrb, C, sums, dgamma, dbeta, gamma, beta, save_mean, save_invstd, M, coef);
    else
        hipLaunchKernelGGL(bn_bwd_final_t_kernel<false>, dim3(cdiv(C, 4)), dim3(256), 0, st, stat, nrb, C, sums, dgamma, dbeta, gamma, beta, save_mean, save_invstd, M, coef);
    const int cpr = C / (dtype == EG_F32 ? 4 : 8);
    const int blocks = bn_apply_blocks((size_t)M, cpr);
    // dy already carries the activation gradient (act = NONE here)
    if (dtype == EG_F32) hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)z, (const float*)dy, (float*)dz, (size_t)M, C, coef, EG_ACT_NONE, 0.f, EG_ACT_NONE, 0.f, (const float*)nullptr);
    else if (dtype == EG_F16) hipLaunchKernelGGL(bn_bwd_apply_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)z, (const f16_t*)dy, (f16_t*)dz, (size_t)M, C, coef, EG_ACT_NONE, 0.f, EG_ACT_NONE, 0.f, (const float*)nullptr);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)dy, (bf16_t*)dz, (size_t)M, C, coef, EG_ACT_NONE, 0.f, EG_ACT_NONE, 0.f, (const float*)nullptr);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_bn_bwd(int dtype, const void* z, const void* da, void* dz, int M, int C, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, int act, float slope, float* dgamma, float* dbeta,
                         float* sums, float* ws, eg_stream_t s) {
    EG_REQUIRE(z && da && dz && gamma && beta && save_mean && save_invstd && sums && ws, "eg_bn_bwd: null pointer");
    bn_bwd_impl(dtype, z, da, dz, M, C, gamma, beta, save_mean, save_invstd, act, slope, dgamma, dbeta, sums, ws, EG_ACT_NONE, 0.f, nullptr, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_bn_bwd_post(int dtype, const void* z, const void* da, void* dz, int M, int C, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* sums, float* ws,
                              int post_act, float post_slope, const float* post_sigma, eg_stream_t s) {
    EG_REQUIRE(z && da && dz && gamma && beta && save_mean && save_invstd && sums && ws, "eg_bn_bwd_post: null pointer");
    bn_bwd_impl(dtype, z, da, dz, M, C, gamma, beta, save_mean, save_invstd, EG_ACT_NONE, 0.f, dgamma, dbeta, sums, ws, post_act, post_slope, post_sigma, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}
