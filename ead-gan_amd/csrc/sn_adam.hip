// Spectral-norm power iteration (torch.nn.utils.spectral_norm hook semantics: one iteration per training
// forward, eps 1e-12, sigma from the UPDATED u,v -- celebA/EAD-GAN_celebA.py:110-120, MNIST/EAD-GAN_rpqmnxy.py:107,
// dSprites/rp.py:95-109) and the fused Adam update (torch.optim.Adam, betas (.5,.999), eps 1e-8 --
// celebA/EAD-GAN_celebA.py:211-217).  W is never divided by sigma in memory: consumers scale in their epilogue.
#include "eg_common.h"

#define SN_NRB 8   // row blocks in the W^T u partial kernel

__global__ void sn_wtu_partial_kernel(const float* __restrict__ W, const float* __restrict__ u, int R, int Kd, float* __restrict__ partial) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int rb = (R + SN_NRB - 1) / SN_NRB;
    const int r0 = blockIdx.y * rb, r1 = min(R, r0 + rb);
    if (k >= Kd) return;
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += W[(size_t)r * Kd + k] * u[r];
    partial[(size_t)blockIdx.y * Kd + k] = a;
}

__global__ void sn_v_final_kernel(const float* __restrict__ partial, int nrb, int Kd, float eps, float* __restrict__ v, float* __restrict__ v_snap) {
    __shared__ float sm[16];
    float ss = 0.f;
    for (int k = threadIdx.x; k < Kd; k += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < nrb; ++r) t += partial[(size_t)r * Kd + k];
        v[k] = t;
        ss += t * t;
    }
    const float nrm = sqrtf(block_sum(ss, sm));
    const float inv = 1.f / fmaxf(nrm, eps);
    for (int k = threadIdx.x; k < Kd; k += blockDim.x) {
        const float vn = v[k] * inv;
        v[k] = vn;
        if (v_snap) v_snap[k] = vn;
    }
}

__global__ void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ v, int Kd, float* __restrict__ s) {
    __shared__ float sm[16];
    const int r = blockIdx.x;
    float a = 0.f;
    for (int k = threadIdx.x; k < Kd; k += blockDim.x) a += W[(size_t)r * Kd + k] * v[k];
    const float tot = block_sum(a, sm);
    if (threadIdx.x == 0) s[r] = tot;
}

__global__ void sn_u_final_kernel(const float* __restrict__ s, int R, float eps, float* __restrict__ u, float* __restrict__ sigma, float* __restrict__ u_snap) {
    __shared__ float sm[16];
    float ss = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) ss += s[r] * s[r];
    const float nrm = sqrtf(block_sum(ss, sm));
    const float inv = 1.f / fmaxf(nrm, eps);
    float d = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const float un = s[r] * inv;
        u[r] = un;
        if (u_snap) u_snap[r] = un;
        d += un * s[r];
    }
    const float sg = block_sum(d, sm);
    if (threadIdx.x == 0) sigma[0] = sg;
}

// sigma only (eval mode / frozen u,v): sigma = u . (W v)
__global__ void sn_sigma_only_kernel(const float* __restrict__ s, const float* __restrict__ u, int R, float* __restrict__ sigma) {
    __shared__ float sm[16];
    float d = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) d += u[r] * s[r];
    const float sg = block_sum(d, sm);
    if (threadIdx.x == 0) sigma[0] = sg;
}

extern "C" size_t eg_sn_ws_floats(int R, int Kd) { return (size_t)SN_NRB * Kd + R; }

/* one power iteration in place on u[R], v[Kd]; writes sigma[0] and (optionally) snapshots of the updated u,v that the
 * backward of THIS forward must use (torch clones them).  training==0: u,v untouched, sigma only. */
extern "C" int eg_sn_power_iter(const float* w_orig, int R, int Kd, float* u, float* v, float* sigma, float* u_snap, float* v_snap, float* ws,
                                int training, float eps, eg_stream_t s) {
    EG_REQUIRE(w_orig && u && v && sigma && ws && R > 0 && Kd > 0, "eg_sn_power_iter: bad argument");
    hipStream_t st = (hipStream_t)s;
    const int nrb = SN_NRB;
    float* partial = ws;
    float* sv = ws + (size_t)nrb * Kd;
    if (training) {
        hipLaunchKernelGGL(sn_wtu_partial_kernel, dim3(cdiv(Kd, 256), nrb), dim3(256), 0, st, w_orig, u, R, Kd, partial);
        hipLaunchKernelGGL(sn_v_final_kernel, dim3(1), dim3(1024), 0, st, partial, nrb, Kd, eps, v, v_snap);
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3(R), dim3(256), 0, st, w_orig, v, Kd, sv);
    if (training) hipLaunchKernelGGL(sn_u_final_kernel, dim3(1), dim3(1024), 0, st, sv, R, eps, u, sigma, u_snap);
    else hipLaunchKernelGGL(sn_sigma_only_kernel, dim3(1), dim3(1024), 0, st, sv, u, R, sigma);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- Adam -----------------------------------------------------------------------------------------
__global__ void adam_tick_kernel(int* step) { step[0] += 1; }

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr, float b1, float b2, float eps, const int* __restrict__ step) {
    const int t = step[0];
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w = 1.f - b1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float mi = m[i];
        // exp_avg.lerp_(grad, 1-beta1) with ATen's two-sided formula
        mi = (w < 0.5f) ? mi + w * (gi - mi) : gi - (gi - mi) * (1.f - w);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

/* In-place Adam over a flat fp32 arena.  `step` is a device int32 incremented by this call (so a captured
 * graph replays with the right bias correction). */
extern "C" int eg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int* step,
                            int tick, eg_stream_t s) {
    EG_REQUIRE(p && g && m && v && step, "eg_adam_step: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, step);
    EG_LAUNCH_CHECK();
    return 0;
}

__global__ void fill_kernel(float* p, size_t n, float val) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = val;
}
extern "C" int eg_fill_f32(float* p, size_t n, float val, eg_stream_t s) {
    EG_REQUIRE(p, "eg_fill_f32: null pointer");
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (n) hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, p, n, val);
    EG_LAUNCH_CHECK();
    return 0;
}
