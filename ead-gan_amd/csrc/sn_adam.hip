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

// ---- all spectrally-normalised layers of a network in ONE launch per stage (grid.z = layer) -------------------------
#define SN_MAXL 8
struct SnMulti {
    const float* w[SN_MAXL];
    float* u[SN_MAXL];
    float* v[SN_MAXL];
    float* sigma[SN_MAXL];
    float* u_snap[SN_MAXL];
    float* v_snap[SN_MAXL];
    int R[SN_MAXL], Kd[SN_MAXL];
    long long ws_off[SN_MAXL];
    float* ws;
    float eps;
};

__global__ void snm_wtu_partial_kernel(const SnMulti p) {
    const int l = blockIdx.z, R = p.R[l], Kd = p.Kd[l];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Kd) return;
    const int rb = (R + SN_NRB - 1) / SN_NRB;
    const int r0 = blockIdx.y * rb, r1 = min(R, r0 + rb);
    const float* __restrict__ W = p.w[l];
    const float* __restrict__ u = p.u[l];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        a0 += W[(size_t)r * Kd + k] * u[r];
        a1 += W[(size_t)(r + 1) * Kd + k] * u[r + 1];
        a2 += W[(size_t)(r + 2) * Kd + k] * u[r + 2];
        a3 += W[(size_t)(r + 3) * Kd + k] * u[r + 3];
    }
    for (; r < r1; ++r) a0 += W[(size_t)r * Kd + k] * u[r];
    p.ws[p.ws_off[l] + (size_t)blockIdx.y * Kd + k] = (a0 + a1) + (a2 + a3);
}

__global__ void snm_v_final_kernel(const SnMulti p) {
    __shared__ float sm[16];
    const int l = blockIdx.z, Kd = p.Kd[l];
    const float* partial = p.ws + p.ws_off[l];
    float* v = p.v[l];
    float* v_snap = p.v_snap[l];
    float ss = 0.f;
    for (int k = threadIdx.x; k < Kd; k += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < SN_NRB; ++r) t += partial[(size_t)r * Kd + k];
        v[k] = t;
        ss += t * t;
    }
    const float nrm = sqrtf(block_sum(ss, sm));
    const float inv = 1.f / fmaxf(nrm, p.eps);
    for (int k = threadIdx.x; k < Kd; k += blockDim.x) {
        const float vn = v[k] * inv;
        v[k] = vn;
        if (v_snap) v_snap[k] = vn;
    }
}

__global__ void snm_wv_kernel(const SnMulti p) {
    __shared__ float sm[16];
    const int l = blockIdx.z, R = p.R[l], Kd = p.Kd[l];
    const int r = blockIdx.x;
    if (r >= R) return;
    const float* __restrict__ W = p.w[l] + (size_t)r * Kd;
    const float* __restrict__ v = p.v[l];
    float a = 0.f;
    for (int k = threadIdx.x; k < Kd; k += blockDim.x) a += W[k] * v[k];
    const float tot = block_sum(a, sm);
    if (threadIdx.x == 0) p.ws[p.ws_off[l] + (size_t)SN_NRB * Kd + r] = tot;
}

__global__ void snm_u_final_kernel(const SnMulti p, int training) {
    __shared__ float sm[16];
    const int l = blockIdx.z, R = p.R[l], Kd = p.Kd[l];
    const float* s = p.ws + p.ws_off[l] + (size_t)SN_NRB * Kd;
    float* u = p.u[l];
    if (!training) {
        float d = 0.f;
        for (int r = threadIdx.x; r < R; r += blockDim.x) d += u[r] * s[r];
        const float sg = block_sum(d, sm);
        if (threadIdx.x == 0) p.sigma[l][0] = sg;
        return;
    }
    float ss = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) ss += s[r] * s[r];
    const float nrm = sqrtf(block_sum(ss, sm));
    const float inv = 1.f / fmaxf(nrm, p.eps);
    float d = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const float un = s[r] * inv;
        u[r] = un;
        if (p.u_snap[l]) p.u_snap[l][r] = un;
        d += un * s[r];
    }
    const float sg = block_sum(d, sm);
    if (threadIdx.x == 0) p.sigma[l][0] = sg;
}

// ---- the same iteration in TWO launches: each stage's per-layer finish by the last workgroup of the layer to arrive ---------------------
// (W^T u partial sums + the norm of v; W v + the norm of u and sigma).  On the chains that wait for a power iteration -- step 3 of the
// CelebA iteration runs three in a row on freshly updated weights, the small networks six per iteration -- every launch is a dependent
// ~5-10 us.  The finishing code is the 1024-thread code of snm_v_final / snm_u_final: stage one runs 1024-thread workgroups (each k's
// partial sum does not depend on the workgroup shape), stage two's 256-thread workgroup walks the 16 "virtual waves" of that code in
// order, so u, v and sigma keep their bits.  Hand-off: partial results are written through (agent-scope stores), vmcnt(0), a relaxed
// counter per layer and stage (caller-owned, zeroed once, left at zero).
__global__ __launch_bounds__(1024) void snm2_wtu_v_kernel(const SnMulti p, unsigned* __restrict__ cnt) {
    __shared__ float sm[16];
    __shared__ unsigned flag;
    const int l = blockIdx.z, R = p.R[l], Kd = p.Kd[l];
    if ((int)blockIdx.x * 1024 >= Kd) return;           // (uniform: this layer has fewer column blocks than the widest one)
    const int k = blockIdx.x * 1024 + threadIdx.x;
    float* partial = p.ws + p.ws_off[l];
    if (k < Kd) {
        const int rb = (R + SN_NRB - 1) / SN_NRB;
        const int r0 = blockIdx.y * rb, r1 = min(R, r0 + rb);
        const float* __restrict__ W = p.w[l];
        const float* __restrict__ u = p.u[l];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            a0 += W[(size_t)r * Kd + k] * u[r];
            a1 += W[(size_t)(r + 1) * Kd + k] * u[r + 1];
            a2 += W[(size_t)(r + 2) * Kd + k] * u[r + 2];
            a3 += W[(size_t)(r + 3) * Kd + k] * u[r + 3];
        }
        for (; r < r1; ++r) a0 += W[(size_t)r * Kd + k] * u[r];
        __hip_atomic_store(partial + (size_t)blockIdx.y * Kd + k, (a0 + a1) + (a2 + a3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned total = (unsigned)((Kd + 1023) / 1024) * SN_NRB;
    if (threadIdx.x == 0) flag = __hip_atomic_fetch_add(cnt + 2 * l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag != total - 1) return;
    if (threadIdx.x == 0) __hip_atomic_store(cnt + 2 * l, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // snm_v_final_kernel's body (1024 threads)
    float* v = p.v[l];
    float* v_snap = p.v_snap[l];
    float ss = 0.f;
    for (int kk = threadIdx.x; kk < Kd; kk += 1024) {
        float t = 0.f;
        for (int r = 0; r < SN_NRB; ++r) t += __hip_atomic_load(partial + (size_t)r * Kd + kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[kk] = t;
        ss += t * t;
    }
    const float nrm = sqrtf(block_sum(ss, sm));
    const float inv = 1.f / fmaxf(nrm, p.eps);
    for (int kk = threadIdx.x; kk < Kd; kk += 1024) {
        const float vn = v[kk] * inv;
        v[kk] = vn;
        if (v_snap) v_snap[kk] = vn;
    }
}

// block_sum of a 1024-thread workgroup, computed by 256 threads: virtual thread vt = 64 * vw + lane of virtual wave vw = w, w + 4, w + 8,
// w + 12 belongs to physical wave w; wave sums go to sm[vw], then the 16 are added in order (block_sum's order)
template <typename F>
__device__ __forceinline__ float block_sum_as_1024(F&& value_of_virtual_thread, float* sm) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int vw = w + 4 * q;
        const float s = wave_sum(value_of_virtual_thread(64 * vw + lane));
        if (lane == 0) sm[vw] = s;
    }
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += sm[i];
    return r;
}

__global__ __launch_bounds__(256) void snm2_wv_u_kernel(const SnMulti p, unsigned* __restrict__ cnt) {
    __shared__ float sm[16];
    __shared__ unsigned flag;
    const int l = blockIdx.z, R = p.R[l], Kd = p.Kd[l];
    const int r = blockIdx.x;
    if (r >= R) return;
    const float* __restrict__ W = p.w[l] + (size_t)r * Kd;
    const float* __restrict__ v = p.v[l];
    float* s = p.ws + p.ws_off[l] + (size_t)SN_NRB * Kd;
    float a = 0.f;
    for (int k = threadIdx.x; k < Kd; k += blockDim.x) a += W[k] * v[k];
    const float tot = block_sum(a, sm);
    if (threadIdx.x == 0) {
        __hip_atomic_store(s + r, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        flag = __hip_atomic_fetch_add(cnt + 2 * l + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (flag != (unsigned)(R - 1)) return;
    if (threadIdx.x == 0) __hip_atomic_store(cnt + 2 * l + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // snm_u_final_kernel's training body as its 1024 threads would run it
    float* u = p.u[l];
    auto sr = [&](int i) { return __hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const float nrm = sqrtf(block_sum_as_1024([&](int vt) { float ss = 0.f; for (int i = vt; i < R; i += 1024) { const float x = sr(i); ss += x * x; } return ss; }, sm));
    const float inv = 1.f / fmaxf(nrm, p.eps);
    const float sg = block_sum_as_1024([&](int vt) {
        float d = 0.f;
        for (int i = vt; i < R; i += 1024) {
            const float x = sr(i), un = x * inv;
            u[i] = un;
            if (p.u_snap[l]) p.u_snap[l][i] = un;
            d += un * x;
        }
        return d; }, sm);
    if (threadIdx.x == 0) p.sigma[l][0] = sg;
}

extern "C" size_t eg_sn_multi_ws_floats(const eg_sn_layer* layers, int nlayers) {
    size_t tot = 0;
    for (int i = 0; i < nlayers; ++i) tot += (size_t)SN_NRB * layers[i].Kd + layers[i].R;
    return tot;
}

extern "C" int eg_sn_power_iter_multi2(const eg_sn_layer* layers, int nlayers, float* ws, unsigned int* counters, int training, float eps,
                                       eg_stream_t s) {
    EG_REQUIRE(layers && ws && nlayers > 0 && nlayers <= SN_MAXL, "eg_sn_power_iter_multi: bad argument");
    SnMulti p;
    memset(&p, 0, sizeof(p));
    long long off = 0;
    int maxK = 0, maxR = 0;
    for (int i = 0; i < nlayers; ++i) {
        const eg_sn_layer& L = layers[i];
        EG_REQUIRE(L.w && L.u && L.v && L.sigma && L.R > 0 && L.Kd > 0, "eg_sn_power_iter_multi: bad layer %d", i);
        p.w[i] = L.w; p.u[i] = L.u; p.v[i] = L.v; p.sigma[i] = L.sigma; p.u_snap[i] = L.u_snap; p.v_snap[i] = L.v_snap;
        p.R[i] = L.R; p.Kd[i] = L.Kd; p.ws_off[i] = off;
        off += (long long)SN_NRB * L.Kd + L.R;
        maxK = L.Kd > maxK ? L.Kd : maxK;
        maxR = L.R > maxR ? L.R : maxR;
    }
    p.ws = ws; p.eps = eps;
    hipStream_t st = (hipStream_t)s;
    if (training && counters) {
        hipLaunchKernelGGL(snm2_wtu_v_kernel, dim3(cdiv(maxK, 1024), SN_NRB, nlayers), dim3(1024), 0, st, p, counters);
        hipLaunchKernelGGL(snm2_wv_u_kernel, dim3(maxR, 1, nlayers), dim3(256), 0, st, p, counters);
        EG_LAUNCH_CHECK();
        return 0;
    }
    if (training) {
        hipLaunchKernelGGL(snm_wtu_partial_kernel, dim3(cdiv(maxK, 256), SN_NRB, nlayers), dim3(256), 0, st, p);
        hipLaunchKernelGGL(snm_v_final_kernel, dim3(1, 1, nlayers), dim3(1024), 0, st, p);
    }
    hipLaunchKernelGGL(snm_wv_kernel, dim3(maxR, 1, nlayers), dim3(256), 0, st, p);
    hipLaunchKernelGGL(snm_u_final_kernel, dim3(1, 1, nlayers), dim3(1024), 0, st, p, training);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_sn_power_iter_multi(const eg_sn_layer* layers, int nlayers, float* ws, int training, float eps, eg_stream_t s) {
    return eg_sn_power_iter_multi2(layers, nlayers, ws, nullptr, training, eps, s);
}

// ---- Adam -----------------------------------------------------------------------------------------
__global__ void adam_tick_kernel(int* step) { step[0] += 1; }

#include "adam.h"

// elements [0, head) and [head + 4*nvec, n) one per thread (the unaligned ends of an arena slice), the middle as float4; the four arrays
// are slices of sibling arenas at the same element offset, so one `head` aligns all of them.  ZERO: the gradient is cleared in the
// same pass (optimizer.zero_grad() of the next backward pass that accumulates into it).
template <bool ZERO, bool V2>
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, size_t head,
                            size_t nvec, float lr, float b1, float b2, float eps, const int* __restrict__ step) {
    const AdamCoef c = adam_coef(lr, b1, b2, eps, step);
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    // 8-byte vectors (two per 16-byte slot `nvec` counts): with 16-byte ones the kernel needs 50-54 VGPRs, with these 48 -- what a wave
    // may use on a SIMD beside the two 232-register waves of a resident 8-wave GEMM workgroup: the updates on the optimizer
    // lane run beside the next sub-step's GEMMs.  Same element arithmetic.
    if constexpr (!V2) {                                 // (the 16-byte form: A/B runs, EG_ADAM_V2=0)
        float4* p4 = reinterpret_cast<float4*>(p + head);
        float4* g4 = reinterpret_cast<float4*>(g + head);
        float4* m4 = reinterpret_cast<float4*>(m + head);
        float4* v4 = reinterpret_cast<float4*>(v + head);
        for (size_t i = tid; i < nvec; i += nth) {
            float4 pi = p4[i], mi = m4[i], vi = v4[i];
            const float4 gi = g4[i];
            adam_elem(pi.x, gi.x, mi.x, vi.x, c);
            adam_elem(pi.y, gi.y, mi.y, vi.y, c);
            adam_elem(pi.z, gi.z, mi.z, vi.z, c);
            adam_elem(pi.w, gi.w, mi.w, vi.w, c);
            m4[i] = mi;
            v4[i] = vi;
            p4[i] = pi;
            if (ZERO) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
    float2* p2 = reinterpret_cast<float2*>(p + head);
    float2* g2 = reinterpret_cast<float2*>(g + head);
    float2* m2 = reinterpret_cast<float2*>(m + head);
    float2* v2 = reinterpret_cast<float2*>(v + head);
    for (size_t i = tid; i < 2 * nvec; i += nth) {
        float2 pi = p2[i], mi = m2[i], vi = v2[i];
        const float2 gi = g2[i];
        adam_elem(pi.x, gi.x, mi.x, vi.x, c);
        adam_elem(pi.y, gi.y, mi.y, vi.y, c);
        m2[i] = mi;
        v2[i] = vi;
        p2[i] = pi;
        if (ZERO) g2[i] = make_float2(0.f, 0.f);
    }
    }
    const size_t tail0 = head + 4 * nvec, nends = head + (n - tail0);
    for (size_t e = tid; e < nends; e += nth) {
        const size_t i = e < head ? e : tail0 + (e - head);
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, g[i], mi, vi, c);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
        if (ZERO) g[i] = 0.f;
    }
}

static void launch_adam(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int* step, int tick,
                        bool zero, hipStream_t st) {
    if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
    size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;
    const bool same = (((uintptr_t)p ^ (uintptr_t)g) & 15) == 0 && (((uintptr_t)p ^ (uintptr_t)m) & 15) == 0 && (((uintptr_t)p ^ (uintptr_t)v) & 15) == 0;
    if (head > n) head = n;
    const size_t nvec = same ? (n - head) / 4 : 0;
    if (!same) head = 0;
    const size_t work = same ? (nvec > 8 ? nvec : 8) : n;          // differently aligned slices: every element through the scalar loop
    const int blocks = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
    static const bool v2 = [] { const char* e = getenv("EG_ADAM_V2"); return !(e && atoi(e) == 0); }();
    if (v2) {
        if (zero) hipLaunchKernelGGL((adam_kernel<true, true>), dim3(blocks), dim3(256), 0, st, p, g, m, v, n, head, nvec, lr, b1, b2, eps, step);
        else hipLaunchKernelGGL((adam_kernel<false, true>), dim3(blocks), dim3(256), 0, st, p, g, m, v, n, head, nvec, lr, b1, b2, eps, step);
    } else {
        if (zero) hipLaunchKernelGGL((adam_kernel<true, false>), dim3(blocks), dim3(256), 0, st, p, g, m, v, n, head, nvec, lr, b1, b2, eps, step);
        else hipLaunchKernelGGL((adam_kernel<false, false>), dim3(blocks), dim3(256), 0, st, p, g, m, v, n, head, nvec, lr, b1, b2, eps, step);
    }
}

/* In-place Adam over a flat fp32 arena.  `step` is a device int32 incremented by this call (so a captured
 * graph replays with the right bias correction). */
extern "C" int eg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int* step,
                            int tick, eg_stream_t s) {
    EG_REQUIRE(p && g && m && v && step, "eg_adam_step: null pointer");
    launch_adam(p, const_cast<float*>(g), m, v, n, lr, b1, b2, eps, step, tick, false, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

/* The same update on a slice of the arena (one gradient bucket), clearing the gradient slice in the same pass when `zero_grad`:
 * torch.optim.Adam.step() + optimizer.zero_grad() of the slice (celebA/EAD-GAN_celebA.py:344-345,365-366,400-401). */
extern "C" int eg_adam_step_zero(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int* step,
                                 int tick, int zero_grad, eg_stream_t s) {
    EG_REQUIRE(p && g && m && v && step, "eg_adam_step_zero: null pointer");
    if (n == 0) return 0;
    launch_adam(p, g, m, v, n, lr, b1, b2, eps, step, tick, zero_grad != 0, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_adam_tick(int* step, eg_stream_t s) {
    EG_REQUIRE(step, "eg_adam_tick: null pointer");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, step);
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
