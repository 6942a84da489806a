// Internal helpers shared by the HIP translation units of libeadgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/eadgan_hip.h"

// ---- error plumbing: never throw / abort across the C ABI ------------------------------------
void eg_set_error(const char* fmt, ...);
#define EG_FAIL(code, ...)        \
    do {                          \
        eg_set_error(__VA_ARGS__); \
        return (code);            \
    } while (0)
#define EG_REQUIRE(cond, ...) \
    do {                      \
        if (!(cond)) EG_FAIL(-1, __VA_ARGS__); \
    } while (0)
#define EG_LAUNCH_CHECK()                                                             \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) EG_FAIL((int)e__, "%s: %s", __func__, hipGetErrorString(e__)); \
    } while (0)

// ---- element types -----------------------------------------------------------------------------
typedef uint16_t bf16_t;   // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; NaN stays NaN (plain integer trick is wrong for NaN, see MI355X guide)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct Elt;
template <> struct Elt<float> {
    static constexpr int VEC = 4;   // elements per 16-byte chunk
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
typedef _Float16 f16_t;    // IEEE binary16 (config "colored dSprites fp16"); same MFMA rate as bf16, fp32 accumulate
template <> struct Elt<f16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }     // round to nearest even
};
template <> struct Elt<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

static inline int ilog2_exact(int v) {   // -1 if v is not a power of two
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// activation codes (shared by kernels and the ABI header)
__device__ __forceinline__ float eg_act(float v, int act, float slope) {
    switch (act) {
        case EG_ACT_LRELU: return v > 0.f ? v : v * slope;
        case EG_ACT_RELU: return v > 0.f ? v : 0.f;
        case EG_ACT_TANH: return tanhf(v);
        case EG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}
// derivative expressed through the *output* a = act(z)
__device__ __forceinline__ float eg_act_grad_from_out(float a, int act, float slope) {
    switch (act) {
        case EG_ACT_LRELU: return a > 0.f ? 1.f : slope;
        case EG_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        case EG_ACT_TANH: return 1.f - a * a;
        case EG_ACT_SIGMOID: return a * (1.f - a);
        default: return 1.f;
    }
}

// wave (64 lanes) and block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// deterministic block sum for blockDim.x <= 1024 (multiple of 64); result valid in all threads
__device__ __forceinline__ float block_sum(float v, float* sm /* >= 16 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}
