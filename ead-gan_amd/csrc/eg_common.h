// Internal helpers shared by the HIP translation units of libeadgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

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
// round-to-nearest-even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (what the __bf16 cast compiles to).  The integer form of the same
// rounding (add 0x7fff + lsb, shift, NaN test) is ~7 VALU operations per element in every epilogue and elementwise kernel; fp16, which
// always had its hardware convert, ran the whole CelebA step 3 % faster than bf16 for that reason alone.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
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
        case EG_ACT_TANH: return fmaf(-a, a, 1.f);      // explicit: the same bits in translation units built with and without fp contraction
        case EG_ACT_SIGMOID: return a * (1.f - a);
        default: return 1.f;
    }
}

// The same two functions for code that applies them to many elements of one launch: `act` is uniform, so the switch above costs a
// chain of scalar branches PER ELEMENT (the 256 x 128 epilogue spent 12-25k cycles in them).  NONE / LRELU / RELU become two selects
// on loop-invariant operands with bit-identical results (x * 1 == x; RELU keeps its +0); tanh / sigmoid keep the switch -- callers
// test `special` once, outside their loops.
struct EgActFast { float neg; bool relu, special; };
__device__ __forceinline__ EgActFast eg_act_fast(int act, float slope) {
    return {act == EG_ACT_LRELU ? slope : 1.f, act == EG_ACT_RELU, act >= EG_ACT_TANH};
}
__device__ __forceinline__ float eg_act_apply(float v, const EgActFast& a) {
    float alt = v * a.neg;
    alt = a.relu ? 0.f : alt;
    return v > 0.f ? v : alt;
}
struct EgGradFast { float neg; bool special; };
__device__ __forceinline__ EgGradFast eg_grad_fast(int act, float slope) {
    return {act == EG_ACT_LRELU ? slope : (act == EG_ACT_RELU ? 0.f : 1.f), act >= EG_ACT_TANH};
}
__device__ __forceinline__ float eg_grad_apply(float a, const EgGradFast& g) { return a > 0.f ? 1.f : g.neg; }
// f(std::true_type) on the fast path, f(std::false_type) when the launch uses tanh / sigmoid
template <typename F>
__device__ __forceinline__ void eg_if_fast(bool special, F&& f) {
    if (special) f(std::false_type{});
    else f(std::true_type{});
}

// wave (64 lanes) and block reductions
// Lane distances 1, 2, 4, 8 as DPP adds (register to register: quad_perm, row_half_mirror, row_mirror -- after each step all lanes of a
// group hold the group's sum, so the mirrored partner carries the same value the xor partner would), then row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3, and lane 63's total broadcast through an SGPR.  The __shfl_xor form is six dependent ds_bpermute round
// trips through LDS and made the reduction-heavy small kernels latency chains.  (gfx950's v_permlane16/32_swap would do the last two steps
// symmetrically, but hipcc 7.2 lowers __builtin_amdgcn_permlane*_swap with both results in one register.)  Every lane returns the same
// value; all 64 lanes must be active.
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float eg_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += eg_dpp<0xB1>(v);             // quad_perm [1,0,3,2]
    v += eg_dpp<0x4E>(v);             // quad_perm [2,3,0,1]
    v += eg_dpp<0x141>(v);            // row_half_mirror
    v += eg_dpp<0x140>(v);            // row_mirror
    v += eg_dpp<0x142, 0xA>(v);       // row_bcast:15 -> rows 1, 3
    v += eg_dpp<0x143, 0xC>(v);       // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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
