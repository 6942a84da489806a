// Shared pieces of the NT implicit-GEMM kernels (igemm.hip, igemm_nt8.hip): launch parameters, LDS image of a K row,
// MFMA step, buffer-descriptor LDS-DMA helpers and the fused epilogue through LDS.  gfx950 only.
#pragma once
#include <type_traits>

#include "eg_common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ------------------------------------------------------------------------------------------------
// geometry derived from eg_conv
// ------------------------------------------------------------------------------------------------
struct NtPhase {
    int TH, TW, dy0, dys, dx0, dxs, ooy, oox, K, Kpad;
    long long w_off;   // element offset of this phase's packed weights
};
struct NtParams {
    const void* src;
    const void* wp;
    void* dst;
    const float* bias;
    const float* sigma;
    const void* mask;
    int B, H, W, C;   // gathered source tensor (NHWC)
    int lOH, lOW;     // log2 of the output lattice
    int sy, sx, up;
    int N, bias_mod;
    int DH, DW, osy, osx;
    int act;
    float slope;
    int mask_act;
    float mask_slope;
    int out_mode;
    int sigma_rows;
    int M;
    float* part;      // split-K partial tiles [split][phase][Mpad][N] fp32 (nsplit > 1)
    int xcd_remap;    // igemm_nt_buf: 1 = every XCD gets a contiguous range of M tiles (workgroups are dispatched round robin over the 8 XCDs)
    size_t part_bytes;
    int nsplit;
    unsigned* split_cnt;   // igemm_nt8s with K splits: arrival counters, one per output tile (tail of the caller's split-K scratch; zero between launches)
    NtPhase ph[4];
};
// the last EG_SPLIT_CNT_BYTES of the caller-lent split-K scratch hold the arrival counters of igemm_nt8s's in-kernel reduction (at most 224
// tiles split); no planner ever places partial tiles there.  The scratch must be zero when it is first lent; every launch leaves the
// counters at zero again.
#define EG_SPLIT_CNT_BYTES 4096

// patch geometry of igemm_nt8p_kernel (igemm_nt8.hip): filter taps grouped into classes that walk one pixel lattice
struct NtClass { int oy0, ox0, AH, AW, ty0, tys, tx0, txs; };       // source = stride * lattice + o0; tap(ay, ax) = (ty0 + ay*tys, tx0 + ax*txs)
struct Nt8pPhase { int ncls; NtClass cls[4]; };
struct Nt8pGeom {
    int nimg, OHt;                  // images per 256-row tile; output rows of one image per tile
    int PH, PW, npix, npp;          // patch lattice per image, pixels per patch, DMA pieces per wave
    unsigned inv_pw, inv_plane;     // x / PW == (x * inv_pw) >> 20, x / (PH*PW) == (x * inv_plane) >> 20  for x < 512
    Nt8pPhase ph[4];
};
#define EG_P8P_SLOTS 448

// launch parameters of igemm_tn8_kernel (igemm_tn8.hip): weight gradients of the 4x4 / stride-2 layers, taps grouped by input parity
struct Tn8Params {
    const void* P;      // [M][N]
    const void* src;    // [B,H,W,C]
    float* slab;        // [nsplit][N][16][C]
    int B, H, W, C, N;
    int lOH, lOW, M;
    int rows_per_split; // multiple of 64
    int ntn, ntc, ch;   // tiles; channels per tile on both sides (128 or 64)
    int nimg, OHt, PH, PW, npix, npp;       // per 64-row K step: images, output rows per image, patch lattice, pixels, X pieces per wave
    unsigned inv_pw, inv_plane;             // x / PW, x / (PH * PW) as (x * inv) >> 20 for x < 512
};

#define EG_TN8_XSLOTS 136                   // patch pixel slots per stage (34 pieces of 4 pixels x 256 B, or 17 of 8 pixels x 128 B)


static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int bk_of(int dtype) { return dtype == EG_F32 ? 32 : 64; }
static inline int vec_of(int dtype) { return dtype == EG_F32 ? 4 : 8; }

// ------------------------------------------------------------------------------------------------
// LDS helpers: rows of 128 bytes, 16-byte chunk c of row r lives at chunk (c ^ ((r>>1)&7))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// D = B-fragment x A-fragment: the weight-panel rows (n) land on the accumulator ROW index (4 consecutive n per lane),
// the gathered rows (m) on the lane -> the epilogue packs 4 consecutive output channels per lane.
template <typename T>
__device__ __forceinline__ void mfma_step(const uint4& a, const uint4& b, f32x4& acc) {
    if constexpr (std::is_same<T, float>::value) {
        const float* af = reinterpret_cast<const float*>(&a);
        const float* bf = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[e], af[e], acc, 0, 0, 0);
    } else if constexpr (std::is_same<T, f16_t>::value) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, b), __builtin_bit_cast(f16x8_t, a), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), acc, 0, 0, 0);
    }
}

// shared epilogue of the LDS-DMA kernels: a BM x BNW window of the fp32 tile goes through LDS (XOR-swizzled 16-byte chunks), then 16-byte
// vector stores with the fused 1/sigma, bias, activation and activation-gradient mask.  Callers __syncthreads() before (the K loop's LDS
// is reused).  row0 / col0: first row / first window column of the calling wave's accumulators (col0 < 0: wave outside the window);
// nw0: global column of the window's first column; NT threads per workgroup.
template <typename T, int BM, int BNW, int TM, int TN, int NT>
__device__ __forceinline__ void nt_epilogue_lds(const NtParams& p, const NtPhase& ph, f32x4 (&acc)[TM][TN], char* smem, int m0, int nw0, int row0,
                                                int col0, int tid, int frow, int fq) {
    constexpr int VEC = Elt<T>::VEC;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    constexpr int SW = BNW / 4 < 32 ? BNW / 4 - 1 : 31;     // 16-byte fp32 chunks per tile row - 1
    float* ct = reinterpret_cast<float*>(smem);
    const EgActFast af = eg_act_fast(p.act, p.slope);
    const EgGradFast gf = eg_grad_fast(p.mask_act, p.mask_slope);
    if (col0 >= 0) {
        float bias[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nw0 + col0 + j * 16 + fq * 4 + r;
                bias[j][r] = (p.bias && n < p.N) ? p.bias[p.bias_mod ? n % p.bias_mod : n] : 0.f;
            }
        eg_if_fast(af.special, [&](auto fast) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = row0 + i * 16 + frow;
                const int mrow = min(m0 + row, p.M - 1);
                const float inv_sigma = p.sigma ? 1.f / p.sigma[p.sigma_rows ? mrow / p.sigma_rows : 0] : 1.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = col0 + j * 16 + fq * 4;
                    float4 v;
                    float* ve = reinterpret_cast<float*>(&v);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = __fmul_rn(acc[i][j][r], inv_sigma);
                        if (p.bias && nw0 + nl + r < p.N) x = __fadd_rn(x, bias[j][r]);
                        if constexpr (decltype(fast)::value) ve[r] = eg_act_apply(x, af);
                        else ve[r] = eg_act(x, p.act, p.slope);
                    }
                    *reinterpret_cast<float4*>(ct + row * BNW + (((nl >> 2) ^ (row & SW)) << 2)) = v;
                }
            }
        });
    }
    __syncthreads();
    constexpr int VPR = BNW / VEC;
    constexpr int RPP = NT / VPR;
    const int vc = tid % VPR, vr = tid / VPR;
    const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
    const int n = nw0 + vc * VEC;
    if (n < p.N) {
#pragma unroll 4
        for (int row = vr; row < BM; row += RPP) {
            const int m = m0 + row;
            if (m >= p.M) break;
            const int b = m >> (p.lOW + p.lOH);
            const int y = ((m >> p.lOW) & OHm) * p.osy + ph.ooy;
            const int x = (m & OWm) * p.osx + ph.oox;
            const size_t o = (((size_t)b * p.DH + y) * p.DW + x) * p.N + n;
            float f[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const int chunk = (vc * (VEC / 4) + q) ^ (row & SW);
                const float4 v = *reinterpret_cast<const float4*>(ct + row * BNW + (chunk << 2));
                f[q * 4 + 0] = v.x; f[q * 4 + 1] = v.y; f[q * 4 + 2] = v.z; f[q * 4 + 3] = v.w;
            }
            if (mask) {
                const uint4 mv = *reinterpret_cast<const uint4*>(mask + o);
                const T* me = reinterpret_cast<const T*>(&mv);
                if (!gf.special) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) f[q] *= eg_grad_apply(Elt<T>::ld(me + q), gf);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) f[q] *= eg_act_grad_from_out(Elt<T>::ld(me + q), p.mask_act, p.mask_slope);
                }
            }
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int q = 0; q < VEC; ++q) Elt<T>::st(oe + q, f[q]);
            *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.dst) + o) = ov;
        }
    }
}

// Epilogue operands fetched ahead of the last K step (igemm_nt_buf): 1/sigma per accumulator row block, the bias of the lane's 4 x TN
// columns and the activation-gradient mask vectors of the first PF store iterations -- their global-load latency (the epilogue's
// critical path: ~40 % of a 16-step launch) overlaps the last MFMA block and the LDS staging.
template <typename T, int TM, int TN, int PF>
struct NtEpiPre {
    float inv_sigma[TM];
    float bias[TN][4];
    uint4 mask[PF];
};

template <typename T, int BM, int BNW, int TM, int TN, int NT, int PF>
__device__ __forceinline__ void nt_epi_prefetch(NtEpiPre<T, TM, TN, PF>& e, const NtParams& p, const NtPhase& ph, int m0, int nw0, int row0, int col0,
                                                int tid, int frow, int fq) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int VPR = BNW / VEC, RPP = NT / VPR;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow = min(m0 + row0 + i * 16 + frow, p.M - 1);
        e.inv_sigma[i] = p.sigma ? p.sigma[p.sigma_rows ? mrow / p.sigma_rows : 0] : 1.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nw0 + col0 + j * 16 + fq * 4 + r;
            e.bias[j][r] = (p.bias && n < p.N) ? p.bias[p.bias_mod ? n % p.bias_mod : n] : 0.f;
        }
    const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
    const int vc = tid % VPR, vr = tid / VPR;
    const int n = nw0 + vc * VEC;
#pragma unroll
    for (int it = 0; it < PF; ++it) {
        const int m = m0 + vr + it * RPP;
        e.mask[it] = make_uint4(0, 0, 0, 0);
        if (mask && n < p.N && m < p.M) {
            const int b = m >> (p.lOW + p.lOH);
            const int y = ((m >> p.lOW) & OHm) * p.osy + ph.ooy;
            const int x = (m & OWm) * p.osx + ph.oox;
            e.mask[it] = *reinterpret_cast<const uint4*>(mask + (((size_t)b * p.DH + y) * p.DW + x) * p.N + n);
        }
    }
}

// nt_epilogue_lds with the operands of NtEpiPre (same arithmetic, same results)
template <typename T, int BM, int BNW, int TM, int TN, int NT, int PF>
__device__ __forceinline__ void nt_epilogue_lds_pre(const NtEpiPre<T, TM, TN, PF>& e, const NtParams& p, const NtPhase& ph, f32x4 (&acc)[TM][TN],
                                                    char* smem, int m0, int nw0, int row0, int col0, int tid, int frow, int fq) {
    constexpr int VEC = Elt<T>::VEC;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    constexpr int SW = BNW / 4 < 32 ? BNW / 4 - 1 : 31;     // 16-byte fp32 chunks per tile row - 1
    float* ct = reinterpret_cast<float*>(smem);
    const EgActFast af = eg_act_fast(p.act, p.slope);
    const EgGradFast gf = eg_grad_fast(p.mask_act, p.mask_slope);
    eg_if_fast(af.special, [&](auto fast) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = row0 + i * 16 + frow;
            const float inv_sigma = p.sigma ? 1.f / e.inv_sigma[i] : 1.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = col0 + j * 16 + fq * 4;
                float4 v;
                float* ve = reinterpret_cast<float*>(&v);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = __fmul_rn(acc[i][j][r], inv_sigma);
                    if (p.bias && nw0 + nl + r < p.N) x = __fadd_rn(x, e.bias[j][r]);
                    if constexpr (decltype(fast)::value) ve[r] = eg_act_apply(x, af);
                    else ve[r] = eg_act(x, p.act, p.slope);
                }
                *reinterpret_cast<float4*>(ct + row * BNW + (((nl >> 2) ^ (row & SW)) << 2)) = v;
            }
        }
    });
    __syncthreads();
    constexpr int VPR = BNW / VEC;
    constexpr int RPP = NT / VPR;
    constexpr int NIT = BM / RPP;
    const int vc = tid % VPR, vr = tid / VPR;
    const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
    const int n = nw0 + vc * VEC;
    if (n < p.N) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = vr + it * RPP;
            const int m = m0 + row;
            if (m >= p.M) break;
            const int b = m >> (p.lOW + p.lOH);
            const int y = ((m >> p.lOW) & OHm) * p.osy + ph.ooy;
            const int x = (m & OWm) * p.osx + ph.oox;
            const size_t o = (((size_t)b * p.DH + y) * p.DW + x) * p.N + n;
            float f[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const int chunk = (vc * (VEC / 4) + q) ^ (row & SW);
                const float4 v = *reinterpret_cast<const float4*>(ct + row * BNW + (chunk << 2));
                f[q * 4 + 0] = v.x; f[q * 4 + 1] = v.y; f[q * 4 + 2] = v.z; f[q * 4 + 3] = v.w;
            }
            if (mask) {
                const uint4 mv = it < PF ? e.mask[it < PF ? it : 0] : *reinterpret_cast<const uint4*>(mask + o);
                const T* me = reinterpret_cast<const T*>(&mv);
                if (!gf.special) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) f[q] *= eg_grad_apply(Elt<T>::ld(me + q), gf);
                } else {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) f[q] *= eg_act_grad_from_out(Elt<T>::ld(me + q), p.mask_act, p.mask_slope);
                }
            }
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int q = 0; q < VEC; ++q) Elt<T>::st(oe + q, f[q]);
            *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.dst) + o) = ov;
        }
    }
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4_t eg_make_srd(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

#define EG_OOB 0x80000000u

// one 1-KiB LDS-DMA piece (64 lanes x 16 B from per-lane offsets into `srd`, + the SGPR offset) to LDS [lds, lds + 1 KiB).  Issued
// through inline asm so that hipcc does not count it: the caller owns the waits (counted s_waitcnt vmcnt + s_barrier before any wave
// reads the bytes).  M0 is saved and restored inside the statement.
__device__ __forceinline__ void eg_bufdma1s(const u32x4_t srd, unsigned v0, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "s"(srd), "s"(soff), "s"(lds)
        : "memory");
}

// the same without saving M0 (three instructions): for kernels in which nothing else reads M0 -- gfx9 LDS instructions do not, and the
// kernels that use this form contain no GWS, s_movrel or s_sendmsg code.  LDS address = base + OFF.
template <int OFF>
__device__ __forceinline__ void eg_bufdma1f(const u32x4_t srd, unsigned v0, unsigned soff, unsigned lds_base) {
    asm volatile(
        "s_add_u32 m0, %3, %4\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, %2 offen lds"
        :
        : "v"(v0), "s"(srd), "s"(soff), "s"(lds_base), "i"(OFF)
        : "memory", "scc");
}

template <int STRIDE>
__device__ __forceinline__ void eg_bufdma4s(const u32x4_t srd, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %4, %5, %6 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(srd), "s"(soff), "s"(lds), "i"(STRIDE)
        : "memory", "scc");
}
template <int STRIDE>
__device__ __forceinline__ void eg_bufdma2s(const u32x4_t srd, unsigned v0, unsigned v1, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
        "s_add_u32 m0, m0, %6\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "s"(srd), "s"(soff), "s"(lds), "i"(STRIDE)
        : "memory", "scc");
}
