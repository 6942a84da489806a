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
    // column statistics of the stored tile (eg_epilogue.stat_*; igemm_nt8s only)
    int stat_mode, stat_nrb;
    float* stat_out;
    const void* stat_aux;
    const float* stat_p[4];
    int stat_act;
    float stat_slope;
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
    uint4 mask[PF > 0 ? PF : 1];
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
// first half of nt_epilogue_lds_pre: the accumulators with 1/sigma, bias and activation applied go to the fp32 tile in LDS
template <typename T, int BNW, int TM, int TN, int PF>
__device__ __forceinline__ void nt_epilogue_stage_pre(const NtEpiPre<T, TM, TN, PF>& e, const NtParams& p, f32x4 (&acc)[TM][TN], char* smem, int nw0,
                                                      int row0, int col0, int frow, int fq) {
    constexpr int SW = BNW / 4 < 32 ? BNW / 4 - 1 : 31;     // 16-byte fp32 chunks per tile row - 1
    float* ct = reinterpret_cast<float*>(smem);
    const EgActFast af = eg_act_fast(p.act, p.slope);
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
}

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

// nt_epilogue_lds_pre + column statistics of the stored tile (NtParams.stat_*, eg_epilogue.stat_mode in the ABI header): the same staging,
// the same stored values (EG_STAT_BN_BWD stores dy = da * act'(bn(z)) instead of da), plus two sums per column over the tile's BM rows,
// taken from the values as they are stored (rounded to T).  Order of the sums: a thread adds its NIT rows top down, then the RPP row
// lanes of a column are added in lane order -> deterministic.  Whole tiles only (the planner guarantees m0 + BM <= M); 16-bit types.
// rb = phase * tiles_m + m_tile: row block of the tile in stat_out; n_tile / tiles_n: for the per-tile dot of EG_STAT_SN_BIAS.
template <typename T, int BM, int BNW, int TM, int TN, int NT, int PF>
__device__ __forceinline__ void nt_epilogue_lds_stat(const NtEpiPre<T, TM, TN, PF>& e, const NtParams& p, const NtPhase& ph, f32x4 (&acc)[TM][TN],
                                                     char* smem, int m0, int nw0, int row0, int col0, int tid, int frow, int fq, int rb, int n_tile,
                                                     int tiles_n) {
    constexpr int VEC = Elt<T>::VEC;
    static_assert(VEC == 8, "16-bit element types");
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    constexpr int SW = BNW / 4 < 32 ? BNW / 4 - 1 : 31;
    float* ct = reinterpret_cast<float*>(smem);
    const EgGradFast gf = eg_grad_fast(p.mask_act, p.mask_slope);
    nt_epilogue_stage_pre<T, BNW, TM, TN, PF>(e, p, acc, smem, nw0, row0, col0, frow, fq);
    __syncthreads();
    constexpr int VPR = BNW / VEC, RPP = NT / VPR, NIT = BM / RPP;
    const int vc = tid % VPR, vr = tid / VPR;
    const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
    const T* __restrict__ aux = reinterpret_cast<const T*>(p.stat_aux);
    const int n = nw0 + vc * VEC;
    const int mode = p.stat_mode;
    auto round_t = [](float v) { T t; Elt<T>::st(&t, v); return Elt<T>::ld(&t); };
    float s1[VEC], s2[VEC], k0[VEC], k1[VEC], k2[VEC], k3[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) { s1[q] = 0.f; s2[q] = 0.f; k0[q] = 0.f; k1[q] = 0.f; k2[q] = 0.f; k3[q] = 0.f; }
    if (mode == EG_STAT_MOMENTS) {
        // pivot = the column's value in the tile's first row (any pivot is exact; this one keeps the sums small)
#pragma unroll
        for (int q = 0; q < VEC; ++q) k0[q] = round_t(ct[vc * VEC + q]);
    } else if (mode == EG_STAT_BN_BWD) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const float mu = p.stat_p[0][n + q], is = p.stat_p[1][n + q], g = p.stat_p[2][n + q], b = p.stat_p[3][n + q];
            k0[q] = mu; k1[q] = is; k2[q] = __fmul_rn(g, is); k3[q] = __fsub_rn(b, __fmul_rn(mu, k2[q]));
        }
    } else {
#pragma unroll
        for (int q = 0; q < VEC; ++q) k0[q] = p.stat_p[0][n + q];
    }
    const EgGradFast sgf = eg_grad_fast(p.stat_act, p.stat_slope);
    const float inv_slope = p.stat_slope != 0.f ? 1.f / p.stat_slope : 1.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = vr + it * RPP;
        const int m = m0 + row;
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
        uint4 mv = make_uint4(0, 0, 0, 0);
        if (mask) {
            mv = it < PF ? e.mask[it < PF ? it : 0] : *reinterpret_cast<const uint4*>(mask + o);
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
        if (mode == EG_STAT_MOMENTS) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                Elt<T>::st(oe + q, f[q]);
                const float d = Elt<T>::ld(oe + q) - k0[q];
                s1[q] += d;
                s2[q] = fmaf(d, d, s2[q]);
            }
        } else if (mode == EG_STAT_BN_BWD) {
            const uint4 zv = *reinterpret_cast<const uint4*>(aux + o);
            const T* ze = reinterpret_cast<const T*>(&zv);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float z = Elt<T>::ld(ze + q);
                const float xh = (z - k0[q]) * k1[q];
                const float dy = round_t(f[q]) * eg_grad_apply(fmaf(z, k2[q], k3[q]), sgf);      // the forward's activation input: z * (g * is) + (b - mu * g * is)
                Elt<T>::st(oe + q, dy);
                s1[q] += dy;
                s2[q] = fmaf(dy, xh, s2[q]);
            }
        } else {
            const T* me = reinterpret_cast<const T*>(&mv);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                Elt<T>::st(oe + q, f[q]);
                const float g = Elt<T>::ld(oe + q);
                float a = Elt<T>::ld(me + q);
                a = a > 0.f ? a : a * inv_slope;
                s1[q] += g;
                s2[q] = fmaf(g, a - k0[q], s2[q]);
            }
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.dst) + o) = ov;
    }
    __syncthreads();                                   // every row of the tile has been read: rows 1.. become the reduction scratch
    float* red = ct + BNW;                             // [2][RPP][BNW]; row 0 of the tile (the pivots) stays
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
        *reinterpret_cast<float4*>(red + (size_t)vr * BNW + vc * VEC + q * 4) = make_float4(s1[q * 4], s1[q * 4 + 1], s1[q * 4 + 2], s1[q * 4 + 3]);
        *reinterpret_cast<float4*>(red + (size_t)(RPP + vr) * BNW + vc * VEC + q * 4) = make_float4(s2[q * 4], s2[q * 4 + 1], s2[q * 4 + 2], s2[q * 4 + 3]);
    }
    __syncthreads();
    float t2 = 0.f;
    if (tid < BNW) {
        float t1 = 0.f;
#pragma unroll 8
        for (int l = 0; l < RPP; ++l) { t1 += red[l * BNW + tid]; t2 += red[(RPP + l) * BNW + tid]; }
        const size_t nrb = (size_t)p.stat_nrb;
        float* o1 = p.stat_out + (size_t)(nw0 + tid) * nrb + rb;
        if (mode == EG_STAT_MOMENTS) {
            const float pv = round_t(ct[tid]), cnt = (float)BM;
            o1[0] = pv + t1 / cnt;
            o1[(size_t)p.N * nrb] = fmaxf(t2 - t1 * t1 / cnt, 0.f);
        } else if (mode == EG_STAT_BN_BWD) {
            o1[0] = t1;
            o1[(size_t)p.N * nrb] = t2;
        } else {
            o1[0] = t1;
        }
    }
    constexpr int NWV = (BNW + 63) / 64;               // waves that hold the column threads (2 for BNW = 128, 1 for 64 / 32)
    if (mode == EG_STAT_SN_BIAS && tid < NWV * 64) {   // the tile's dot, added in wave order (whole waves: the threads beyond BNW add zeros)
        const float w = wave_sum(tid < BNW ? t2 : 0.f);
        if ((tid & 63) == 0) red[2 * RPP * BNW + (tid >> 6)] = w;
    }
    if (mode == EG_STAT_SN_BIAS) {
        __syncthreads();
        if (tid == 0) {
            float d = red[2 * RPP * BNW];
#pragma unroll
            for (int w = 1; w < NWV; ++w) d += red[2 * RPP * BNW + w];
            p.stat_out[(size_t)p.N * p.stat_nrb + (size_t)rb * tiles_n + n_tile] = d;
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
