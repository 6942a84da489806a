// Implicit-GEMM convolution family for gfx950 (MFMA, 64-wide waves, LDS-staged 128-byte K rows).
//
//   igemm_nt : C[m][n] = sum_k A[m][k] * Wp[n][k]       A rows gathered from an NHWC tensor
//              (conv forward, conv backward-data == ConvTranspose forward, linear)
//   igemm_tn : S[n][t][c] = sum_m P[m][n] * X[pix(m,t)][c]   (weight gradients, split over m)
//
// Both run in fp32 (v_mfma_f32_16x16x4_f32, exact fp32) or bf16 (v_mfma_f32_16x16x32_bf16, fp32
// accumulate).  K rows in LDS are always 128 bytes (32 fp32 / 64 bf16) and XOR-swizzled in 16-byte
// chunks so that ds_read_b128 fragment reads are bank-conflict free without padding.
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <type_traits>

#include "eg_common.h"
#include "igemm_nt.h"
#include "adam.h"

// This file is compiled with -ffp-contract=off (Makefile): every NT variant must round the epilogue (acc / sigma + bias) alike -- the
// variants are tested bit for bit against each other -- and clang contracts `a * b + c` depending on where the operands come from.



// K extent of a packed weight panel row: K rounded up to the K tile.  (Skewing pitches that are multiples of 2 KiB by one extra tile
// was tried against L2 channel aliasing of the 128-byte column slices: no effect on MI355X.)
static int kpad_of(int K, int N, int dtype) {
    (void)N;
    return round_up(K > 0 ? K : 1, bk_of(dtype));
}

static int conv_out_dim(const eg_conv* c, int in) { return ((in << c->up) + 2 * c->pad - c->k) / c->stride + 1; }

enum { NEED_CIN = 1, NEED_COUT = 2 };
static int check_conv(const eg_conv* c, int dtype, int need) {
    EG_REQUIRE(c && c->B > 0 && c->k > 0 && c->stride > 0 && c->pad >= 0 && (c->up == 0 || c->up == 1), "eg_conv: bad field");
    EG_REQUIRE(dtype == EG_F32 || dtype == EG_BF16 || dtype == EG_F16, "dtype must be EG_F32, EG_BF16 or EG_F16");
    const int OH = conv_out_dim(c, c->H), OW = conv_out_dim(c, c->W);
    EG_REQUIRE(ilog2_exact(c->H) >= 0 && ilog2_exact(c->W) >= 0 && ilog2_exact(OH) >= 0 && ilog2_exact(OW) >= 0,
               "spatial extents must be powers of two (H=%d W=%d OH=%d OW=%d)", c->H, c->W, OH, OW);
    EG_REQUIRE((!(need & NEED_CIN) || c->Cin % vec_of(dtype) == 0) && (!(need & NEED_COUT) || c->Cout % vec_of(dtype) == 0),
               "gathered channel count must be a multiple of %d for this dtype (Cin=%d Cout=%d); pad the tensor", vec_of(dtype), c->Cin, c->Cout);
    return 0;
}

// forward: one phase, taps = k x k, source = X
static void geom_fwd(const eg_conv* c, int dtype, NtParams& p) {
    const int OH = conv_out_dim(c, c->H), OW = conv_out_dim(c, c->W);
    p.B = c->B; p.H = c->H; p.W = c->W; p.C = c->Cin;
    p.lOH = ilog2_exact(OH); p.lOW = ilog2_exact(OW);
    p.sy = p.sx = c->stride; p.up = c->up;
    p.N = c->Cout;
    p.DH = OH; p.DW = OW; p.osy = p.osx = 1;
    p.M = c->B * OH * OW;
    NtPhase& f = p.ph[0];
    f.TH = f.TW = c->k; f.dy0 = f.dx0 = -c->pad; f.dys = f.dxs = 1; f.ooy = f.oox = 0;
    f.K = c->k * c->k * c->Cin; f.Kpad = kpad_of(f.K, c->Cout, dtype); f.w_off = 0;
}

struct BwdAxis { int k0, T, d0; };   // first kernel index, tap count, source offset for tap 0 (then -1 per tap)
static BwdAxis bwd_axis(const eg_conv* c, int r) {
    BwdAxis a;
    a.k0 = (r + c->pad) % c->stride;
    a.T = a.k0 < c->k ? (c->k - a.k0 + c->stride - 1) / c->stride : 0;
    a.d0 = (r + c->pad - a.k0) / c->stride;
    return a;
}

// backward-data: stride*stride phases; source = dY (lattice == dY's grid), dst = dX interleaved
static int geom_bwd(const eg_conv* c, int dtype, NtParams& p, int* nphase) {
    const int OH = conv_out_dim(c, c->H), OW = conv_out_dim(c, c->W);
    const int s = c->stride;
    EG_REQUIRE(s == 1 || s == 2, "stride must be 1 or 2");
    const int XH = c->H << c->up, XW = c->W << c->up;   // dX is produced at the (upsampled) conv-input resolution
    EG_REQUIRE(XH % s == 0 && XW % s == 0 && XH / s == OH && XW / s == OW,
               "backward-data needs OH == H/stride (k=%d s=%d p=%d)", c->k, s, c->pad);
    p.B = c->B; p.H = OH; p.W = OW; p.C = c->Cout;
    p.lOH = ilog2_exact(OH); p.lOW = ilog2_exact(OW);
    p.sy = p.sx = 1; p.up = 0;
    p.N = c->Cin;
    p.DH = XH; p.DW = XW; p.osy = p.osx = s;
    p.M = c->B * OH * OW;
    long long off = 0;
    int n = 0;
    for (int ry = 0; ry < s; ++ry)
        for (int rx = 0; rx < s; ++rx) {
            const BwdAxis ay = bwd_axis(c, ry), ax = bwd_axis(c, rx);
            NtPhase& f = p.ph[n++];
            f.TH = ay.T; f.TW = ax.T; f.dy0 = ay.d0; f.dx0 = ax.d0; f.dys = f.dxs = -1; f.ooy = ry; f.oox = rx;
            f.K = ay.T * ax.T * c->Cout; f.Kpad = kpad_of(f.K, c->Cin, dtype); f.w_off = off;
            off += (long long)c->Cin * f.Kpad;
        }
    *nphase = n;
    return 0;
}


// ------------------------------------------------------------------------------------------------
// igemm_nt
// ------------------------------------------------------------------------------------------------
// STAT: the epilogue with column statistics (eg_epilogue.stat_mode; 16-bit types, whole 128-row tiles, ONE column tile: N == BN) -- the
// small networks' 32- and 64-channel layers, whose BatchNorm / bias-gradient reductions were 2-3 extra launches per layer on a chain
// that is launch-bound end to end
// SPLITK: blockIdx.z = phase * nsplit + split; every workgroup runs a slice of the K loop, the last of a tile's workgroups to arrive adds
// the fp32 partial tiles in split order (its own from registers) and runs the epilogue (protocol and comments: igemm_nt8s.hip).  For the
// small networks' few-row / deep-K launches (a 4 x 4 output map: 16-48 tiles on 256 CUs, 16 dependent K steps of ~1.4 us each).
template <typename T, int BM, int BN, int WGM, int WGN, bool STAT = false, bool SPLITK = false>
__global__ __launch_bounds__(256) void igemm_nt_kernel(const NtParams p) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int TM = BM / WGM / 16, TN = BN / WGN / 16;
    constexpr int A_LD = BM / 32;
    constexpr int B_LD = (BN + 31) / 32;
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(WGM * WGN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nsplit = SPLITK && p.nsplit > 1 ? p.nsplit : 1;
    const int phase = blockIdx.z / nsplit, split = blockIdx.z - phase * nsplit;
    const NtPhase ph = p.ph[phase];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int chunk = tid & 7, rbase = tid >> 3;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const int HU = p.H << p.up, WU = p.W << p.up;

    // per-thread A rows
    int a_pix0[A_LD];   // b*H*W (pixel base of the image) or -1 when the row is out of range
    int a_y[A_LD], a_x[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m0 + rbase + 32 * i;
        const int b = m >> (p.lOW + p.lOH);
        a_pix0[i] = (m < p.M) ? b * p.H * p.W : -1;
        a_y[i] = ((m >> p.lOW) & OHm) * p.sy + ph.dy0;
        a_x[i] = (m & OWm) * p.sx + ph.dx0;
    }
    const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
    const T* __restrict__ wp = reinterpret_cast<const T*>(p.wp) + ph.w_off;
    const int nk_all = ph.Kpad / BK;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = split * per;
    const int nk = max(0, min(per, nk_all - kt0));
    // tap state of this thread's 16-byte chunk at K tile kt0
    int kc = kt0 * BK + chunk * VEC, ty = 0, tx = 0;
    if (kc >= p.C) {
        const int tap = kc / p.C;
        kc -= tap * p.C;
        ty = tap / ph.TW;
        tx = tap - ty * ph.TW;
    }

    uint4 ra[A_LD], rb[B_LD];
    auto gload = [&](int kt) {
        const bool tap_ok = ty < ph.TH;
        const int oy = ty * ph.dys, ox = tx * ph.dxs;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int iy = a_y[i] + oy, ix = a_x[i] + ox;
            const bool ok = tap_ok && a_pix0[i] >= 0 && iy >= 0 && iy < HU && ix >= 0 && ix < WU;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const size_t pix = (size_t)a_pix0[i] + (size_t)((iy >> p.up) * p.W + (ix >> p.up));
                v = *reinterpret_cast<const uint4*>(src + pix * p.C + kc);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int r = rbase + 32 * i;
            const int n = n0 + r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < BN && n < p.N) v = *reinterpret_cast<const uint4*>(wp + (size_t)n * ph.Kpad + (size_t)kt * BK + chunk * VEC);
            rb[i] = v;
        }
        kc += BK;
        while (kc >= p.C) { kc -= p.C; if (++tx == ph.TW) { tx = 0; ++ty; } }
    };
    auto lstore = [&](int stage) {
        char* sa = smem + stage * STAGE;
        char* sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<uint4*>(sa + lds_off(rbase + 32 * i, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int r = rbase + 32 * i;
            if (r < BN) *reinterpret_cast<uint4*>(sb + lds_off(r, chunk)) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    if (nk > 0) {
        gload(kt0);
        lstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt0 + kt + 1);
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(sa + lds_off((wm * TM + i) * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off((wn * TN + j) * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mfma_step<T>(af[i], bfr[j], acc[i][j]);
        }
        if (kt + 1 < nk) lstore((kt + 1) & 1);
        __syncthreads();
    }

    if (SPLITK && nsplit > 1) {
        const int nphase = gridDim.z / nsplit;
        const size_t Mpad = (size_t)gridDim.x * BM;
        const size_t slab = (size_t)nphase * Mpad * p.N;
        float* part = p.part + ((size_t)phase * Mpad + m0) * p.N + n0;
        float* mine = part + (size_t)split * slab;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wm * TM + i) * 16 + frow;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* a = mine + (size_t)row * p.N + (wn * TN + j) * 16 + fq * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) __hip_atomic_store(a + r, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // written through
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* cnt = p.split_cnt + ((size_t)phase * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(smem);
        if (tid == 0) flag[0] = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (flag[0] != (unsigned)(nsplit - 1)) return;
        if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x4 sum[TM][TN];
        for (int s = 0; s < nsplit; ++s) {             // uniform; own partial from registers: the order does not depend on who is last
            if (s == split) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) sum[i][j] = s == 0 ? acc[i][j] : sum[i][j] + acc[i][j];
                continue;
            }
            const float* other = part + (size_t)s * slab;
            f32x4 ld[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 16 + frow;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float* a = other + (size_t)row * p.N + (wn * TN + j) * 16 + fq * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ld[i][j][r] = __hip_atomic_load(a + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) sum[i][j] = s == 0 ? ld[i][j] : sum[i][j] + ld[i][j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = sum[i][j];
        __syncthreads();
    }

    // ---- epilogue ----
    // acc[i][j][r] = C[m = m0 + (wm*TM+i)*16 + frow][n = n0 + (wn*TN+j)*16 + fq*4 + r]
    if constexpr (STAT) {
        NtEpiPre<T, TM, TN, 0> epi;
        nt_epi_prefetch<T, BM, BN, TM, TN, 256, 0>(epi, p, ph, m0, n0, wm * TM * 16, wn * TN * 16, tid, frow, fq);
        nt_epilogue_lds_stat<T, BM, BN, TM, TN, 256, 0>(epi, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * TN * 16, tid, frow, fq,
                                                        phase * gridDim.x + blockIdx.x, 0, 1);
        return;
    }
    if (p.out_mode == EG_OUT_NHWC && (p.N % VEC) == 0) {
        // the shared epilogue: fp32 tile through LDS (16-byte chunks XOR-swizzled by row), then whole 16-byte vector stores
        nt_epilogue_lds<T, BM, BN, TM, TN, 256>(p, ph, acc, smem, m0, n0, wm * TM * 16, wn * TN * 16, tid, frow, fq);
        return;
    }
    // scalar path: NCHW fp32 image outputs (N = 1..4) and channel counts that are not a multiple of the vector width
    const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
    const EgActFast af = eg_act_fast(p.act, p.slope);
    const EgGradFast gf = eg_grad_fast(p.mask_act, p.mask_slope);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 16 + frow;
        if (m >= p.M) continue;
        const float inv_sigma = p.sigma ? 1.f / p.sigma[p.sigma_rows ? m / p.sigma_rows : 0] : 1.f;
        const int b = m >> (p.lOW + p.lOH);
        const int y = ((m >> p.lOW) & OHm) * p.osy + ph.ooy;
        const int x = (m & OWm) * p.osx + ph.oox;
        const size_t pix = ((size_t)b * p.DH + y) * p.DW + x;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + (wn * TN + j) * 16 + fq * 4 + r;
                if (n >= p.N) continue;
                float v = __fmul_rn(acc[i][j][r], inv_sigma);
                if (p.bias) v = __fadd_rn(v, p.bias[p.bias_mod ? n % p.bias_mod : n]);
                v = af.special ? eg_act(v, p.act, p.slope) : eg_act_apply(v, af);
                if (p.out_mode == EG_OUT_NHWC) {
                    const size_t o = pix * p.N + n;
                    if (mask) v *= gf.special ? eg_act_grad_from_out(Elt<T>::ld(mask + o), p.mask_act, p.mask_slope) : eg_grad_apply(Elt<T>::ld(mask + o), gf);
                    Elt<T>::st(reinterpret_cast<T*>(p.dst) + o, v);
                } else {
                    const size_t o = (((size_t)b * p.N + n) * p.DH + y) * p.DW + x;
                    if (p.mask) {
                        const float a = reinterpret_cast<const float*>(p.mask)[o];
                        v *= gf.special ? eg_act_grad_from_out(a, p.mask_act, p.mask_slope) : eg_grad_apply(a, gf);
                    }
                    reinterpret_cast<float*>(p.dst)[o] = v;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// igemm_nt_buf: 128x128 tile, 2-stage LDS-DMA ring like igemm_nt_dma<.,128,2>, but the gather goes through buffer descriptors
// (`buffer_load_dwordx4 ... offen lds`): the per-lane part of every address is a 32-bit byte offset that only changes when the
// K loop moves to the next filter tap, the walk along the channels of a tap is the instruction's SGPR offset, and padded /
// out-of-range rows carry an offset beyond num_records (the hardware range check returns zeros) -> no branches, no 64-bit
// address math and no EXEC juggling in the K loop.  Needs C % BK == 0 (every lane of a K step sits in the same tap) and
// tensors below 2 GiB; the dispatcher falls back to igemm_nt_dma otherwise.
// ------------------------------------------------------------------------------------------------

// four 1-KiB LDS-DMA pieces (tile rows (j*4+wave)*8 .. +7, j = 0..3) from one descriptor: LDS bases lds, lds+4K, lds+8K, lds+12K
__device__ __forceinline__ void eg_bufdma4(const u32x4_t srd, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
        "buffer_load_dwordx4 %4, %5, %6 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(srd), "s"(soff), "s"(lds)
        : "memory", "scc");
}

__device__ __forceinline__ void eg_bufdma1(const u32x4_t srd, unsigned v0, unsigned soff, unsigned lds) {
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


template <typename T, bool PROF = false, bool SPLITK = false>   // SPLITK: its own instantiation so that profilers list the split launches apart
__global__ __launch_bounds__(256) void igemm_nt_buf_kernel(const NtParams p, unsigned long long* prof = nullptr) {
    unsigned long long t_begin = 0, t_wait = 0, t_issue = 0, t_comp = 0, t0 = 0, t1 = 0;
    if (PROF) t_begin = __builtin_amdgcn_s_memtime();
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int BM = 128, BN = 128;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int TM = 4, TN = 4;                  // waves 2 x 2, wave tile 64 x 64
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nsplit = SPLITK && p.nsplit > 1 ? p.nsplit : 1;
    const int phase = blockIdx.z / nsplit, split = blockIdx.z - phase * nsplit;
    const NtPhase ph = p.ph[phase];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroup ids go round robin over the 8 XCDs (each with its own L2), so neighbouring M tiles -- which share input
    // halo rows in an implicit convolution -- would sit on different L2s.  With gridDim.x a multiple of 8, XCD c takes tiles [c, c+1) * gridDim.x / 8.
    const int bx = (p.xcd_remap && (gridDim.x & 7) == 0) ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int m0 = bx * BM, n0 = blockIdx.y * BN;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const int HU = p.H << p.up, WU = p.W << p.up;
    const int rsub = lane >> 3, pos = lane & 7;
    const int srcchunk = pos ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);

    int a_pix0[4], a_y[4], a_x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + (j * 4 + wave) * 8 + rsub;
        const int b = m >> (p.lOW + p.lOH);
        a_pix0[j] = (m < p.M) ? b * p.H * p.W : -1;
        a_y[j] = ((m >> p.lOW) & OHm) * p.sy + ph.dy0;
        a_x[j] = (m & OWm) * p.sx + ph.dx0;
    }
    const unsigned row_bytes = (unsigned)p.C * sizeof(T);
    unsigned va[4], vb[4];
    auto tap_offsets = [&](int ty, int tx) {
        const int oy = ty * ph.dys, ox = tx * ph.dxs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = a_y[j] + oy, ix = a_x[j] + ox;
            const bool ok = a_pix0[j] >= 0 && iy >= 0 && iy < HU && ix >= 0 && ix < WU;
            const unsigned pix = (unsigned)(a_pix0[j] + (iy >> p.up) * p.W + (ix >> p.up));
            va[j] = ok ? pix * row_bytes + (unsigned)srcchunk * 16u : EG_OOB;
        }
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + (j * 4 + wave) * 8 + rsub;
        vb[j] = n < p.N ? (unsigned)n * (unsigned)ph.Kpad * (unsigned)sizeof(T) + (unsigned)srcchunk * 16u : EG_OOB;
    }
    const u32x4_t srdA = eg_make_srd(p.src, (unsigned)((size_t)p.B * p.H * p.W * p.C * sizeof(T)));
    const u32x4_t srdB = eg_make_srd(reinterpret_cast<const T*>(p.wp) + ph.w_off, (unsigned)((size_t)p.N * ph.Kpad * sizeof(T)));
    // this block's K steps [kt0, kt0 + nk)
    const int nk_all = ph.Kpad / BK;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = split * per;
    const int nk = min(per, nk_all - kt0);

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 1024u;
    // wave-uniform walk over (tap, channel block)
    const int steps_per_tap = p.C / BK;
    const int tap0 = kt0 / steps_per_tap;
    int ty = tap0 / ph.TW, tx = tap0 - ty * ph.TW;
    unsigned kc_bytes = (unsigned)(kt0 - tap0 * steps_per_tap) * 128u;
    if (ty < ph.TH) tap_offsets(ty, tx);
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = EG_OOB;
    }
    auto issue = [&](int kt, int stage) {
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage * STAGE);
        eg_bufdma4(srdA, va[0], va[1], va[2], va[3], kc_bytes, sa);
        eg_bufdma4(srdB, vb[0], vb[1], vb[2], vb[3], (unsigned)(kt0 + kt) * 128u, sa + BM * 128);
        kc_bytes += 128u;
        if (kc_bytes >= row_bytes) {               // next tap (uniform branch)
            kc_bytes = 0;
            if (++tx == ph.TW) { tx = 0; ++ty; }
            if (ty < ph.TH) tap_offsets(ty, tx);
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) va[j] = EG_OOB;     // K padding beyond the last tap
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 bfr[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off((wn * TN + j) * 16 + frow, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const uint4 af = *reinterpret_cast<const uint4*>(sa + lds_off((wm * TM + i) * 16 + frow, ks * 4 + fq));
#pragma unroll
                for (int j = 0; j < TN; ++j) mfma_step<T>(af, bfr[j], acc[i][j]);
            }
        }
    };
    if (nk > 0) issue(0, 0);
    for (int kt = 0; kt + 1 < nk; ++kt) {
        if (PROF) t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // everyone's stage kt landed; everyone finished reading stage kt-1
        __builtin_amdgcn_sched_barrier(0);
        if (PROF) { t1 = __builtin_amdgcn_s_memtime(); t_wait += t1 - t0; }
        issue(kt + 1, (kt + 1) & 1);
        if (PROF) { __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); t_issue += t0 - t1; __builtin_amdgcn_sched_barrier(0); }
        compute(kt & 1);
        if (PROF) { __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); t_comp += t1 - t0; __builtin_amdgcn_sched_barrier(0); }
    }
    // last K step: nothing left to issue -> fetch the epilogue operands instead, their latency hides behind the MFMAs
    constexpr int PF = 8;
    NtEpiPre<T, TM, TN, PF> epi;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (nsplit == 1) nt_epi_prefetch<T, BM, BN, TM, TN, 256, PF>(epi, p, ph, m0, n0, wm * TM * 16, wn * TN * 16, tid, frow, fq);
    if (nk > 0) compute((nk - 1) & 1);
    if (PROF) t0 = __builtin_amdgcn_s_memtime();
    if (nsplit > 1) {
        // raw fp32 partial tile; nt_splitk_epilogue_kernel sums the splits and applies the epilogue
        const int nphase = gridDim.z / nsplit;
        float* part = p.part + ((size_t)(split * nphase + phase) * (gridDim.x * BM) + m0) * p.N + n0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wm * TM + i) * 16 + frow;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *reinterpret_cast<f32x4*>(part + (size_t)row * p.N + (wn * TN + j) * 16 + fq * 4) = acc[i][j];
        }
        return;
    }
    __syncthreads();
    nt_epilogue_lds_pre<T, BM, BN, TM, TN, 256, PF>(epi, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * TN * 16, tid, frow, fq);
    if (PROF && lane == 0) {
        t1 = __builtin_amdgcn_s_memtime();
        unsigned long long* o = prof + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        o[0] = t1 - t_begin; o[1] = t_wait; o[2] = t_issue; o[3] = t_comp; o[4] = t1 - t0; o[5] = t_begin; o[6] = t1; o[7] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// igemm_nt_pers: the 128x128 buffer-descriptor kernel as a persistent pipeline.  A workgroup walks tiles t = blockIdx.x, +gridDim.x, ...
// with ONE continuous 2-stage LDS-DMA ring: the first K step of the next tile is in flight while the last step of the current tile is
// computed, the epilogue operands (1/sigma, bias, activation-gradient mask in accumulator layout) are fetched during that last step,
// and the results leave straight from the accumulators as 8/16-byte stores that nobody waits for -- no LDS staging, no barriers and no
// idle MFMA pipe at tile seams (in igemm_nt_buf the epilogue of every workgroup ran at the same time and cost 15-45 % of a launch).
// vmcnt discipline: the epilogue's TM*TN stores are the wave's youngest operations when the next step's DMA must have landed, so that
// wait is vmcnt(TM*TN) (vmcnt(0) for ragged tiles, where waves may skip stores).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void igemm_nt_pers_kernel(const NtParams p, int tiles_m, int tiles_n, int ntiles) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int BM = 128, BN = 128;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int TM = 4, TN = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const int HU = p.H << p.up, WU = p.W << p.up;
    const int rsub = lane >> 3, pos = lane & 7;
    const int srcchunk = pos ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned row_bytes = (unsigned)p.C * sizeof(T);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 1024u;
    const u32x4_t srdA = eg_make_srd(p.src, (unsigned)((size_t)p.B * p.H * p.W * p.C * sizeof(T)));

    // ---- load side: the tile whose K steps are being issued ----
    int lt = blockIdx.x, lkt = 0, lnk = 0;
    int l_dys = 0, l_dxs = 0, l_TW = 1, l_TH = 0;
    int a_pix0[4], a_y[4], a_x[4];
    unsigned va[4], vb[4];
    u32x4_t srdB = srdA;
    int ty = 0, tx = 0;
    unsigned kc_bytes = 0;
    auto tap_offsets = [&]() {
        const int oy = ty * l_dys, ox = tx * l_dxs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = a_y[j] + oy, ix = a_x[j] + ox;
            const bool ok = ty < l_TH && a_pix0[j] >= 0 && iy >= 0 && iy < HU && ix >= 0 && ix < WU;
            const unsigned pix = (unsigned)(a_pix0[j] + (iy >> p.up) * p.W + (ix >> p.up));
            va[j] = ok ? pix * row_bytes + (unsigned)srcchunk * 16u : EG_OOB;
        }
    };
    auto setup_load = [&]() {
        const int mt = lt % tiles_m, q = lt / tiles_m;
        const int nt = q % tiles_n, phase = q / tiles_n;
        const NtPhase& ph = p.ph[phase];
        const int m0 = mt * BM, n0 = nt * BN;
        l_dys = ph.dys; l_dxs = ph.dxs; l_TW = ph.TW; l_TH = ph.TH;
        lnk = ph.Kpad / BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + (j * 4 + wave) * 8 + rsub;
            const int b = m >> (p.lOW + p.lOH);
            a_pix0[j] = (m < p.M) ? b * p.H * p.W : -1;
            a_y[j] = ((m >> p.lOW) & OHm) * p.sy + ph.dy0;
            a_x[j] = (m & OWm) * p.sx + ph.dx0;
            const int n = n0 + (j * 4 + wave) * 8 + rsub;
            vb[j] = n < p.N ? (unsigned)n * (unsigned)ph.Kpad * (unsigned)sizeof(T) + (unsigned)srcchunk * 16u : EG_OOB;
        }
        srdB = eg_make_srd(reinterpret_cast<const T*>(p.wp) + ph.w_off, (unsigned)((size_t)p.N * ph.Kpad * sizeof(T)));
        ty = 0; tx = 0; kc_bytes = 0; lkt = 0;
        tap_offsets();
    };
    // K step lkt of tile lt -> LDS stage: 8 pieces (A slots 0..3, B slots 0..3), issued in one burst right after the barrier (spreading
    // them between the MFMA groups of the step was measured 8 % slower)
    auto issue_piece = [&](int q, int stage) {
        const unsigned sa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage * STAGE);
        // (single-tap launches whose row is not a whole number of K tiles: chunks past the end of the row read as zeros)
        if (q < 4) eg_bufdma1(srdA, kc_bytes + (unsigned)srcchunk * 16u < row_bytes ? va[q] : EG_OOB, kc_bytes, sa + q * 0x1000);
        else eg_bufdma1(srdB, vb[q - 4], (unsigned)lkt * 128u, sa + BM * 128 + (q - 4) * 0x1000);
    };
    auto issue_advance = [&]() {
        ++lkt;
        kc_bytes += 128u;
        if (kc_bytes >= row_bytes) {               // next tap (uniform branch)
            kc_bytes = 0;
            if (++tx == l_TW) { tx = 0; ++ty; }
            tap_offsets();
        }
    };

    // ---- compute side: the tile whose accumulators are live ----
    int ct = blockIdx.x, ckt = 0, cnk = 0, c_m0 = 0, c_n0 = 0, c_ooy = 0, c_oox = 0;
    auto setup_comp = [&]() {
        const int mt = ct % tiles_m, q = ct / tiles_m;
        const int nt = q % tiles_n, phase = q / tiles_n;
        c_m0 = mt * BM; c_n0 = nt * BN;
        c_ooy = p.ph[phase].ooy; c_oox = p.ph[phase].oox;
        cnk = p.ph[phase].Kpad / BK;
        ckt = 0;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (lt < ntiles) {
        setup_load();
#pragma unroll
        for (int q = 0; q < 8; ++q) issue_piece(q, 0);
        issue_advance();
    }
    if (ct < ntiles) setup_comp();
    unsigned g = 0;                                   // global K step: LDS stage = g & 1
    bool stores_pending = false;
    const T* __restrict__ maskp = reinterpret_cast<const T*>(p.mask);
    typedef typename std::conditional<std::is_same<T, float>::value, uint4, uint2>::type OutVec;   // 4 outputs of one accumulator

    while (ct < ntiles) {
        if (stores_pending) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // everyone's stage g landed; everyone finished reading stage g-1
        __builtin_amdgcn_sched_barrier(0);
        stores_pending = false;
        if (lkt == lnk) {                             // load side moves on to this workgroup's next tile
            lt += gridDim.x;
            if (lt < ntiles) setup_load();
        }
        if (lt < ntiles) {
#pragma unroll
            for (int q = 0; q < 8; ++q) issue_piece(q, (g + 1) & 1);
            issue_advance();
        }
        const bool last = ckt + 1 == cnk;
        // epilogue operands in accumulator layout, fetched under the last MFMA block of the tile
        float e_sigma[TM], e_bias[TN][4];
        OutVec e_mask[TM][TN];
        size_t e_off[TM];
        bool e_ok[TM];
        if (last) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = c_m0 + (wm * TM + i) * 16 + frow;
                e_ok[i] = m < p.M;
                const int mc = min(m, p.M - 1);
                e_sigma[i] = p.sigma ? p.sigma[p.sigma_rows ? mc / p.sigma_rows : 0] : 1.f;
                const int b = mc >> (p.lOW + p.lOH);
                const int y = ((mc >> p.lOW) & OHm) * p.osy + c_ooy;
                const int x = (mc & OWm) * p.osx + c_oox;
                e_off[i] = (((size_t)b * p.DH + y) * p.DW + x) * p.N + c_n0 + wn * 64 + fq * 4;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = c_n0 + (wn * TN + j) * 16 + fq * 4 + r;
                    e_bias[j][r] = p.bias ? p.bias[p.bias_mod ? n % p.bias_mod : n] : 0.f;
                }
            if (maskp) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) e_mask[i][j] = *reinterpret_cast<const OutVec*>(maskp + e_off[i] + j * 16);
            }
        }
        {
            const char* sa = smem + (g & 1) * STAGE;
            const char* sb = sa + BM * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 bfr[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(sb + lds_off((wn * TN + j) * 16 + frow, ks * 4 + fq));
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const uint4 af = *reinterpret_cast<const uint4*>(sa + lds_off((wm * TM + i) * 16 + frow, ks * 4 + fq));
#pragma unroll
                    for (int j = 0; j < TN; ++j) mfma_step<T>(af, bfr[j], acc[i][j]);
                }
            }
        }
        ++g;
        if (!last) { ++ckt; continue; }
        // ---- epilogue straight from the accumulators: acc[i][j][r] = C[c_m0 + (wm*TM+i)*16 + frow][c_n0 + (wn*TN+j)*16 + fq*4 + r]
        const EgActFast eaf = eg_act_fast(p.act, p.slope);
        const EgGradFast egf = eg_grad_fast(p.mask_act, p.mask_slope);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float inv_sigma = p.sigma ? 1.f / e_sigma[i] : 1.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float f[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = __fmul_rn(acc[i][j][r], inv_sigma);
                    if (p.bias) x = __fadd_rn(x, e_bias[j][r]);
                    f[r] = eaf.special ? eg_act(x, p.act, p.slope) : eg_act_apply(x, eaf);
                }
                if (maskp) {
                    const T* me = reinterpret_cast<const T*>(&e_mask[i][j]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        f[r] *= egf.special ? eg_act_grad_from_out(Elt<T>::ld(me + r), p.mask_act, p.mask_slope) : eg_grad_apply(Elt<T>::ld(me + r), egf);
                }
                OutVec ov;
                T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
                for (int r = 0; r < 4; ++r) Elt<T>::st(oe + r, f[r]);
                if (e_ok[i]) *reinterpret_cast<OutVec*>(reinterpret_cast<T*>(p.dst) + e_off[i] + j * 16) = ov;
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        stores_pending = c_m0 + BM <= p.M;            // full tile: every wave issued exactly TM*TN stores
        ct += gridDim.x;
        if (ct < ntiles) setup_comp();
    }
}

// sum of the split-K partial tiles + the fused epilogue (1/sigma, bias, activation, activation-gradient mask), NHWC store.
// blockIdx.y = phase; 32-bit index arithmetic (M * N / VEC < 2^31, checked by the planner's size limits), shifts when N / VEC is a power
// of two (lvpr >= 0), one bias modulo per vector: with 64-bit divisions and a modulo per element this copy was ALU-bound.
template <typename T>
__global__ __launch_bounds__(256) void nt_splitk_epilogue_kernel(const NtParams p, int nphase, int Mpad, int lvpr) {
    constexpr int VEC = Elt<T>::VEC;
    const EgActFast saf = eg_act_fast(p.act, p.slope);
    const EgGradFast sgf = eg_grad_fast(p.mask_act, p.mask_slope);
    const unsigned vpr = p.N / VEC;
    const unsigned total = (unsigned)p.M * vpr;
    const int phase = blockIdx.y;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const T* __restrict__ mask = reinterpret_cast<const T*>(p.mask);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned vc = lvpr >= 0 ? (i & (vpr - 1)) : (i % vpr);
        const int m = (int)(lvpr >= 0 ? (i >> lvpr) : (i / vpr));
        const int n = (int)vc * VEC;
        float f[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) f[q] = 0.f;
        for (int sp = 0; sp < p.nsplit; ++sp) {
            const float* src = p.part + ((size_t)(sp * nphase + phase) * Mpad + m) * p.N + n;
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
                f[4 * q] += v.x; f[4 * q + 1] += v.y; f[4 * q + 2] += v.z; f[4 * q + 3] += v.w;
            }
        }
        const float inv_sigma = p.sigma ? 1.f / p.sigma[p.sigma_rows ? m / p.sigma_rows : 0] : 1.f;
        const int b = m >> (p.lOW + p.lOH);
        const int y = ((m >> p.lOW) & OHm) * p.osy + p.ph[phase].ooy;
        const int x = (m & OWm) * p.osx + p.ph[phase].oox;
        const size_t o = (((size_t)b * p.DH + y) * p.DW + x) * p.N + n;
        // bias index of element q: (n + q) % bias_mod; n is a multiple of VEC, so when VEC divides bias_mod it is (n % bias_mod) + q
        const bool bias_vec = p.bias_mod == 0 || (p.bias_mod % VEC) == 0;
        const int nb = p.bias_mod ? n % p.bias_mod : n;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float v = __fmul_rn(f[q], inv_sigma);
            if (p.bias) v = __fadd_rn(v, p.bias[bias_vec ? nb + q : (n + q) % p.bias_mod]);
            f[q] = saf.special ? eg_act(v, p.act, p.slope) : eg_act_apply(v, saf);
        }
        if (mask) {
            const uint4 mv = *reinterpret_cast<const uint4*>(mask + o);
            const T* me = reinterpret_cast<const T*>(&mv);
#pragma unroll
            for (int q = 0; q < VEC; ++q)
                f[q] *= sgf.special ? eg_act_grad_from_out(Elt<T>::ld(me + q), p.mask_act, p.mask_slope) : eg_grad_apply(Elt<T>::ld(me + q), sgf);
        }
        uint4 ov;
        T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int q = 0; q < VEC; ++q) Elt<T>::st(oe + q, f[q]);
        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.dst) + o) = ov;
    }
}

// ------------------------------------------------------------------------------------------------
// dispatch.  The planner is a pure function of the problem and of the caller's per-call hints (eg_epilogue.nt_variant / nt_splitk):
// the library keeps no tuning state, so two trainers (or a test and a trainer) in one process cannot interfere.
//   EG_NT_REG     register-staged 128 x {16,32,64,128} tiles: everything (N < 128, NCHW image outputs, ragged channel counts); the
//                 bit-exact reference of the other variants
//   EG_NT_BUF128  128 x 128, 4 waves, 2-stage buffer-descriptor LDS-DMA ring (+ split-K): launches too small for 256-row tiles
//   EG_NT_PERS    persistent 128 x 128 pipeline: the 1-2-step image-side layers
//   EG_NT_S8      igemm_nt8s.hip: 256 x 128 tiles, 8 waves, 3-K-tile ring, one barrier per K tile, register double buffering (+ split-K)
//   EG_NT_S8P     the same with A held in LDS as an input patch shared by the taps of a class
// ------------------------------------------------------------------------------------------------
bool eg_nt8p_geometry(const NtParams& p, int nphase, Nt8pGeom& g);
template <typename T> void eg_launch_nt8s(const NtParams& p, const Nt8pGeom& g, bool patch, int nphase, int ns, hipStream_t st);
template <typename T> void eg_launch_nt8h(const NtParams& p, int nphase, int ns, hipStream_t st);

struct NtPlan { int kind, ns; };

// geometry facts every DMA variant needs
struct NtFacts { bool dma_ok, c_tiles; int nk_min, nk_max; long long tiles128; bool half; };
static NtFacts nt_facts(const NtParams& p, int nphase, int vec, size_t esize) {
    NtFacts f{false, false, 1 << 30, 0, 0, esize == 2};
    f.tiles128 = (long long)cdiv(p.M, 128) * cdiv(p.N, 128) * nphase;
    if (p.out_mode != EG_OUT_NHWC || (p.N % 128) != 0 || (p.C % vec) != 0) return f;
    if ((size_t)p.B * p.H * p.W * p.C * esize >= 0x7fffffffull) return f;
    bool single_tap = true;
    for (int i = 0; i < nphase; ++i) {
        if ((size_t)p.N * p.ph[i].Kpad * esize >= 0x7fffffffull) return f;
        f.nk_min = std::min(f.nk_min, p.ph[i].Kpad / (8 * vec));
        f.nk_max = std::max(f.nk_max, p.ph[i].Kpad / (8 * vec));
        single_tap = single_tap && p.ph[i].TH * p.ph[i].TW == 1;
    }
    f.c_tiles = (p.C % (8 * vec)) == 0;              // every lane of a K step sits in the same filter tap
    f.dma_ok = f.c_tiles || single_tap;
    return f;
}

// split count: enough workgroups to reach `target`, at least 8 K steps per split, partial tiles must fit the lent scratch
static int nt_splits(long long wgs, int target, int nk_min, size_t bytes_per_split, size_t ws_bytes, int forced) {
    if (ws_bytes == 0 || bytes_per_split == 0) return 1;
    int ns = 1;
    if (forced > 0) {
        while (ns * 2 <= forced && nk_min / (ns * 2) >= 1) ns *= 2;
    } else {
        while (wgs * ns < target && ns < 16 && nk_min / (ns * 2) >= 8) ns *= 2;
    }
    while (ns > 1 && (size_t)ns * bytes_per_split > ws_bytes) ns /= 2;
    return ns;
}

static NtPlan nt_plan_auto(const NtParams& p, int nphase, const NtFacts& f, size_t ws_bytes, int splitk);

// variant > 0: that kernel or an error; variant < 0: that kernel where it can run the problem, else the planner's choice (A/B runs)
static NtPlan nt_plan(const NtParams& p, int nphase, int vec, size_t esize, size_t ws_bytes, int variant, int splitk) {
    const NtFacts f = nt_facts(p, nphase, vec, esize);
    const bool prefer = variant < 0;
    if (prefer) variant = -variant;
    const NtPlan bad = prefer ? nt_plan_auto(p, nphase, f, ws_bytes, splitk) : NtPlan{-1, 1};
    const size_t part128 = (size_t)nphase * cdiv(p.M, 128) * 128 * p.N * 4, part256 = (size_t)nphase * cdiv(p.M, 256) * 256 * p.N * 4;
    switch (variant) {
        case EG_NT_REG: return {EG_NT_REG, 1};
        case EG_NT_BUF128:
            if (!f.dma_ok || !f.c_tiles || f.nk_max < 2) return bad;
            return {EG_NT_BUF128, nt_splits(f.tiles128, 512, f.nk_min, part128, ws_bytes, splitk)};
        case EG_NT_PERS: return f.dma_ok ? NtPlan{EG_NT_PERS, 1} : bad;
        case EG_NT_S8P: {
            Nt8pGeom g;
            if (!f.dma_ok || !f.c_tiles || !eg_nt8p_geometry(p, nphase, g)) return bad;
            return {variant, nt_splits((long long)cdiv(p.M, 256) * (p.N / 128) * nphase, 224, f.nk_min, part256, ws_bytes, splitk)};
        }
        case EG_NT_S8:
            if (!f.dma_ok || !f.c_tiles) return bad;
            return {variant, nt_splits((long long)cdiv(p.M, 256) * (p.N / 128) * nphase, 224, f.nk_min, part256, ws_bytes, splitk)};
        case EG_NT_S8H:
            if (!f.dma_ok || !f.c_tiles) return bad;
            return {variant, nt_splits(f.tiles128, 224, f.nk_min, part128, ws_bytes, splitk)};
        case EG_NT_AUTO: return nt_plan_auto(p, nphase, f, ws_bytes, splitk);
        default: return {-1, 1};
    }
}

static NtPlan nt_plan_auto(const NtParams& p, int nphase, const NtFacts& f, size_t ws_bytes, int splitk) {
    const size_t part128 = (size_t)nphase * cdiv(p.M, 128) * 128 * p.N * 4, part256 = (size_t)nphase * cdiv(p.M, 256) * 256 * p.N * 4;
    if (!f.dma_ok) return {EG_NT_REG, 1};
    // shallow launches (1-2 K steps: the image-side layers as 1x1 convolutions over patches) are all prologue and epilogue for a
    // workgroup-per-tile kernel: only the persistent pipeline overlaps them
    if (f.nk_max < 3 || !f.c_tiles) return f.tiles128 >= 128 ? NtPlan{EG_NT_PERS, 1} : NtPlan{EG_NT_REG, 1};
    // 256 x 128 tiles, one 8-wave workgroup per CU (igemm_nt8s): where the launch fills the chip in whole rounds and the K loop is long
    // enough to pay for a prologue and an epilogue that nothing overlaps (measured per shape: profiles/r02_d_layers_nt8s.txt); shorter
    // or fewer tiles run better as 128 x 128 tiles with two workgroups per CU (one's epilogue beside the other's K loop)
    const long long wgs256 = (long long)cdiv(p.M, 256) * (p.N / 128) * nphase;
    const long long rounds = (wgs256 + 255) / 256;
    // Measured in the whole (overlapped) step, the big tile wins beyond the shapes where it wins back to back: S8 wherever it can run
    // 4.70 ms, this rule set 4.76, 128 x 128 everywhere 4.93 (profiles/r02_r_ab_nt_variant.txt; EG_NT_AUTO_S8=0 restores the rule set)
    static const bool all_s8 = [] { const char* e = getenv("EG_NT_AUTO_S8"); return !(e && atoi(e) == 0); }();
    // (16-bit launches of at least 48 such tiles: below that -- the small networks' layers -- and in fp32 the rule set below stays ahead,
    //  profiles/r02_r_ab_auto_s8.txt)
    // fewer than ~200 tiles of 256 x 128 (the single-tape layers: 64 or 128 such tiles on 256 CUs): the same 8-wave design on 128 x 128
    // tiles (igemm_nt8h) -- whole K loops in place of one K split level (profiles/r03_n_nt8h_layers.txt; EG_NT_AUTO_S8H=0: igemm_nt8s)
    static const bool use_h = [] { const char* e = getenv("EG_NT_AUTO_S8H"); return !(e && atoi(e) == 0); }();
    static const int h_below = [] { const char* e = getenv("EG_NT_S8H_BELOW"); return e ? atoi(e) : 200; }();
    if (all_s8 && use_h && f.half && wgs256 >= 48 && wgs256 < h_below && f.nk_min >= 8)
        return {EG_NT_S8H, nt_splits(f.tiles128, 224, f.nk_min, part128, ws_bytes, splitk)};
    if (all_s8 && f.half && wgs256 >= 48) return {EG_NT_S8, nt_splits(wgs256, 224, f.nk_min, part256, ws_bytes, splitk)};
    if (splitk <= 1 && f.nk_min >= 32 && wgs256 >= 200 && wgs256 * 100 >= rounds * 256 * 85) return {EG_NT_S8, 1};
    // split K only below one workgroup per CU: at 256..511 tiles the unsplit launch wins or ties (M=8192 N=512 K=4096: 52 vs 58 us,
    // the 4-phase M=2048 N=512 K=4096: 50 vs 48 us -- profiles/r02_i_t1_splits.txt) and saves the slab round trip and the epilogue launch
    static const int split_below = [] { const char* e = getenv("EG_NT_SPLIT_BELOW"); return e ? atoi(e) : 256; }();
    // ... except very long K loops (>= 128 K tiles: the 512 -> 1024 layer, K = 8192), which still gain from two splits at 256 tiles
    // (in-step 110 -> 79 us, profiles/r02_q_step_detail.txt)
    const bool long_k = f.nk_min >= 128 && f.tiles128 <= 256;         // (at 384 tiles the split loses: 143 vs 116 us)
    if ((f.tiles128 < (splitk > 1 ? 512 : split_below) || long_k) && ws_bytes > 0) {
        const int ns = nt_splits(f.tiles128, 512, f.nk_min, part128, ws_bytes, splitk);
        if (ns > 1) return {EG_NT_BUF128, ns};
    }
    return f.tiles128 >= 256 ? NtPlan{EG_NT_BUF128, 1} : NtPlan{EG_NT_REG, 1};
}

template <typename T, int BM, int BN, int WGM, int WGN>
static void launch_nt_cfg(const NtParams& p, int nphase, hipStream_t st) {
    const int ns = p.nsplit > 1 ? p.nsplit : 1;
    dim3 grid(cdiv(p.M, BM), cdiv(p.N, BN), nphase * ns);
    const size_t lds = 2 * (BM + BN) * 128 > BM * BN * 4 ? 2 * (BM + BN) * 128 : BM * BN * 4;
    if constexpr (sizeof(T) == 2 && (BN == 64 || BN == 32)) {
        if (p.stat_mode != EG_STAT_NONE) {              // (nt_stat_blocks has checked: whole tiles, N == BN)
            // fp32 tile + the statistics' reduction scratch [2][rows lanes][BN] behind its first row
            const size_t need = (size_t)(1 + 2 * (256 / (BN / 8))) * BN * 4;
            if (ns > 1) hipLaunchKernelGGL((igemm_nt_kernel<T, BM, BN, WGM, WGN, true, true>), grid, dim3(256), lds > need ? lds : need, st, p);
            else hipLaunchKernelGGL((igemm_nt_kernel<T, BM, BN, WGM, WGN, true>), grid, dim3(256), lds > need ? lds : need, st, p);
            return;
        }
    }
    if constexpr (BN == 64 || BN == 32) {
        if (ns > 1) {
            hipLaunchKernelGGL((igemm_nt_kernel<T, BM, BN, WGM, WGN, false, true>), grid, dim3(256), lds, st, p);
            return;
        }
    }
    hipLaunchKernelGGL((igemm_nt_kernel<T, BM, BN, WGM, WGN>), grid, dim3(256), lds, st, p);
}

// K splits of the register-staged kernel (N = 32 / 64, NHWC output): few tiles and a deep K loop
// EXPERIMENT, default off (EG_NT_REG_SPLIT=1 or eg_epilogue.nt_splitk > 1 turn it on): correct (tested) but slower in the small networks'
// steps -- dSprites 1.415 -> 1.503 ms, colored / MNIST +0.2-0.8 % (profiles/r03_zg_ab_reg_split.txt): the 16-48-tile launches it targets
// are not the ~22 us the profiler shows for them once the profiler is off, and the split adds partial-tile traffic and a serial finish.
template <typename T>
static int nt_reg_splits(const NtParams& p, int nphase, int forced) {
    static const bool env_on = [] { const char* e = getenv("EG_NT_REG_SPLIT"); return e && atoi(e) != 0; }();
    const bool on = env_on || forced > 1;
    if (!on || p.out_mode != EG_OUT_NHWC || !(p.N == 64 || p.N == 32) || (p.N % Elt<T>::VEC) != 0 || !p.part || p.part_bytes <= EG_SPLIT_CNT_BYTES) return 1;
    const long long tiles = (long long)cdiv(p.M, 128) * nphase;
    int nk = 1 << 30;
    for (int i = 0; i < nphase; ++i) nk = std::min(nk, p.ph[i].Kpad / (8 * Elt<T>::VEC));
    if (tiles >= 96 || nk < 8 || tiles > EG_SPLIT_CNT_BYTES / 4) return 1;
    int ns = 1;
    while (tiles * ns < 128 && ns < 16 && nk / (ns * 2) >= 2 && (forced <= 1 || ns * 2 <= forced)) ns *= 2;
    const size_t per_split = (size_t)nphase * cdiv(p.M, 128) * 128 * p.N * 4;
    while (ns > 1 && (size_t)ns * per_split > p.part_bytes - EG_SPLIT_CNT_BYTES) ns /= 2;
    return ns;
}

template <typename T>
static void launch_splitk_epilogue(const NtParams& q, int nphase, int Mpad, hipStream_t st) {
    // split launches have at most a few thousand tiles: M * N / VEC is far below 2^31
    const int vpr = q.N / Elt<T>::VEC;
    const long long vecs = (long long)q.M * vpr;
    const int blocks = (int)std::min<long long>((vecs + 255) / 256, std::max(1, 256 * 16 / nphase));
    int lvpr = -1;
    if ((vpr & (vpr - 1)) == 0) { lvpr = 0; while ((1 << lvpr) < vpr) ++lvpr; }
    hipLaunchKernelGGL((nt_splitk_epilogue_kernel<T>), dim3(blocks, nphase), dim3(256), 0, st, q, nphase, Mpad, lvpr);
}

static bool nt_split_inkernel();
static int nt_stat_blocks(const NtParams& p, int nphase, const NtPlan& plan, bool half);

template <typename T>
static int launch_nt(const NtParams& p, int nphase, int variant, int splitk, hipStream_t st) {
    // the tail of the scratch is kept for igemm_nt8s's arrival counters: no variant's partial tiles may reach it
    const NtPlan plan = nt_plan(p, nphase, Elt<T>::VEC, sizeof(T), (p.part && p.part_bytes > EG_SPLIT_CNT_BYTES) ? p.part_bytes - EG_SPLIT_CNT_BYTES : 0, variant, splitk);
    EG_REQUIRE(plan.kind > 0, "eg_epilogue.nt_variant %d cannot run this problem (M=%d N=%d C=%d)", variant, p.M, p.N, p.C);
    const int stat_nrb = nt_stat_blocks(p, nphase, plan, sizeof(T) == 2);
    EG_REQUIRE(p.stat_mode == EG_STAT_NONE || stat_nrb > 0, "eg_epilogue.stat_mode is set but this launch cannot fuse column statistics (eg_conv_stat_blocks() == 0: M=%d N=%d C=%d)", p.M, p.N, p.C);
    static const int xcd = [] { const char* e = getenv("EG_XCD_REMAP"); return e ? atoi(e) : 1; }();   // default on (7: + diagnostic piece skipping in PROF builds)
    if (plan.kind == EG_NT_S8 || plan.kind == EG_NT_S8P) {
        NtParams q = p;
        q.nsplit = plan.ns;
        q.xcd_remap = xcd;
        q.stat_nrb = stat_nrb;
        Nt8pGeom g;
        memset(&g, 0, sizeof(g));
        if (plan.kind == EG_NT_S8P) EG_REQUIRE(eg_nt8p_geometry(p, nphase, g), "patch geometry");
        const bool inkernel = nt_split_inkernel();
        q.split_cnt = (plan.ns > 1 && inkernel) ? reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.part) + p.part_bytes - EG_SPLIT_CNT_BYTES) : nullptr;
        eg_launch_nt8s<T>(q, g, plan.kind == EG_NT_S8P, nphase, plan.ns, st);      // K splits are reduced inside the launch (last-arriving workgroup)
        if (plan.ns > 1 && !inkernel) launch_splitk_epilogue<T>(q, nphase, cdiv(p.M, 256) * 256, st);
        return 0;
    }
    if (plan.kind == EG_NT_S8H) {
        NtParams q = p;
        q.nsplit = plan.ns;
        q.xcd_remap = xcd;
        q.stat_nrb = stat_nrb;
        EG_REQUIRE(plan.ns == 1 || nt_split_inkernel(), "igemm_nt8h reduces its K splits inside the launch only (EG_NT_SPLIT_INKERNEL=0 is an igemm_nt8s experiment)");
        q.split_cnt = plan.ns > 1 ? reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.part) + p.part_bytes - EG_SPLIT_CNT_BYTES) : nullptr;
        eg_launch_nt8h<T>(q, nphase, plan.ns, st);
        return 0;
    }
    if (plan.kind == EG_NT_PERS) {
        static bool attr_set = false;
        const size_t lds = 2 * (128 + 128) * 128;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt_pers_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        const int tm = cdiv(p.M, 128), tn = p.N / 128, ntiles = tm * tn * nphase;
        hipLaunchKernelGGL((igemm_nt_pers_kernel<T>), dim3(std::min(ntiles, 512)), dim3(256), lds, st, p, tm, tn, ntiles);
        return 0;
    }
    if (plan.kind == EG_NT_BUF128) {
        const int ns = plan.ns;
        static bool attr_set = false;
        const size_t lds = 2 * (128 + 128) * 128;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt_buf_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        NtParams q = p;
        q.nsplit = ns;
        q.xcd_remap = xcd;
        // diagnostic (EG_NT_PROF=1 in the environment): run the instrumented instantiation synchronously and print the per-wave averages of
        // its phase timers (wait+barrier / LDS-DMA issue / ds_read+MFMA / epilogue) -- how DESIGN.md section 6's round-1 breakdown was measured
        static const char* prof_env = getenv("EG_NT_PROF");
        if (prof_env && ns == 1) {
            const dim3 grid(cdiv(p.M, 128), p.N / 128, nphase);
            const size_t nw = (size_t)grid.x * grid.y * grid.z * 4;
            unsigned long long* dbuf = nullptr;
            (void)hipMalloc(&dbuf, nw * 64);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt_buf_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((igemm_nt_buf_kernel<T, true>), grid, dim3(256), lds, st, q, dbuf);
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> h(nw * 8);
            (void)hipMemcpy(h.data(), dbuf, nw * 64, hipMemcpyDeviceToHost);
            (void)hipFree(dbuf);
            double tot = 0, wt = 0, is = 0, cp = 0, ep = 0;
            for (size_t w = 0; w < nw; ++w) { tot += h[w * 8]; wt += h[w * 8 + 1]; is += h[w * 8 + 2]; cp += h[w * 8 + 3]; ep += h[w * 8 + 4]; }
            const int nk = p.ph[0].Kpad / (8 * Elt<T>::VEC);
            fprintf(stderr, "[nt_prof] grid %ux%ux%u nk %d | per wave (s_memtime ticks): total %.0f  wait+barrier %.0f (%.1f/step)  issue %.0f (%.1f/step)  compute %.0f (%.1f/step)  epilogue %.0f\n",
                    grid.x, grid.y, grid.z, nk, tot / nw, wt / nw, wt / nw / nk, is / nw, is / nw / nk, cp / nw, cp / nw / nk, ep / nw);
            return 0;
        }
        if (ns > 1) {
            static bool attr_split = false;
            if (!attr_split) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt_buf_kernel<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_split = true;
            }
            hipLaunchKernelGGL((igemm_nt_buf_kernel<T, false, true>), dim3(cdiv(p.M, 128), p.N / 128, nphase * ns), dim3(256), lds, st, q);
            launch_splitk_epilogue<T>(q, nphase, cdiv(p.M, 128) * 128, st);
        } else
            hipLaunchKernelGGL((igemm_nt_buf_kernel<T>), dim3(cdiv(p.M, 128), p.N / 128, nphase), dim3(256), lds, st, q);
        return 0;
    }
    NtParams q = p;
    q.stat_nrb = stat_nrb;
    q.nsplit = nt_reg_splits<T>(p, nphase, splitk);
    q.split_cnt = q.nsplit > 1 ? reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.part) + p.part_bytes - EG_SPLIT_CNT_BYTES) : nullptr;
    if (p.N <= 16)
        launch_nt_cfg<T, 128, 16, 4, 1>(q, nphase, st);
    else if (p.N <= 32)
        launch_nt_cfg<T, 128, 32, 4, 1>(q, nphase, st);
    else if (p.N <= 64 || (variant != EG_NT_REG && (long long)cdiv(p.M, 128) * cdiv(p.N, 128) * nphase < 512))
        launch_nt_cfg<T, 128, 64, 2, 2>(q, nphase, st);
    else
        launch_nt_cfg<T, 128, 128, 2, 2>(q, nphase, st);
    return 0;
}

/* which igemm_nt instantiation eg_conv_fwd (bwd = 0) / eg_conv_bwd_data (bwd = 1) dispatches this problem to under the given hints
 * (profiling labels and tests; the same planner as the launches, with unlimited split-K scratch): BM * 1000 + code, code = BN of the
 * register-staged kernels, 131 / 132 = 128 x 128 buffer-descriptor kernel (plain / split-K), 135 = persistent pipeline, 147 / 148 =
 * igemm_nt8s, 149 / 150 = igemm_nt8s with the input patch; -1 = the forced variant cannot run the problem. */
extern "C" int eg_igemm_nt_tile(const eg_conv* c, int dtype, int bwd, int variant, int splitk) {
    if (!c || check_conv(c, dtype, bwd ? NEED_COUT : NEED_CIN)) return -1;
    NtParams p;
    memset(&p, 0, sizeof(p));
    int nphase = 1;
    if (bwd) { if (geom_bwd(c, dtype, p, &nphase)) return -1; }
    else geom_fwd(c, dtype, p);
    p.out_mode = EG_OUT_NHWC;
    const NtPlan plan = nt_plan(p, nphase, vec_of(dtype), dtype == EG_F32 ? 4 : 2, (size_t)1 << 40, variant, splitk);
    if (plan.kind < 0) return -1;
    if (plan.kind == EG_NT_S8) return 256 * 1000 + (plan.ns > 1 ? 148 : 147);
    if (plan.kind == EG_NT_S8P) return 256 * 1000 + (plan.ns > 1 ? 150 : 149);
    if (plan.kind == EG_NT_S8H) return 128 * 1000 + (plan.ns > 1 ? 152 : 151);
    if (plan.kind == EG_NT_PERS) return 128 * 1000 + 135;
    if (plan.kind == EG_NT_BUF128) return 128 * 1000 + (plan.ns > 1 ? 132 : 131);
    const int M = p.M, N = p.N;
    if (N <= 16) return 128 * 1000 + 16;
    if (N <= 32) return 128 * 1000 + 32;
    if (N <= 64 || (variant != EG_NT_REG && (long long)cdiv(M, 128) * cdiv(N, 128) * nphase < 512)) return 128 * 1000 + 64;
    return 128 * 1000 + 128;
}

static void fill_epilogue(NtParams& p, const eg_epilogue* ep) {
    p.bias = ep ? ep->bias : nullptr;
    p.bias_mod = ep ? ep->bias_mod : 0;
    p.sigma = ep ? ep->sigma : nullptr;
    p.act = ep ? ep->act : EG_ACT_NONE;
    p.slope = ep ? ep->slope : 0.f;
    p.mask = ep ? ep->mask : nullptr;
    p.mask_act = ep ? ep->mask_act : EG_ACT_NONE;
    p.mask_slope = ep ? ep->mask_slope : 0.f;
    p.out_mode = ep ? ep->out_mode : EG_OUT_NHWC;
    p.sigma_rows = ep ? ep->sigma_rows : 0;
    p.part = ep ? reinterpret_cast<float*>(ep->splitk_ws) : nullptr;
    p.part_bytes = ep ? ep->splitk_ws_bytes : 0;
    p.stat_mode = ep ? ep->stat_mode : EG_STAT_NONE;
    p.stat_out = ep ? ep->stat_out : nullptr;
    p.stat_aux = ep ? ep->stat_aux : nullptr;
    p.stat_p[0] = ep ? ep->stat_p0 : nullptr; p.stat_p[1] = ep ? ep->stat_p1 : nullptr;
    p.stat_p[2] = ep ? ep->stat_p2 : nullptr; p.stat_p[3] = ep ? ep->stat_p3 : nullptr;
    p.stat_act = ep ? ep->stat_act : EG_ACT_NONE;
    p.stat_slope = ep ? ep->stat_slope : 0.f;
}

// row blocks of the fused column statistics if this plan can produce them (the 8-wave kernel on whole 256-row tiles, K splits reduced
// inside the launch), else 0
static bool nt_split_inkernel() {
    static const bool inkernel = [] { const char* e = getenv("EG_NT_SPLIT_INKERNEL"); return !(e && atoi(e) == 0); }();
    return inkernel;
}
static int nt_stat_blocks(const NtParams& p, int nphase, const NtPlan& plan, bool half) {
    if (!half || p.out_mode != EG_OUT_NHWC) return 0;
    if (plan.kind == EG_NT_S8H) return (p.M % 128) == 0 ? nphase * (p.M / 128) : 0;          // row blocks of 128
    // the register-staged kernel on 128 x 64 / 128 x 32 tiles (the small networks' layers): one column tile, whole row tiles
    static const bool reg_stat = [] { const char* e = getenv("EG_NT_REG_STAT"); return !(e && atoi(e) == 0); }();
    if (plan.kind == EG_NT_REG) return (reg_stat && (p.N == 64 || p.N == 32) && (p.M % 128) == 0) ? nphase * (p.M / 128) : 0;
    if (plan.kind != EG_NT_S8 || (p.M % 256) != 0) return 0;
    if (plan.ns > 1 && !nt_split_inkernel()) return 0;
    return nphase * (p.M / 256);
}
static int check_stat(const NtParams& p) {
    if (p.stat_mode == EG_STAT_NONE) return 0;
    EG_REQUIRE(p.stat_mode >= EG_STAT_MOMENTS && p.stat_mode <= EG_STAT_SN_BIAS && p.stat_out, "eg_epilogue.stat_mode %d: bad mode or no stat_out", p.stat_mode);
    if (p.stat_mode == EG_STAT_BN_BWD)
        EG_REQUIRE(p.stat_aux && p.stat_p[0] && p.stat_p[1] && p.stat_p[2] && p.stat_p[3] && !p.mask, "EG_STAT_BN_BWD needs stat_aux (z), stat_p0..p3 (mean, invstd, gamma, beta) and no mask");
    if (p.stat_mode == EG_STAT_SN_BIAS)
        EG_REQUIRE(p.mask && p.mask_act == EG_ACT_LRELU && p.mask_slope > 0.f && p.stat_p[0] && p.stat_slope == p.mask_slope, "EG_STAT_SN_BIAS needs the LeakyReLU mask, stat_p0 (bias) and stat_slope == mask_slope");
    return 0;
}

extern "C" int eg_conv_fwd(const eg_conv* c, int dtype, const void* X, const void* wp_fwd, void* Y,
                           const eg_epilogue* ep, eg_stream_t s) {
    if (int e = check_conv(c, dtype, NEED_CIN)) return e;
    EG_REQUIRE(X && wp_fwd && Y, "eg_conv_fwd: null pointer");
    NtParams p;
    memset(&p, 0, sizeof(p));
    geom_fwd(c, dtype, p);
    p.src = X; p.wp = wp_fwd; p.dst = Y;
    fill_epilogue(p, ep);
    if (int e = check_stat(p)) return e;
    const int variant = ep ? ep->nt_variant : EG_NT_AUTO, splitk = ep ? ep->nt_splitk : 0;
    int rc;
    if (dtype == EG_F32) rc = launch_nt<float>(p, 1, variant, splitk, (hipStream_t)s);
    else if (dtype == EG_F16) rc = launch_nt<f16_t>(p, 1, variant, splitk, (hipStream_t)s);
    else rc = launch_nt<bf16_t>(p, 1, variant, splitk, (hipStream_t)s);
    if (rc) return rc;
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_conv_bwd_data(const eg_conv* c, int dtype, const void* dY, const void* wp_bwd, void* dX,
                                const eg_epilogue* ep, eg_stream_t s) {
    if (int e = check_conv(c, dtype, NEED_COUT)) return e;
    EG_REQUIRE(dY && wp_bwd && dX, "eg_conv_bwd_data: null pointer");
    NtParams p;
    memset(&p, 0, sizeof(p));
    int nphase = 0;
    if (int e = geom_bwd(c, dtype, p, &nphase)) return e;
    p.src = dY; p.wp = wp_bwd; p.dst = dX;
    fill_epilogue(p, ep);
    if (int e = check_stat(p)) return e;
    const int variant = ep ? ep->nt_variant : EG_NT_AUTO, splitk = ep ? ep->nt_splitk : 0;
    int rc;
    if (dtype == EG_F32) rc = launch_nt<float>(p, nphase, variant, splitk, (hipStream_t)s);
    else if (dtype == EG_F16) rc = launch_nt<f16_t>(p, nphase, variant, splitk, (hipStream_t)s);
    else rc = launch_nt<bf16_t>(p, nphase, variant, splitk, (hipStream_t)s);
    if (rc) return rc;
    EG_LAUNCH_CHECK();
    return 0;
}

/* eg_igemm_nt_tile for one concrete call: the planner exactly as the launch runs it -- the epilogue's kernel hints AND its split-K scratch
 * (a launch whose scratch cannot hold the partial tiles does not split) */
extern "C" int eg_igemm_nt_tile_ep(const eg_conv* c, int dtype, int bwd, const eg_epilogue* ep) {
    if (!c || check_conv(c, dtype, bwd ? NEED_COUT : NEED_CIN)) return -1;
    NtParams p;
    memset(&p, 0, sizeof(p));
    int nphase = 1;
    if (bwd) { if (geom_bwd(c, dtype, p, &nphase)) return -1; }
    else geom_fwd(c, dtype, p);
    fill_epilogue(p, ep);
    const size_t ws = (p.part && p.part_bytes > EG_SPLIT_CNT_BYTES) ? p.part_bytes - EG_SPLIT_CNT_BYTES : 0;
    const int variant = ep ? ep->nt_variant : EG_NT_AUTO;
    const NtPlan plan = nt_plan(p, nphase, vec_of(dtype), dtype == EG_F32 ? 4 : 2, ws, variant, ep ? ep->nt_splitk : 0);
    if (plan.kind < 0) return -1;
    if (plan.kind == EG_NT_S8) return 256 * 1000 + (plan.ns > 1 ? 148 : 147);
    if (plan.kind == EG_NT_S8P) return 256 * 1000 + (plan.ns > 1 ? 150 : 149);
    if (plan.kind == EG_NT_S8H) return 128 * 1000 + (plan.ns > 1 ? 152 : 151);
    if (plan.kind == EG_NT_PERS) return 128 * 1000 + 135;
    if (plan.kind == EG_NT_BUF128) return 128 * 1000 + (plan.ns > 1 ? 132 : 131);
    const int M = p.M, N = p.N;
    if (N <= 16) return 128 * 1000 + 16;
    if (N <= 32) return 128 * 1000 + 32;
    if (N <= 64 || (variant != EG_NT_REG && (long long)cdiv(M, 128) * cdiv(N, 128) * nphase < 512)) return 128 * 1000 + 64;
    return 128 * 1000 + 128;
}

extern "C" int eg_conv_stat_blocks(const eg_conv* c, int dtype, int bwd, const eg_epilogue* ep) {
    if (!c || check_conv(c, dtype, bwd ? NEED_COUT : NEED_CIN)) return 0;
    NtParams p;
    memset(&p, 0, sizeof(p));
    int nphase = 1;
    if (bwd) { if (geom_bwd(c, dtype, p, &nphase)) return 0; }
    else geom_fwd(c, dtype, p);
    fill_epilogue(p, ep);
    // the planner exactly as launch_nt runs it for this call: same hints, same scratch
    const size_t ws = (p.part && p.part_bytes > EG_SPLIT_CNT_BYTES) ? p.part_bytes - EG_SPLIT_CNT_BYTES : 0;
    const NtPlan plan = nt_plan(p, nphase, vec_of(dtype), dtype == EG_F32 ? 4 : 2, ws, ep ? ep->nt_variant : EG_NT_AUTO, ep ? ep->nt_splitk : 0);
    if (plan.kind < 0) return 0;
    return nt_stat_blocks(p, nphase, plan, dtype != EG_F32);
}

extern "C" size_t eg_conv_splitk_ws_bytes(const eg_conv* c, int dtype, int bwd) {
    if (!c || check_conv(c, dtype, bwd ? NEED_COUT : NEED_CIN)) return 0;
    NtParams p;
    memset(&p, 0, sizeof(p));
    int nphase = 1;
    if (bwd) { if (geom_bwd(c, dtype, p, &nphase)) return 0; }
    else geom_fwd(c, dtype, p);
    p.out_mode = EG_OUT_NHWC;
    // what the planner would split into with unlimited scratch (callers size one shared scratch from the maximum over their layers)
    const NtPlan plan = nt_plan(p, nphase, vec_of(dtype), dtype == EG_F32 ? 4 : 2, (size_t)1 << 40, EG_NT_AUTO, 0);
    if (plan.ns <= 1) return 0;
    const int bm = plan.kind >= EG_NT_S8 ? 256 : 128;
    return (size_t)plan.ns * nphase * cdiv(p.M, bm) * bm * p.N * 4 + EG_SPLIT_CNT_BYTES;      // + the arrival counters at the scratch's tail
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 master [Cout][Cin][k][k]  ->  K-contiguous dtype-T panels
// ------------------------------------------------------------------------------------------------
struct PackPhase { int TH, TW, kh0, kw0, K, Kpad; long long w_off; };
struct PackParams {
    const float* w;
    void* wp;
    int Nrows;           // rows of the packed panel
    int Crow;            // channels per tap in a row
    int khs, kws, k;
    long long n_stride, c_stride;   // master strides of the (row, channel) indices
    int nphase;
    PackPhase ph[4];
};

// (the pack kernels' bodies take their block coordinates as arguments: pack_multi_kernel below runs several layers' packs in one launch)
template <typename T>
__device__ __forceinline__ void pack_gather_body(const PackParams& p, int bx, int by, int gx) {
    const PackPhase ph = p.ph[by];
    const long long total = (long long)p.Nrows * ph.Kpad;
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < total; i += (long long)gx * blockDim.x) {
        const int n = (int)(i / ph.Kpad), kk = (int)(i % ph.Kpad);
        float v = 0.f;
        if (kk < ph.K) {
            const int t = kk / p.Crow, c = kk % p.Crow;
            const int ty = t / ph.TW, tx = t % ph.TW;
            const int kh = ph.kh0 + ty * p.khs, kw = ph.kw0 + tx * p.kws;
            v = p.w[n * p.n_stride + c * p.c_stride + kh * p.k + kw];
        }
        Elt<T>::st(reinterpret_cast<T*>(p.wp) + ph.w_off + i, v);
    }
}
template <typename T>
__global__ void pack_kernel(const PackParams p) { pack_gather_body<T>(p, blockIdx.x, blockIdx.y, gridDim.x); }

struct PackTileParams;
static bool pack_record_gather(const PackParams& p, int dtype, int gx, int gy);
static bool pack_record_strided(int kind, int dtype, const float* w, void* wp, int N, int K, int Kpad, int n_div, long long s_hi, long long s_lo,
                                long long s_k, int k_div, long long s_khi, long long s_klo, int gx);
static bool pack_record_tile(const PackTileParams& p, int dtype, int gx, int gy);

static void launch_pack(const PackParams& p, int dtype, hipStream_t st) {
    long long total = 0;
    for (int i = 0; i < p.nphase; ++i) total = total > (long long)p.Nrows * p.ph[i].Kpad ? total : (long long)p.Nrows * p.ph[i].Kpad;
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    if (pack_record_gather(p, dtype, blocks, p.nphase)) return;
    if (dtype == EG_F32) hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks, p.nphase), dim3(256), 0, st, p);
    else if (dtype == EG_F16) hipLaunchKernelGGL(pack_kernel<f16_t>, dim3(blocks, p.nphase), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(blocks, p.nphase), dim3(256), 0, st, p);
}

extern "C" size_t eg_pack_fwd_elems(const eg_conv* c, int dtype) {
    return (size_t)c->Cout * kpad_of(c->k * c->k * c->Cin, c->Cout, dtype);
}
extern "C" size_t eg_pack_bwd_elems(const eg_conv* c, int dtype) {
    size_t tot = 0;
    for (int ry = 0; ry < c->stride; ++ry)
        for (int rx = 0; rx < c->stride; ++rx) {
            const int K = bwd_axis(c, ry).T * bwd_axis(c, rx).T * c->Cout;
            tot += (size_t)c->Cin * kpad_of(K, c->Cin, dtype);
        }
    return tot;
}

extern "C" int eg_pack_fwd(const eg_conv* c, int dtype, const float* w, void* wp, eg_stream_t s) {
    EG_REQUIRE(c && w && wp, "eg_pack_fwd: null pointer");
    PackParams p;
    memset(&p, 0, sizeof(p));
    p.w = w; p.wp = wp; p.Nrows = c->Cout; p.Crow = c->Cin;
    p.khs = p.kws = 1; p.k = c->k;
    p.n_stride = (long long)c->Cin * c->k * c->k; p.c_stride = c->k * c->k;
    p.nphase = 1;
    p.ph[0].TH = p.ph[0].TW = c->k; p.ph[0].kh0 = p.ph[0].kw0 = 0;
    p.ph[0].K = c->k * c->k * c->Cin; p.ph[0].Kpad = kpad_of(p.ph[0].K, c->Cout, dtype); p.ph[0].w_off = 0;
    launch_pack(p, dtype, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_pack_bwd(const eg_conv* c, int dtype, const float* w, void* wp, eg_stream_t s) {
    EG_REQUIRE(c && w && wp && c->stride <= 2, "eg_pack_bwd: bad argument");
    PackParams p;
    memset(&p, 0, sizeof(p));
    p.w = w; p.wp = wp; p.Nrows = c->Cin; p.Crow = c->Cout;
    p.khs = p.kws = c->stride; p.k = c->k;
    p.n_stride = c->k * c->k; p.c_stride = (long long)c->Cin * c->k * c->k;
    long long off = 0;
    int n = 0;
    for (int ry = 0; ry < c->stride; ++ry)
        for (int rx = 0; rx < c->stride; ++rx) {
            const BwdAxis ay = bwd_axis(c, ry), ax = bwd_axis(c, rx);
            PackPhase& f = p.ph[n++];
            f.TH = ay.T; f.TW = ax.T > 0 ? ax.T : 1; f.kh0 = ay.k0; f.kw0 = ax.k0;
            f.K = ay.T * ax.T * c->Cout; f.Kpad = kpad_of(f.K, c->Cin, dtype); f.w_off = off;
            off += (long long)c->Cin * f.Kpad;
        }
    p.nphase = n;
    launch_pack(p, dtype, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Both panels of a conv layer in one pass over its fp32 master [Cout][Cin][k*k] (conv view): a workgroup stages a 16 (Cout) x 32 (Cin)
// x taps tile in LDS with coalesced reads and writes it out twice as 16-byte vectors -- K-contiguous rows [n][tap][c] of the forward
// panel and [c][tap'][n] rows of every backward (sub-pixel phase) panel.  The per-element gather above reads the master with a
// 64-byte (forward) or Cin*64-byte (backward) stride and was 5-9 % of a CelebA iteration.
// ------------------------------------------------------------------------------------------------
#define EG_PACKM_TC 8       // Cin tile of the tile jobs inside pack_multi_kernel (the stand-alone tile kernel uses 32)
struct PackTileParams {
    const float* w;
    void* wp_fwd;          // may be null
    void* wp_bwd;          // may be null
    int Cout, Cin, T;      // T = k*k
    int nq;                // (phase, tap') combinations of the backward panels
    int t_of[16];          // master tap index of combination q
    int pitch[16];         // row pitch (Kpad) of q's phase
    long long off[16];     // element offset of (c = 0, n = 0) of combination q inside wp_bwd
};

// TTC: taps (k*k) as a compile-time constant (16 for every 4x4 layer; 0 = run time): the index arithmetic below divides by it per element
template <typename T, int TTC, int TC = 32>
__device__ __forceinline__ void pack_conv_tile_body(const PackTileParams& p, float* tile, int bx, int by) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int TN = 16;
    const int TT = TTC ? TTC : p.T, TP = TT + 1;
    const int n0 = bx * TN, c0 = by * TC;
    const int tid = threadIdx.x;
    for (int e = tid; e < TN * TC * TT; e += 256) {
        const int n = e / (TC * TT), rem = e - n * (TC * TT);
        const int c = rem / TT, t = rem - c * TT;
        tile[(n * TC + c) * TP + t] = p.w[((size_t)(n0 + n) * p.Cin + c0) * TT + rem];
    }
    __syncthreads();
    if (p.wp_fwd) {
        T* dst = reinterpret_cast<T*>(p.wp_fwd);
        const size_t pitch = (size_t)TT * p.Cin;
        for (int it = tid; it < TN * TT * (TC / VEC); it += 256) {
            const int cg = it % (TC / VEC), r = it / (TC / VEC);
            const int t = r % TT, n = r / TT;
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int j = 0; j < VEC; ++j) Elt<T>::st(oe + j, tile[(n * TC + cg * VEC + j) * TP + t]);
            *reinterpret_cast<uint4*>(dst + (size_t)(n0 + n) * pitch + (size_t)t * p.Cin + c0 + cg * VEC) = ov;
        }
    }
    if (p.wp_bwd) {
        T* dst = reinterpret_cast<T*>(p.wp_bwd);
        for (int it = tid; it < TC * p.nq * (TN / VEC); it += 256) {
            const int ng = it % (TN / VEC), r = it / (TN / VEC);
            const int q = r % p.nq, c = r / p.nq;
            const int t = p.t_of[q];
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int j = 0; j < VEC; ++j) Elt<T>::st(oe + j, tile[((ng * VEC + j) * TC + c) * TP + t]);
            *reinterpret_cast<uint4*>(dst + p.off[q] + (size_t)(c0 + c) * p.pitch[q] + n0 + ng * VEC) = ov;
        }
    }
}
template <typename T, int TTC, int TC = 32>
__global__ __launch_bounds__(256) void pack_conv_tile_kernel(const PackTileParams p) {
    extern __shared__ float tile[];                 // [TN][TC][T + 1]
    pack_conv_tile_body<T, TTC, TC>(p, tile, blockIdx.x, blockIdx.y);
}

// tile-kernel parameters of a layer, or false where only the per-element gather kernels can pack it (ragged channel counts, K padding)
static bool pack_tile_params(const eg_conv* c, int dtype, void* wp_fwd, void* wp_bwd, PackTileParams& p) {
    const int T = c->k * c->k, bk = bk_of(dtype);
    bool fast = (c->Cout % 16) == 0 && (c->Cin % 32) == 0 && T <= 16 && (!wp_bwd || c->stride <= 2);
    memset(&p, 0, sizeof(p));
    p.wp_fwd = wp_fwd; p.wp_bwd = wp_bwd; p.Cout = c->Cout; p.Cin = c->Cin; p.T = T;
    if (fast && wp_fwd && (T * c->Cin) % bk != 0) fast = false;          // K padding: the gather kernel zero-fills
    if (fast && wp_bwd) {
        long long off = 0;
        for (int ry = 0; ry < c->stride && fast; ++ry)
            for (int rx = 0; rx < c->stride && fast; ++rx) {
                const BwdAxis ay = bwd_axis(c, ry), ax = bwd_axis(c, rx);
                const int K = ay.T * ax.T * c->Cout;
                const int Kpad = kpad_of(K, c->Cin, dtype);
                if (K != Kpad) { fast = false; break; }
                for (int ty = 0; ty < ay.T; ++ty)
                    for (int tx = 0; tx < ax.T; ++tx) {
                        const int kh = ay.k0 + ty * c->stride, kw = ax.k0 + tx * c->stride;
                        p.t_of[p.nq] = kh * c->k + kw;
                        p.pitch[p.nq] = Kpad;
                        p.off[p.nq] = off + (long long)(ty * ax.T + tx) * c->Cout;
                        ++p.nq;
                    }
                off += (long long)c->Cin * Kpad;
            }
    }
    return fast;
}

extern "C" int eg_pack_conv(const eg_conv* c, int dtype, const float* w, void* wp_fwd, void* wp_bwd, eg_stream_t s) {
    EG_REQUIRE(c && w && (wp_fwd || wp_bwd), "eg_pack_conv: null pointer");
    EG_REQUIRE(dtype == EG_F32 || dtype == EG_BF16 || dtype == EG_F16, "dtype must be EG_F32, EG_BF16 or EG_F16");
    const int T = c->k * c->k;
    PackTileParams p;
    const bool fast = pack_tile_params(c, dtype, wp_fwd, wp_bwd, p);
    p.w = w;
    if (!fast) {
        if (wp_fwd) if (int e = eg_pack_fwd(c, dtype, w, wp_fwd, s)) return e;
        if (wp_bwd) if (int e = eg_pack_bwd(c, dtype, w, wp_bwd, s)) return e;
        return 0;
    }
    // Cin tile: 32 (64-byte panel stores, 34 KiB of LDS) or 8 (16-byte stores, 8.5 KiB: fits on a CU beside a resident 8-wave GEMM workgroup:
    // the re-packing on the optimizer lane then runs beside the next sub-step's GEMMs; EG_PACK_TC, A/B in profiles/r03_zm_ab_pack_tc.txt)
    static const int tc = [] { const char* e = getenv("EG_PACK_TC"); return (e && atoi(e) == 32) ? 32 : 8; }();
    const dim3 grid(c->Cout / 16, c->Cin / tc);
    const size_t lds = (size_t)16 * tc * (T + 1) * sizeof(float);
    if (pack_record_tile(p, dtype, grid.x, c->Cin / EG_PACKM_TC)) return 0;     // (the joint launch tiles Cin by EG_PACKM_TC)
#define EG_PACK_TILE(TY) do { if (tc == 32) { if (T == 16) hipLaunchKernelGGL((pack_conv_tile_kernel<TY, 16>), grid, dim3(256), lds, (hipStream_t)s, p); \
                                              else hipLaunchKernelGGL((pack_conv_tile_kernel<TY, 0>), grid, dim3(256), lds, (hipStream_t)s, p); } \
                              else { if (T == 16) hipLaunchKernelGGL((pack_conv_tile_kernel<TY, 16, 8>), grid, dim3(256), lds, (hipStream_t)s, p); \
                                     else hipLaunchKernelGGL((pack_conv_tile_kernel<TY, 0, 8>), grid, dim3(256), lds, (hipStream_t)s, p); } } while (0)
    if (dtype == EG_F32) EG_PACK_TILE(float);
    else if (dtype == EG_F16) EG_PACK_TILE(f16_t);
    else EG_PACK_TILE(bf16_t);
#undef EG_PACK_TILE
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Adam + re-packing in one pass (torch.optim.Adam.step() on a convolution's weight, celebA/EAD-GAN_celebA.py:344,365,400, followed by
// the panel refresh every consumer of the weight needs): the pack_conv_tile tiling with the optimizer update applied while the master
// tile is on its way into LDS -- p, g, m, v are read once, p, m, v (and the cleared g) written once, both panels written from the tile.
// Same element update as adam_kernel (adam.h), same panel bytes as eg_pack_conv of the updated master.
// ------------------------------------------------------------------------------------------------
template <typename T, int TTC, bool ZERO, bool V4, int TC>
__global__ __launch_bounds__(256) void adam_pack_conv_tile_kernel(const PackTileParams p, float* __restrict__ w, float* __restrict__ g,
                                                                  float* __restrict__ m, float* __restrict__ v, float lr, float b1, float b2,
                                                                  float eps, const int* __restrict__ step) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int TN = 16;
    extern __shared__ float tile[];                 // [TN][TC][T + 1]
    const int TT = TTC ? TTC : p.T, TP = TT + 1;
    const int n0 = blockIdx.x * TN, c0 = blockIdx.y * TC;
    const int tid = threadIdx.x;
    const AdamCoef ac = adam_coef(lr, b1, b2, eps, step);
    if (V4) {
        // 16-byte loads and stores (the four slices are 16-byte aligned and TT is a multiple of 4: a float4 stays inside one (n, c) run),
        // four vectors per thread in flight -- the scalar form below moved 1.2 TB/s, the flat-arena Adam kernel moves 5.9
        const int total4 = TN * TC * TT / 4;
        constexpr int U = TC >= 32 ? 4 : 2;           // vectors per thread in flight (the narrow tile keeps the kernel under 48 registers)
#pragma unroll 1
        for (int e0 = tid; e0 < total4; e0 += U * 256) {
            float4 pi[U], gi[U], mi[U], vi[U];
            size_t idx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = (e0 + u * 256) * 4;
                const int n = e / (TC * TT), rem = e - n * (TC * TT);
                idx[u] = ((size_t)(n0 + n) * p.Cin + c0) * TT + rem;
                if (e0 + u * 256 < total4) {
                    pi[u] = *reinterpret_cast<const float4*>(w + idx[u]); gi[u] = *reinterpret_cast<const float4*>(g + idx[u]);
                    mi[u] = *reinterpret_cast<const float4*>(m + idx[u]); vi[u] = *reinterpret_cast<const float4*>(v + idx[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (e0 + u * 256 >= total4) continue;
                adam_elem(pi[u].x, gi[u].x, mi[u].x, vi[u].x, ac);
                adam_elem(pi[u].y, gi[u].y, mi[u].y, vi[u].y, ac);
                adam_elem(pi[u].z, gi[u].z, mi[u].z, vi[u].z, ac);
                adam_elem(pi[u].w, gi[u].w, mi[u].w, vi[u].w, ac);
                *reinterpret_cast<float4*>(w + idx[u]) = pi[u];
                *reinterpret_cast<float4*>(m + idx[u]) = mi[u];
                *reinterpret_cast<float4*>(v + idx[u]) = vi[u];
                if (ZERO) *reinterpret_cast<float4*>(g + idx[u]) = make_float4(0.f, 0.f, 0.f, 0.f);
                const int e = (e0 + u * 256) * 4;
                const int n = e / (TC * TT), rem = e - n * (TC * TT);
                const int c = rem / TT, t = rem - c * TT;
                float* tp = tile + (n * TC + c) * TP + t;
                tp[0] = pi[u].x; tp[1] = pi[u].y; tp[2] = pi[u].z; tp[3] = pi[u].w;
            }
        }
    } else
    for (int e = tid; e < TN * TC * TT; e += 256) {
        const int n = e / (TC * TT), rem = e - n * (TC * TT);
        const int c = rem / TT, t = rem - c * TT;
        const size_t i = ((size_t)(n0 + n) * p.Cin + c0) * TT + rem;
        float pi = w[i], mi = m[i], vi = v[i];
        adam_elem(pi, g[i], mi, vi, ac);
        w[i] = pi; m[i] = mi; v[i] = vi;
        if (ZERO) g[i] = 0.f;
        tile[(n * TC + c) * TP + t] = pi;
    }
    __syncthreads();
    if (p.wp_fwd) {
        T* dst = reinterpret_cast<T*>(p.wp_fwd);
        const size_t pitch = (size_t)TT * p.Cin;
        for (int it = tid; it < TN * TT * (TC / VEC); it += 256) {
            const int cg = it % (TC / VEC), r = it / (TC / VEC);
            const int t = r % TT, n = r / TT;
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int j = 0; j < VEC; ++j) Elt<T>::st(oe + j, tile[(n * TC + cg * VEC + j) * TP + t]);
            *reinterpret_cast<uint4*>(dst + (size_t)(n0 + n) * pitch + (size_t)t * p.Cin + c0 + cg * VEC) = ov;
        }
    }
    if (p.wp_bwd) {
        T* dst = reinterpret_cast<T*>(p.wp_bwd);
        for (int it = tid; it < TC * p.nq * (TN / VEC); it += 256) {
            const int ng = it % (TN / VEC), r = it / (TN / VEC);
            const int q = r % p.nq, c = r / p.nq;
            const int t = p.t_of[q];
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int j = 0; j < VEC; ++j) Elt<T>::st(oe + j, tile[((ng * VEC + j) * TC + c) * TP + t]);
            *reinterpret_cast<uint4*>(dst + p.off[q] + (size_t)(c0 + c) * p.pitch[q] + n0 + ng * VEC) = ov;
        }
    }
}

extern "C" int eg_adam_pack_conv_ok(const eg_conv* c, int dtype, int has_fwd, int has_bwd) {
    PackTileParams p;
    if (!c || (!has_fwd && !has_bwd) || c->k * c->k > 16) return 0;
    return pack_tile_params(c, dtype, has_fwd ? (void*)1 : nullptr, has_bwd ? (void*)1 : nullptr, p) ? 1 : 0;
}

extern "C" int eg_adam_pack_conv(const eg_conv* c, int dtype, float* w, float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                                 const int* step, int zero_grad, void* wp_fwd, void* wp_bwd, eg_stream_t s) {
    EG_REQUIRE(c && w && g && m && v && step && (wp_fwd || wp_bwd), "eg_adam_pack_conv: null pointer");
    EG_REQUIRE(dtype == EG_F32 || dtype == EG_BF16 || dtype == EG_F16, "dtype must be EG_F32, EG_BF16 or EG_F16");
    const int T = c->k * c->k;
    PackTileParams p;
    EG_REQUIRE(pack_tile_params(c, dtype, wp_fwd, wp_bwd, p), "eg_adam_pack_conv: this layer needs the gather kernels (eg_adam_pack_conv_ok() == 0): run eg_adam_step_zero + eg_pack_conv");
    // tile width along Cin: 32 (64-byte panel stores, 34 KiB of LDS) or 8 (16-byte stores, 8.5 KiB: fits beside a resident 147-KiB GEMM
    // workgroup, so the update can share CUs with the main chain's convolutions like the LDS-free flat Adam kernel does)
    static const int tc_env = [] { const char* e = getenv("EG_ADAM_PACK_TC"); return e ? atoi(e) : 8; }();
    const int tc = tc_env == 32 ? 32 : 8;
    const dim3 grid(c->Cout / 16, c->Cin / tc);
    const size_t lds = (size_t)16 * tc * (T + 1) * sizeof(float);
    const bool v4 = T == 16 && ((((uintptr_t)w | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
#define EG_AP_TC(TY, Z, TCC) do { if (v4) hipLaunchKernelGGL((adam_pack_conv_tile_kernel<TY, 16, Z, true, TCC>), grid, dim3(256), lds, (hipStream_t)s, p, w, g, m, v, lr, b1, b2, eps, step); \
                               else if (T == 16) hipLaunchKernelGGL((adam_pack_conv_tile_kernel<TY, 16, Z, false, TCC>), grid, dim3(256), lds, (hipStream_t)s, p, w, g, m, v, lr, b1, b2, eps, step); \
                               else hipLaunchKernelGGL((adam_pack_conv_tile_kernel<TY, 0, Z, false, TCC>), grid, dim3(256), lds, (hipStream_t)s, p, w, g, m, v, lr, b1, b2, eps, step); } while (0)
#define EG_AP_TILE(TY, Z) do { if (tc == 32) EG_AP_TC(TY, Z, 32); else EG_AP_TC(TY, Z, 8); } while (0)
#define EG_AP_TYPE(Z) do { if (dtype == EG_F32) EG_AP_TILE(float, Z); else if (dtype == EG_F16) EG_AP_TILE(f16_t, Z); else EG_AP_TILE(bf16_t, Z); } while (0)
    if (zero_grad) EG_AP_TYPE(true);
    else EG_AP_TYPE(false);
#undef EG_AP_TC
#undef EG_AP_TYPE
#undef EG_AP_TILE
    EG_LAUNCH_CHECK();
    return 0;
}

// Adam + re-packing of a weight whose panel is a row-permuted transpose of the master: master w[K][N] (N contiguous; ConvTranspose2d on
// a 1x1 input, celebA/EAD-GAN_celebA.py:76: K = input channels, N = Cout * 16), panel wp[n'][Kpad] with n' = (n % n_mod) * n_mul + n / n_mod
// and k < K (the K padding keeps the zeros of the first eg_pack_strided).  A workgroup takes 32 master rows x 256 columns: coalesced
// reads along N, the update, 16-byte panel stores along K.
template <typename T, bool ZERO>
__global__ __launch_bounds__(256) void adam_pack_rows_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                             T* __restrict__ wp, int K, int N, int Kpad, int n_mod, int n_mul, float lr, float b1,
                                                             float b2, float eps, const int* __restrict__ step) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int TK = 8, TNN = 256;                   // 8 KiB of LDS: fits beside a resident GEMM workgroup
    __shared__ float tile[TK][TNN + 1];
    const int k0 = blockIdx.y * TK, n0 = blockIdx.x * TNN;
    const int tid = threadIdx.x;
    const AdamCoef ac = adam_coef(lr, b1, b2, eps, step);
    const int n = n0 + tid;
#pragma unroll 1
    for (int kk = 0; kk < TK; kk += 4) {                 // four rows' loads in flight per thread
        float pi[4], gi[4], mi[4], vi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + kk + u;
            pi[u] = 0.f; gi[u] = 0.f; mi[u] = 0.f; vi[u] = 0.f;
            if (k < K && n < N) {
                const size_t i = (size_t)k * N + n;
                pi[u] = w[i]; gi[u] = g[i]; mi[u] = m[i]; vi[u] = v[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + kk + u;
            if (k < K && n < N) {
                const size_t i = (size_t)k * N + n;
                adam_elem(pi[u], gi[u], mi[u], vi[u], ac);
                w[i] = pi[u]; m[i] = mi[u]; v[i] = vi[u];
                if (ZERO) g[i] = 0.f;
            }
            tile[kk + u][tid] = pi[u];
        }
    }
    __syncthreads();
    // panel rows n' of this tile's columns, TK / VEC 16-byte vectors each
    for (int it = tid; it < TNN * (TK / VEC); it += 256) {
        const int kg = it % (TK / VEC), nl = it / (TK / VEC);
        const int n = n0 + nl, kb = k0 + kg * VEC;
        if (n >= N || kb >= K) continue;
        const int np = (n % n_mod) * n_mul + n / n_mod;
        T* dst = wp + (size_t)np * Kpad + kb;
        if (kb + VEC <= K) {
            uint4 ov;
            T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
            for (int j = 0; j < VEC; ++j) Elt<T>::st(oe + j, tile[kg * VEC + j][nl]);
            *reinterpret_cast<uint4*>(dst) = ov;
        } else {
            for (int j = 0; kb + j < K; ++j) Elt<T>::st(dst + j, tile[kg * VEC + j][nl]);
        }
    }
}

extern "C" int eg_adam_pack_rows(int dtype, float* w, float* g, float* m, float* v, void* wp, int K, int N, int Kpad, int n_mod, int n_mul,
                                 float lr, float b1, float b2, float eps, const int* step, int zero_grad, eg_stream_t s) {
    EG_REQUIRE(w && g && m && v && wp && step && K > 0 && N > 0 && Kpad >= K && n_mod > 0 && n_mul > 0 && (N % n_mod) == 0, "eg_adam_pack_rows: bad argument");
    EG_REQUIRE(dtype == EG_F32 || dtype == EG_BF16 || dtype == EG_F16, "dtype must be EG_F32, EG_BF16 or EG_F16");
    EG_REQUIRE((Kpad % vec_of(dtype)) == 0, "eg_adam_pack_rows: Kpad must be a multiple of the 16-byte vector width");
    const dim3 grid(cdiv(N, 256), cdiv(K, 8));
#define EG_APR(TY, Z) hipLaunchKernelGGL((adam_pack_rows_kernel<TY, Z>), grid, dim3(256), 0, (hipStream_t)s, w, g, m, v, (TY*)wp, K, N, Kpad, n_mod, n_mul, lr, b1, b2, eps, step)
    if (zero_grad) { if (dtype == EG_F32) EG_APR(float, true); else if (dtype == EG_F16) EG_APR(f16_t, true); else EG_APR(bf16_t, true); }
    else { if (dtype == EG_F32) EG_APR(float, false); else if (dtype == EG_F16) EG_APR(f16_t, false); else EG_APR(bf16_t, false); }
#undef EG_APR
    EG_LAUNCH_CHECK();
    return 0;
}

// generic strided pack: wp[n][k] = w[(n / n_div) * s_hi + (n % n_div) * s_lo + k * s_k]  (k < K, else 0)
template <typename T>
__device__ __forceinline__ void pack_strided_body(const float* __restrict__ w, T* __restrict__ wp, int N, int K, int Kpad, int n_div, long long s_hi,
                                                  long long s_lo, long long s_k, int bx, int gx) {
    const long long total = (long long)N * Kpad;
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < total; i += (long long)gx * blockDim.x) {
        const int n = (int)(i / Kpad), kk = (int)(i % Kpad);
        float v = 0.f;
        if (kk < K) v = w[(n / n_div) * s_hi + (n % n_div) * s_lo + kk * s_k];
        Elt<T>::st(wp + i, v);
    }
}
template <typename T>
__global__ void pack_strided_kernel(const float* __restrict__ w, T* __restrict__ wp, int N, int K, int Kpad, int n_div, long long s_hi,
                                    long long s_lo, long long s_k) {
    pack_strided_body<T>(w, wp, N, K, Kpad, n_div, s_hi, s_lo, s_k, blockIdx.x, gridDim.x);
}

// same with a decomposed column index: + (k / k_div) * s_khi + (k % k_div) * s_klo
template <typename T>
__device__ __forceinline__ void pack_strided2_body(const float* __restrict__ w, T* __restrict__ wp, int N, int K, int Kpad, int n_div, long long s_hi,
                                                   long long s_lo, int k_div, long long s_khi, long long s_klo, int bx, int gx) {
    const long long total = (long long)N * Kpad;
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < total; i += (long long)gx * blockDim.x) {
        const int n = (int)(i / Kpad), kk = (int)(i % Kpad);
        float v = 0.f;
        if (kk < K) v = w[(n / n_div) * s_hi + (n % n_div) * s_lo + (kk / k_div) * s_khi + (kk % k_div) * s_klo];
        Elt<T>::st(wp + i, v);
    }
}
template <typename T>
__global__ void pack_strided2_kernel(const float* __restrict__ w, T* __restrict__ wp, int N, int K, int Kpad, int n_div, long long s_hi,
                                     long long s_lo, int k_div, long long s_khi, long long s_klo) {
    pack_strided2_body<T>(w, wp, N, K, Kpad, n_div, s_hi, s_lo, k_div, s_khi, s_klo, blockIdx.x, gridDim.x);
}
extern "C" int eg_pack_strided2(int dtype, const float* w, void* wp, int N, int K, int Kpad, int n_div, long long s_hi, long long s_lo, int k_div,
                                long long s_khi, long long s_klo, eg_stream_t s) {
    EG_REQUIRE(w && wp && N > 0 && K > 0 && Kpad >= K && n_div > 0 && k_div > 0, "eg_pack_strided2: bad argument");
    const long long total = (long long)N * Kpad;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (pack_record_strided(2, dtype, w, wp, N, K, Kpad, n_div, s_hi, s_lo, 0, k_div, s_khi, s_klo, blocks)) return 0;
    if (dtype == EG_F32) hipLaunchKernelGGL(pack_strided2_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (float*)wp, N, K, Kpad, n_div, s_hi, s_lo, k_div, s_khi, s_klo);
    else if (dtype == EG_F16) hipLaunchKernelGGL(pack_strided2_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (f16_t*)wp, N, K, Kpad, n_div, s_hi, s_lo, k_div, s_khi, s_klo);
    else hipLaunchKernelGGL(pack_strided2_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)wp, N, K, Kpad, n_div, s_hi, s_lo, k_div, s_khi, s_klo);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_pack_strided(int dtype, const float* w, void* wp, int N, int K, int Kpad, int n_div, long long s_hi, long long s_lo,
                               long long s_k, eg_stream_t s) {
    EG_REQUIRE(w && wp && N > 0 && K > 0 && Kpad >= K && n_div > 0, "eg_pack_strided: bad argument");
    const long long total = (long long)N * Kpad;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (pack_record_strided(1, dtype, w, wp, N, K, Kpad, n_div, s_hi, s_lo, s_k, 1, 0, 0, blocks)) return 0;
    if (dtype == EG_F32) hipLaunchKernelGGL(pack_strided_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (float*)wp, N, K, Kpad, n_div, s_hi, s_lo, s_k);
    else if (dtype == EG_F16) hipLaunchKernelGGL(pack_strided_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (f16_t*)wp, N, K, Kpad, n_div, s_hi, s_lo, s_k);
    else hipLaunchKernelGGL(pack_strided_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)wp, N, K, Kpad, n_div, s_hi, s_lo, s_k);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Several packs in ONE launch.  The small networks re-pack 7-12 panels per optimizer update, each a 5 us launch of a few workgroups on a
// chain that is launch-bound end to end (dSprites: 46 of 290 launches per iteration).  eg_pack_record_begin() turns the pack entry points
// above into recorders (nothing is launched), eg_pack_record_end() hands the recorded jobs to the caller, who keeps them in device memory;
// eg_pack_multi() runs them: workgroup -> job by the jobs' first-workgroup table, then the SAME body the stand-alone kernel runs.
// ------------------------------------------------------------------------------------------------
struct PackStridedParams {
    const float* w;
    void* wp;
    int N, K, Kpad, n_div, k_div;
    long long s_hi, s_lo, s_k, s_khi, s_klo;
};
struct PackJob {
    int kind;           // 0 gather (pack_kernel), 1 strided, 2 strided2, 3 tile
    int dtype;
    int gx, gy;         // the stand-alone launch's grid
    int block0;         // first workgroup of this job in the joint launch
    int pad;
    union {
        PackParams g;
        PackStridedParams s;
        PackTileParams t;
    };
};
static thread_local std::vector<PackJob>* g_pack_rec = nullptr;

static bool pack_record_gather(const PackParams& p, int dtype, int gx, int gy) {
    if (!g_pack_rec) return false;
    PackJob j;
    memset(&j, 0, sizeof(j));
    j.kind = 0; j.dtype = dtype; j.gx = gx; j.gy = gy; j.g = p;
    g_pack_rec->push_back(j);
    return true;
}
static bool pack_record_strided(int kind, int dtype, const float* w, void* wp, int N, int K, int Kpad, int n_div, long long s_hi, long long s_lo,
                                long long s_k, int k_div, long long s_khi, long long s_klo, int gx) {
    if (!g_pack_rec) return false;
    PackJob j;
    memset(&j, 0, sizeof(j));
    j.kind = kind; j.dtype = dtype; j.gx = gx; j.gy = 1;
    j.s = PackStridedParams{w, wp, N, K, Kpad, n_div, k_div, s_hi, s_lo, s_k, s_khi, s_klo};
    g_pack_rec->push_back(j);
    return true;
}
static bool pack_record_tile(const PackTileParams& p, int dtype, int gx, int gy) {
    if (!g_pack_rec) return false;
    PackJob j;
    memset(&j, 0, sizeof(j));
    j.kind = 3; j.dtype = dtype; j.gx = gx; j.gy = gy; j.t = p;
    g_pack_rec->push_back(j);
    return true;
}

template <typename T>
__device__ __forceinline__ void pack_job_run(const PackJob& j, float* tile, int bx, int by) {
    switch (j.kind) {
        case 0: pack_gather_body<T>(j.g, bx, by, j.gx); break;
        case 1: pack_strided_body<T>(j.s.w, reinterpret_cast<T*>(j.s.wp), j.s.N, j.s.K, j.s.Kpad, j.s.n_div, j.s.s_hi, j.s.s_lo, j.s.s_k, bx, j.gx); break;
        case 2: pack_strided2_body<T>(j.s.w, reinterpret_cast<T*>(j.s.wp), j.s.N, j.s.K, j.s.Kpad, j.s.n_div, j.s.s_hi, j.s.s_lo, j.s.k_div, j.s.s_khi, j.s.s_klo, bx, j.gx); break;
        default:
            if (j.t.T == 16) pack_conv_tile_body<T, 16, EG_PACKM_TC>(j.t, tile, bx, by);
            else pack_conv_tile_body<T, 0, EG_PACKM_TC>(j.t, tile, bx, by);
    }
}

// (tile jobs 16 x 8 channels x taps: 8.5 KiB of LDS and 33 VGPRs -- the joint launch fits on a CU beside a resident 8-wave GEMM workgroup,
//  like the per-element pack kernels it replaces did; the stand-alone tile kernel's 16 x 32 tile needs 34 KiB)
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJob* __restrict__ jobs, int njobs) {
    __shared__ float tile[16 * EG_PACKM_TC * 17];
    int ji = 0;
    while (ji + 1 < njobs && (int)blockIdx.x >= jobs[ji + 1].block0) ++ji;      // (uniform: scalar loads)
    const PackJob& j = jobs[ji];
    const int local = blockIdx.x - j.block0, bx = local % j.gx, by = local / j.gx;
    if (j.dtype == EG_F32) pack_job_run<float>(j, tile, bx, by);
    else if (j.dtype == EG_F16) pack_job_run<f16_t>(j, tile, bx, by);
    else pack_job_run<bf16_t>(j, tile, bx, by);
}

extern "C" int eg_pack_record_begin(void) {
    EG_REQUIRE(!g_pack_rec, "eg_pack_record_begin: already recording");
    g_pack_rec = new std::vector<PackJob>();
    return 0;
}
extern "C" size_t eg_pack_job_bytes(void) { return sizeof(PackJob); }
/* ends the recording; copies the jobs (eg_pack_job_bytes() each) into host memory `jobs_out` of `cap_bytes` bytes; *njobs / *nblocks: what
 * eg_pack_multi wants back together with a DEVICE copy of jobs_out */
extern "C" int eg_pack_record_end(void* jobs_out, size_t cap_bytes, int* njobs, int* nblocks) {
    EG_REQUIRE(g_pack_rec, "eg_pack_record_end: not recording");
    std::vector<PackJob>* rec = g_pack_rec;
    g_pack_rec = nullptr;
    int nb = 0;
    for (PackJob& j : *rec) { j.block0 = nb; nb += j.gx * j.gy; }
    const size_t need = rec->size() * sizeof(PackJob);
    const bool ok = jobs_out && njobs && nblocks && need <= cap_bytes;
    if (ok) {
        memcpy(jobs_out, rec->data(), need);
        *njobs = (int)rec->size();
        *nblocks = nb;
    }
    delete rec;
    EG_REQUIRE(ok, "eg_pack_record_end: the job table does not fit (or null argument)");
    return 0;
}
extern "C" int eg_pack_multi(const void* jobs_dev, int njobs, int nblocks, eg_stream_t s) {
    EG_REQUIRE(jobs_dev && njobs > 0 && nblocks > 0, "eg_pack_multi: bad argument");
    hipLaunchKernelGGL(pack_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)s, reinterpret_cast<const PackJob*>(jobs_dev), njobs);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// igemm_tn : weight-gradient partial slabs
// ------------------------------------------------------------------------------------------------
struct TnParams {
    const void* P;     // [M][N]
    const void* src;   // [B,H,W,C]
    float* slab;       // [nsplit][N][ntaps][C]
    int B, H, W, C, N;
    int lOH, lOW, M;
    int TW, ntaps;
    int sy, sx, dy0, dx0, up;
    int rows_per_split;
    int ntn, ntc;
};

// swizzle of a [32][BW] element tile in units of 16 elements
__device__ __forceinline__ int tn_fsw(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T, int BNT, int BCT, int KR>   // KR = rows of m per K step (32, or 64: half the barriers per MFMA)
__global__ __launch_bounds__(256) void igemm_tn_kernel(const TnParams p) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int TM = BNT / 32, TN = BCT / 32;            // 2x2 waves
    constexpr int CPR_P = BNT / VEC, CPR_S = BCT / VEC;    // 16-byte chunks per row
    constexpr int LD_P = KR * CPR_P / 256 > 0 ? KR * CPR_P / 256 : 1;
    constexpr int LD_S = KR * CPR_S / 256 > 0 ? KR * CPR_S / 256 : 1;
    constexpr int MASK_P = BNT / 16 - 1, MASK_S = BCT / 16 - 1;
    constexpr int STAGE = KR * (BNT + BCT) * (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tn_i = blockIdx.x / p.ntc, tc_i = blockIdx.x % p.ntc;
    const int n0 = tn_i * BNT, c0 = tc_i * BCT;
    const int t = blockIdx.y, ty = t / p.TW, tx = t % p.TW;
    const int mbeg = blockIdx.z * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int nk = (mend - mbeg + KR - 1) / KR;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const int HU = p.H << p.up, WU = p.W << p.up;
    const T* __restrict__ P = reinterpret_cast<const T*>(p.P);
    const T* __restrict__ src = reinterpret_cast<const T*>(p.src);

    uint4 rp[LD_P], rs[LD_S];
    auto gload = [&](int kt) {
        const int mb = mbeg + kt * KR;
#pragma unroll
        for (int i = 0; i < LD_P; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPR_P, ch = idx % CPR_P;
            const int m = mb + row, n = n0 + ch * VEC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < KR && m < mend && n < p.N) v = *reinterpret_cast<const uint4*>(P + (size_t)m * p.N + n);
            rp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < LD_S; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPR_S, ch = idx % CPR_S;
            const int m = mb + row, c = c0 + ch * VEC;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < KR && m < mend && c < p.C) {
                const int b = m >> (p.lOW + p.lOH);
                const int iy = ((m >> p.lOW) & OHm) * p.sy + p.dy0 + ty;
                const int ix = (m & OWm) * p.sx + p.dx0 + tx;
                if (iy >= 0 && iy < HU && ix >= 0 && ix < WU) {
                    const size_t pix = ((size_t)b * p.H + (iy >> p.up)) * p.W + (ix >> p.up);
                    v = *reinterpret_cast<const uint4*>(src + pix * p.C + c);
                }
            }
            rs[i] = v;
        }
    };
    auto lstore = [&](int stage) {
        T* sp = reinterpret_cast<T*>(smem + stage * STAGE);
        T* ss = sp + KR * BNT;
#pragma unroll
        for (int i = 0; i < LD_P; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPR_P, e = (idx % CPR_P) * VEC;
            if (row < KR) *reinterpret_cast<uint4*>(sp + row * BNT + ((((e >> 4) ^ (tn_fsw(row) & MASK_P)) << 4) | (e & 15))) = rp[i];
        }
#pragma unroll
        for (int i = 0; i < LD_S; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CPR_S, e = (idx % CPR_S) * VEC;
            if (row < KR) *reinterpret_cast<uint4*>(ss + row * BCT + ((((e >> 4) ^ (tn_fsw(row) & MASK_S)) << 4) | (e & 15))) = rs[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    const int g = lane >> 4, li = lane & 15;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const T* sp = reinterpret_cast<const T*>(smem + (kt & 1) * STAGE);
        const T* ss = sp + KR * BNT;
        if constexpr (std::is_same<T, float>::value) {
#pragma unroll
            for (int kg = 0; kg < KR / 4; ++kg) {
                const int row = kg * 4 + g;
                const int f = tn_fsw(row);
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int blk = (wm * TM + i);
                    av[i] = sp[row * BNT + (((blk ^ (f & MASK_P)) << 4) | li)];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int blk = (wn * TN + j);
                    bv[j] = ss[row * BCT + (((blk ^ (f & MASK_S)) << 4) | li)];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // transposed LDS reads: each 16-lane group g fetches k rows 8g..8g+7 (two 4-row blocks) of a 16-column block
            const int q = li >> 2, pc = li & 3;
#pragma unroll
            for (int kb = 0; kb < KR; kb += 32) {
            uint4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int blk = (wm * TM + i);
                s16x4 lo, hi;
                {
                    const int row = kb + 8 * g + q;
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sp + row * BNT + (((blk ^ (tn_fsw(row) & MASK_P)) << 4) | (pc * 4))));
                }
                {
                    const int row = kb + 8 * g + 4 + q;
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sp + row * BNT + (((blk ^ (tn_fsw(row) & MASK_P)) << 4) | (pc * 4))));
                }
                af[i] = make_uint4(((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                                   ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int blk = (wn * TN + j);
                s16x4 lo, hi;
                {
                    const int row = kb + 8 * g + q;
                    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ss + row * BCT + (((blk ^ (tn_fsw(row) & MASK_S)) << 4) | (pc * 4))));
                }
                {
                    const int row = kb + 8 * g + 4 + q;
                    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ss + row * BCT + (((blk ^ (tn_fsw(row) & MASK_S)) << 4) | (pc * 4))));
                }
                bfr[j] = make_uint4(((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                                    ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if constexpr (std::is_same<T, f16_t>::value)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, af[i]), __builtin_bit_cast(f16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) lstore((kt + 1) & 1);
        __syncthreads();
    }

    float* slab = p.slab + (size_t)blockIdx.z * p.N * p.ntaps * p.C;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + (wm * TM + i) * 16 + g * 4 + r;
            if (n >= p.N) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = c0 + (wn * TN + j) * 16 + li;
                if (c < p.C) slab[((size_t)n * p.ntaps + t) * p.C + c] = acc[i][j][r];
            }
        }
}

static void tn_tiles(const eg_conv* c, int* bnt, int* bct) {
    *bnt = c->Cout >= 128 ? 128 : (c->Cout > 32 ? 64 : 32);
    *bct = c->Cin >= 128 ? 128 : (c->Cin > 32 ? 64 : 32);
}

static void tn_plan(const eg_conv* c, int* nsplit, int* rps) {
    const int OH = conv_out_dim(c, c->H), OW = conv_out_dim(c, c->W);
    const int M = c->B * OH * OW;
    int bnt, bct;
    tn_tiles(c, &bnt, &bct);
    const long long base = (long long)cdiv(c->Cout, bnt) * cdiv(c->Cin, bct) * c->k * c->k;
    static const int target = [] { const char* e = getenv("EG_TN_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 768; }();   // workgroups per launch
    long long want = base >= target ? 1 : (target + base - 1) / base;
    long long cap = M / 256 > 0 ? M / 256 : 1;
    if (want > cap) want = cap;
    // the cap binds where one output tile (x a tap or two) is all there is -- the image-side layers as GEMMs over patch rows, and there it IS
    // the grid: 128 workgroups of a 32 x 64 tile keep ~1.5 MB in flight and stream the 100-134 MB of the small networks' image-side layers at
    // ~1 TB/s; 512 of them: colored dSprites 3.08 -> 2.98 ms, dSprites 1.39 -> 1.38, MNIST neutral (profiles/r03_zv_tn_maxsplit_small.txt).
    // Big tiles (CelebA's 128-channel image-side layer, on a lane beside the main chain's GEMMs) keep 128: more took CUs from the main chain
    // (5.50 -> 5.65 ms at 512, profiles/r01_timeline_notes.md item 14).  EG_TN_MAXSPLIT overrides both.
    static const int max_split_env = [] { const char* e = getenv("EG_TN_MAXSPLIT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    const int max_split = max_split_env ? max_split_env : (bnt * bct <= 64 * 64 ? 512 : 128);
    if (want > max_split) want = max_split;
    int r = round_up((int)((M + want - 1) / want), 32);
    *rps = r;
    *nsplit = cdiv(M, r);
}

bool eg_tn8_plan(const eg_conv* c, int dtype, Tn8Params& p, int* nsplit, int wgs_target);
template <typename T> void eg_launch_tn8(const Tn8Params& p, int nsplit, hipStream_t st);

/* 2: the parity-class kernel igemm_tn8 runs this weight gradient, 1: the per-tap kernel igemm_tn (profiling labels and tests) */
extern "C" int eg_conv_wgrad_variant(const eg_conv* c, int dtype) {
    int ns;
    Tn8Params p8;
    if (!c || check_conv(c, dtype, NEED_CIN | NEED_COUT)) return -1;
    return eg_tn8_plan(c, dtype, p8, &ns, 0) ? 2 : 1;
}

extern "C" size_t eg_conv_wgrad_ws_bytes(const eg_conv* c, int dtype) {
    int ns, rps;
    Tn8Params p8;
    if (c && !check_conv(c, dtype, NEED_CIN | NEED_COUT) && eg_tn8_plan(c, dtype, p8, &ns, 0))      // the whole-chip target splits furthest
        return (size_t)ns * c->Cout * 16 * c->Cin * sizeof(float);
    tn_plan(c, &ns, &rps);
    return (size_t)ns * c->Cout * c->k * c->k * c->Cin * sizeof(float);
}

template <typename T, int BNT, int BCT, int KR>
static void launch_tn_kr(TnParams& p, const dim3& grid, hipStream_t st) {
    const size_t lds = 2 * KR * (BNT + BCT) * sizeof(T);
    if (lds > 65536) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_tn_kernel<T, BNT, BCT, KR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((igemm_tn_kernel<T, BNT, BCT, KR>), grid, dim3(256), lds, st, p);
}

template <typename T, int BNT, int BCT>
static void launch_tn_cfg(TnParams& p, int nsplit, hipStream_t st) {
    p.ntn = cdiv(p.N, BNT); p.ntc = cdiv(p.C, BCT);
    dim3 grid(p.ntn * p.ntc, p.ntaps, nsplit);
    // 32 rows per pipeline step; 64 (half the barriers) measured 10-20 % slower on the CelebA layers.  Launches with ONE output tile (the
    // image-side layers: the grid is just the split count, <= 128 workgroups streaming the whole activation) are latency-bound on bytes in
    // flight per workgroup: those get 64 rows per step (colored dSprites B=512: 5.22 -> 5.00 ms; CelebA, dSprites, MNIST neutral to +1 %;
    // EG_TN_KR64=0 restores 32).
    static const bool kr64 = [] { const char* e = getenv("EG_TN_KR64"); return !(e && atoi(e) == 0); }();
    if (kr64 && p.ntn * p.ntc * p.ntaps == 1 && sizeof(T) == 2 && BNT + BCT <= 192)
        launch_tn_kr<T, BNT, BCT, 64>(p, grid, st);
    else
        launch_tn_kr<T, BNT, BCT, 32>(p, grid, st);
}

template <typename T>
static void launch_tn(TnParams& p, int bnt, int bct, int nsplit, hipStream_t st) {
#define EG_TN_CASE(a, b) if (bnt == a && bct == b) { launch_tn_cfg<T, a, b>(p, nsplit, st); return; }
    EG_TN_CASE(128, 128) EG_TN_CASE(128, 64) EG_TN_CASE(128, 32)
    EG_TN_CASE(64, 128) EG_TN_CASE(64, 64) EG_TN_CASE(64, 32)
    EG_TN_CASE(32, 128) EG_TN_CASE(32, 64) EG_TN_CASE(32, 32)
#undef EG_TN_CASE
}

extern "C" int eg_conv_wgrad(const eg_conv* c, int dtype, const void* X, const void* dY, float* slab, int* nsplit_out,
                             eg_stream_t s) {
    return eg_conv_wgrad_target(c, dtype, X, dY, slab, nsplit_out, 0, s);
}

extern "C" int eg_conv_wgrad_target(const eg_conv* c, int dtype, const void* X, const void* dY, float* slab, int* nsplit_out,
                                    int wgs_target, eg_stream_t s) {
    if (int e = check_conv(c, dtype, NEED_CIN | NEED_COUT)) return e;
    EG_REQUIRE(X && dY && slab && nsplit_out && wgs_target >= 0, "eg_conv_wgrad: bad argument");
    const int OH = conv_out_dim(c, c->H), OW = conv_out_dim(c, c->W);
    {
        // 16-bit 4x4 / stride-2 layers with channel counts in multiples of 128: the parity-class kernel (igemm_tn8.hip), same slab layout
        Tn8Params p8;
        int ns8 = 0;
        if (eg_tn8_plan(c, dtype, p8, &ns8, wgs_target)) {
            p8.P = dY; p8.src = X; p8.slab = slab;
            if (dtype == EG_F16) eg_launch_tn8<f16_t>(p8, ns8, (hipStream_t)s);
            else eg_launch_tn8<bf16_t>(p8, ns8, (hipStream_t)s);
            *nsplit_out = ns8;
            EG_LAUNCH_CHECK();
            return 0;
        }
    }
    TnParams p;
    memset(&p, 0, sizeof(p));
    p.P = dY; p.src = X; p.slab = slab;
    p.B = c->B; p.H = c->H; p.W = c->W; p.C = c->Cin; p.N = c->Cout;
    p.lOH = ilog2_exact(OH); p.lOW = ilog2_exact(OW); p.M = c->B * OH * OW;
    p.TW = c->k; p.ntaps = c->k * c->k;
    p.sy = p.sx = c->stride; p.dy0 = p.dx0 = -c->pad; p.up = c->up;
    int ns, rps, bnt, bct;
    tn_plan(c, &ns, &rps);
    tn_tiles(c, &bnt, &bct);
    p.rows_per_split = rps;
    if (dtype == EG_F32) launch_tn<float>(p, bnt, bct, ns, (hipStream_t)s);
    else if (dtype == EG_F16) launch_tn<f16_t>(p, bnt, bct, ns, (hipStream_t)s);
    else launch_tn<bf16_t>(p, bnt, bct, ns, (hipStream_t)s);
    *nsplit_out = ns;
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// slab reduction into the master-layout gradient (+ spectral-norm correction)
// ------------------------------------------------------------------------------------------------
#define EG_SN_NPART 1024

// One block per (n, chunk of 64 gathered channels): reads the [t][c] slab rows coalesced along c, sums the splits, subtracts
// the rank-1 spectral-norm terms, transposes through LDS and read-modify-writes the master-layout gradient [n][c][t] as
// one contiguous run of 64*T floats.  MODE 0: out += a; MODE 1: out = a (gtmp) + <a,W> partials; MODE 2: out += a - rank1.
#define EG_RC 64
__device__ int eg_reduce_chain_flag;                    // EG_REDUCE_RAGGED=0 (A/B runs): the one-chain loop for ragged channel counts
// LEAN: the instantiation for launches with < 16 splits (every big layer: 2-8 slabs): no split-group scratch (4.4 KiB of LDS instead of
// 24) and four loads in flight instead of eight (<= 48 VGPRs) -- its workgroups then fit on a CU beside a resident 8-wave GEMM workgroup
// (147 KiB of LDS, 2 x 232 VGPRs per SIMD lane), so a reduction forked beside the main chain's GEMMs no longer waits for their tiles to
// retire.  Same summation order as the eight-deep loop (one add per slab, in slab order): same bits.
template <int MODE, bool LEAN = false>
__global__ __launch_bounds__(LEAN ? 256 : 1024) void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int NS, int N, int C, int T,
                                                           float* __restrict__ out, int accumulate, const float* __restrict__ w_orig,
                                                           float* __restrict__ partials, int ntapes, const float* __restrict__ coef,
                                                           const float* __restrict__ u, const float* __restrict__ v, int row_div, int row_mul, int c_row) {
    extern __shared__ float tile[];      // [EG_RC][T+1]
    __shared__ float sm[16];
    const int cchunks = (C + EG_RC - 1) / EG_RC;
    const int n = blockIdx.x / cchunks, c0 = (blockIdx.x % cchunks) * EG_RC;
    const int cw = min(EG_RC, C - c0);
    const size_t split_stride = (size_t)NS * T * C;
    float un[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 2)
        for (int q = 0; q < ntapes; ++q) un[q] = coef[q] * u[(size_t)q * N + n];
    const int crow = c_row ? c_row : C;                 // row length of the destination (<= C when the slab is column padded)
    const int nv = T * (EG_RC / 4);                     // float4 elements of a full tile
    const int NTH = blockDim.x;                          // 256, or 1024 for the many-split form (the host picks)
    if (!LEAN && cw == EG_RC && c0 + EG_RC <= crow && (C & 3) == 0 && nv * 2 <= NTH && nsplit >= 16) {
        // many splits of a small tile (the image-side layers as 1x1 convolutions: T = 1, 128 splits; the 32- and 64-channel layers of the
        // small networks: one 64 x 16 tile per output channel, 128 splits): the loop below is a chain of nsplit loads per thread with most
        // of the chip idle.  Here blockDim / nv thread groups each take every (blockDim / nv)-th split, eight loads in flight, and the
        // groups are added in group order through LDS (deterministic).
        __shared__ float4 part[1024];
        const int G = NTH / nv, e4 = threadIdx.x % nv, sg = threadIdx.x / nv;
        const int t = e4 / (EG_RC / 4), c = (e4 % (EG_RC / 4)) * 4;
        const size_t si = ((size_t)n * T + t) * C + c0 + c;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sg < G) {
            int z = sg;
            for (; z + 7 * G < nsplit; z += 8 * G) {
                float4 x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const float4*>(slab + (size_t)(z + q * G) * split_stride + si);
#pragma unroll
                for (int q = 0; q < 8; ++q) { a.x += x[q].x; a.y += x[q].y; a.z += x[q].z; a.w += x[q].w; }
            }
            for (; z < nsplit; z += G) {
                const float4 x = *reinterpret_cast<const float4*>(slab + (size_t)z * split_stride + si);
                a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            }
        }
        part[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < nv) {
            a = part[threadIdx.x];
            for (int gq = 1; gq < G; ++gq) { const float4 x = part[gq * nv + threadIdx.x]; a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; }
            if (MODE == 2)
                for (int q = 0; q < ntapes; ++q) {
                    const float* vq = v + (size_t)q * crow * T + (size_t)(c0 + c) * T + t;
                    a.x -= un[q] * vq[0]; a.y -= un[q] * vq[T]; a.z -= un[q] * vq[2 * T]; a.w -= un[q] * vq[3 * T];
                }
            tile[c * (T + 1) + t] = a.x; tile[(c + 1) * (T + 1) + t] = a.y; tile[(c + 2) * (T + 1) + t] = a.z; tile[(c + 3) * (T + 1) + t] = a.w;
        }
    } else if (cw == EG_RC && c0 + EG_RC <= crow && (C & 3) == 0) {
        // full 64-channel block: 16-byte loads along the channels (one float4 per thread and slab for k = 4), same summation order
        for (int e4 = threadIdx.x; e4 < T * (EG_RC / 4); e4 += NTH) {
            const int t = e4 / (EG_RC / 4), c = (e4 % (EG_RC / 4)) * 4;
            const size_t si = ((size_t)n * T + t) * C + c0 + c;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            // eight (LEAN: four) slabs' loads in flight, added in slab order (the sum's order is unchanged: bit-identical to the one-by-one
            // loop; a thread owns ONE float4 of the tile, so without this the loop is a chain of nsplit exposed HBM latencies)
            constexpr int DEPTH = LEAN ? 4 : 8;
            int z = 0;
            for (; z + DEPTH <= nsplit; z += DEPTH) {
                float4 x[DEPTH];
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) x[q] = *reinterpret_cast<const float4*>(slab + (size_t)(z + q) * split_stride + si);
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) { a.x += x[q].x; a.y += x[q].y; a.z += x[q].z; a.w += x[q].w; }
            }
            for (; z < nsplit; ++z) {
                const float4 x = *reinterpret_cast<const float4*>(slab + z * split_stride + si);
                a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            }
            if (MODE == 2 && !LEAN) {
#pragma unroll 1
                for (int q = 0; q < ntapes; ++q) {
                    const float* vq = v + (size_t)q * crow * T + (size_t)(c0 + c) * T + t;
                    a.x -= un[q] * vq[0]; a.y -= un[q] * vq[T]; a.z -= un[q] * vq[2 * T]; a.w -= un[q] * vq[3 * T];
                }
            }
            tile[c * (T + 1) + t] = a.x; tile[(c + 1) * (T + 1) + t] = a.y; tile[(c + 2) * (T + 1) + t] = a.z; tile[(c + 3) * (T + 1) + t] = a.w;
        }
    } else if (!LEAN && nsplit >= 16 && T * EG_RC * 2 <= NTH && !eg_reduce_chain_flag) {
        // many splits of a small tile whose channel count is not a multiple of 64 (the image-side layers: 48 = 3 x 16 gathered channels,
        // 128 splits): the loop below is ONE chain of nsplit dependent loads per thread (39 us for a 6 K-parameter gradient, at the end
        // of every sub-step's backward pass).  As above: thread groups take every G-th split, eight loads in flight, added in group order.
        __shared__ float parts[1024];
        const int ne = T * EG_RC, G = min(NTH / ne, 16), e = threadIdx.x % ne, sg = threadIdx.x / ne;
        const int t = e / EG_RC, cc = e % EG_RC;
        const bool valid = cc < cw && c0 + cc < crow;
        const size_t si = ((size_t)n * T + t) * C + c0 + cc;
        float a = 0.f;
        if (sg < G && valid) {
            int z = sg;
            for (; z + 7 * G < nsplit; z += 8 * G) {
                float x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = slab[(size_t)(z + q * G) * split_stride + si];
#pragma unroll
                for (int q = 0; q < 8; ++q) a += x[q];
            }
            for (; z < nsplit; z += G) a += slab[(size_t)z * split_stride + si];
        }
        parts[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < ne && valid) {
            a = parts[threadIdx.x];
            for (int gq = 1; gq < G; ++gq) a += parts[gq * ne + threadIdx.x];
            if (MODE == 2)
                for (int q = 0; q < ntapes; ++q) a -= un[q] * v[(size_t)q * crow * T + (size_t)(c0 + cc) * T + t];
            tile[cc * (T + 1) + t] = a;
        }
    } else
    for (int e = threadIdx.x; e < T * EG_RC; e += NTH) {
        const int t = e / EG_RC, c = e % EG_RC;
        if (c < cw && c0 + c < crow) {
            const size_t si = ((size_t)n * T + t) * C + c0 + c;
            float a = 0.f;
            for (int z = 0; z < nsplit; ++z) a += slab[z * split_stride + si];
            if (MODE == 2 && !LEAN)
                for (int q = 0; q < ntapes; ++q) a -= un[q] * v[(size_t)q * crow * T + (size_t)(c0 + c) * T + t];
            tile[c * (T + 1) + t] = a;
        }
    }
    __syncthreads();
    float dot = 0.f;
    const int nout = row_div ? (n % row_div) * row_mul + n / row_div : n;
    const size_t obase = ((size_t)nout * crow + c0) * T;
    const int cwo = min(cw, crow - c0);
    for (int e = threadIdx.x; e < cwo * T; e += NTH) {
        const int c = e / T, t = e % T;
        float a = tile[c * (T + 1) + t];
        if (MODE == 2 && LEAN) {
            // the rank-1 spectral-norm terms here, on the way out: v[q][(c0 + c) * T + t] is v_q[c0 * T + e] -- contiguous along e (the
            // load phase gathers it with stride T); the same subtractions in the same order on the same values -> the same bits
#pragma unroll 1
            for (int q = 0; q < ntapes; ++q) a -= un[q] * v[(size_t)q * crow * T + (size_t)c0 * T + e];
        }
        if (MODE == 1) {
            out[obase + e] = a;
            dot += a * w_orig[obase + e];
        } else {
            out[obase + e] = (MODE == 2 || accumulate) ? out[obase + e] + a : a;
        }
    }
    if (MODE == 1) {
        const float tot = block_sum(dot, sm);
        if (threadIdx.x == 0) partials[blockIdx.x] = tot;
    }
}

__global__ void sn_grad_apply_kernel(const float* __restrict__ gtmp, const float* __restrict__ partials, int npart,
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

static inline int reduce_blocks(int n_rows, int C) {
    static const bool once = [] {
        const char* e = getenv("EG_REDUCE_RAGGED");
        const int v = (e && atoi(e) == 0) ? 1 : 0;
        if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(eg_reduce_chain_flag), &v, sizeof(v));
        return true;
    }();
    (void)once;
    return n_rows * ((C + EG_RC - 1) / EG_RC);
}
// threads per workgroup: 1024 (several split groups per tile vector) where a launch has many splits and few, small tiles
static inline int reduce_threads(int nsplit, int n_rows, int C, int T) {
    const int nv = T * (EG_RC / 4);
    return (nsplit >= 16 && nv * 2 <= 1024 && nv * 2 > 256 && reduce_blocks(n_rows, C) <= 2048) ? 1024 : 256;
}
static inline size_t reduce_lds(int T) { return (size_t)EG_RC * (T + 1) * sizeof(float); }
static inline bool reduce_lean(int nsplit) {
    static const bool on = [] { const char* e = getenv("EG_REDUCE_LEAN"); return !(e && atoi(e) == 0); }();
    return on && nsplit < 16;
}

extern "C" int eg_wgrad_reduce(const float* slab, int nsplit, int n_slab, int n_rows, int C, int T, float* grad, int accumulate, eg_stream_t s) {
    EG_REQUIRE(slab && grad && nsplit > 0 && n_rows <= n_slab && T > 0 && T <= 64, "eg_wgrad_reduce: bad argument");
    if (reduce_lean(nsplit))
        hipLaunchKernelGGL((wgrad_reduce_kernel<0, true>), dim3(reduce_blocks(n_rows, C)), dim3(256), reduce_lds(T), (hipStream_t)s, slab, nsplit, n_slab, n_rows, C, T,
                           grad, accumulate, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0, 0);
    else
    hipLaunchKernelGGL(wgrad_reduce_kernel<0>, dim3(reduce_blocks(n_rows, C)), dim3(reduce_threads(nsplit, n_rows, C, T)), reduce_lds(T), (hipStream_t)s, slab, nsplit, n_slab, n_rows, C, T,
                       grad, accumulate, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0, 0);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_wgrad_reduce_perm(const float* slab, int nsplit, int n_slab, int n_rows, int C, int T, float* grad, int row_div, int row_mul, int c_row, eg_stream_t s) {
    EG_REQUIRE(slab && grad && nsplit > 0 && n_rows <= n_slab && T > 0 && T <= 64 && row_div >= 0 && c_row >= 0 && c_row <= C, "eg_wgrad_reduce_perm: bad argument");
    hipLaunchKernelGGL(wgrad_reduce_kernel<0>, dim3(reduce_blocks(n_rows, C)), dim3(reduce_threads(nsplit, n_rows, C, T)), reduce_lds(T), (hipStream_t)s, slab, nsplit, n_slab, n_rows, C, T,
                       grad, 1, (const float*)nullptr, (float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, row_div, row_mul, c_row);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_wgrad_reduce_rank1(const float* slab, int nsplit, int n_slab, int n_rows, int C, int T, float* grad, int ntapes,
                                     const float* coef, const float* u, const float* v, int c_row, eg_stream_t s) {
    EG_REQUIRE(slab && grad && nsplit > 0 && n_rows <= n_slab && ntapes >= 0 && ntapes <= 4 && (ntapes == 0 || (coef && u && v)) && T > 0 && T <= 64,
               "eg_wgrad_reduce_rank1: bad argument");
    if (reduce_lean(nsplit))
        hipLaunchKernelGGL((wgrad_reduce_kernel<2, true>), dim3(reduce_blocks(n_rows, C)), dim3(256), reduce_lds(T), (hipStream_t)s, slab, nsplit, n_slab, n_rows, C, T,
                           grad, 1, (const float*)nullptr, (float*)nullptr, ntapes, coef, u, v, 0, 0, c_row);
    else
    hipLaunchKernelGGL(wgrad_reduce_kernel<2>, dim3(reduce_blocks(n_rows, C)), dim3(reduce_threads(nsplit, n_rows, C, T)), reduce_lds(T), (hipStream_t)s, slab, nsplit, n_slab, n_rows, C, T,
                       grad, 1, (const float*)nullptr, (float*)nullptr, ntapes, coef, u, v, 0, 0, c_row);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_sn_partials(void) { return 1 << 17; }

extern "C" int eg_wgrad_reduce_sn(const eg_conv* c, const float* slab, int nsplit, const float* w_orig, const float* sigma,
                                  const float* u, const float* v, float* gtmp, float* partials, float* grad, eg_stream_t s) {
    EG_REQUIRE(c && slab && w_orig && sigma && u && v && gtmp && partials && grad && nsplit > 0, "eg_wgrad_reduce_sn: bad argument");
    const int T = c->k * c->k;
    const int blocks = reduce_blocks(c->Cout, c->Cin);
    EG_REQUIRE(blocks <= (1 << 17), "eg_wgrad_reduce_sn: layer too large for the partials buffer");
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(blocks), dim3(256), reduce_lds(T), (hipStream_t)s, slab, nsplit, c->Cout, c->Cout, c->Cin, T, gtmp, 0,
                       w_orig, partials, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0, 0);
    const long long total = (long long)c->Cout * c->Cin * T;
    const int blocks2 = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(sn_grad_apply_kernel, dim3(blocks2), dim3(256), 0, (hipStream_t)s, gtmp, partials, blocks, sigma, u, v, total,
                       c->Cin * T, grad);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bias gradient: column sums of a [rows][N] tensor (two deterministic stages)
// ------------------------------------------------------------------------------------------------
// rows per workgroup: 256, halved (down to 16) until the launch has >= 512 workgroups -- the head layers have few rows (B*16) of
// 1024 columns, where 256-row blocks left 8-24 workgroups walking a dependent-load chain
static int bg_rpb(int rows_per_group, int ngroups, int gx) {
    int rpb = 256;
    while (rpb > 16 && (long long)gx * ngroups * cdiv(rows_per_group, rpb) < 512) rpb >>= 1;
    return rpb;
}
static int bg_gx(int N, int dtype) {
    const int cpr = N / (dtype == EG_F32 ? 4 : 8);
    return cdiv(cpr, cpr < 256 ? cpr : 256);
}

// block = 256 threads = (N/VEC chunk columns) x (row lanes); 16-byte loads; LDS combine of the row lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int rows, int N, int rpb, float* __restrict__ partials) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ float sm[256 * VEC];
    const int cpr = N / VEC;                       // chunks per row (host guarantees cpr <= 256 and 256 % cpr == 0 via tiling in x)
    const int ccols = cpr < 256 ? cpr : 256;
    const int lanes = 256 / ccols;
    const int cj = threadIdx.x % ccols, rl = threadIdx.x / ccols;
    const int chunk = blockIdx.x * ccols + cj;
    const int r0 = blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
    float a[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) a[j] = 0.f;
    if (chunk < cpr)
        for (int r = r0 + rl; r < r1; r += lanes) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)r * N + (size_t)chunk * VEC);
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[j] += Elt<T>::ld(e + j);
        }
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[threadIdx.x * VEC + j] = a[j];
    __syncthreads();
    if (rl == 0 && chunk < cpr) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += sm[(l * ccols + cj) * VEC + j];
            partials[(size_t)blockIdx.y * N + (size_t)chunk * VEC + j] = t;
        }
    }
}

// gb[j] += scale * sum_{n == j mod nb} sum_r partials[r][n]; one wave per output j
__global__ void colsum_final_kernel(const float* __restrict__ partials, int nrb, int N, int nb, const float* __restrict__ scale, float* __restrict__ gb) {
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= nb) return;
    float a = 0.f;
    for (int n = j; n < N; n += nb)
        for (int r = lane; r < nrb; r += 64) a += partials[(size_t)r * N + n];
    a = wave_sum(a);
    if (lane == 0) gb[j] += a * (scale ? scale[0] : 1.f);
}

extern "C" size_t eg_bias_grad_ws_floats(int rows, int N) {
    // bound for every launch with <= rows rows and <= N columns, either dtype: bg_rpb stops halving once there are 512 workgroups, so
    // a launch has at most max(rows / 256, 1024) row blocks
    return (size_t)std::max(cdiv(rows, 256), 1024) * N;
}

extern "C" int eg_bias_grad(int dtype, const void* dY, int rows, int N, int bias_mod, float* partials, float* gb, eg_stream_t s) {
    EG_REQUIRE(dY && partials && gb && rows > 0 && N > 0, "eg_bias_grad: bad argument");
    const int vecw = dtype == EG_F32 ? 4 : 8;
    EG_REQUIRE(N % vecw == 0, "eg_bias_grad: N must be a multiple of the 16-byte vector width");
    const int cpr = N / vecw;
    const int ccols = cpr < 256 ? cpr : 256;
    EG_REQUIRE(256 % ccols == 0, "eg_bias_grad: N/vec must divide 256 or be a multiple of it");
    const int rpb = bg_rpb(rows, 1, cdiv(cpr, ccols));
    const int nrb = cdiv(rows, rpb);
    dim3 grid(cdiv(cpr, ccols), nrb);
    if (dtype == EG_F32) hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)dY, rows, N, rpb, partials);
    else if (dtype == EG_F16) hipLaunchKernelGGL(colsum_partial_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)s, (const f16_t*)dY, rows, N, rpb, partials);
    else hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)dY, rows, N, rpb, partials);
    const int nb = bias_mod > 0 ? bias_mod : N;
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(nb, 4)), dim3(256), 0, (hipStream_t)s, partials, nrb, N, nb, (const float*)nullptr, gb);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- tape-segmented bias gradient + <G,W>/sigma^2 coefficient from activations (spectrally normalised layers) --------
template <typename T>
__global__ __launch_bounds__(256) void colsum_sn_partial_kernel(const T* __restrict__ x, const T* __restrict__ act, const float* __restrict__ bias,
                                                                int N, int rows_per_tape, int blocks_per_tape, int rpb, float inv_slope,
                                                                float* __restrict__ partials, float* __restrict__ dots) {
    constexpr int VEC = Elt<T>::VEC;
    __shared__ float sm[256 * VEC];
    __shared__ float smd[16];
    const int cpr = N / VEC;
    const int ccols = cpr < 256 ? cpr : 256;
    const int lanes = 256 / ccols;
    const int cj = threadIdx.x % ccols, rl = threadIdx.x / ccols;
    const int chunk = blockIdx.x * ccols + cj;
    const int tape = blockIdx.y / blocks_per_tape, blk = blockIdx.y % blocks_per_tape;
    const int r0 = tape * rows_per_tape + blk * rpb;
    const int r1 = min(tape * rows_per_tape + rows_per_tape, r0 + rpb);
    float a[VEC], bv[VEC];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { a[j] = 0.f; bv[j] = (chunk < cpr) ? bias[chunk * VEC + j] : 0.f; }
    if (chunk < cpr)
        for (int r = r0 + rl; r < r1; r += lanes) {
            const size_t o = (size_t)r * N + (size_t)chunk * VEC;
            const uint4 v = *reinterpret_cast<const uint4*>(x + o);
            const uint4 w = *reinterpret_cast<const uint4*>(act + o);
            const T* e = reinterpret_cast<const T*>(&v);
            const T* ae = reinterpret_cast<const T*>(&w);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float g = Elt<T>::ld(e + j);
                float z = Elt<T>::ld(ae + j);
                z = z > 0.f ? z : z * inv_slope;
                a[j] += g;
                dot += g * (z - bv[j]);
            }
        }
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[threadIdx.x * VEC + j] = a[j];
    const float dtot = block_sum(dot, smd);      // contains the __syncthreads that also publishes sm
    if (rl == 0 && chunk < cpr) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += sm[(l * ccols + cj) * VEC + j];
            partials[(size_t)blockIdx.y * N + (size_t)chunk * VEC + j] = t;
        }
    }
    if (threadIdx.x == 0) dots[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = dtot;
}

// gb[n] += sum_rb sigma[tape(rb)] * partials[rb][n]   (one wave per n);  coef[t] = sum of the tape's dot partials (block z)
__global__ void colsum_sn_final_kernel(const float* __restrict__ partials, const float* __restrict__ dots, int nrb, int N, int blocks_per_tape,
                                       int ndot_per_rb, int ntapes, const float* __restrict__ sigma, float* __restrict__ gb, float* __restrict__ coef) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j < N) {
        float a = 0.f;
        for (int r = lane; r < nrb; r += 64) a += partials[(size_t)r * N + j] * sigma[r / blocks_per_tape];
        a = wave_sum(a);
        if (lane == 0) gb[j] += a;
    } else if (j < N + ntapes) {
        const int t = j - N;
        float a = 0.f;
        const int n = blocks_per_tape * ndot_per_rb;
        for (int r = lane; r < n; r += 64) a += dots[(size_t)t * n + r];
        a = wave_sum(a);
        if (lane == 0) coef[t] = a;
    }
}

extern "C" size_t eg_bias_grad_sn_ws_floats(int rows, int N, int rows_per_tape) {
    // same bound as eg_bias_grad_ws_floats (row blocks over all tapes), plus the per-block dot partials
    const int ntapes = rows / rows_per_tape;
    return (size_t)std::max(ntapes * cdiv(rows_per_tape, 256), 1024 + ntapes) * (N + 64);
}

extern "C" int eg_bias_grad_sn(int dtype, const void* dzs, const void* a, const float* bias, int rows, int N, int rows_per_tape,
                               const float* sigma, float slope, float* ws, float* gb, float* coef, eg_stream_t s) {
    EG_REQUIRE(dzs && a && bias && sigma && ws && gb && coef && rows_per_tape > 0 && rows % rows_per_tape == 0 && slope > 0.f, "eg_bias_grad_sn: bad argument");
    const int vecw = dtype == EG_F32 ? 4 : 8;
    EG_REQUIRE(N % vecw == 0, "eg_bias_grad_sn: N must be a multiple of the 16-byte vector width");
    const int cpr = N / vecw;
    const int ccols = cpr < 256 ? cpr : 256;
    EG_REQUIRE(256 % ccols == 0, "eg_bias_grad_sn: N/vec must divide 256 or be a multiple of it");
    const int ntapes = rows / rows_per_tape;
    EG_REQUIRE(ntapes <= 4, "eg_bias_grad_sn: at most 4 tapes");
    const int gx = cdiv(cpr, ccols);
    const int rpb = bg_rpb(rows_per_tape, ntapes, gx);
    const int bpt = cdiv(rows_per_tape, rpb);
    const int nrb = ntapes * bpt;
    float* partials = ws;
    float* dots = ws + (size_t)nrb * N;
    dim3 grid(gx, nrb);
    if (dtype == EG_F32) hipLaunchKernelGGL(colsum_sn_partial_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)dzs, (const float*)a, bias, N, rows_per_tape, bpt, rpb, 1.f / slope, partials, dots);
    else if (dtype == EG_F16) hipLaunchKernelGGL(colsum_sn_partial_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)s, (const f16_t*)dzs, (const f16_t*)a, bias, N, rows_per_tape, bpt, rpb, 1.f / slope, partials, dots);
    else hipLaunchKernelGGL(colsum_sn_partial_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)dzs, (const bf16_t*)a, bias, N, rows_per_tape, bpt, rpb, 1.f / slope, partials, dots);
    hipLaunchKernelGGL(colsum_sn_final_kernel, dim3(cdiv(N + ntapes, 4)), dim3(256), 0, (hipStream_t)s, partials, dots, nrb, N, bpt, gx, ntapes, sigma, gb, coef);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- the same sums from the epilogue of the convolution that produced dzs (eg_epilogue.stat_mode = EG_STAT_SN_BIAS) -----------------
// stat: [N][nrb] column sums, then [nrb][tiles_n] tile dots.  gb[n] += sum_rb sigma[tape(rb)] * stat[n][rb] (one wave per n, contiguous
// reads); coef[t] = sum of the dots of tape t's row blocks (waves N .. N + ntapes - 1).  tape(rb) = (rb % tiles_m) / tiles_per_tape.
// WIDE: one 256-thread workgroup per sum instead of one wave (>= 1024 row blocks: the small networks' 32 / 64-channel layers at B = 128 .. 512
// have up to 12288 of them; one wave walked 192 partials per lane with an integer division each: 25 us per launch in the colored dSprites step);
// fewer row blocks keep the wave form and its bits
template <bool WIDE>
__global__ void colsum_sn_final_t_kernel(const float* __restrict__ stat, int nrb, int N, int tiles_m, int tiles_per_tape, int tiles_n, int ntapes,
                                         const float* __restrict__ sigma, float* __restrict__ gb, float* __restrict__ coef) {
    __shared__ float sh[4];
    const int lane = WIDE ? threadIdx.x : threadIdx.x & 63, step = WIDE ? 256 : 64;
    const int j = WIDE ? blockIdx.x : blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    float acc = 0.f;
    if (j < N) {
        const float* __restrict__ a = stat + (size_t)j * nrb;
        for (int r = lane; r < nrb; r += step) acc += a[r] * sigma[(r % tiles_m) / tiles_per_tape];
    } else if (j < N + ntapes) {
        const int t = j - N;
        const float* __restrict__ dots = stat + (size_t)N * nrb;
        for (int r = lane; r < nrb; r += step)
            if ((r % tiles_m) / tiles_per_tape == t)
                for (int q = 0; q < tiles_n; ++q) acc += dots[(size_t)r * tiles_n + q];
    } else if (!WIDE) {
        return;
    }
    acc = wave_sum(acc);
    if (WIDE) {
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
        __syncthreads();
        acc = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    }
    if (lane != 0) return;
    if (j < N) gb[j] += acc;
    else if (j < N + ntapes) coef[j - N] = acc;
}

extern "C" int eg_bias_grad_sn_fused(const float* stat, int nrb, int N, int tiles_m, int tiles_per_tape, int ntapes, const float* sigma, float* gb,
                                     float* coef, eg_stream_t s) {
    EG_REQUIRE(stat && sigma && gb && coef && nrb > 0 && N > 0 && ((N % 128) == 0 || N == 64 || N == 32) && tiles_m > 0 && (nrb % tiles_m) == 0 &&
               tiles_per_tape > 0 && ntapes > 0 && ntapes <= 4 && tiles_per_tape * ntapes == tiles_m, "eg_bias_grad_sn_fused: bad argument");
    // (column tiles of the producing launch: 128 wide for the 8-wave kernels, the whole row for the register-staged kernel's N = 32 / 64)
    if (nrb >= 1024)
        hipLaunchKernelGGL(colsum_sn_final_t_kernel<true>, dim3(N + ntapes), dim3(256), 0, (hipStream_t)s, stat, nrb, N, tiles_m, tiles_per_tape,
                           N >= 128 ? N / 128 : 1, ntapes, sigma, gb, coef);
    else
        hipLaunchKernelGGL(colsum_sn_final_t_kernel<false>, dim3(cdiv(N + ntapes, 4)), dim3(256), 0, (hipStream_t)s, stat, nrb, N, tiles_m, tiles_per_tape,
                           N >= 128 ? N / 128 : 1, ntapes, sigma, gb, coef);
    EG_LAUNCH_CHECK();
    return 0;
}
