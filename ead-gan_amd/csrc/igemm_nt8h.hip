// igemm_nt8h: the 8-wave NT implicit GEMM of igemm_nt8s.hip on 128 x 128 tiles (waves 4 x 2, wave tile 32 x 64) with a ring of FOUR
// 32-KiB K tiles, for launches that have too few 256-row tiles to fill the chip: every single-tape layer of the CelebA step is 64 or 128
// tiles of 256 x 128 on 256 CUs.  igemm_nt8s ran those as two or four K splits per tile -- half-length K loops that still pay a whole
// prologue each, 128 KiB of fp32 partials per workgroup written and read back, an arrival-counter hand-off and the epilogue of a 256-row
// tile on the last arriver -- at 580-740 TFLOP/s.  Here the same launches are 128 or 256 tiles with whole K loops (one split fewer or none).
//
// Same K order and the same MFMA operand placement as igemm_nt8s / the register-staged kernel: bit-identical without a K split.
// Per K tile and wave: 16 MFMAs (4 groups of 4: k half g >> 1, row tile g & 1, four column tiles), 12 ds_read_b128 of K tile t + 1 into
// the other fragment set, 4 LDS-DMA pieces of K tile t + 3 (A rows 8w + 64j, B rows 8w + 64j; j = 0, 1), one counted wait, ONE barrier.
// Hazards as in igemm_nt8s.hip with one more stage of slack: K tile t + 3 goes into stage (t + 3) % 4, last read in iteration t - 2.
#include <stdlib.h>
#include <string.h>

#include "eg_common.h"
#include "igemm_nt.h"

template <int VM>
__device__ __forceinline__ void nt8h_wait() {           // s_waitcnt vmcnt(VM) lgkmcnt(0) through the builtin (see igemm_nt8s.hip)
    __builtin_amdgcn_s_waitcnt((VM & 15) | ((VM >> 4) << 14) | (7 << 4) | (0 << 8));
}
__device__ __forceinline__ void nt8h_wait_dyn(int n) {  // wave-uniform n in {0, 4, 8}
    if (n >= 8) nt8h_wait<8>();
    else if (n >= 4) nt8h_wait<4>();
    else nt8h_wait<0>();
}

template <typename T, bool SPLITK, bool STAT>
__global__ __launch_bounds__(512) void igemm_nt8h_kernel(const NtParams p, int tiles_m, int tiles_n) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int BM = 128, BN = 128;
    constexpr int TM = 2, TN = 4;
    constexpr int SLOT = 128 * 128;                      // 16 KiB: 128 K rows
    constexpr int STAGE = 2 * SLOT, NST = 4;             // A | B; ring of four
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nsplit = SPLITK && p.nsplit > 1 ? p.nsplit : 1;
    const int phase = blockIdx.z / nsplit, split = blockIdx.z - phase * nsplit;
    const NtPhase ph = p.ph[phase];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int L = blockIdx.x;                                  // XCD-contiguous logical tile order (igemm_nt8s.hip)
    if (p.xcd_remap) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = L & 7, j = L >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m_tile = L / tiles_n, n_tile = L - m_tile * tiles_n;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const int HU = p.H << p.up, WU = p.W << p.up;
    const int rsub = lane >> 3, pos = lane & 7;
    const int srcchunk = pos ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned row_bytes = (unsigned)p.C * sizeof(T);
    const int ncb = p.C / BK;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 1024u;

    unsigned vb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (j * 8 + wave) * 8 + rsub;
        vb[j] = n < p.N ? (unsigned)n * (unsigned)ph.Kpad * (unsigned)sizeof(T) + (unsigned)srcchunk * 16u : EG_OOB;
    }
    const u32x4_t srdA = eg_make_srd(p.src, (unsigned)((size_t)p.B * p.H * p.W * p.C * sizeof(T)));
    const u32x4_t srdB = eg_make_srd(reinterpret_cast<const T*>(p.wp) + ph.w_off, (unsigned)((size_t)p.N * ph.Kpad * sizeof(T)));

    const int nk_all = ph.Kpad / BK;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = split * per;
    const int nk = max(0, min(per, nk_all - kt0));

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];      // fragment sets [k half][tile]

    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    int a_pix0[2], a_y[2], a_x[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + (j * 8 + wave) * 8 + rsub;
        const int b = m >> (p.lOW + p.lOH);
        a_pix0[j] = (m < p.M) ? b * p.H * p.W : -1;
        a_y[j] = ((m >> p.lOW) & OHm) * p.sy + ph.dy0;
        a_x[j] = (m & OWm) * p.sx + ph.dx0;
    }
    unsigned va[2];
    auto tap_offsets = [&](int ty, int tx) {
        const int oy = ty * ph.dys, ox = tx * ph.dxs;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = a_y[j] + oy, ix = a_x[j] + ox;
            const bool ok = a_pix0[j] >= 0 && iy >= 0 && iy < HU && ix >= 0 && ix < WU;
            const unsigned pix = (unsigned)(a_pix0[j] + (iy >> p.up) * p.W + (ix >> p.up));
            va[j] = ok ? pix * row_bytes + (unsigned)srcchunk * 16u : EG_OOB;
        }
    };
    const int tap0 = kt0 / ncb;
    int ty = tap0 / ph.TW, tx = tap0 - ty * ph.TW;
    unsigned kc_bytes = (unsigned)(kt0 - tap0 * ncb) * 128u;
    unsigned kb_bytes = (unsigned)kt0 * 128u;
    if (ty < ph.TH) tap_offsets(ty, tx);
    else { va[0] = EG_OOB; va[1] = EG_OOB; }
    auto advance_a = [&]() {
        kc_bytes += 128u;
        if (kc_bytes >= row_bytes) {                     // next tap (uniform branch)
            kc_bytes = 0;
            if (++tx == ph.TW) { tx = 0; ++ty; }
            if (ty < ph.TH) tap_offsets(ty, tx);
            else { va[0] = EG_OOB; va[1] = EG_OOB; }     // K padding beyond the last tap
        }
    };
    // piece q = 0..3 of the K tile being issued into the stage at `base`: A pieces (rows 8w, 8w + 64), then B pieces
    auto issue_piece = [&](unsigned base, auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < 2) {
            eg_bufdma1f<q * 0x2000>(srdA, va[q], kc_bytes, base);
            if (q == 1) advance_a();
        } else {
            eg_bufdma1f<SLOT + (q - 2) * 0x2000>(srdB, vb[q - 2], kb_bytes, base);
            if (q == 3) kb_bytes += 128u;
        }
    };
    auto read_b4 = [&](uint4 (&nb)[2][TN], const char* sb, int ks) {
#pragma unroll
        for (int j = 0; j < TN; ++j) nb[ks][j] = *reinterpret_cast<const uint4*>(sb + lds_off(j * 16 + frow, ks * 4 + fq));
    };
    auto read_a2 = [&](uint4 (&na)[2][TM], const char* sa, int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) na[ks][i] = *reinterpret_cast<const uint4*>(sa + lds_off(i * 16 + frow, ks * 4 + fq));
    };
    auto mma4 = [&](uint4 (&ca)[2][TM], uint4 (&cb)[2][TN], int g) {     // group g: k half g >> 1, row tile g & 1, four column tiles
        const int ks = g >> 1, i = g & 1;
#pragma unroll
        for (int j = 0; j < TN; ++j) mfma_step<T>(ca[ks][i], cb[ks][j], acc[i][j]);
    };
    auto stage_of = [&](int t) { return t & (NST - 1); };
    auto iter = [&](uint4 (&ca)[2][TM], uint4 (&cb)[2][TN], uint4 (&na)[2][TM], uint4 (&nb)[2][TN], int t, auto dmc, bool dm_rt) {
        constexpr bool DM = decltype(dmc)::value;        // true: main loop, every piece issued; false: tail, pieces guarded by dm_rt
        const int st_r = stage_of(t + 1);
        const char* sa = smem + st_r * STAGE + wm * (32 * 128);
        const char* sb = smem + st_r * STAGE + SLOT + wn * (64 * 128);
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage_of(t + 3) * STAGE);
        read_b4(nb, sb, 0);
        mma4(ca, cb, 0); __builtin_amdgcn_sched_barrier(0);
        read_b4(nb, sb, 1);
        if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 0>{});
        mma4(ca, cb, 1); __builtin_amdgcn_sched_barrier(0);
        read_a2(na, sa, 0);
        read_a2(na, sa, 1);
        if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 1>{});
        mma4(ca, cb, 2); __builtin_amdgcn_sched_barrier(0);
        if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 2>{});
        if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 3>{});
        mma4(ca, cb, 3); __builtin_amdgcn_sched_barrier(0);
        if (DM || dm_rt) nt8h_wait<4>();
        else nt8h_wait<0>();
        barrier();
    };
    // prologue: K tiles 0, 1, 2 in flight; fragments of K tile 0 in registers; K tile 1 visible
    const int npro = min(nk, 3);
    for (int s = 0; s < npro; ++s) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)s * STAGE);
        issue_piece(base, std::integral_constant<int, 0>{}); issue_piece(base, std::integral_constant<int, 1>{});
        issue_piece(base, std::integral_constant<int, 2>{}); issue_piece(base, std::integral_constant<int, 3>{});
    }
    nt8h_wait_dyn(4 * max(npro - 1, 0));
    barrier();
    if (nk > 0) {
        const char* sa = smem + wm * (32 * 128);
        const char* sb = smem + SLOT + wn * (64 * 128);
        read_b4(fb0, sb, 0); read_b4(fb0, sb, 1); read_a2(fa0, sa, 0); read_a2(fa0, sa, 1);
    }
    nt8h_wait_dyn(4 * max(npro - 2, 0));
    barrier();
    int t = 0;
    for (; t + 4 < nk;) {
        iter(fa0, fb0, fa1, fb1, t, std::true_type{}, true); ++t;
        iter(fa1, fb1, fa0, fb0, t, std::true_type{}, true); ++t;
    }
    for (; t < nk;) {                                   // tail (t is even): DMA only while K tiles t + 3 exist
        iter(fa0, fb0, fa1, fb1, t, std::false_type{}, t + 3 < nk); ++t;
        if (t < nk) { iter(fa1, fb1, fa0, fb0, t, std::false_type{}, t + 3 < nk); ++t; }
    }

    // ---- epilogue: all DMA has landed, all reads are retired, every wave is past the last barrier: the LDS is free ----
    if (SPLITK && nsplit > 1) {
        // K splits reduced inside the launch by the last-arriving workgroup (protocol and comments: igemm_nt8s.hip)
        const int nphase = gridDim.z / nsplit;
        const size_t slab = (size_t)nphase * ((size_t)tiles_m * BM) * p.N;
        float* part = p.part + ((size_t)phase * ((size_t)tiles_m * BM) + m0) * p.N + n0;
        float* mine = part + (size_t)split * slab;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wm * TM + i) * 16 + frow;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* a = mine + (size_t)row * p.N + (wn * TN + j) * 16 + fq * 4;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(a), "v"(acc[i][j]) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* cnt = p.split_cnt + (size_t)phase * gridDim.x + blockIdx.x;
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
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(ld[i][j]) : "v"(a) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the loads above are asm outputs: for the compiler they are ready at once.  Pin every use behind the wait (volatile asm
            // statements keep their order; the uses below depend on these outputs) -- without it the scheduler may hoist an add above it.
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(ld[i][j]));
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
    constexpr int PF = 4;
    NtEpiPre<T, TM, TN, PF> epi;
    nt_epi_prefetch<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
    if constexpr (STAT) {
        nt_epilogue_lds_stat<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq, phase * tiles_m + m_tile, n_tile,
                                                         tiles_n);
        return;
    }
    nt_epilogue_lds_pre<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
}

template <typename T, bool SPLITK, bool STAT>
static void launch_h(const NtParams& p, int nphase, int ns, hipStream_t st) {
    constexpr size_t lds = 4 * 2 * 16384;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt8h_kernel<T, SPLITK, STAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int tm = (p.M + 127) / 128, tn = p.N / 128;
    hipLaunchKernelGGL((igemm_nt8h_kernel<T, SPLITK, STAT>), dim3(tm * tn, 1, nphase * ns), dim3(512), lds, st, p, tm, tn);
}

template <typename T>
void eg_launch_nt8h(const NtParams& p, int nphase, int ns, hipStream_t st) {
    NtParams q = p;
    q.nsplit = ns;
    const bool stat = p.stat_mode != EG_STAT_NONE;
    if constexpr (std::is_same<T, float>::value) {
        if (ns > 1) launch_h<T, true, false>(q, nphase, ns, st);
        else launch_h<T, false, false>(q, nphase, ns, st);
    } else {
        if (ns > 1) { if (stat) launch_h<T, true, true>(q, nphase, ns, st); else launch_h<T, true, false>(q, nphase, ns, st); }
        else { if (stat) launch_h<T, false, true>(q, nphase, ns, st); else launch_h<T, false, false>(q, nphase, ns, st); }
    }
}
template void eg_launch_nt8h<float>(const NtParams&, int, int, hipStream_t);
template void eg_launch_nt8h<bf16_t>(const NtParams&, int, int, hipStream_t);
template void eg_launch_nt8h<f16_t>(const NtParams&, int, int, hipStream_t);
