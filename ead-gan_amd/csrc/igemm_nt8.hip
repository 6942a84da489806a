// igemm_nt8: the NT implicit GEMM (conv forward, conv backward-data == ConvTranspose forward, linear) on 256 x BN tiles with
// EIGHT waves per workgroup (512 threads, one workgroup per CU), BN = 128 or 256.
//
//   C[m][n] = sum_k A[m][k] * Wp[n][k]      A rows gathered from an NHWC tensor through a buffer descriptor
//
// Why this shape (measured in round 1 on the 128 x 128 / 4-wave kernel): the L2 -> LDS path delivers ~70 GB/s
// per CU, a 128 x 128 tile needs 1 byte per 64 FLOP and its 2-stage ring drains `vmcnt(0)` every K step, so that kernel sat at
// 650 TFLOP/s whatever its inner loop did.  Here a K step of 64 bf16 moves (256 + BN) * 128 B for 2 * 256 * BN * 64 FLOP
// (85 FLOP/B at BN = 128, 128 FLOP/B at BN = 256), up to two whole K tiles stay in flight ACROSS the barriers behind a counted
// `s_waitcnt vmcnt(N)` that is never 0 inside the loop, and the two halves of the workgroup (waves 0-3 / 4-7: one wave of each
// half on every SIMD) run one half-phase apart, so that on each SIMD one wave issues its `ds_read_b128`s and LDS-DMA pieces while
// its partner issues MFMAs (guide: "The 256^2 8-phase template" -- the schedule is re-derived here for a gathered A operand, the
// example file is not in this image).
//
// Phase (16 MFMAs of 16x16x32 per wave for 16-bit types):
//     L: ds_read_b128 fragments of this phase | LDS-DMA pieces of a later K tile | [last phase of a K tile: counted vmcnt + lgkmcnt(0)]
//     s_barrier
//     M: setprio 1, 16 MFMAs, setprio 0
//     s_barrier
// Waves 4-7 pass one extra barrier before their first phase and waves 0-3 one after their last: half 1's L runs beside half 0's M.
//
// LDS: K rows of 128 bytes, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128, swizzle applied on the
// DMA source side).  BN = 128: ring of three K tiles of 48 KiB (A rows 0..255 | B rows 0..127).  BN = 256: ring of ten 16-KiB slots
// (128 rows each), four per K tile in the order A0 A1 B0 B1, filled six slots ahead of the K tile that is being read.
//
// Hazards (one per rule, both hold for either half because of where the waits sit):
//   RAW  a slot is read only after EVERY wave has executed the counted vmcnt that retires its own pieces of it and then a barrier that
//        the reader passes afterwards: the wait sits at the end of the L part of a K tile's LAST phase (half 0 passes barrier 2g there,
//        half 1 barrier 2g+1; the first read of the next K tile comes after barrier 2g+1 / 2g+2).
//   WAR  a slot is refilled only after every wave has retired its last ds_read of it: `lgkmcnt(0)` sits in front of the same barrier,
//        and the refill is issued at the earliest in the next phase's L part.
#include <stdlib.h>
#include <algorithm>

#include "eg_common.h"
#include "igemm_nt.h"

template <int N>
__device__ __forceinline__ void eg_wait_vm_lgkm0() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}

// DBG (diagnostic builds, EG_NT8_DBG in the environment, wrong results by design): 1 = no LDS-DMA inside the K loop, 2 = no MFMA,
// 3 = no fragment ds_reads -- what each of the three streams costs a phase when the other two are left alone.
template <typename T, int BN, bool SPLITK, int DBG = 0>
__global__ __launch_bounds__(512) void igemm_nt8_kernel(const NtParams p) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int BM = 256;
    constexpr int WGN = BN / 64, WGM = 8 / WGN;          // waves along N (2 or 4) and M (4 or 2): wave tile (256 / WGM) x 64
    constexpr int TM = BM / WGM / 16, TN = 4;            // 16-row / 16-column MFMA tiles per wave: 4 x 4 or 8 x 4
    constexpr int NPH = TM / 2;                          // phases per K tile: two row tiles x four column tiles x two k halves = 16 MFMAs
    constexpr int SLOT = 128 * 128;                      // 16 KiB: 128 K rows
    constexpr int B_SL = BN / 64;                        // B pieces per wave and K tile (A: 4)
    static_assert(BN == 128 || BN == 256, "BN");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nsplit = SPLITK && p.nsplit > 1 ? p.nsplit : 1;
    const int phase = blockIdx.z / nsplit, split = blockIdx.z - phase * nsplit;
    const NtPhase ph = p.ph[phase];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const bool late = wave >= 4;                         // the half that runs one barrier behind
    // every XCD gets a contiguous range of M tiles (workgroup ids go round robin over the 8 XCDs; neighbouring tiles share halo rows and
    // both N tiles of an M tile share all of A)
    const int bx = (p.xcd_remap && (gridDim.x & 7) == 0) ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int m0 = bx * BM, n0 = blockIdx.y * BN;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const int HU = p.H << p.up, WU = p.W << p.up;
    // DMA piece = 8 K rows x 128 B; piece q of a 128-row slot covers rows 8q..8q+7; wave w issues pieces w and w + 8 of every slot.
    // lane -> (row = lane / 8, position = lane % 8); (row >> 1) & 7 == ((w & 1) * 4 + lane / 16) & 7 for both pieces.
    const int rsub = lane >> 3, pos = lane & 7;
    const int srcchunk = pos ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);

    int a_pix0[4], a_y[4], a_x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + (j * 8 + wave) * 8 + rsub;
        const int b = m >> (p.lOW + p.lOH);
        a_pix0[j] = (m < p.M) ? b * p.H * p.W : -1;
        a_y[j] = ((m >> p.lOW) & OHm) * p.sy + ph.dy0;
        a_x[j] = (m & OWm) * p.sx + ph.dx0;
    }
    const unsigned row_bytes = (unsigned)p.C * sizeof(T);
    unsigned va[4], vb[B_SL];
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
    for (int j = 0; j < B_SL; ++j) {
        const int n = n0 + (j * 8 + wave) * 8 + rsub;
        vb[j] = n < p.N ? (unsigned)n * (unsigned)ph.Kpad * (unsigned)sizeof(T) + (unsigned)srcchunk * 16u : EG_OOB;
    }
    const u32x4_t srdA = eg_make_srd(p.src, (unsigned)((size_t)p.B * p.H * p.W * p.C * sizeof(T)));
    const u32x4_t srdB = eg_make_srd(reinterpret_cast<const T*>(p.wp) + ph.w_off, (unsigned)((size_t)p.N * ph.Kpad * sizeof(T)));
    // this block's K tiles [kt0, kt0 + nk)
    const int nk_all = ph.Kpad / BK;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = split * per;
    const int nk = max(0, min(per, nk_all - kt0));

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 1024u;
    // wave-uniform walk of the A gather over (tap, channel block); it runs ahead of the K tile that is being multiplied
    const int steps_per_tap = p.C / BK;
    const int tap0 = kt0 / steps_per_tap;
    int ty = tap0 / ph.TW, tx = tap0 - ty * ph.TW;
    unsigned kc_bytes = (unsigned)(kt0 - tap0 * steps_per_tap) * 128u;
    if (ty < ph.TH) tap_offsets(ty, tx);
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = EG_OOB;
    }
    auto advance_a = [&]() {
        kc_bytes += 128u;
        if (kc_bytes >= row_bytes) {                   // next tap (uniform branch)
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
    uint4 bfr[2][TN], afr[2][2];                       // [k half][tile]
    if (DBG == 3) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[ks][j] = make_uint4(lane, 1, 2, 3);
            afr[ks][0] = afr[ks][1] = make_uint4(3, lane, 1, 0);
        }
    }
    auto read_b = [&](const char* sb) {
        if (DBG == 3) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[ks][j] = *reinterpret_cast<const uint4*>(sb + lds_off(j * 16 + frow, ks * 4 + fq));
    };
    auto read_a = [&](const char* sa, int h) {         // row tiles 2h, 2h + 1 of the wave
        if (DBG == 3) return;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) afr[ks][i] = *reinterpret_cast<const uint4*>(sa + lds_off((2 * h + i) * 16 + frow, ks * 4 + fq));
    };
    auto mma = [&](int h) {
        if (DBG == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(afr[ks][i].x), "v"(afr[ks][i].y), "v"(afr[ks][i].z), "v"(afr[ks][i].w));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bfr[ks][j].x), "v"(bfr[ks][j].y), "v"(bfr[ks][j].z), "v"(bfr[ks][j].w));
            }
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mfma_step<T>(afr[ks][i], bfr[ks][j], acc[2 * h + i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (BN == 128) {
        // ---- ring of three K tiles: [A 256 rows | B 128 rows] = 48 KiB each --------------------------------------------------------
        constexpr int STAGE = 3 * SLOT;
        auto issue_a = [&](int stage) {
            eg_bufdma4s<0x2000>(srdA, va[0], va[1], va[2], va[3], kc_bytes, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage * STAGE));
            advance_a();
        };
        auto issue_b = [&](int stage, int kt) {
            eg_bufdma2s<0x2000>(srdB, vb[0], vb[1], (unsigned)(kt0 + kt) * 128u, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage * STAGE + 2 * SLOT));
        };
        if (nk > 0) { issue_a(0); issue_b(0, 0); }
        if (nk > 1) { issue_a(1); issue_b(1, 1); }
        if (nk > 1) eg_wait_vm_lgkm0<6>();
        else eg_wait_vm_lgkm0<0>();
        barrier();
        if (late) barrier();
        int st_use = 0, st_fill = 2;
        for (int kt = 0; kt < nk; ++kt) {
            const char* sa = smem + st_use * STAGE + wm * (TM * 16 * 128);
            const char* sb = smem + st_use * STAGE + 2 * SLOT + wn * (64 * 128);
            const bool more = kt + 2 < nk;
            // phase 0
            read_b(sb);
            read_a(sa, 0);
            if (more && DBG != 1) issue_a(st_fill);
            barrier();
            mma(0);
            barrier();
            // phase 1 (last of the K tile): retire this wave's pieces of K tile kt + 1 and its reads of K tile kt before the barrier
            read_a(sa, 1);
            if (more) {
                if (DBG != 1) issue_b(st_fill, kt + 2);
                eg_wait_vm_lgkm0<6>();
            } else
                eg_wait_vm_lgkm0<0>();
            barrier();
            mma(1);
            barrier();
            st_use = st_use == 2 ? 0 : st_use + 1;
            st_fill = st_fill == 2 ? 0 : st_fill + 1;
        }
        if (!late) barrier();
    } else {
        // ---- ring of ten 16-KiB slots; K tile t owns slot indices 4t .. 4t+3 = A0 A1 B0 B1, filled six slots ahead -------------------
        // issue order per K tile t (one slot = two pieces per phase): B0(t+1) B1(t+1) A0(t+2) A1(t+2); the prologue issues K tile 0
        // and A0 A1 of K tile 1.  `pos_*` are ring positions (index mod 10) of the next slot to fill / of the K tile being read.
        auto slot_lds = [&](int posn) { return __builtin_amdgcn_readfirstlane(lds0 + (unsigned)posn * SLOT); };
        auto ring = [](int v) { return v >= 10 ? v - 10 : v; };
        int ka = 0, kb = 0;                            // K tiles whose A / B slots have been issued
        int pa = 0, pb = 2;                            // ring positions of the next A0 / B0 slot to fill
        auto issue_a_half = [&](int hf) {              // A0 (hf = 0) or A1 (hf = 1) of K tile ka
            eg_bufdma2s<0x2000>(srdA, va[2 * hf], va[2 * hf + 1], kc_bytes, slot_lds(ring(pa + hf)));
            if (hf == 1) { advance_a(); ++ka; pa = ring(pa + 4); }
        };
        auto issue_b_half = [&](int hf) {
            eg_bufdma2s<0x2000>(srdB, vb[2 * hf], vb[2 * hf + 1], (unsigned)(kt0 + kb) * 128u, slot_lds(ring(pb + hf)));
            if (hf == 1) { ++kb; pb = ring(pb + 4); }
        };
        if (nk > 0) { issue_a_half(0); issue_a_half(1); issue_b_half(0); issue_b_half(1); }
        if (nk > 1) { issue_a_half(0); issue_a_half(1); }
        if (nk > 1) eg_wait_vm_lgkm0<4>();
        else eg_wait_vm_lgkm0<0>();
        barrier();
        if (late) barrier();
        int pu = 0;                                    // ring position of A0 of the K tile being read
        for (int kt = 0; kt < nk; ++kt) {
            const char* sa = smem + ring(pu + wm) * SLOT;                            // A half wm: the wave's 128 rows
            const char* sb = smem + ring(pu + 2 + (wn >> 1)) * SLOT + (wn & 1) * (64 * 128);
            const bool more_b = kt + 1 < nk, more_a = kt + 2 < nk;
            // phase 0
            read_b(sb);
            read_a(sa, 0);
            if (more_b && DBG != 1) issue_b_half(0);
            barrier();
            mma(0);
            barrier();
            // phase 1
            read_a(sa, 1);
            if (more_b && DBG != 1) issue_b_half(1);
            barrier();
            mma(1);
            barrier();
            // phase 2
            read_a(sa, 2);
            if (more_a && DBG != 1) issue_a_half(0);
            barrier();
            mma(2);
            barrier();
            // phase 3 (last): K tile kt + 1 must have landed; only A0 A1 of K tile kt + 2 may still be in flight
            read_a(sa, 3);
            if (more_a) {
                if (DBG != 1) issue_a_half(1);
                eg_wait_vm_lgkm0<4>();
            } else
                eg_wait_vm_lgkm0<0>();
            barrier();
            mma(3);
            barrier();
            pu = ring(pu + 4);
        }
        if (!late) barrier();
    }

    // ---- epilogue: every DMA has landed and every ds_read has been retired (the last phase waited for both), all waves are past the
    // last barrier -> the ring memory is free --------------------------------------------------------------------------------------
    if (SPLITK && nsplit > 1) {
        // raw fp32 partial tile; nt_splitk_epilogue_kernel sums the splits in a fixed order and applies the epilogue
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
    if constexpr (BN == 128) {
        constexpr int PF = 8;                          // all eight store iterations' mask vectors in flight while the tile goes through LDS
        NtEpiPre<T, TM, TN, PF> epi;
        nt_epi_prefetch<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
        nt_epilogue_lds_pre<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
    } else {
        // 256 x 256 fp32 does not fit the LDS: two 128-column windows
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col0 = (wn >> 1) == h ? (wn & 1) * 64 : -1;
            nt_epilogue_lds<T, BM, 128, TM, TN, 512>(p, ph, acc, smem, m0, n0 + h * 128, wm * TM * 16, col0, tid, frow, fq);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <typename T, int BN, bool SPLITK, int DBG>
static void launch_dbg(const NtParams& p, const dim3& grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt8_kernel<T, BN, SPLITK, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_nt8_kernel<T, BN, SPLITK, DBG>), grid, dim3(512), lds, st, p);
}

template <typename T, int BN, bool SPLITK>
static void launch_cfg(const NtParams& p, int nphase, int ns, hipStream_t st) {
    constexpr size_t lds = BN == 128 ? 3 * 3 * 16384 : 10 * 16384;
    const dim3 grid((p.M + 255) / 256, p.N / BN, nphase * ns);
    if constexpr (std::is_same<T, bf16_t>::value && !SPLITK) {
        static const int dbg = [] { const char* e = getenv("EG_NT8_DBG"); return e ? atoi(e) : 0; }();
        if (dbg == 1) return launch_dbg<T, BN, SPLITK, 1>(p, grid, lds, st);
        if (dbg == 2) return launch_dbg<T, BN, SPLITK, 2>(p, grid, lds, st);
        if (dbg == 3) return launch_dbg<T, BN, SPLITK, 3>(p, grid, lds, st);
    }
    launch_dbg<T, BN, SPLITK, 0>(p, grid, lds, st);
}

template <typename T>
void eg_launch_nt8(const NtParams& p, int nphase, int bn, int ns, hipStream_t st) {
    NtParams q = p;
    q.nsplit = ns;
    if (bn == 256) {
        if (ns > 1) launch_cfg<T, 256, true>(q, nphase, ns, st);
        else launch_cfg<T, 256, false>(q, nphase, ns, st);
    } else {
        if (ns > 1) launch_cfg<T, 128, true>(q, nphase, ns, st);
        else launch_cfg<T, 128, false>(q, nphase, ns, st);
    }
}

template void eg_launch_nt8<float>(const NtParams&, int, int, int, hipStream_t);
template void eg_launch_nt8<bf16_t>(const NtParams&, int, int, int, hipStream_t);
template void eg_launch_nt8<f16_t>(const NtParams&, int, int, int, hipStream_t);
