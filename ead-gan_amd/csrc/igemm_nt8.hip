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
// igemm_nt8p: the 256 x 128 kernel with the A operand held in LDS as an INPUT PATCH instead of im2col rows.
//
// Measured on the kernel above (profiles/r02_a_nt8_ablation.txt): a phase is bound by the LDS-DMA path (~28 cycles per 1-KiB piece and
// CU, 44-53 GB/s), not by MFMA -- and 32 of the 48 KiB a K tile moves are A rows, although in a convolution every input pixel serves
// (k / stride)^2 filter taps of neighbouring output positions of the SAME tile.  So the filter taps are grouped into classes that walk
// one pixel lattice (4x4 stride 2: four classes of 2 x 2 taps, one per input parity; a backward-data phase: its 2 x 2 taps; 3x3
// stride 1: all nine), and for one class and one 64-channel block the tile's pixels are loaded ONCE:
//     256 output positions = nimg images x OHt rows x OW columns  ->  patch of nimg x (OHt + AH - 1) x (OW + AW - 1) pixels x 128 B
// (16x16 lattice: 289 pixels instead of 4 x 256 rows = 3.5x fewer A bytes, 8x8: 3.2x, 4x4: 2.6x).  The K loop runs class by class,
// channel block by channel block, tap by tap; the A fragment of lattice row m for tap (ay, ax) is patch pixel  pb(m) + ay * PW + ax,
// a per-lane LDS address like any other ds_read_b128.  B (weights) is staged as before, one 16-KiB K tile per tap from the packed
// panel's column block (tap * C + cb * 64) -- the panel layout does not change, only the ORDER in which K is accumulated, so this
// variant is deterministic but not bit-identical to the tap-major kernels.
//
// LDS: two patch buffers of 56 KiB (448 pixel slots = 7 pieces per wave) + a ring of three B tiles = 160 KiB.  The next (class, channel
// block)'s patch is issued during the first two K tiles of the current one, B two K tiles ahead; one counted vmcnt per K tile:
// everything older than this K tile's own issues must have landed (that is B of the next K tile), and at the last K tile of a
// (class, channel block) step the next patch as well.  Phases, barriers and the half-phase stagger are those of igemm_nt8_kernel.
// ------------------------------------------------------------------------------------------------

template <int N>
__device__ __forceinline__ void eg_wait_vm_lgkm0_n() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void eg_wait_vm_lgkm0_dyn(int n) {      // wave-uniform n in 0..9
    switch (n) {
        case 0: eg_wait_vm_lgkm0_n<0>(); break;
        case 1: eg_wait_vm_lgkm0_n<1>(); break;
        case 2: eg_wait_vm_lgkm0_n<2>(); break;
        case 3: eg_wait_vm_lgkm0_n<3>(); break;
        case 4: eg_wait_vm_lgkm0_n<4>(); break;
        case 5: eg_wait_vm_lgkm0_n<5>(); break;
        case 6: eg_wait_vm_lgkm0_n<6>(); break;
        case 7: eg_wait_vm_lgkm0_n<7>(); break;
        case 8: eg_wait_vm_lgkm0_n<8>(); break;
        default: eg_wait_vm_lgkm0_n<9>(); break;
    }
}

// DBG as in igemm_nt8_kernel, plus 4 = A fragments read at im2col-aligned patch addresses (no tap offset: wrong rows, conflict-free)
template <typename T, bool SPLITK, int DBG = 0>
__global__ __launch_bounds__(512) void igemm_nt8p_kernel(const NtParams p, const Nt8pGeom g, int tiles_m, int tiles_n) {
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int BM = 256, BN = 128;
    constexpr int TM = 4, TN = 4;                        // waves 4 (M) x 2 (N), wave tile 64 x 64
    constexpr int SLOT = 128 * 128;                      // one B K tile
    constexpr int PCAP = EG_P8P_SLOTS * 128;             // one patch buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nsplit = SPLITK && p.nsplit > 1 ? p.nsplit : 1;
    const int phase = blockIdx.z / nsplit, split = blockIdx.z - phase * nsplit;
    const NtPhase ph = p.ph[phase];
    const Nt8pPhase& gp = g.ph[phase];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool late = wave >= 4;
    // workgroup ids go round robin over the 8 XCDs: XCD x runs ids x, x + 8, ...  Give each XCD a CONTIGUOUS range of logical tiles
    // (bijective for any grid size), N tiles of one M tile adjacent: they share the whole patch through that XCD's L2.
    int L = blockIdx.x;
    if (p.xcd_remap) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = L & 7, j = L >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int m_tile = L / tiles_n, n_tile = L - m_tile * tiles_n;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const int HU = p.H << p.up, WU = p.W << p.up;
    const int b0 = m0 >> (p.lOW + p.lOH);
    const int oyt0 = g.nimg == 1 ? ((m0 >> p.lOW) & OHm) : 0;
    const int rsub = lane >> 3, pos = lane & 7;
    const int srcchunk = pos ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
    const int frow = lane & 15, fq = lane >> 4;
    const unsigned row_bytes = (unsigned)p.C * sizeof(T);
    const int ncb = p.C / BK;

    // patch pixel of every lattice row this lane reads A fragments for (row tiles i = 0..3 of the wave)
    int pb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 64 + i * 16 + frow;
        const int img = (m >> (p.lOW + p.lOH)) - b0, oy = ((m >> p.lOW) & OHm) - oyt0, ox = m & OWm;
        pb[i] = (img * g.PH + oy) * g.PW + ox;
        if (DBG == 4) pb[i] = wm * 64 + i * 16 + frow;
    }
    // per-lane source offsets of this wave's patch pieces for one class (pieces w, w + 8, ... of the patch: 8 pixel slots each)
    unsigned vp[7];
    auto patch_offsets = [&](const NtClass& c) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const unsigned ps = (unsigned)(((wave + 8 * q) << 3) + rsub);
            const unsigned img = (ps * g.inv_plane) >> 20;
            const unsigned rem = ps - img * (unsigned)(g.PH * g.PW);
            const unsigned qy = (rem * g.inv_pw) >> 20;
            const unsigned qx = rem - qy * (unsigned)g.PW;
            const int iy = ((int)qy + oyt0) * p.sy + c.oy0, ix = (int)qx * p.sx + c.ox0;
            const bool ok = (int)ps < g.npix && b0 + (int)img < p.B && iy >= 0 && iy < HU && ix >= 0 && ix < WU;
            const unsigned pix = (unsigned)((b0 + (int)img) * p.H * p.W + (iy >> p.up) * p.W + (ix >> p.up));
            vp[q] = ok ? pix * row_bytes + (unsigned)srcchunk * 16u : EG_OOB;
        }
    };
    unsigned vb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (j * 8 + wave) * 8 + rsub;
        vb[j] = n < p.N ? (unsigned)n * (unsigned)ph.Kpad * (unsigned)sizeof(T) + (unsigned)srcchunk * 16u : EG_OOB;
    }
    const u32x4_t srdA = eg_make_srd(p.src, (unsigned)((size_t)p.B * p.H * p.W * p.C * sizeof(T)));
    const u32x4_t srdB = eg_make_srd(reinterpret_cast<const T*>(p.wp) + ph.w_off, (unsigned)((size_t)p.N * ph.Kpad * sizeof(T)));

    // K tiles of this phase in class order; this block's range [kt0, kt0 + nk)
    int nk_all = 0;
    for (int c = 0; c < gp.ncls; ++c) nk_all += gp.cls[c].AH * gp.cls[c].AW * ncb;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = split * per;
    const int nk = max(0, min(per, nk_all - kt0));

    // iterator over the K tiles (class, channel block, tap row, tap column); the class's fields are cached so that the loop does not
    // re-read kernel arguments
    struct It { int cls, cb, ay, ax, AH, AW, ty0, tys, tx0, txs; };
    auto load_cls = [&](It& it) {
        const NtClass& c = gp.cls[min(it.cls, gp.ncls - 1)];
        it.AH = c.AH; it.AW = c.AW; it.ty0 = c.ty0; it.tys = c.tys; it.tx0 = c.tx0; it.txs = c.txs;
    };
    auto advance = [&](It& it) {
        if (++it.ax == it.AW) {
            it.ax = 0;
            if (++it.ay == it.AH) {
                it.ay = 0;
                if (++it.cb == ncb) { it.cb = 0; ++it.cls; load_cls(it); }
            }
        }
    };
    It itc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    {
        int rem = kt0;
        while (itc.cls < gp.ncls - 1 && rem >= gp.cls[itc.cls].AH * gp.cls[itc.cls].AW * ncb) {
            rem -= gp.cls[itc.cls].AH * gp.cls[itc.cls].AW * ncb;
            ++itc.cls;
        }
        load_cls(itc);
        const int taps = itc.AH * itc.AW;
        itc.cb = rem / taps;
        rem -= itc.cb * taps;
        itc.ay = rem / itc.AW;
        itc.ax = rem - itc.ay * itc.AW;
    }
    It itb = itc;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 1024u;
    auto issue_b = [&](int stage) {                    // B of K tile `itb`, then step the iterator
        const unsigned tap = (unsigned)((itb.ty0 + itb.ay * itb.tys) * ph.TW + (itb.tx0 + itb.ax * itb.txs));
        eg_bufdma2s<0x2000>(srdB, vb[0], vb[1], tap * row_bytes + (unsigned)itb.cb * 128u,
                            __builtin_amdgcn_readfirstlane(lds0 + 2u * PCAP + (unsigned)stage * SLOT));
        advance(itb);
    };
    auto issue_patch = [&](int par, unsigned soff, int q0, int q1) {    // pieces q0 .. q1-1 of this wave (those below npp)
#pragma unroll
        for (int q = 0; q < 7; ++q)
            if (q >= q0 && q < q1 && q < g.npp)
                eg_bufdma1s(srdA, vp[q], soff, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)par * PCAP + (unsigned)q * 0x2000u));
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 bfr[2][TN], afr[2][2];
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
    auto read_a = [&](const char* pbuf, int tapoff, int h) {     // row tiles 2h, 2h + 1 of the wave from the patch
        if (DBG == 3) return;
        if (DBG == 4) tapoff = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = pb[2 * h + i] + tapoff;
            const int a0 = (pp << 7) + ((fq ^ ((pp >> 1) & 7)) << 4);
            afr[0][i] = *reinterpret_cast<const uint4*>(pbuf + a0);
            afr[1][i] = *reinterpret_cast<const uint4*>(pbuf + (a0 ^ 64));
        }
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

    // ---- prologue: the first patch (all pieces), B of K tiles 0 and 1 ----
    int par = 0;                                       // patch buffer being read
    int vp_cls = -1;                                   // class vp[] currently describes
    if (nk > 0) {
        patch_offsets(gp.cls[itc.cls]);
        vp_cls = itc.cls;
        issue_patch(0, (unsigned)itc.cb * 128u, 0, 7);
        issue_b(0);
    }
    if (nk > 1) issue_b(1);
    eg_wait_vm_lgkm0_dyn(nk > 1 ? 2 : 0);
    barrier();
    if (late) barrier();

    int st_use = 0, st_fill = 2;
    int since = 0;                                     // K tiles this block has spent in the current (class, channel block) step
    int left = 0;                                      // K tiles left in the step, this one included
    bool have_next = false;                            // a further step exists inside this block's K range
    int ncls_n = 0, ncb_n = 0;                         // ... and which
    for (int kt = 0; kt < nk; ++kt) {
        if (since == 0) {
            // step entry: how long it lasts for this block, and which patch comes next
            left = (itc.AH - itc.ay) * itc.AW - itc.ax;
            have_next = kt + left < nk;
            if (have_next) {
                ncb_n = itc.cb + 1;
                ncls_n = itc.cls;
                if (ncb_n == ncb) { ncb_n = 0; ++ncls_n; }
                if (ncls_n != vp_cls) {                // the current patch is complete: its offsets are not needed any more
                    patch_offsets(gp.cls[ncls_n]);
                    vp_cls = ncls_n;
                }
            }
        }
        const char* pbuf = smem + par * PCAP;
        const char* sb = smem + 2 * PCAP + st_use * SLOT + wn * (64 * 128);
        const int tapoff = itc.ay * g.PW + itc.ax;
        const bool more_b = kt + 2 < nk;
        const bool last = left == 1;
        // phase 0
        read_b(sb);
        read_a(pbuf, tapoff, 0);
        int n_p = 0;
        if (have_next && DBG != 1) {
            const unsigned soff = (unsigned)ncb_n * 128u;
            if (since == 0) {
                issue_patch(par ^ 1, soff, 0, last ? 7 : 4);
                n_p = min(g.npp, last ? 7 : 4);
            } else if (since == 1) {
                issue_patch(par ^ 1, soff, 4, 7);
                n_p = max(0, g.npp - 4);
            }
        }
        barrier();
        mma(0);
        barrier();
        // phase 1 (last of the K tile): B of K tile kt + 2 goes out, then everything older than this K tile's own issues must have
        // landed (B of K tile kt + 1) -- at the end of a step the next patch too -- and this wave's reads must be retired
        read_a(pbuf, tapoff, 1);
        if (more_b && DBG != 1) issue_b(st_fill);
        eg_wait_vm_lgkm0_dyn(DBG == 1 ? 0 : (more_b ? 2 : 0) + (last ? 0 : n_p));
        barrier();
        mma(1);
        barrier();
        st_use = st_use == 2 ? 0 : st_use + 1;
        st_fill = st_fill == 2 ? 0 : st_fill + 1;
        advance(itc);
        ++since;
        --left;
        if (left == 0) { since = 0; par ^= 1; }
    }
    if (!late) barrier();

    if (SPLITK && nsplit > 1) {
        const int nphase = gridDim.z / nsplit;
        float* part = p.part + ((size_t)(split * nphase + phase) * ((size_t)tiles_m * BM) + m0) * p.N + n0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wm * TM + i) * 16 + frow;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *reinterpret_cast<f32x4*>(part + (size_t)row * p.N + (wn * TN + j) * 16 + fq * 4) = acc[i][j];
        }
        return;
    }
    constexpr int PF = 8;
    NtEpiPre<T, TM, TN, PF> epi;
    nt_epi_prefetch<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
    nt_epilogue_lds_pre<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
}

// patch geometry of a launch; false if the problem does not fit the patch buffers (the caller then uses igemm_nt8_kernel)
bool eg_nt8p_geometry(const NtParams& p, int nphase, Nt8pGeom& g) {
    memset(&g, 0, sizeof(g));
    const int OH = 1 << p.lOH, OW = 1 << p.lOW;
    if (OW > 256 || p.sy > 2 || p.sx > 2 || p.sy < 1 || p.sx < 1) return false;
    if (OH * OW >= 256) { g.nimg = 1; g.OHt = 256 / OW; }
    else { g.nimg = 256 / (OH * OW); g.OHt = OH; }
    int amax_h = 1, amax_w = 1;
    for (int f = 0; f < nphase; ++f) {
        const NtPhase& ph = p.ph[f];
        struct Ax { int t0, ts, A, o0; } ys[2], xs[2];
        int ny = 0, nx = 0;
        auto axis = [](int T, int d0, int ds, int s, Ax* out, int& n) -> bool {
            n = 0;
            if (ds > 0) {                              // forward: taps r, r + s, ... walk the lattice of residue r
                if (ds != 1) return false;
                for (int r = 0; r < s && r < T; ++r) out[n++] = {r, s, (T - r + s - 1) / s, d0 + r};
            } else {                                   // backward-data phase: consecutive source pixels, taps in reverse
                if (ds != -1 || s != 1) return false;
                out[n++] = {T - 1, -1, T, d0 - (T - 1)};
            }
            return true;
        };
        if (!axis(ph.TH, ph.dy0, ph.dys, p.sy, ys, ny) || !axis(ph.TW, ph.dx0, ph.dxs, p.sx, xs, nx)) return false;
        Nt8pPhase& gp = g.ph[f];
        gp.ncls = 0;
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) {
                NtClass& c = gp.cls[gp.ncls++];
                c.oy0 = ys[a].o0; c.ox0 = xs[b].o0; c.AH = ys[a].A; c.AW = xs[b].A;
                c.ty0 = ys[a].t0; c.tys = ys[a].ts; c.tx0 = xs[b].t0; c.txs = xs[b].ts;
                amax_h = std::max(amax_h, c.AH); amax_w = std::max(amax_w, c.AW);
            }
    }
    g.PH = g.OHt + amax_h - 1;
    g.PW = OW + amax_w - 1;
    g.npix = g.nimg * g.PH * g.PW;
    if (g.npix > EG_P8P_SLOTS) return false;
    g.npp = ((g.npix + 7) / 8 + 7) / 8;
    g.inv_pw = (1u << 20) / (unsigned)g.PW + 1;
    g.inv_plane = (1u << 20) / (unsigned)(g.PH * g.PW) + 1;
    for (unsigned x = 0; x < 512; ++x)                 // the two magic divisions are exact over the slot range
        if (((x * g.inv_pw) >> 20) != x / (unsigned)g.PW || ((x * g.inv_plane) >> 20) != x / (unsigned)(g.PH * g.PW)) return false;
    return true;
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

template <typename T, bool SPLITK, int DBG>
static void launch_p_dbg(const NtParams& p, const Nt8pGeom& g, int nphase, int ns, hipStream_t st) {
    constexpr size_t lds = 2 * EG_P8P_SLOTS * 128 + 3 * 16384;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt8p_kernel<T, SPLITK, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int tm = (p.M + 255) / 256, tn = p.N / 128;
    hipLaunchKernelGGL((igemm_nt8p_kernel<T, SPLITK, DBG>), dim3(tm * tn, 1, nphase * ns), dim3(512), lds, st, p, g, tm, tn);
}

template <typename T, bool SPLITK>
static void launch_p(const NtParams& p, const Nt8pGeom& g, int nphase, int ns, hipStream_t st) {
    if constexpr (std::is_same<T, bf16_t>::value && !SPLITK) {
        static const int dbg = [] { const char* e = getenv("EG_NT8_DBG"); return e ? atoi(e) : 0; }();
        if (dbg == 1) return launch_p_dbg<T, SPLITK, 1>(p, g, nphase, ns, st);
        if (dbg == 2) return launch_p_dbg<T, SPLITK, 2>(p, g, nphase, ns, st);
        if (dbg == 3) return launch_p_dbg<T, SPLITK, 3>(p, g, nphase, ns, st);
        if (dbg == 4) return launch_p_dbg<T, SPLITK, 4>(p, g, nphase, ns, st);
    }
    launch_p_dbg<T, SPLITK, 0>(p, g, nphase, ns, st);
}

// patch variant (eg_nt8p_geometry must have accepted the problem)
template <typename T>
void eg_launch_nt8p(const NtParams& p, const Nt8pGeom& g, int nphase, int ns, hipStream_t st) {
    NtParams q = p;
    q.nsplit = ns;
    if (ns > 1) launch_p<T, true>(q, g, nphase, ns, st);
    else launch_p<T, false>(q, g, nphase, ns, st);
}
template void eg_launch_nt8p<float>(const NtParams&, const Nt8pGeom&, int, int, hipStream_t);
template void eg_launch_nt8p<bf16_t>(const NtParams&, const Nt8pGeom&, int, int, hipStream_t);
template void eg_launch_nt8p<f16_t>(const NtParams&, const Nt8pGeom&, int, int, hipStream_t);

template void eg_launch_nt8<float>(const NtParams&, int, int, int, hipStream_t);
template void eg_launch_nt8<bf16_t>(const NtParams&, int, int, int, hipStream_t);
template void eg_launch_nt8<f16_t>(const NtParams&, int, int, int, hipStream_t);
