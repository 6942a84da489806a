// igemm_nt8s: 256 x 128 NT implicit GEMM, 8 waves (4 x 2, wave tile 64 x 64), ONE barrier per K tile, fragments double-buffered in
// registers: while a wave issues the 32 MFMAs of K tile t it reads the fragments of K tile t+1 from LDS and issues the LDS-DMA pieces
// of K tile t+3, interleaved group by group (4 MFMAs | 4 ds_read_b128 | 1 DMA piece).
//
// Why (profiles/r02_b_pmc_nt_variants.txt, r02_a_nt8_ablation.txt): with two barriers per 16 MFMAs (igemm_nt8.hip) the waves spend
// 44 % of their cycles parked -- the three streams of a phase (fragment reads, DMA issue, MFMA) overlap only through the partner wave,
// and only as well as the two halves of a phase happen to balance.  Here every wave overlaps its own streams and meets the others
// once per K tile (the barrier that publishes the next K tile's DMA pieces).
//
// PATCH = false: im2col rows, K in tap-major order: bit-identical to the register-staged kernel.
// PATCH = true : the A operand lives in LDS as an input patch shared by the filter taps of a class (see Nt8pGeom in igemm_nt.h):
//                2.6-3.5 x fewer A bytes through the LDS-DMA path, K accumulated class by class (deterministic, other rounding order).
//
// LDS (ring positions rotate with the K tile index):
//   !PATCH  3 stages x [A 256 rows | B 128 rows] x 128 B                          = 144 KiB
//    PATCH  2 patch buffers x 448 pixel slots x 128 B + 3 B stages x 128 rows     = 160 KiB
// Hazards.  Iteration t = { reads of K tile t+1 ; DMA of K tile t+3 into the stage K tile t was read from ; MFMA of K tile t ;
// s_waitcnt vmcnt(pieces issued in this iteration) lgkmcnt(0) ; s_barrier }.  RAW: K tile t+2 (issued in iteration t-1) has landed for
// every wave before the barrier that ends iteration t, its first read is in iteration t+1.  WAR: stage t % 3 was last read in iteration
// t-1 (fragments of K tile t), retired by that iteration's lgkmcnt(0) in front of its barrier.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "eg_common.h"
#include "igemm_nt.h"

// s_waitcnt vmcnt(VM) lgkmcnt(0) through the BUILTIN, not inline asm: hipcc's own wait-count bookkeeping has to see that the fragment
// reads issued in this iteration are retired -- behind an asm wait it still believes them outstanding at the loop's back edge and
// makes the next iteration's first MFMA groups wait for that iteration's fresh ds_reads (lgkmcnt(3..0) in front of every MFMA).
// gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4, expcnt = [6:4] (7 = do not wait), lgkmcnt = [11:8].
template <int VM>
__device__ __forceinline__ void eg_wait_vm_lgkm0() {
    __builtin_amdgcn_s_waitcnt((VM & 15) | ((VM >> 4) << 14) | (7 << 4) | (0 << 8));
}
__device__ __forceinline__ void eg_wait_vm_dyn_lgkm0(int n) {      // wave-uniform n in 0..12
    switch (n) {
        case 0: eg_wait_vm_lgkm0<0>(); break;
        case 1: eg_wait_vm_lgkm0<1>(); break;
        case 2: eg_wait_vm_lgkm0<2>(); break;
        case 3: eg_wait_vm_lgkm0<3>(); break;
        case 4: eg_wait_vm_lgkm0<4>(); break;
        case 5: eg_wait_vm_lgkm0<5>(); break;
        case 6: eg_wait_vm_lgkm0<6>(); break;
        case 7: eg_wait_vm_lgkm0<7>(); break;
        case 8: eg_wait_vm_lgkm0<8>(); break;
        case 9: eg_wait_vm_lgkm0<9>(); break;
        case 10: eg_wait_vm_lgkm0<10>(); break;
        case 11: eg_wait_vm_lgkm0<11>(); break;
        default: eg_wait_vm_lgkm0<12>(); break;
    }
}

// PROF (diagnostic instantiation, EG_NT8_PROF=1): wave 0 of every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz)
// at kernel entry, after the prologue, after the K loop and at the end: where a workgroup's time goes and at which clock it ran.
// STAT: its own instantiation for launches with eg_epilogue.stat_mode set (column statistics of the stored tile in the epilogue): the
// plain kernels keep their register allocation (230 VGPRs, no scratch; with the statistics epilogue inlined behind a branch they spilled)
template <typename T, bool PATCH, bool SPLITK, bool PROF = false, bool STAT = false>
__global__ __launch_bounds__(512) void igemm_nt8s_kernel(const NtParams p, const Nt8pGeom g, int tiles_m, int tiles_n, unsigned long long* prof = nullptr) {
    unsigned long long pt[4] = {0, 0, 0, 0}, pr[4] = {0, 0, 0, 0}, pe[2] = {0, 0};
    unsigned long long lp[3] = {0, 0, 0};              // PROF: K loop split into groups (reads + DMA issue + MFMA) | counted wait | barrier
    if (PROF) { pt[0] = __builtin_amdgcn_s_memtime(); pr[0] = __builtin_amdgcn_s_memrealtime(); }
    constexpr int VEC = Elt<T>::VEC;
    constexpr int BK = 8 * VEC;
    constexpr int BM = 256, BN = 128;
    constexpr int TM = 4, TN = 4;
    constexpr int SLOT = 128 * 128;                      // 16 KiB: 128 K rows
    constexpr int PCAP = EG_P8P_SLOTS * 128;             // one patch buffer
    constexpr int STAGE = PATCH ? SLOT : 3 * SLOT;       // ring stage: B only, or A (2 slots) + B
    constexpr int RING0 = PATCH ? 2 * PCAP : 0;          // byte offset of the ring
    constexpr int BOFF = PATCH ? 0 : 2 * SLOT;           // B inside a stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nsplit = SPLITK && p.nsplit > 1 ? p.nsplit : 1;
    const int phase = blockIdx.z / nsplit, split = blockIdx.z - phase * nsplit;
    const NtPhase ph = p.ph[phase];
    const Nt8pPhase& gp = g.ph[phase];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // workgroup ids go round robin over the 8 XCDs: XCD x runs ids x, x + 8, ...  Every XCD gets a CONTIGUOUS range of logical tiles
    // (bijective for any grid size) with the N tiles of one M tile adjacent: they share the gathered rows through that XCD's L2.
    int L = blockIdx.x;
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

    // this block's K tiles [kt0, kt0 + nk) of the phase's sequence (tap-major, or class by class in PATCH mode: same count)
    const int nk_all = ph.Kpad / BK;
    const int per = (nk_all + nsplit - 1) / nsplit;
    const int kt0 = split * per;
    const int nk = max(0, min(per, nk_all - kt0));

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 fa0[2][4], fb0[2][4], fa1[2][4], fb1[2][4];      // fragment sets [k half][tile]

    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mma4 = [&](uint4 (&ca)[2][4], uint4 (&cb)[2][4], int gidx) {     // group gidx: k half gidx >> 2, row tile gidx & 3, all four column tiles
        const int ks = gidx >> 2, i = gidx & 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) mfma_step<T>(ca[ks][i], cb[ks][j], acc[i][j]);
    };
    auto read_b4 = [&](uint4 (&nb)[2][4], const char* sb, int ks) {
#pragma unroll
        for (int j = 0; j < TN; ++j) nb[ks][j] = *reinterpret_cast<const uint4*>(sb + lds_off(j * 16 + frow, ks * 4 + fq));
    };

    if constexpr (!PATCH) {
        // ---------------------------------------------------------------- im2col rows ----------------------------------------------
        int a_pix0[4], a_y[4], a_x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + (j * 8 + wave) * 8 + rsub;
            const int b = m >> (p.lOW + p.lOH);
            a_pix0[j] = (m < p.M) ? b * p.H * p.W : -1;
            a_y[j] = ((m >> p.lOW) & OHm) * p.sy + ph.dy0;
            a_x[j] = (m & OWm) * p.sx + ph.dx0;
        }
        unsigned va[4];
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
        // wave-uniform walk of the gather over (tap, channel block): three K tiles ahead of the MFMAs
        const int tap0 = kt0 / ncb;
        int ty = tap0 / ph.TW, tx = tap0 - ty * ph.TW;
        unsigned kc_bytes = (unsigned)(kt0 - tap0 * ncb) * 128u;
        unsigned kb_bytes = (unsigned)kt0 * 128u;      // byte offset of the next B K tile in a panel row
        if (ty < ph.TH) tap_offsets(ty, tx);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) va[j] = EG_OOB;
        }
        auto advance_a = [&]() {
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
        // piece q = 0..5 of the K tile being issued into ring stage `st`: four A pieces (rows 8w + 64q), two B pieces
        auto issue_piece = [&](unsigned base, auto qc) {       // base: this wave's first piece address in the stage (SGPR)
            constexpr int q = decltype(qc)::value;
            if constexpr (q < 4) {
                eg_bufdma1f<q * 0x2000>(srdA, va[q], kc_bytes, base);
                if (q == 3) advance_a();
            } else {
                eg_bufdma1f<2 * SLOT + (q - 4) * 0x2000>(srdB, vb[q - 4], kb_bytes, base);
                if (q == 5) kb_bytes += 128u;
            }
        };
        auto read_a4 = [&](uint4 (&na)[2][4], const char* sa, int ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) na[ks][i] = *reinterpret_cast<const uint4*>(sa + lds_off(i * 16 + frow, ks * 4 + fq));
        };
        auto read_all = [&](uint4 (&na)[2][4], uint4 (&nb)[2][4], int st) {
            const char* sa = smem + st * STAGE + wm * (64 * 128);
            const char* sb = smem + st * STAGE + BOFF + wn * (64 * 128);
            read_b4(nb, sb, 0); read_b4(nb, sb, 1); read_a4(na, sa, 0); read_a4(na, sa, 1);
        };
        // one iteration: MFMAs of the current set | reads of K tile t+1 into the other set | DMA of K tile t+3 (DM: the main loop;
        // the last three iterations have nothing left to issue)
        auto iter = [&](uint4 (&ca)[2][4], uint4 (&cb)[2][4], uint4 (&na)[2][4], uint4 (&nb)[2][4], int st, auto dmc, bool dm_rt) {
            constexpr bool DM = decltype(dmc)::value;      // true: main loop, every piece issued; false: tail, pieces guarded by dm_rt
            const int st_r = st == 2 ? 0 : st + 1;     // stage of K tile t+1; K tile t+3 goes into stage st (K tile t's)
            const char* sa = smem + st_r * STAGE + wm * (64 * 128);
            const char* sb = smem + st_r * STAGE + BOFF + wn * (64 * 128);
            const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)st * STAGE);
            unsigned long long q0 = 0, q1 = 0, q2 = 0;
            if (PROF) q0 = __builtin_amdgcn_s_memtime();
            // (reads are unconditional: behind a branch hipcc's wait-count pass makes every MFMA group wait for this iteration's fresh
            // reads; past the last K tile they fetch stale bytes of a valid stage that nobody uses)
            read_b4(nb, sb, 0);
            mma4(ca, cb, 0); __builtin_amdgcn_sched_barrier(0);
            read_b4(nb, sb, 1);
            mma4(ca, cb, 1); __builtin_amdgcn_sched_barrier(0);
            read_a4(na, sa, 0);
            if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 0>{});
            mma4(ca, cb, 2); __builtin_amdgcn_sched_barrier(0);
            read_a4(na, sa, 1);
            if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 1>{});
            mma4(ca, cb, 3); __builtin_amdgcn_sched_barrier(0);
            if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 2>{});
            mma4(ca, cb, 4); __builtin_amdgcn_sched_barrier(0);
            if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 3>{});
            mma4(ca, cb, 5); __builtin_amdgcn_sched_barrier(0);
            if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 4>{});
            mma4(ca, cb, 6); __builtin_amdgcn_sched_barrier(0);
            if (DM || dm_rt) issue_piece(base, std::integral_constant<int, 5>{});
            mma4(ca, cb, 7); __builtin_amdgcn_sched_barrier(0);
            if (PROF) q1 = __builtin_amdgcn_s_memtime();
            if (DM || dm_rt) eg_wait_vm_lgkm0<6>();
            else eg_wait_vm_lgkm0<0>();
            if (PROF) q2 = __builtin_amdgcn_s_memtime();
            barrier();
            if (PROF) { const unsigned long long q3 = __builtin_amdgcn_s_memtime(); lp[0] += q1 - q0; lp[1] += q2 - q1; lp[2] += q3 - q2; }
        };
        // prologue: K tiles 0, 1, 2 in flight; fragments of K tile 0 in registers; K tile 1 visible
        const int npro = min(nk, 3);
        for (int s = 0; s < npro; ++s) {
            const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)s * STAGE);
            issue_piece(base, std::integral_constant<int, 0>{}); issue_piece(base, std::integral_constant<int, 1>{});
            issue_piece(base, std::integral_constant<int, 2>{}); issue_piece(base, std::integral_constant<int, 3>{});
            issue_piece(base, std::integral_constant<int, 4>{}); issue_piece(base, std::integral_constant<int, 5>{});
        }
        eg_wait_vm_dyn_lgkm0(6 * max(npro - 1, 0));
        barrier();
        if (nk > 0) read_all(fa0, fb0, 0);
        eg_wait_vm_dyn_lgkm0(6 * max(npro - 2, 0));
        barrier();
        if (PROF) { pt[1] = __builtin_amdgcn_s_memtime(); pr[1] = __builtin_amdgcn_s_memrealtime(); }
        // main loop in pairs (the two fragment sets swap roles); K tiles t+3 exist for t < nk - 3
        int st = 0, t = 0;
        auto next = [&]() { st = st == 2 ? 0 : st + 1; ++t; };
        for (; t + 4 < nk;) {
            iter(fa0, fb0, fa1, fb1, st, std::true_type{}, true); next();
            iter(fa1, fb1, fa0, fb0, st, std::true_type{}, true); next();
        }
        // tail (t is even): the last iterations, DMA only while K tiles t+3 exist
        for (; t < nk;) {
            iter(fa0, fb0, fa1, fb1, st, std::false_type{}, t + 3 < nk); next();
            if (t < nk) { iter(fa1, fb1, fa0, fb0, st, std::false_type{}, t + 3 < nk); next(); }
        }
    } else {
        // ---------------------------------------------------------------- input patch ----------------------------------------------
        const int b0 = m0 >> (p.lOW + p.lOH);
        const int oyt0 = g.nimg == 1 ? ((m0 >> p.lOW) & OHm) : 0;
        int pb[TM];                                    // patch pixel of the lattice row this lane reads A fragments for, per row tile
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * 64 + i * 16 + frow;
            const int img = (m >> (p.lOW + p.lOH)) - b0, oy = ((m >> p.lOW) & OHm) - oyt0, ox = m & OWm;
            pb[i] = (img * g.PH + oy) * g.PW + ox;
        }
        unsigned vp[7];                                // source offsets of this wave's patch pieces (pieces w, w + 8, ...) for one class
        auto patch_offsets = [&](int cls) {
            const NtClass& c = gp.cls[cls];
            const int coy = c.oy0, cox = c.ox0;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const unsigned ps = (unsigned)(((wave + 8 * q) << 3) + rsub);
                const unsigned img = (ps * g.inv_plane) >> 20;
                const unsigned rem = ps - img * (unsigned)(g.PH * g.PW);
                const unsigned qy = (rem * g.inv_pw) >> 20;
                const unsigned qx = rem - qy * (unsigned)g.PW;
                const int iy = ((int)qy + oyt0) * p.sy + coy, ix = (int)qx * p.sx + cox;
                const bool ok = (int)ps < g.npix && b0 + (int)img < p.B && iy >= 0 && iy < HU && ix >= 0 && ix < WU;
                const unsigned pix = (unsigned)((b0 + (int)img) * p.H * p.W + (iy >> p.up) * p.W + (ix >> p.up));
                vp[q] = ok ? pix * row_bytes + (unsigned)srcchunk * 16u : EG_OOB;
            }
        };
        // K-tile iterator (class, channel block, tap row, tap column) with the class's fields cached
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
        It itr{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // K tile whose fragments are read next
        {
            int rem = kt0;
            while (itr.cls < gp.ncls - 1 && rem >= gp.cls[itr.cls].AH * gp.cls[itr.cls].AW * ncb) {
                rem -= gp.cls[itr.cls].AH * gp.cls[itr.cls].AW * ncb;
                ++itr.cls;
            }
            load_cls(itr);
            const int taps = itr.AH * itr.AW;
            itr.cb = rem / taps;
            rem -= itr.cb * taps;
            itr.ay = rem / itr.AW;
            itr.ax = rem - itr.ay * itr.AW;
        }
        It itb = itr;                                  // K tile whose B pieces are issued next
        auto issue_b = [&](int st, int j) {
            const unsigned tap = (unsigned)((itb.ty0 + itb.ay * itb.tys) * ph.TW + (itb.tx0 + itb.ax * itb.txs));
            eg_bufdma1s(srdB, vb[j], tap * row_bytes + (unsigned)itb.cb * 128u,
                        __builtin_amdgcn_readfirstlane(lds0 + RING0 + (unsigned)st * STAGE + (unsigned)j * 0x2000u));
            if (j == 1) advance(itb);
        };
        // state of the step (class, channel block) the read iterator is in
        int par = 0;                                   // patch buffer it reads
        int left = 0;                                  // K tiles left in the step, the one at `itr` included
        bool have_next = false;                        // a further step starts inside this block's K range ...
        int ncb_n = 0, vp_cls = -1, pp_done = 0;       // ... its channel block; class vp[] describes; pieces of its patch issued so far
        auto enter_step = [&](int kpos) {              // kpos: index of the K tile at itr
            left = (itr.AH - itr.ay) * itr.AW - itr.ax;
            have_next = kpos + left < nk;
            pp_done = 0;
            if (have_next) {
                int ncls_n = itr.cls;
                ncb_n = itr.cb + 1;
                if (ncb_n == ncb) { ncb_n = 0; ++ncls_n; }
                if (ncls_n != vp_cls) { patch_offsets(ncls_n); vp_cls = ncls_n; }
            }
        };
        auto issue_patch_piece = [&](int buf, unsigned soff, int q) {
            // q is wave-uniform: the register array is indexed through the scalar index register (s_set_gpr_idx / v_movrel), not by a
            // chain of seven compares per call
            const int qu = __builtin_amdgcn_readfirstlane(q);
            eg_bufdma1s(srdA, vp[qu], soff, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * PCAP + (unsigned)qu * 0x2000u));
        };
        auto read_a4 = [&](uint4 (&na)[2][4], const char* pbuf, int tapoff, int ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pp = pb[i] + tapoff;
                na[ks][i] = *reinterpret_cast<const uint4*>(pbuf + (pp << 7) + (((ks * 4 + fq) ^ ((pp >> 1) & 7)) << 4));
            }
        };
        auto iter = [&](uint4 (&ca)[2][4], uint4 (&cb)[2][4], uint4 (&na)[2][4], uint4 (&nb)[2][4], int t, int st) {
            const bool rd = t + 1 < nk, dm = t + 3 < nk;
            const int st_r = st == 2 ? 0 : st + 1;
            const char* sb = smem + RING0 + st_r * STAGE + wn * (64 * 128);
            const char* pbuf = smem + par * PCAP;
            const int tapoff = itr.ay * g.PW + itr.ax;
            // next-patch pieces of this iteration: up to four, or all that are left when the step ends with this read
            int n_p = 0;
            if (rd && have_next) n_p = left == 1 ? g.npp - pp_done : min(4, g.npp - pp_done);
            const unsigned soff = (unsigned)ncb_n * 128u;
            unsigned long long q0 = 0, q1 = 0, q2 = 0;
            if (PROF) q0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) {
                if (gi == 0) read_b4(nb, sb, 0);
                if (gi == 1) read_b4(nb, sb, 1);
                if (gi == 2) read_a4(na, pbuf, tapoff, 0);
                if (gi == 3) read_a4(na, pbuf, tapoff, 1);
                if (gi < 6) {                          // patch pieces first (they must be older than this iteration's B pieces)
                    if (gi < n_p) issue_patch_piece(par ^ 1, soff, pp_done + gi);
                    if (gi == 5 && n_p > 6) issue_patch_piece(par ^ 1, soff, pp_done + 6);
                } else if (dm)
                    issue_b(st, gi - 6);
                mma4(ca, cb, gi);
                __builtin_amdgcn_sched_barrier(0);
            }
            pp_done += n_p;
            // everything older than this iteration's own issues has to land (B of K tile t+2); at the end of a step the next patch too
            if (PROF) q1 = __builtin_amdgcn_s_memtime();
            eg_wait_vm_dyn_lgkm0((dm ? 2 : 0) + (left == 1 ? 0 : n_p));
            if (PROF) q2 = __builtin_amdgcn_s_memtime();
            barrier();
            if (PROF) { const unsigned long long q3 = __builtin_amdgcn_s_memtime(); lp[0] += q1 - q0; lp[1] += q2 - q1; lp[2] += q3 - q2; }
            if (rd) {
                advance(itr);
                if (--left == 0) { par ^= 1; enter_step(t + 2); }
            }
        };
        // prologue: the first step's patch (all pieces), the first pieces of the next one (all of them if the step is one K tile long),
        // B of K tiles 0..2; fragments of K tile 0 into registers; K tile 1 visible
        const int npro = min(nk, 3);
        if (nk > 0) {
            patch_offsets(itr.cls);
            vp_cls = itr.cls;
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (q < g.npp) issue_patch_piece(0, (unsigned)itr.cb * 128u, q);
            enter_step(0);
            if (have_next) {
                const int n_p = left == 1 ? g.npp : min(4, g.npp);
                for (int q = 0; q < n_p; ++q) issue_patch_piece(1, (unsigned)ncb_n * 128u, q);
                pp_done = n_p;
            }
        }
        for (int s = 0; s < npro; ++s) { issue_b(s, 0); issue_b(s, 1); }
        eg_wait_vm_dyn_lgkm0(2 * max(npro - 1, 0));
        barrier();
        if (nk > 0) {
            const char* sb = smem + RING0 + wn * (64 * 128);
            const int tapoff = itr.ay * g.PW + itr.ax;
            read_b4(fb0, sb, 0); read_b4(fb0, sb, 1);
            read_a4(fa0, smem, tapoff, 0); read_a4(fa0, smem, tapoff, 1);
            advance(itr);
            if (--left == 0) { par ^= 1; enter_step(1); }
        }
        eg_wait_vm_dyn_lgkm0(2 * max(npro - 2, 0));
        barrier();
        if (PROF) { pt[1] = __builtin_amdgcn_s_memtime(); pr[1] = __builtin_amdgcn_s_memrealtime(); }
        int st = 0;
        for (int t = 0; t < nk; t += 2) {
            iter(fa0, fb0, fa1, fb1, t, st);
            st = st == 2 ? 0 : st + 1;
            if (t + 1 < nk) {
                iter(fa1, fb1, fa0, fb0, t + 1, st);
                st = st == 2 ? 0 : st + 1;
            }
        }
    }

    if (PROF) { pt[2] = __builtin_amdgcn_s_memtime(); pr[2] = __builtin_amdgcn_s_memrealtime(); }
    // ---- epilogue: all DMA has landed, all reads are retired, every wave is past the last barrier: the LDS is free ----
    if (SPLITK && nsplit > 1) {
        // K splits reduce INSIDE the launch: every workgroup stores its partial tile, the one that arrives last at the tile's counter sums
        // the partials in split order (its own from registers: the order does not depend on who is last -> deterministic) and runs the
        // ordinary epilogue.  Cross-workgroup hand-off as measured safe on gfx950 (MI355X_MICROARCH.md, hand-off table, first row): 16-byte
        // sc1 stores, every storing wave waits vmcnt(0), workgroup barrier, ONE lane adds to the tile's counter at agent scope; the
        // workgroup whose add came last loads (sc1, 16 bytes) after its add has returned and a workgroup barrier.  One workgroup per CU.
        const int nphase = gridDim.z / nsplit;
        const size_t slab = (size_t)nphase * ((size_t)tiles_m * BM) * p.N;       // floats per split
        float* part = p.part + ((size_t)phase * ((size_t)tiles_m * BM) + m0) * p.N + n0;
        float* mine = part + (size_t)split * slab;
        if (p.split_cnt == nullptr) {                  // A/B path (EG_NT_SPLIT_INKERNEL=0): plain partial stores, a second launch sums them
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 16 + frow;
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(mine + (size_t)row * p.N + (wn * TN + j) * 16 + fq * 4) = acc[i][j];
            }
            return;
        }
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
        if (tid == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero again for the next launch
        f32x4 sum[TM][TN];
        for (int s = 0; s < nsplit; ++s) {             // uniform
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
        __syncthreads();                               // the flag word is read; the epilogue stages through the same LDS
    }
    if constexpr (STAT) {
        // column statistics of the stored tile (eg_epilogue.stat_mode)
        NtEpiPre<T, TM, TN, 8> epi0;
        nt_epi_prefetch<T, BM, 128, TM, TN, 512, 8>(epi0, p, ph, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
        nt_epilogue_lds_stat<T, BM, 128, TM, TN, 512, 8>(epi0, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq, phase * tiles_m + m_tile,
                                                         n_tile, tiles_n);
        return;
    }
    constexpr int PF = 8;
    NtEpiPre<T, TM, TN, PF> epi;
    nt_epi_prefetch<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
    if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pe[0] = __builtin_amdgcn_s_memtime(); }
    nt_epilogue_lds_pre<T, BM, 128, TM, TN, 512, PF>(epi, p, ph, acc, smem, m0, n0, wm * TM * 16, wn * 64, tid, frow, fq);
    if (PROF) pe[1] = __builtin_amdgcn_s_memtime();
    if (PROF && (tid == 0 || tid == 448)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pt[3] = __builtin_amdgcn_s_memtime(); pr[3] = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = prof + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2 + (tid ? 1 : 0)) * 13;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = pt[i]; o[4 + i] = pr[i]; }
        o[8] = pe[0]; o[9] = pe[1]; o[10] = lp[0]; o[11] = lp[1]; o[12] = lp[2];
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// patch geometry of a launch; false if the problem does not fit the patch buffers (the caller then uses the im2col mode)
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

template <typename T, bool PATCH, bool SPLITK>
static void launch_s(const NtParams& p, const Nt8pGeom& g, int nphase, int ns, hipStream_t st) {
    constexpr size_t lds = PATCH ? 2 * EG_P8P_SLOTS * 128 + 3 * 16384 : 9 * 16384;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt8s_kernel<T, PATCH, SPLITK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int tm = (p.M + 255) / 256, tn = p.N / 128;
    if constexpr (std::is_same<T, bf16_t>::value && !SPLITK) {
        static const char* prof_env = getenv("EG_NT8_PROF");
        if (prof_env) {
            // diagnostic: synchronous instrumented launch, prints medians over workgroups (ticks of the shader clock; 100 MHz real time)
            const size_t nwg = (size_t)tm * tn * nphase;
            unsigned long long* dbuf = nullptr;
            (void)hipMalloc(&dbuf, nwg * 208);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt8s_kernel<T, PATCH, SPLITK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((igemm_nt8s_kernel<T, PATCH, SPLITK, true>), dim3(tm * tn, 1, nphase), dim3(512), lds, st, p, g, tm, tn, dbuf);
            (void)hipStreamSynchronize(st);
            std::vector<unsigned long long> h(nwg * 26);
            (void)hipMemcpy(h.data(), dbuf, nwg * 208, hipMemcpyDeviceToHost);
            (void)hipFree(dbuf);
            std::vector<double> pro, loop, epi, clk, start, endt, e_pf, e_st, e_dr, l_g, l_w, l_b, l7_g, l7_w, l7_b;
            unsigned long long r0 = ~0ull, r1 = 0;
            for (size_t w = 0; w < nwg; ++w) {
                const unsigned long long* o = &h[w * 26];
                const unsigned long long* o7 = &h[w * 26 + 13];
                l7_g.push_back((double)o7[10]); l7_w.push_back((double)o7[11]); l7_b.push_back((double)o7[12]);
                l_g.push_back((double)o[10]); l_w.push_back((double)o[11]); l_b.push_back((double)o[12]);
                e_pf.push_back((double)(o[8] - o[2])); e_st.push_back((double)(o[9] - o[8])); e_dr.push_back((double)(o[3] - o[9]));
                pro.push_back((double)(o[1] - o[0])); loop.push_back((double)(o[2] - o[1])); epi.push_back((double)(o[3] - o[2]));
                clk.push_back((double)(o[3] - o[0]) / (double)(o[7] - o[4]) * 100.0);
                r0 = std::min(r0, o[4]); r1 = std::max(r1, o[7]);
            }
            for (size_t w = 0; w < nwg; ++w) { start.push_back((double)(h[w * 26 + 4] - r0) / 100.0); endt.push_back((double)(h[w * 26 + 7] - r0) / 100.0); }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            auto mx = [](std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
            const int nk = p.ph[0].Kpad / (8 * Elt<T>::VEC);
            fprintf(stderr, "[nt8s_prof] patch %d wgs %zu nk %d | median ticks: prologue %.0f  K loop %.0f (%.0f per K tile: wave 0 groups %.0f, counted wait %.0f, barrier %.0f; wave 7 %.0f / %.0f / %.0f)  epilogue %.0f (operand fetch %.0f, staging + store issue %.0f, store drain %.0f) | clock %.0f MHz | "
                    "first->last wave-0 stamp %.1f us, median WG start %.1f us, median end %.1f us, last end %.1f us\n",
                    (int)PATCH, nwg, nk, med(pro), med(loop), med(loop) / nk, med(l_g) / nk, med(l_w) / nk, med(l_b) / nk, med(l7_g) / nk, med(l7_w) / nk, med(l7_b) / nk, med(epi), med(e_pf), med(e_st), med(e_dr), med(clk), (double)(r1 - r0) / 100.0, med(start), med(endt), mx(endt));
            return;
        }
    }
    if constexpr (!PATCH && !std::is_same<T, float>::value) {
        if (p.stat_mode != EG_STAT_NONE) {
            static bool attr_stat = false;
            if (!attr_stat) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_nt8s_kernel<T, PATCH, SPLITK, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_stat = true;
            }
            hipLaunchKernelGGL((igemm_nt8s_kernel<T, PATCH, SPLITK, false, true>), dim3(tm * tn, 1, nphase * ns), dim3(512), lds, st, p, g, tm, tn, (unsigned long long*)nullptr);
            return;
        }
    }
    hipLaunchKernelGGL((igemm_nt8s_kernel<T, PATCH, SPLITK>), dim3(tm * tn, 1, nphase * ns), dim3(512), lds, st, p, g, tm, tn, (unsigned long long*)nullptr);
}

// patch = true needs a geometry eg_nt8p_geometry() accepted; patch = false ignores g's classes
template <typename T>
void eg_launch_nt8s(const NtParams& p, const Nt8pGeom& g, bool patch, int nphase, int ns, hipStream_t st) {
    NtParams q = p;
    q.nsplit = ns;
    if (patch) {
        if (ns > 1) launch_s<T, true, true>(q, g, nphase, ns, st);
        else launch_s<T, true, false>(q, g, nphase, ns, st);
    } else {
        if (ns > 1) launch_s<T, false, true>(q, g, nphase, ns, st);
        else launch_s<T, false, false>(q, g, nphase, ns, st);
    }
}
template void eg_launch_nt8s<float>(const NtParams&, const Nt8pGeom&, bool, int, int, hipStream_t);
template void eg_launch_nt8s<bf16_t>(const NtParams&, const Nt8pGeom&, bool, int, int, hipStream_t);
template void eg_launch_nt8s<f16_t>(const NtParams&, const Nt8pGeom&, bool, int, int, hipStream_t);
