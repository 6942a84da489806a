// igemm_tn8: weight gradients of the 4x4 / stride-2 / pad-1 convolutions (every big layer of the CelebA and dSprites networks, in
// conv view also the transposed convolutions), 16-bit types:
//
//     S[n][t][c] = sum_m P[m][n] * X[pix(m, t)][c]        P = output gradient [M][N], X = layer input [B,H,W,C], t = filter tap
//
// split over m into fp32 slabs [split][N][16][C] (the layout of igemm_tn_kernel: the slab reductions are shared).
//
// What igemm_tn_kernel does per workgroup and tap -- 32 rows of P and 32 gathered rows of X per barrier through registers, 16 MFMAs
// per wave between barriers, every input pixel fetched once per tap (16 x) and every P row once per tap and channel tile -- is
// replaced by:
//   * one workgroup = 128 output channels x 128 input channels x the FOUR taps of one input-parity class (ty = ry + 2 ay,
//     tx = rx + 2 ax): the four taps of a class read one pixel lattice at offsets (ay, ax), so the 64 lattice rows of a K step need
//     ONE input patch of (rows + 1) x (OW + 1) pixels (81..130 pixels instead of 4 x 64 gathered rows) and ONE 64 x 128 tile of P
//     for 4 x 2 x 64 x 128 x 128 FLOP: 200+ FLOP per staged byte instead of 64;
//   * 8 waves: wave w owns tap w / 2 and output-channel half w % 2 -> 64 x 128 accumulators (32 MFMA tiles), 64 MFMAs per wave and
//     barrier; both operands are K-major in memory, fragments come from `ds_read_b64_tr_b16` (transposed LDS reads);
//   * both operands staged by LDS-DMA (buffer descriptors, 256-byte rows, 32-byte blocks XOR-swizzled on the source side) into a ring
//     of three stages, two K steps in flight behind a counted vmcnt, one barrier per K step.
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "eg_common.h"
#include "igemm_nt.h"

__device__ __forceinline__ int tn8_fsw(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ void tn8_wait(int n) {        // vmcnt(n) lgkmcnt(0) through the builtin (see igemm_nt8s.hip), n in 0..7
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(0 | (7 << 4)); break;
        case 1: __builtin_amdgcn_s_waitcnt(1 | (7 << 4)); break;
        case 2: __builtin_amdgcn_s_waitcnt(2 | (7 << 4)); break;
        case 3: __builtin_amdgcn_s_waitcnt(3 | (7 << 4)); break;
        case 4: __builtin_amdgcn_s_waitcnt(4 | (7 << 4)); break;
        case 5: __builtin_amdgcn_s_waitcnt(5 | (7 << 4)); break;
        case 6: __builtin_amdgcn_s_waitcnt(6 | (7 << 4)); break;
        default: __builtin_amdgcn_s_waitcnt(7 | (7 << 4)); break;
    }
}

// CH: channels per tile on both sides (128: the CelebA layers; 64: the dSprites generators; 32: the first trunk layers of the dSprites networks,
// whose 64 x 32 tile of P is four DMA pieces: waves 0..3 issue one each) = 16-bit elements per LDS row
template <typename T, int CH>
__global__ __launch_bounds__(512) void igemm_tn8_kernel(const Tn8Params p) {
    constexpr int ROWB = CH * 2;                         // bytes per LDS row (a K row of P, a patch pixel of X)
    constexpr int RPP = 1024 / ROWB;                     // rows per 1 KiB DMA piece
    constexpr int CPR = ROWB / 16;                       // 16-byte chunks per row
    constexpr int NBLK = ROWB / 32;                      // 32-byte (16-channel) blocks per row: the swizzle unit
    constexpr int NI = CH / 32, NJ = CH / 16;            // MFMA tiles per wave: output channels (half of CH) x input channels
    constexpr int NPP_P = (64 / RPP + 7) / 8;            // P pieces per wave and K step
    constexpr int PW_P = 64 / RPP < 8 ? 64 / RPP : 8;    // waves that issue P pieces
    constexpr int STAGE_P = 64 * ROWB, STAGE_X = EG_TN8_XSLOTS * ROWB, STAGE = STAGE_P + STAGE_X;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tp = wave >> 1, nh = wave & 1;             // this wave's tap of the class and half of the 128 output channels
    const int ay = tp >> 1, ax = tp & 1;
    const int tn_i = blockIdx.x / p.ntc, tc_i = blockIdx.x - tn_i * p.ntc;
    const int n0 = tn_i * CH, c0 = tc_i * CH;
    const int ry = blockIdx.y >> 1, rx = blockIdx.y & 1;  // parity class: taps (ry + 2 ay, rx + 2 ax), source = 2 * lattice + (r - 1)
    const int mbeg = blockIdx.z * p.rows_per_split;
    const int mend = min(p.M, mbeg + p.rows_per_split);
    const int nk = (mend - mbeg) >> 6;
    const int OWm = (1 << p.lOW) - 1, OHm = (1 << p.lOH) - 1;
    const unsigned row_bytes = (unsigned)p.C * 2u;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pc = li & 3;

    // ---- fragment addresses (loop invariant): K rows 8g + q (+4) of both 32-row MFMA steps, as P rows and as patch pixels ----
    int prow[4], xrow[4];                                // byte offset of the row + its swizzle key in the low bits (rows are 256-byte aligned)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (i >> 1) * 32 + 8 * g + (i & 1) * 4 + q;
        prow[i] = r * ROWB + (tn8_fsw(r) & (NBLK - 1));
        int img, oy;
        if (p.nimg == 1) { img = 0; oy = r >> p.lOW; }
        else { img = r >> (p.lOH + p.lOW); oy = (r >> p.lOW) & OHm; }
        const int px = (img * p.PH + oy + ay) * p.PW + (r & OWm) + ax;
        xrow[i] = px * ROWB + (tn8_fsw(px) & (NBLK - 1));
    }
    auto frag_addr = [&](int rowkey, int blk) { return (rowkey & ~7) + (((blk ^ (rowkey & 7)) << 5) | (pc << 3)); };

    // ---- DMA source offsets ----
    const int c16 = lane % CPR, rsub = lane / CPR;       // a piece = RPP rows x ROWB bytes; lane -> (row, 16-byte chunk)
    const u32x4_t srdP = eg_make_srd(p.P, (unsigned)((size_t)p.M * p.N * 2));
    const u32x4_t srdX = eg_make_srd(p.src, (unsigned)((size_t)p.B * p.H * p.W * p.C * 2));
    unsigned vP[NPP_P];
#pragma unroll
    for (int j = 0; j < NPP_P; ++j) {
        const int r = RPP * (wave + 8 * j) + rsub;
        const int src16 = (((c16 >> 1) ^ (tn8_fsw(r) & (NBLK - 1))) << 1) | (c16 & 1);
        vP[j] = (unsigned)r * (unsigned)p.N * 2u + (unsigned)n0 * 2u + (unsigned)src16 * 16u;
    }
    // patch pieces w, w + 8, ...: lane part of the source offset, source row relative to the step's first source row, x validity
    int xa[5], xdy[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const unsigned ps = (unsigned)(RPP * (wave + 8 * j) + rsub);
        const unsigned img = (ps * p.inv_plane) >> 20;
        const unsigned rem = ps - img * (unsigned)(p.PH * p.PW);
        const unsigned qy = (rem * p.inv_pw) >> 20, qx = rem - qy * (unsigned)p.PW;
        const int ix = (int)qx * 2 + rx - 1;
        const int src16 = (((c16 >> 1) ^ (tn8_fsw((int)ps) & (NBLK - 1))) << 1) | (c16 & 1);
        const bool ok = (int)ps < p.npix && ix >= 0 && ix < p.W;
        xdy[j] = (int)qy * 2 + ry - 1;                  // source row = 2 * (first lattice row of the step) + xdy
        xa[j] = ok ? (int)(((img * (unsigned)(p.H * p.W) + (unsigned)ix) * row_bytes) + (unsigned)c0 * 2u + (unsigned)src16 * 16u) : -1;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 1024u;

    auto issue = [&](int s, int stage) {                // K step s of this block into ring stage `stage`
        const int m0s = mbeg + (s << 6);
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage * STAGE);
        const unsigned soffP = (unsigned)m0s * (unsigned)p.N * 2u;
        if (PW_P == 8 || wave < PW_P) eg_bufdma1f<0>(srdP, vP[0], soffP, base);
        if constexpr (NPP_P > 1) eg_bufdma1f<0x2000>(srdP, vP[NPP_P - 1], soffP, base);
        const int b_s = m0s >> (p.lOH + p.lOW);
        const int oy_s = p.nimg == 1 ? ((m0s >> p.lOW) & OHm) : 0;
        const int ybase = oy_s * 2;
        const int pixbase = (b_s * p.H + ybase) * p.W;   // source pixel of (image b_s, row 2 * oy_s, column 0)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j < p.npp && (wave + 8 * j) * RPP < p.npix) {       // (pieces past the patch would land in the next stage)
                const int iy = ybase + xdy[j];
                const bool ok = xa[j] >= 0 && iy >= 0 && iy < p.H;
                const unsigned v = ok ? (unsigned)(xa[j] + (pixbase + xdy[j] * p.W) * (int)row_bytes) : EG_OOB;
                if (j == 0) eg_bufdma1f<STAGE_P>(srdX, v, 0u, base);
                if (j == 1) eg_bufdma1f<STAGE_P + 0x2000>(srdX, v, 0u, base);
                if (j == 2) eg_bufdma1f<STAGE_P + 0x4000>(srdX, v, 0u, base);
                if (j == 3) eg_bufdma1f<STAGE_P + 0x6000>(srdX, v, 0u, base);
                if (j == 4) eg_bufdma1f<STAGE_P + 0x8000>(srdX, v, 0u, base);
            }
        }
    };

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    auto tr2 = [&](const char* base, int a_lo, int a_hi) {          // 8 K-consecutive 16-bit elements of one column: two transposed reads
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + a_lo));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + a_hi));
        return make_uint4(((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                          ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16));
    };
    auto compute = [&](int stage) {
        const char* sp = smem + stage * STAGE;
        const char* sx = sp + STAGE_P;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 af[NI], bfr[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = tr2(sp, frag_addr(prow[2 * kb], nh * NI + i), frag_addr(prow[2 * kb + 1], nh * NI + i));
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = tr2(sx, frag_addr(xrow[2 * kb], j), frag_addr(xrow[2 * kb + 1], j));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (std::is_same<T, f16_t>::value)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, af[i]), __builtin_bit_cast(f16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                }
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ring: K step s lives in stage s % 3; steps s + 1 and s + 2 are in flight while step s is multiplied
    int per = (PW_P == 8 || wave < PW_P) ? NPP_P : 0;    // pieces this wave issues per K step
    for (int j = 0; j < p.npp; ++j) per += (wave + 8 * j) * RPP < p.npix;
    if (nk > 0) issue(0, 0);
    if (nk > 1) issue(1, 1);
    tn8_wait(nk > 1 ? per : 0);
    barrier();
    int st = 0;
    for (int t = 0; t < nk; ++t) {
        const int st2 = st == 0 ? 2 : st - 1;            // (t + 2) % 3: last read in iteration t - 1, retired before its barrier
        const bool more = t + 2 < nk;
        if (more) issue(t + 2, st2);
        compute(st);
        // everything but this iteration's own pieces has landed (K step t + 1); this wave's fragment reads of stage st are retired
        tn8_wait(more ? per : 0);
        barrier();
        st = st == 2 ? 0 : st + 1;
    }

    // ---- slab[split][n][tap][c] ----
    const int tap = (ry + 2 * ay) * 4 + (rx + 2 * ax);
    float* slab = p.slab + (size_t)blockIdx.z * p.N * 16 * p.C;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + (nh * NI + i) * 16 + g * 4 + r;
            float* row = slab + ((size_t)n * 16 + tap) * p.C + c0 + li;
#pragma unroll
            for (int j = 0; j < NJ; ++j) row[j * 16] = acc[i][j][r];
        }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// geometry + split plan; false if this convolution is not a 16-bit 4x4 / stride-2 / pad-1 layer of the supported sizes
bool eg_tn8_plan(const eg_conv* c, int dtype, Tn8Params& p, int* nsplit, int wgs_target) {
    static const bool enabled = [] { const char* e = getenv("EG_TN8"); return !(e && atoi(e) == 0); }();
    if (!enabled || dtype == EG_F32 || c->k != 4 || c->stride != 2 || c->pad != 1 || c->up != 0) return false;
    static const bool ch32 = [] { const char* e = getenv("EG_TN8_CH32"); return !(e && atoi(e) == 0); }();
    const int ch = ((c->Cin % 128) == 0 && (c->Cout % 128) == 0) ? 128 : (((c->Cin % 64) == 0 && (c->Cout % 64) == 0) ? 64
                   : ((ch32 && (c->Cin % 32) == 0 && (c->Cout % 32) == 0) ? 32 : 0));
    if (ch == 0 || (c->H & 1) || (c->W & 1)) return false;
    const int OH = c->H / 2, OW = c->W / 2;
    const int lOH = ilog2_exact(OH), lOW = ilog2_exact(OW);
    const long long M = (long long)c->B * OH * OW;
    if (lOH < 0 || lOW < 0 || OW > 64 || (M % 64) != 0) return false;
    if ((size_t)c->B * c->H * c->W * c->Cin * 2 >= 0x7fffffffull || (size_t)M * c->Cout * 2 >= 0x7fffffffull) return false;
    memset(&p, 0, sizeof(p));
    p.B = c->B; p.H = c->H; p.W = c->W; p.C = c->Cin; p.N = c->Cout;
    p.lOH = lOH; p.lOW = lOW; p.M = (int)M;
    if (OH * OW >= 64) { p.nimg = 1; p.OHt = 64 / OW; }
    else { p.nimg = 64 / (OH * OW); p.OHt = OH; }
    p.PH = p.OHt + 1; p.PW = OW + 1;
    p.npix = p.nimg * p.PH * p.PW;
    p.ch = ch;
    const int rpp = 1024 / (ch * 2);                    // patch pixels per DMA piece
    if ((p.npix + rpp - 1) / rpp * rpp > EG_TN8_XSLOTS) return false;      // (whole pieces land in the stage)
    p.npp = ((p.npix + rpp - 1) / rpp + 7) / 8;
    if (p.npp > 5) return false;
    p.inv_pw = (1u << 20) / (unsigned)p.PW + 1;
    p.inv_plane = (1u << 20) / (unsigned)(p.PH * p.PW) + 1;
    for (unsigned x = 0; x < 512; ++x)
        if (((x * p.inv_pw) >> 20) != x / (unsigned)p.PW || ((x * p.inv_plane) >> 20) != x / (unsigned)(p.PH * p.PW)) return false;
    p.ntn = c->Cout / ch; p.ntc = c->Cin / ch;
    // one workgroup per CU (150 KiB of LDS): split m until about 256 workgroups exist, at least 4 K steps each
    const long long base = (long long)p.ntn * p.ntc * 4;
    // (wgs_target: the caller's share of the chip -- a launch forked beside the main chain's GEMMs runs the step fastest at 128: half the
    //  slab bytes to write and reduce, and the other CUs stay with the main chain; profiles/r02_i_ab_tn8_target.txt)
    static const int env_target = [] { const char* e = getenv("EG_TN8_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    const int target = env_target ? env_target : (wgs_target > 0 ? wgs_target : 256);
    long long want = base >= target ? 1 : (target + base - 1) / base;
    const long long steps = M / 64;
    want = std::min(want, std::max(1LL, steps / 4));
    const long long sps = (steps + want - 1) / want;    // K steps per split
    p.rows_per_split = (int)(sps * 64);
    *nsplit = (int)((steps + sps - 1) / sps);
    return true;
}

template <typename T, int CH>
static void launch_tn8_ch(const Tn8Params& p, int nsplit, hipStream_t st) {
    constexpr size_t lds = 3 * (size_t)(64 + EG_TN8_XSLOTS) * CH * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_tn8_kernel<T, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_tn8_kernel<T, CH>), dim3(p.ntn * p.ntc, 4, nsplit), dim3(512), lds, st, p);
}

template <typename T>
void eg_launch_tn8(const Tn8Params& p, int nsplit, hipStream_t st) {
    if (p.ch == 32) launch_tn8_ch<T, 32>(p, nsplit, st);
    else if (p.ch == 64) launch_tn8_ch<T, 64>(p, nsplit, st);
    else launch_tn8_ch<T, 128>(p, nsplit, st);
}
template void eg_launch_tn8<bf16_t>(const Tn8Params&, int, hipStream_t);
template void eg_launch_tn8<f16_t>(const Tn8Params&, int, hipStream_t);
