/* libeadgan_hip.so -- C ABI of the MI355X-native EAD-GAN training hot path (gfx950 / CDNA4).
 *
 * The reference (letao1991/EAD-GAN) is pure PyTorch: it has no FFI of its own.  Its hot path calls the
 * torch operators listed per entry point below (file:line relative to the reference root); each entry
 * point here is the hand-written HIP replacement that the Python host layer (ead-gan_amd/) binds with
 * ctypes.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (tensor.data_ptr()); nothing is allocated,
 *     freed or retained by the library; workspaces are passed in explicitly;
 *   - every call is an asynchronous enqueue on `stream` (hipStream_t passed as void*), no device sync,
 *     graph-capturable;
 *   - return value: 0 ok, <0 argument error, >0 hipError_t; eg_last_error() gives the text;
 *   - activations are NHWC ("pixel-major") in dtype T (EG_F32, EG_BF16 or EG_F16); images at the module boundary
 *     are NCHW fp32 like the reference's tensors; master weights / gradients / optimizer state are fp32
 *     in the reference's own layouts ([Cout][Cin][kh][kw] for Conv2d, [Cin][Cout][kh][kw] for
 *     ConvTranspose2d);
 *   - "conv view": a ConvTranspose2d(Ci->Co,k,s,p) layer is described as the Conv2d(Co->Ci,k,s,p) whose
 *     backward-data it is; its forward is eg_conv_bwd_data, its input gradient eg_conv_fwd.
 */
#ifndef EADGAN_HIP_H
#define EADGAN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* eg_stream_t;

enum { EG_F32 = 0, EG_BF16 = 1, EG_F16 = 2 };   /* compute / activation dtype T; masters, statistics, losses and Adam are always fp32 */
enum { EG_ACT_NONE = 0, EG_ACT_LRELU = 1, EG_ACT_RELU = 2, EG_ACT_TANH = 3, EG_ACT_SIGMOID = 4 };
enum { EG_OUT_NHWC = 0, EG_OUT_NCHW_F32 = 1 };

int eg_version(void);
const char* eg_last_error(void);
/* drain HIP's sticky last-error slot after a failed hipGraph capture (returns the number of pending errors) */
int eg_clear_errors(void);

/* One square 2-D convolution in conv view.  X is [B,H,W,Cin] (NHWC); if up==1 the conv reads the
 * nearest-2x-upsampled X (nn.Upsample fused, MNIST/EAD-GAN_rpqmnxy.py:81,85).  Y is [B,OH,OW,Cout],
 * OH = ((H<<up) + 2*pad - k)/stride + 1.  All spatial extents must be powers of two. */
typedef struct eg_conv {
    int B, H, W, Cin, Cout, k, stride, pad, up;
} eg_conv;

/* Fused epilogue of the implicit-GEMM kernels:  v = acc [/ sigma[tape]] [+ bias[n % bias_mod]] ;
 * v = act(v) ; [v *= act'(mask)] ; store.  `mask` holds the activation OUTPUT of the layer whose
 * gradient is being formed (same dtype and layout as dst), so LeakyReLU/ReLU backward never needs a
 * separate pass.  Several forwards of one network ("tapes", each with its own spectral-norm sigma) can be
 * batched along M: tape = lattice_row / sigma_rows (sigma_rows == 0: one sigma for the whole launch). */
typedef struct eg_epilogue {
    const float* bias;
    int bias_mod;
    const float* sigma;
    int act;
    float slope;
    const void* mask;
    int mask_act;
    float mask_slope;
    int out_mode;
    int sigma_rows;
    void* splitk_ws;          /* optional caller-owned scratch: lets small-M / deep-K launches split K across workgroups (fp32 partial */
    size_t splitk_ws_bytes;   /* tiles, summed in split order -- inside the launch by the last-arriving workgroup, or by a second launch); */
                              /* NULL / 0 = never split.  ZERO it once before the first use: its last 4 KiB hold arrival counters that */
                              /* every launch leaves at zero.  One scratch must not be lent to two launches that may run concurrently. */
    int nt_variant;           /* EG_NT_AUTO (0): the planner picks the kernel; otherwise force one (the call fails if it cannot run the problem) */
    int nt_splitk;            /* 0: the planner picks the K split; n >= 1: at most n splits (1 = never) */
    /* Column statistics of the stored tile, fused into the epilogue (EG_STAT_*; 0 = none): the reductions that follow a convolution in the
     * reference's graph -- BatchNorm batch statistics (celebA/EAD-GAN_celebA.py:79,83,87), the two sums of the BatchNorm backward, the bias
     * gradient + spectral-norm coefficient of a spectrally normalised layer (:110-122) -- taken from the tile while it is in LDS instead of
     * by a kernel that re-reads the whole tensor.  Only launches that eg_conv_stat_blocks() answers > 0 for may set it (the 8-wave kernel
     * on whole 256-row tiles); such a launch fails otherwise.  Per row block rb = phase * tiles_m + m_tile (256 lattice rows) and column n:
     *   stat_out[(0 * N + n) * nrb + rb], stat_out[(1 * N + n) * nrb + rb]     (nrb = eg_conv_stat_blocks())
     * deterministic (fixed order inside a tile; the consumers eg_bn_fwd_train_fused / eg_bn_bwd_fused / eg_bias_grad_sn_fused sum the row
     * blocks in a fixed order).
     *   EG_STAT_MOMENTS : (mean, M2) of the 256 stored values of the column
     *   EG_STAT_BN_BWD  : the launch computes da = the gradient w.r.t. a BatchNorm layer's activation output.  stat_aux = z (the BatchNorm
     *                     input, layout of dst), stat_p0..p3 = save_mean, save_invstd, gamma, beta, stat_act / stat_slope = the activation
     *                     behind the BatchNorm.  STORES dy = da * act'(bn(z)) instead of da; sums (sum dy, sum dy * xhat)
     *   EG_STAT_SN_BIAS : the launch computes dzs = dz / sigma of a spectrally normalised LeakyReLU layer (mask = its activation output a,
     *                     mask_act = EG_ACT_LRELU); stat_p0 = the layer's bias.  Sums (sum dzs, sum dzs * (lrelu^-1(a) - bias)); the second
     *                     sum is further reduced over the tile's 128 columns: stat_out[N * nrb + rb * (N / 128) + n_tile] */
    int stat_mode;
    float* stat_out;
    const void* stat_aux;
    const float* stat_p0;
    const float* stat_p1;
    const float* stat_p2;
    const float* stat_p3;
    int stat_act;
    float stat_slope;
} eg_epilogue;
enum { EG_STAT_NONE = 0, EG_STAT_MOMENTS = 1, EG_STAT_BN_BWD = 2, EG_STAT_SN_BIAS = 3 };

/* kernels behind eg_conv_fwd / eg_conv_bwd_data.  The choice is a pure function of the problem and of these two per-call fields:
 * the library holds no tuning state.  All variants accumulate K in the same order: without a K split they are bit-identical. */
#define EG_NT_AUTO 0
#define EG_NT_REG 1      /* register-staged 128 x {16,32,64,128} tiles: every problem; the reference of the others */
#define EG_NT_BUF128 2   /* 128 x 128, 4 waves (two workgroups per CU), 2-stage buffer-descriptor LDS-DMA ring */
#define EG_NT_PERS 3     /* persistent 128 x 128 pipeline (1-2-step image-side layers) */
#define EG_NT_S8 4       /* 256 x 128, 8 waves, 3-K-tile LDS-DMA ring, one barrier per K tile, fragments double-buffered in registers */
#define EG_NT_S8H 6      /* the 8-wave design on 128 x 128 tiles (waves 4 x 2, wave tile 32 x 64), ring of four K tiles: launches with too few 256-row tiles */
#define EG_NT_S8P 5      /* the same with the A operand held in LDS as an input patch shared by the filter taps of a class: 2.6-3.5x
                          * fewer A bytes through the LDS-DMA path; K is accumulated class by class (deterministic, not bit-identical
                          * to the tap-major variants) */

/* --- implicit-GEMM convolution family (MFMA) ---------------------------------------------------
 * replaces torch.nn.functional.conv2d / conv_transpose2d / linear and their autograd backward:
 *   celebA/EAD-GAN_celebA.py:76-90,110-122   dSprites/rp.py:95-110,129-146
 *   MNIST/EAD-GAN_rpqmnxy.py:77-90,107-124,143-163 */
size_t eg_pack_fwd_elems(const eg_conv* c, int dtype);                 /* elements of Wp_fwd */
size_t eg_pack_bwd_elems(const eg_conv* c, int dtype);                 /* elements of Wp_bwd */
int eg_pack_fwd(const eg_conv* c, int dtype, const float* w_master, void* wp, eg_stream_t s);
int eg_pack_bwd(const eg_conv* c, int dtype, const float* w_master, void* wp, eg_stream_t s);
/* both panels in one pass over the master (either destination may be NULL); same bytes as eg_pack_fwd + eg_pack_bwd */
int eg_pack_conv(const eg_conv* c, int dtype, const float* w_master, void* wp_fwd, void* wp_bwd, eg_stream_t s);
/* Y = conv(X, W) */
int eg_conv_fwd(const eg_conv* c, int dtype, const void* X, const void* wp_fwd, void* Y,
                const eg_epilogue* ep, eg_stream_t s);
/* dX = conv^T(dY, W)  (== ConvTranspose2d forward); dX has spatial dims (H<<up, W<<up) */
int eg_conv_bwd_data(const eg_conv* c, int dtype, const void* dY, const void* wp_bwd, void* dX,
                     const eg_epilogue* ep, eg_stream_t s);
/* which kernel eg_conv_fwd (bwd = 0) / eg_conv_bwd_data (bwd = 1) runs this problem on under the given hints (same planner as the
 * launches, unlimited split-K scratch): BM * 1000 + code; code = BN of the register-staged kernels, 131 / 132 = 128 x 128
 * buffer-descriptor kernel (plain / split-K), 135 = persistent pipeline, 147 / 148 = igemm_nt8s (plain / split-K), 149 / 150 =
 * igemm_nt8s with the input patch, 151 / 152 = igemm_nt8h (plain / split-K); -1 = the forced variant cannot run the problem.  Profiling labels and tests. */
int eg_igemm_nt_tile(const eg_conv* c, int dtype, int bwd, int variant, int splitk);
/* the same for ONE concrete call: the epilogue's kernel hints and ITS split-K scratch (what the launch itself will do) */
int eg_igemm_nt_tile_ep(const eg_conv* c, int dtype, int bwd, const eg_epilogue* ep);
/* row blocks (nrb) of the column statistics that THIS call -- eg_conv_fwd (bwd = 0) / eg_conv_bwd_data (bwd = 1) with this epilogue, its
 * kernel hints and its split-K scratch -- would write if stat_mode were set; 0 = the launch cannot fuse them (another kernel than the
 * 8-wave one, ragged row tiles, splits reduced by a second launch): the caller then runs the stand-alone reduction kernels. */
int eg_conv_stat_blocks(const eg_conv* c, int dtype, int bwd, const eg_epilogue* ep);
/* bytes of eg_epilogue.splitk_ws that let eg_conv_fwd (bwd = 0) / eg_conv_bwd_data (bwd = 1) split as far as the policy wants */
size_t eg_conv_splitk_ws_bytes(const eg_conv* c, int dtype, int bwd);
/* dW partial slabs: slab[split][Cout][k*k][Cin] fp32.  Returns the split count through *nsplit.  16-bit 4x4 / stride-2 / pad-1 layers
 * whose channel counts are multiples of 128 run on the parity-class kernel (igemm_tn8.hip: the four taps of an input-parity class share
 * one input patch and one tile of dY per K step), everything else on the per-tap kernel; eg_conv_wgrad_variant tells which (2 / 1). */
int eg_conv_wgrad_variant(const eg_conv* c, int dtype);
size_t eg_conv_wgrad_ws_bytes(const eg_conv* c, int dtype);
int eg_conv_wgrad(const eg_conv* c, int dtype, const void* X, const void* dY, float* slab, int* nsplit,
                  eg_stream_t s);
/* the same with the caller's share of the chip: wgs_target = workgroups the parity-class kernel should aim for (0 = one per CU).  A
 * launch forked onto a side stream beside the main chain's GEMMs passes 128 (half the slab to write and reduce, the other CUs stay
 * with the main chain); never more splits than eg_conv_wgrad_ws_bytes provides for. */
int eg_conv_wgrad_target(const eg_conv* c, int dtype, const void* X, const void* dY, float* slab, int* nsplit,
                         int wgs_target, eg_stream_t s);
/* grad[n][c][t] (+)= sum_split slab[split][n][t][c]  for n < n_rows (slab rows: n_slab >= n_rows);
 * C = gathered channels, T = taps.  Master layouts [Cout][Cin][k][k] / [Cin_T][Cout_T][k][k] are both [n][c][t]. */
int eg_wgrad_reduce(const float* slab, int nsplit, int n_slab, int n_rows, int C, int T, float* grad,
                    int accumulate, eg_stream_t s);
/* same with a row permutation: slab row n lands in gradient row (n % row_div) * row_mul + n / row_div
 * (Linear whose output is viewed [B,C,H,W] and kept NHWC on device: MNIST/EAD-GAN_rpqmnxy.py:77,95-96) */
int eg_wgrad_reduce_perm(const float* slab, int nsplit, int n_slab, int n_rows, int C, int T, float* grad,
                         int row_div, int row_mul, int c_row /* destination row length in channels, 0 = C */, eg_stream_t s);
/* out[i] += src[(i / div) * s_div + (i % div) * s_mod]   (un-permute a bias gradient) */
int eg_gather_add(float* out, const float* src, int n, int div, int s_div, int s_mod, eg_stream_t s);
/* y[B,H,W,C] = 2x2 sum-pool of x[B,2H,2W,C]  (backward of nn.Upsample(scale_factor=2), MNIST/EAD-GAN_rpqmnxy.py:81,85) */
int eg_sumpool2x2(int dtype, const void* x, void* y, int B, int H, int W, int C, eg_stream_t s);
/* weight gradient of Conv2d(64, 1, 3, 1, 1) -- the MNIST generator's last layer (MNIST/EAD-GAN_rpqmnxy.py:88) -- with the activation read ONCE
 * (lane = input channel, nine per-lane accumulators, dy [B][H][W] fp32 broadcast from LDS after a pass through the compute dtype): x
 * [B][H][W][64] dtype T; writes slab[split][9][64] for eg_wgrad_reduce(slab, *nsplit_out, 1, 1, 64, 9, grad); the slab must hold
 * eg_wgrad_c1_splits(B, H) * 576 floats.  Replaces eg_cast_pad + the per-tap eg_conv_wgrad over the output padded to 8 channels. */
int eg_wgrad_c1_ok(int dtype, int C, int H, int W, int Cout, int k, int stride, int pad);
int eg_wgrad_c1_splits(int B, int H);
int eg_wgrad_c1(int dtype, const void* x, const float* dy, float* slab, int B, int H, int W, int C, int* nsplit_out, eg_stream_t s);
/* nn.Upsample(scale_factor=2) + Conv2d(Cin -> Cout, 3, 1, 1) run as ConvTranspose2d(Cin -> Cout, 4, 2, 1) with summed taps
 * (MNIST/EAD-GAN_rpqmnxy.py:81-82, 85-86): eg_up3_expand writes the effective master w4t[Cin][Cout][4][4] (W4[kh] = sum of the 3x3 taps
 * t in [max(0, 2-kh), min(2, 3-kh)], rows and columns alike) from w3[Cout][Cin][3][3]; eg_up3_contract takes the weight gradient of the
 * transposed convolution back through the transpose of that map: dw3 (+)= E^T dw4t E.  fp32 masters / gradients only. */
int eg_up3_expand(const float* w3, float* w4t, int Cout, int Cin, eg_stream_t s);
int eg_up3_contract(const float* dw4t, float* dw3, int Cout, int Cin, int accumulate, eg_stream_t s);
/* spectral-norm variant (torch.nn.utils.spectral_norm backward, celebA/EAD-GAN_celebA.py:110-120):
 * G = sum slab ;  grad += G/sigma - (<G,W_orig>/sigma^2) u v^T.  gtmp: Cout*Cin*k*k floats,
 * partials: >= eg_sn_partials() floats. */
int eg_sn_partials(void);
int eg_wgrad_reduce_sn(const eg_conv* c, const float* slab, int nsplit, const float* w_orig,
                       const float* sigma, const float* u, const float* v, float* gtmp, float* partials,
                       float* grad, eg_stream_t s);
/* per-output-channel bias gradient: gb[n % bias_mod] += sum_rows dY[row][n]  (dY is [rows][N] dtype T) */
size_t eg_bias_grad_ws_floats(int rows, int N);
int eg_bias_grad(int dtype, const void* dY, int rows, int N, int bias_mod, float* partials, float* gb,
                 eg_stream_t s);
/* Batched-tape form for spectrally normalised layers.  dzs holds dL/dz ALREADY divided by the tape's sigma
 * (dzs = dz / sigma[tape]); a = activation output (LeakyReLU), bias = the layer's bias.  Computes
 *   gb[n]   += sum_tape sigma[tape] * sum_{rows of tape} dzs[row][n]
 *   coef[t]  = sum_{rows of tape t, n} dzs[row][n] * (lrelu^-1(a[row][n]) - bias[n])      (== <G_t, W_orig> / sigma_t^2)
 * ws: eg_bias_grad_sn_ws_floats(rows, N, rows_per_tape) floats. */
size_t eg_bias_grad_sn_ws_floats(int rows, int N, int rows_per_tape);
int eg_bias_grad_sn(int dtype, const void* dzs, const void* a, const float* bias, int rows, int N, int rows_per_tape,
                    const float* sigma, float slope, float* ws, float* gb, float* coef, eg_stream_t s);
/* eg_bias_grad_sn with the per-tile sums taken from the epilogue of the convolution that produced dzs (EG_STAT_SN_BIAS): stat = its
 * stat_out (nrb = nphase * tiles_m row blocks of 256 lattice rows; tape of a row block = (rb % tiles_m) / tiles_per_tape) */
int eg_bias_grad_sn_fused(const float* stat, int nrb, int N, int tiles_m, int tiles_per_tape, int ntapes, const float* sigma, float* gb,
                          float* coef, eg_stream_t s);
/* grad[n][c][t] += sum_split slab[split][n][t][c] - sum_tape coef[tape] * u[tape][n] * v[tape][c*T + t]
 * (u: [ntapes][n_rows], v: [ntapes][c_row*T]); single pass, deterministic.  c_row: destination row length in channels
 * (0 = C; < C when the gathered operand was zero padded, e.g. 9 of 16 im2col columns). */
int eg_wgrad_reduce_rank1(const float* slab, int nsplit, int n_slab, int n_rows, int C, int T, float* grad,
                          int ntapes, const float* coef, const float* u, const float* v, int c_row, eg_stream_t s);
/* wp[n][k] = w[(n / n_div) * s_hi + (n % n_div) * s_lo + k * s_k], zero for K <= k < Kpad
 * (ConvTranspose2d on a 1x1 input as a GEMM: celebA/EAD-GAN_celebA.py:76; view-permuted Linear outputs) */
int eg_pack_strided(int dtype, const float* w, void* wp, int N, int K, int Kpad, int n_div, long long s_hi,
                    long long s_lo, long long s_k, eg_stream_t s);

/* wp[n][k] = w[(n / n_div) * s_hi + (n % n_div) * s_lo + (k / k_div) * s_khi + (k % k_div) * s_klo]  (row AND column permuted) */
int eg_pack_strided2(int dtype, const float* w, void* wp, int N, int K, int Kpad, int n_div, long long s_hi, long long s_lo,
                     int k_div, long long s_khi, long long s_klo, eg_stream_t s);

/* --- image-side (1..4 channel, NCHW fp32) convolution and its weight gradient ---------------------
 * first Discriminator/Encoder conv (celebA/EAD-GAN_celebA.py:110, dSprites/rp.py:95, MNIST/EAD-GAN_rpqmnxy.py:107)
 * and input gradient / weight gradient of the Generator's last ConvTranspose2d (celebA/EAD-GAN_celebA.py:90) */
int eg_conv_img_fwd(int dtype, const float* img, const float* w_master, void* out, int B, int CI, int H, int W,
                    int N, int k, int stride, int pad, const eg_epilogue* ep, eg_stream_t s);
size_t eg_conv_img_wgrad_ws_bytes(int B, int CI, int N, int k);
int eg_conv_img_wgrad(int dtype, const void* dz, const float* img, float* slab, int B, int CI, int H, int W,
                      int N, int k, int stride, int pad, eg_stream_t s);
/* patch rows [B*OH*OW][Kp] (dtype T, K = CI*k*k in master weight order, zero padded to Kp): lets the image-side layers
 * run on the MFMA kernels as 1x1 convolutions over Kp channels */
int eg_im2col_img(int dtype, const float* img, void* out, int B, int CI, int H, int W, int k, int stride, int pad, int Kp,
                  eg_stream_t s);
/* the same convolution WITHOUT patch rows in HBM, 16-bit types: Conv2d(C <= 4 -> 128, 4, 2, 1) of up to three fp32 NCHW image tensors
 * ("tapes": independent forwards batched into one launch, each with its own spectral-norm sigma -- ep->sigma[tape]) straight on the MFMA
 * units; out [ntapes*B][H/2][W/2][128] dtype T.  wp: the [128][64] panel of eg_pack_strided (K order = master weight order).  gate_t != NULL:
 * the convolution's input is img_t * act'(gate_t) (the input gradient of the Generator's last ConvTranspose2d + Tanh, celebA.py:90-91: img =
 * d(loss)/d(image), gate = the image).  ep: bias, sigma, act / slope only.  Bit-identical to eg_im2col_img + eg_conv_fwd on the patch rows. */
int eg_conv_img_mfma_ok(int dtype, int C, int H, int W, int N, int k, int stride, int pad);
/* ep->stat_mode == EG_STAT_BN_BWD is honoured too (the output feeds a BatchNorm backward: dy stored, the two sums per tile of 64 pixels
 * to ep->stat_out[(which * 128 + n) * nrb + tile]); nrb = eg_conv_img_mfma_stat_blocks() */
int eg_conv_img_mfma_stat_blocks(int B, int H, int W, int ntapes);
int eg_conv_img_mfma(int dtype, const float* img0, const float* img1, const float* img2, const float* gate0, const float* gate1,
                     const float* gate2, int ntapes, const void* wp, void* out, int B, int C, int H, int W, const eg_epilogue* ep,
                     int gate_act, float gate_slope, eg_stream_t s);
/* ... with N = 32, 64 or 128 output channels (wp: the [N][64] panel; out [ntapes*B][H/2][W/2][N]): the first trunk layer of the dSprites
 * networks, Conv2d(C -> 32, 4, 2, 1) (dSprites/rp.py:95-97, 165-167; rp_color.py likewise).  EG_STAT_BN_BWD: N = 128 only. */
int eg_conv_img_mfma_n(int dtype, const float* img0, const float* img1, const float* img2, const float* gate0, const float* gate1,
                       const float* gate2, int ntapes, const void* wp, void* out, int B, int C, int H, int W, int N, const eg_epilogue* ep,
                       int gate_act, float gate_slope, eg_stream_t s);
/* weight gradient of the image-side 4x4 / stride-2 / pad-1 layers of the dSprites networks WITHOUT patch rows in HBM (16-bit types, 64 x 64
 * images, C <= 4): S[n][c*16 + ky*4 + kx] = sum over output pixels of P[pixel][n] * img[b][c][2 oy - 1 + ky][2 ox - 1 + kx] for up to three
 * tapes (img_t fp32 NCHW; P [ntapes*B*1024][N] dtype T, tape-major rows) -- Conv2d(C -> 32, 4, 2, 1) of the trunks (dSprites/rp.py:95-97,
 * 165-167; N = 32, P = d(loss)/d(pre-activation)) and ConvTranspose2d(64 -> C, 4, 2, 1) of the generator (:139-140; N = 64, img = the image
 * gradient, P = the layer's input).  Writes *nsplit_out = eg_wgrad_img_splits_n(ntapes * B, N) slabs [N][16 C] (the per-tap kernel's layout over
 * patch rows: finish with eg_wgrad_reduce / _rank1 / _perm as after eg_im2col_img + eg_conv_wgrad, which this replaces). */
int eg_wgrad_img_ok(int dtype, int C, int H, int W, int N, int k, int stride, int pad);
int eg_wgrad_img_splits(int images);                 /* N = 32 / 64 */
int eg_wgrad_img_splits_n(int images, int N);        /* ... and N = 128: the first Discriminator layer of the CelebA script (celebA/EAD-GAN_celebA.py:110) */
int eg_wgrad_img(int dtype, const float* img0, const float* img1, const float* img2, int ntapes, const void* P, float* slab, int B, int C,
                 int H, int W, int N, int* nsplit_out, eg_stream_t s);
/* ConvTranspose2d(128 -> C <= 3, 4, 2, 1) from 16-bit NHWC activations a [B][Hin][Win][128] to an fp32 NCHW image [B][C][2 Hin][2 Win] in ONE
 * launch (the GEMM's columns stay in LDS): out = act(bias + transposed convolution); wp = the [16 * C][128] panel (row t * C + c) of
 * eg_pack_strided.  The Generator's last layer + Tanh (celebA.py:90-91) and the backward-to-image of the first Discriminator layer (:110).
 * Bit-identical to eg_conv_fwd (N = 16 C columns) + eg_col2im_img. */
int eg_convt_img_mfma_ok(int dtype, int C, int Hin, int Win, int K, int k, int stride, int pad);
/* ... with K = 64 or 128 input channels (a [B][Hin][Win][K], wp [16 * C][K]): the dSprites generators' last layer ConvTranspose2d(64 -> C, 4, 2, 1) +
 * Sigmoid (dSprites/rp.py:139-141) */
int eg_convt_img_mfma_k(int dtype, const void* a, const void* wp, const float* bias, float* out, int B, int C, int Hin, int Win, int K, int act,
                        float slope, eg_stream_t s);
int eg_convt_img_mfma(int dtype, const void* a, const void* wp, const float* bias, float* out, int B, int C, int Hin, int Win, int act,
                      float slope, eg_stream_t s);
int eg_cast_pad(int dtype, const float* src, void* dst, int rows, int n, int npad, eg_stream_t s);
/* col2im of a transposed convolution with C = 1 or 3 output channels (ConvTranspose2d(128 -> 3, 4, 2, 1): celebA/EAD-GAN_celebA.py:90-91; the
 * backward-to-image of Conv2d(3 -> 128, 4, 2, 1): :110).  cols [B*Hin*Win][k*k*C] (dtype T; column t*C + c, t = kh*k + kw) is the output of
 * ONE GEMM over the input pixels (eg_conv_fwd on a 1x1 geometry with Cout = k*k*C), so every activation is read once:
 *   out[b][c][oy][ox] = act(bias[c] + sum of cols[(b,iy,ix)][t*C + c] over the taps with oy = iy*stride - pad + kh, ox = ix*stride - pad + kw)
 * out: fp32 NCHW [B][C][(Hin-1)*stride - 2*pad + k][(Win-1)*stride - 2*pad + k]; bias may be NULL. */
int eg_col2im_img(int dtype, const void* cols, int B, int C, int Hin, int Win, int k, int stride, int pad, const float* bias, int act,
                  float slope, float* out, eg_stream_t s);
/* out = g * act'(a) over an NCHW fp32 tensor and gb[c] += sum_{b,hw} out  (partial: B*C floats) */
int eg_act_grad_mul_bias_nchw(const float* g, const float* a, float* out, int B, int C, int HW, int act, float slope,
                              float* partial, float* gb, eg_stream_t s);
int eg_flat_reduce(const float* slab, int nslab, size_t total, float* grad, int accumulate, eg_stream_t s);
int eg_flat_reduce_sn(const float* slab, int nslab, int rows, int Kdim, const float* w_orig, const float* sigma,
                      const float* u, const float* v, float* gtmp, float* partials, float* grad, eg_stream_t s);
int eg_bias_grad_nchw(const float* x, int B, int C, int HW, float* gb, eg_stream_t s);

/* --- small-N dense heads (celebA/EAD-GAN_celebA.py:122; MNIST/EAD-GAN_rpqmnxy.py:124,161-163; dSprites/rp.py:109-110,180-183)
 * y[b][n] = sum_k x[b][k] Wp[n][k] + bias[n];  x dtype T [B][K]; Wp dtype T [N][Kpad] (eg_pack_fwd order);  y fp32
 * ws (optional, ws_floats >= 16*B*N lets the planner go as far as it wants): scratch for splitting K over workgroups when B/4 row groups
 * do not fill the GPU; the slices are added in a fixed order by a second launch. */
int eg_dense_small_fwd(int dtype, const void* x, const void* wp, const float* bias, float* y, int B, int K,
                       int Kpad, int N, float* ws, size_t ws_floats, eg_stream_t s);
/* spectrally normalised variant: y = (x Wp^T) / sigma[b / sigma_rows] + bias */
int eg_dense_small_fwd_sn(int dtype, const void* x, const void* wp, const float* bias, float* y, int B, int K,
                          int Kpad, int N, const float* sigma, int sigma_rows, float* ws, size_t ws_floats, eg_stream_t s);
/* gradient prep of a dense head: dys = dy / sigma[tape] (sigma == NULL: plain head) written as dtype T at column col0 of a
 * [rows][npad] buffer and, if dys32 != NULL, as fp32 at column col0 of a [rows][ld32] buffer; gb[n] += sum_rows dy;
 * coef[tape] = sum dys * (y - bias)  (== <G_t,W>/sigma_t^2, the spectral-norm rank-1 coefficient) */
int eg_head_prep_sn(int dtype, const float* dy, int ldy, const float* y, int ldyy, const float* bias, int rows, int N,
                    const float* sigma, int rows_per_tape, void* dys, int npad, int col0, float* gb, float* coef,
                    float* dys32, int ld32, eg_stream_t s);
/* Several weight packs in ONE launch (the small networks re-pack 7-12 panels per optimizer update, each a launch of a few workgroups on
 * a launch-bound chain).  Between eg_pack_record_begin() and eg_pack_record_end() the calling thread's eg_pack_fwd / eg_pack_bwd /
 * eg_pack_conv / eg_pack_strided / eg_pack_strided2 calls launch nothing and are recorded as jobs; _end copies the jobs (eg_pack_job_bytes()
 * bytes each) to host memory jobs_out and returns their count and the joint launch's workgroup count; eg_pack_multi runs a DEVICE copy of the
 * job table: the same panel bytes as the recorded calls launched one by one. */
int eg_pack_record_begin(void);
size_t eg_pack_job_bytes(void);
int eg_pack_record_end(void* jobs_out, size_t cap_bytes, int* njobs, int* nblocks);
int eg_pack_multi(const void* jobs_dev, int njobs, int nblocks, eg_stream_t s);
/* the K-slice sums of eg_dense_small_fwd without the combine launch: partials[slice][b][n] (ws_floats >= 16 * B * N), *nslice_out slices;
 * eg_head_fused adds them (slice order, + bias) */
int eg_dense_small_fwd_slices(int dtype, const void* x, const void* wp, int B, int K, int Kpad, int N, float* partials, size_t ws_floats,
                              int* nslice_out, eg_stream_t s);
/* dx[b] = (dy[b] Wp) * act'(mask[b]) [/ sigma[b / sigma_rows]] */
int eg_dense_small_bwd(int dtype, const float* dy, const void* wp, const void* mask, void* dx, int B, int K,
                       int Kpad, int N, int mask_act, float mask_slope, const float* sigma, int sigma_rows,
                       eg_stream_t s);
int eg_dense_small_wgrad(int dtype, const float* dy, const void* x, float* gw, float* gb, int B, int K, int N,
                         int Cin, int taps, eg_stream_t s);
int eg_dense_small_bgrad(const float* dy, float* gb, int B, int N, eg_stream_t s);   /* gb[n] += sum_b dy[b][n] */

/* --- BatchNorm2d, training mode (celebA/EAD-GAN_celebA.py:79,83,87; MNIST/EAD-GAN_rpqmnxy.py:80,83,87,145) ----
 * x,y: [M][C] dtype T; updates running stats (momentum, unbiased var) and num_batches_tracked; fused activation */
size_t eg_bn_ws_floats(int M, int C);
int eg_bn_fwd_train(int dtype, const void* x, void* y, int M, int C, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                    float* save_mean, float* save_invstd, float* ws, int act, float slope, eg_stream_t s);
/* the same with the batch statistics taken from the producing convolution's epilogue (eg_epilogue.stat_mode = EG_STAT_MOMENTS):
 * stat = its stat_out, nrb row blocks of rows_per_block rows each.  Two launches (statistics + running stats, apply) instead of three,
 * and x is read once instead of twice.  ws: 2*C floats. */
int eg_bn_fwd_train_fused(int dtype, const void* x, void* y, int M, int C, const float* stat, int nrb, int rows_per_block,
                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                          long long* num_batches_tracked, float* save_mean, float* save_invstd, float* ws, int act, float slope,
                          eg_stream_t s);
/* backward with the two sums taken from the epilogue of the convolution that produced dy (EG_STAT_BN_BWD: dy already carries the
 * activation gradient): dz = gamma * invstd * (dy - sum(dy)/M - xhat * sum(dy*xhat)/M); dgamma / dbeta accumulate.  ws: 5*C floats. */
int eg_bn_bwd_fused(int dtype, const void* z, const void* dy, void* dz, int M, int C, const float* stat, int nrb,
                    const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                    float* sums, float* ws, eg_stream_t s);
/* synchronised BatchNorm for data parallel runs (statistics over the global batch: N ranks x B/N images == 1 rank x B images).
 * Forward: eg_bn_stats_local writes this rank's (count, mean, M2) per channel to stats[3*C]; the host gathers every rank's block
 * (stats_all[nranks][3*C]); eg_bn_fwd_from_stats combines them (Chan, fp64), updates the running statistics with the GLOBAL
 * batch (M_global rows) and normalises the local rows.  Backward: eg_bn_bwd_sums_local writes the local sum(dy), sum(dy*xhat)
 * to sums[2*C] and adds them to dbeta / dgamma (parameter gradients stay local; the gradient all-reduce averages them); the host
 * all-reduces sums; eg_bn_bwd_from_sums produces dz of the local rows.  ws: eg_bn_ws_floats(M, C) floats. */
int eg_bn_stats_local(int dtype, const void* x, int M, int C, float* ws, float* stats, eg_stream_t s);
int eg_bn_fwd_from_stats(int dtype, const void* x, void* y, int M_local, int C, const float* stats_all, int nranks, int M_global,
                         const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float* save_mean, float* save_invstd, float* ws, int act, float slope,
                         eg_stream_t s);
int eg_bn_bwd_sums_local(int dtype, const void* z, const void* da, int M, int C, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, int act, float slope, float* dgamma, float* dbeta,
                         float* sums, float* ws, eg_stream_t s);
int eg_bn_bwd_from_sums(int dtype, const void* z, const void* da, void* dz, int M_local, int C, const float* sums_global, int M_global,
                        const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int act, float slope,
                        float* ws, eg_stream_t s);
/* eval mode (module.eval(): generate_image.py:146-154, gen_imgs.py:106-120 of the reference): y = act((x - running_mean) /
 * sqrt(running_var + eps) * gamma + beta); nothing is updated; ws: 2*C floats */
int eg_bn_fwd_eval(int dtype, const void* x, void* y, int M, int C, const float* gamma, const float* beta, float eps,
                   const float* running_mean, const float* running_var, float* ws, int act, float slope, eg_stream_t s);
/* dz from da (gradient w.r.t. the activation output); dgamma/dbeta accumulate; sums: 2*C floats scratch */
int eg_bn_bwd(int dtype, const void* z, const void* da, void* dz, int M, int C, const float* gamma, const float* beta,
              const float* save_mean, const float* save_invstd, int act, float slope, float* dgamma, float* dbeta,
              float* sums, float* ws, eg_stream_t s);
/* as eg_bn_bwd, then dz *= act'(z as an activation OUTPUT) / post_sigma  -- for blocks ordered conv -> LeakyReLU -> BN
 * (MNIST/EAD-GAN_rpqmnxy.py:143-146): the BN input IS the LeakyReLU output, so its backward mask is fused here */
int eg_bn_bwd_post(int dtype, const void* z, const void* da, void* dz, int M, int C, const float* gamma, const float* beta,
                   const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* sums, float* ws,
                   int post_act, float post_slope, const float* post_sigma, eg_stream_t s);

/* --- spectral norm power iteration (torch.nn.utils.spectral_norm, celebA/EAD-GAN_celebA.py:110-120) -------- */
size_t eg_sn_ws_floats(int R, int Kd);
int eg_sn_power_iter(const float* w_orig, int R, int Kd, float* u, float* v, float* sigma, float* u_snap,
                     float* v_snap, float* ws, int training, float eps, eg_stream_t s);

/* all spectrally-normalised layers of a network in one launch per stage (4 launches per forward instead of 4 per layer) */
typedef struct eg_sn_layer {
    const float* w;
    float* u;
    float* v;
    float* sigma;
    float* u_snap;
    float* v_snap;
    int R, Kd;
} eg_sn_layer;
size_t eg_sn_multi_ws_floats(const eg_sn_layer* layers, int nlayers);
int eg_sn_power_iter_multi(const eg_sn_layer* layers, int nlayers, float* ws, int training, float eps, eg_stream_t s);
/* the same iteration (same u, v, sigma bits) in TWO launches instead of four: each stage's per-layer finish runs in the last workgroup of
 * the layer to arrive.  counters: >= 2 * nlayers unsigned, zeroed once by the caller, left at zero; NULL (or training == 0) = the four
 * launches of eg_sn_power_iter_multi.  One counters array must not be lent to two calls that may run concurrently. */
int eg_sn_power_iter_multi2(const eg_sn_layer* layers, int nlayers, float* ws, unsigned int* counters, int training, float eps,
                            eg_stream_t s);

/* --- Adam (torch.optim.Adam, celebA/EAD-GAN_celebA.py:211-217) over a flat fp32 arena ----------------------- */
int eg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                 int* step, int tick, eg_stream_t s);
/* the same update on a slice of the arena (one gradient bucket); zero_grad: the gradient slice is cleared in the same pass
 * (optimizer.step() + optimizer.zero_grad(), celebA/EAD-GAN_celebA.py:344-345,365-366,400-401) */
int eg_adam_step_zero(float* p, float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                      int* step, int tick, int zero_grad, eg_stream_t s);
/* optimizer.step() on ONE convolution weight (conv view master [Cout][Cin][k][k]) fused with the refresh of its packed panels: the update
 * of eg_adam_step_zero (same bits) applied on the way through the tile transpose of eg_pack_conv (same panel bytes); p / g / m / v are
 * the parameter's slices of the four arenas; `step` is the optimizer's device step counter, already ticked (eg_adam_tick).  Only layers
 * eg_adam_pack_conv_ok() accepts (channel counts in multiples of 16 / 32, no K padding); either panel may be NULL. */
int eg_adam_pack_conv_ok(const eg_conv* c, int dtype, int has_fwd, int has_bwd);
int eg_adam_pack_conv(const eg_conv* c, int dtype, float* p, float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                      const int* step, int zero_grad, void* wp_fwd, void* wp_bwd, eg_stream_t s);
/* the same for a weight whose panel is a row-permuted transpose of the master: master p[K][N] (N contiguous; the ConvTranspose2d on a
 * 1x1 input, celebA/EAD-GAN_celebA.py:76), panel wp[n'][Kpad], n' = (n % n_mod) * n_mul + n / n_mod, columns k < K (the padding keeps
 * the zeros eg_pack_strided wrote once) */
int eg_adam_pack_rows(int dtype, float* p, float* g, float* m, float* v, void* wp, int K, int N, int Kpad, int n_mod, int n_mul,
                      float lr, float b1, float b2, float eps, const int* step, int zero_grad, eg_stream_t s);
int eg_adam_tick(int* step, eg_stream_t s);            /* step[0] += 1 (once per optimizer.step(), before its slice-wise updates) */
int eg_fill_f32(float* p, size_t n, float val, eg_stream_t s);

/* --- utility: generator input concat+cast, elementwise activation gradient, layout conversion -------------- */
int eg_concat_cast(int dtype, const float* a, int wa, const float* b, int wb, const float* c, int wc, int B,
                   int Cpad, void* out, eg_stream_t s);
int eg_act_grad_mul_f32(const float* g, const float* a, float* out, size_t n, int act, float slope, eg_stream_t s);
int eg_nchw_to_nhwc(int dtype, const float* x, void* y, int B, int C, int HW, int Cpad, eg_stream_t s);
int eg_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int HW, int Cpad, eg_stream_t s);

/* --- affine codes, warp and loss heads ---------------------------------------------------------------------
 * eg_theta_rpqxy : celebA/utils_rpqxy.py:59-80 (rows 0,1 of R*Z*T) ;  eg_warp_affine : transformation_2D,
 * celebA/EAD-GAN_celebA.py:146-152 ;  losses : celebA/EAD-GAN_celebA.py:161-169,342,355-362,383-395 */
int eg_theta_rpqxy(const float* code, int ldc, int B, float* theta, eg_stream_t s);
int eg_warp_affine(const float* img, const float* theta, float* out, int B, int C, int H, int W, eg_stream_t s);
/* eg_theta_rpqxy + eg_warp_affine in one launch (same arithmetic; H * W a multiple of 256); theta_out optional; zero / zero_n: floats the
 * launch clears first (the iteration's loss accumulators) */
int eg_warp_affine_rpqxy(const float* img, const float* code, int ldc, float* theta_out, float* out, int B, int C, int H, int W,
                         float* zero, int zero_n, eg_stream_t s);
int eg_loss_bce_sigmoid(const float* o, int ld, int col, int B, float target, float scale, float* loss, float* dout,
                        int zero_rows, eg_stream_t s);
int eg_loss_mse(const float* o, int ld, int col0, int n, int B, const float* tgt, int ldt, float tconst, float scale,
                float* loss, float* dout, int zero_rows, eg_stream_t s);
int eg_loss_ce_softmaxed(const float* o, int ld, int c0, int n, int B, const long long* labels, float scale,
                         float* loss, float* dout, eg_stream_t s);
/* MNIST variant, 7 codes (theta,p,q,m,n,x,y): MNIST/utils_rpqmnxy.py:46-134.  eg_theta_rpqmnxy = rows 0,1 of R Z S T.
 * eg_loss_affine_rpqmnxy: MSE(latent(MLP(flat(rel))), code) * scale with the frozen 6-256-256-256-256-7 LeakyReLU(0.01)
 * approximator; mlp = eg_mlp_rpqmnxy_floats() floats: W1[256][6] b1 W2[256][256] b2 W3 b3 W4 b4 W5[7][256] b5, then the
 * transposes W2t W3t W4t ([in][out]) for the forward sweep. */
size_t eg_mlp_rpqmnxy_floats(void);
int eg_theta_rpqmnxy(const float* code, int ldc, int B, float* theta, eg_stream_t s);
int eg_loss_affine_rpqmnxy(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc,
                           const float* mlp, float scale, float* loss, float* d_real, float* d_trans, float* pred_out,
                           float* ws /* B floats */, eg_stream_t s);
/* dSprites variants (dSprites/utils_rp.py:38-59,94-147, utils_pxy.py:69-87, rp.py:225-232,375-377) */
int eg_theta_rp(const float* code, int ldc, int B, float* theta, eg_stream_t s);                 /* rows 0,1 of R Z(p,p) T */
int eg_theta_pxy_align_inv(const float* code, int ldc, int B, float* theta, eg_stream_t s);      /* rows 0,1 of inverse(T(x,y)) */
int eg_loss_affine_rp(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc, float scale,
                      float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s);
int eg_loss_mutual_info(const float* o, int ld, int c0, int n, int B, const float* tgt, int ldt, int t0, int target_logits,
                        float scale, float* loss, float* dout, eg_stream_t s);
/* colored dSprites (colored_dSprites/rp_color.py:368-394,415-424; utils_rp_color.py:38-75,100-139) */
int eg_u8_colorize(const unsigned char* sprites, const float* gain, float* out, int B, int C, int HW, eg_stream_t s);
int eg_color_scale(const float* in, const float* code, int ldc, int c0, float factor, int divide, float* out, int B, int C, int HW,
                   eg_stream_t s);
int eg_loss_affine_rp_color(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc, float scale,
                            float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s);
int eg_add_f32(float* out, const float* a, const float* b, size_t n, eg_stream_t s);
int eg_u8_to_f32(const unsigned char* x, float* y, size_t n, eg_stream_t s);            /* uint8 sprites -> float (rp.py:369-370) */
int eg_loss_affine_rpqxy(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc,
                         float scale, float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s);
/* the info step's three losses in one launch (celebA.py:390-396): lcon * MSE(o_gen[:, c_cont : c_cont+n_cont], code) + lcat *
 * CE(softmax(o_gen[:, c_cont+n_cont : +n_cat]), labels) into d_gen (rows zeroed first), laff * MSE(affine_regularzier(o_real, o_trans),
 * code[:, :5]) into d_real / d_trans; all three add to loss[0] in that order -- the same numbers as eg_loss_mse, eg_loss_ce_softmaxed and
 * eg_loss_affine_rpqxy launched one after the other */
int eg_loss_info_rpqxy(const float* o_gen, const float* o_trans, const float* o_real, int ld, int c_cont, int n_cont, int n_cat, int B,
                       const float* code, int ldc, const long long* labels, float lcat, float lcon, float laff, float* loss, float* d_gen,
                       float* d_trans, float* d_real, eg_stream_t s);

/* The tail of the discriminator's head in ONE launch behind eg_dense_small_fwd_slices (celebA.py:110-122 the 4x4 head convolution as a
 * dense layer; :334-345, :353-366 the adversarial BCE terms; :375-401 the info step's three losses): the workgroups of sample index b work
 * on the T rows t*B + b (the tapes of that sample):
 *   y[t*B+b][n]  = sum_slices partials[slice][t*B+b][n] + bias[n]                               (= the slice combine of eg_dense_small_fwd)
 *   dout rows    = d(loss)/d(y) of the rows, other columns zero                                 (= eg_loss_bce_sigmoid / eg_loss_info_rpqxy)
 *   dx[t*B+b][k] = (sum_n dout[t*B+b][n] * wp[n][k]) * act'(x[t*B+b][k]) / sigma[t]              (= eg_dense_small_bwd with mask = x)
 * and the last workgroup to finish adds the batch's loss terms to loss[0] in the order of the stand-alone loss kernels: the same bits as
 * the four launches it replaces (combine, loss, dense backward; the loss sums for B <= 256, affine term B <= 128 -- beyond, the same
 * terms added in another order).  mode 0: tape t carries BCE(sigmoid(y[:, 0]), target[t]) * scale[t] (T <= 3).  mode 1 (T = 3, tapes:
 * generated, transformed, real): lcon * MSE(y_gen[:, c_cont : +n_cont], code) + lcat * CE(softmax(y_gen[:, c_cont+n_cont : +n_cat]),
 * labels) + laff * MSE(affine_regularzier(y_real, y_trans), code[:, :5]).  terms: >= 3*B floats of scratch, counter: one zeroed unsigned
 * (left zero).  Supported heads: N = 19 (CelebA), K a multiple of 8 elements (4 for fp32). */
typedef struct eg_head {
    const void* x;
    const void* wp;
    const float* bias;
    const float* partials;
    int nslice;
    float* y;
    float* dout;
    void* dx;
    const float* sigma;
    int B, T, K, Kpad, N, mode;
    float target[3], scale[3];
    int c_cont, n_cont, n_cat;
    const float* code;
    int ldc;
    const long long* labels;
    float lcat, lcon, laff;
    float* loss;
    float* terms;
    unsigned int* counter;
    int mask_act;
    float mask_slope;
} eg_head;
int eg_head_fused(int dtype, const eg_head* h, eg_stream_t s);
int eg_head_fused_ok(int dtype, int T, int K, int N);

/* regression target of the approximator fit (MNIST/approximate_rpqmnxy.py:43-60,119-136): affine parameters of a code */
int eg_affine_para_rpqmnxy(const float* code, int ldc, int B, float* para, eg_stream_t s);

/* stage-1 trainer of Encoder_pxy (dSprites/pxy.py:156-191, dSprites/utils_pxy.py:24-66,107-126): theta = rows 0,1 of
 * get_matrix_pxy(code); affine_regularzier_pxy + MSE with fused gradients w.r.t. both codes */
int eg_theta_pxy(const float* code, int ldc, int B, float* theta, eg_stream_t s);
/* ncol = 3: three colour-gain entries follow (p, x, y) (colored_dSprites/utils_pxy.py:150-176, pxy_color.py:185-213) */
int eg_loss_affine_pxy(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc, int ncol, float scale,
                       float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s);
/* transformation_2D with grid_sample(padding_mode='zeros') (colored_dSprites/pxy_color.py:86-92) */
int eg_warp_affine_zeros(const float* img, const float* theta, float* out, int B, int C, int H, int W, eg_stream_t s);

/* --- image grids of the sampling tools: torchvision.utils.make_grid / save_image (0.8.2, not vendored) as the reference
 * calls them -- MNIST/EAD-GAN_rpqmnxy.py:281-330, MNIST/generate_image.py:122-138, celebA/EAD-GAN_celebA.py:238-287,
 * celebA/gen_imgs.py:183-199, dSprites/rp.py:299-353, colored_dSprites/rp_color.py:297-353.
 * eg_make_grid : img [B,C,H,W] fp32 -> grid [Cg][ymaps*(H+padding)+padding][xmaps*(W+padding)+padding] fp32, xmaps = min(nrow,B),
 *   ymaps = ceil(B/xmaps), Cg = 3 when C == 1 (replicated) else C; gaps and unused cells = pad_value.  range != NULL ({lo, hi} on the
 *   device, e.g. from eg_minmax_f32): image pixels are normalised like eg_quantize_u8 does while tiling (make_grid(normalize=True)).
 * eg_minmax_f32: out2 = {min, max} of x[0..n) (the normalize=True range); ws: eg_minmax_ws_floats() floats.
 * eg_quantize_u8: x [C][H][W] fp32 -> out [H][W][C] uint8 = clamp(v' * 255 + 0.5, 0, 255) truncated, v' = v or, with range = {lo, hi}
 *   on the device, (clamp(v, lo, hi) - lo) / (hi - lo + 1e-5).  Every step separately rounded: bytes equal the CPU restatement's. */
int eg_make_grid(const float* img, int B, int C, int H, int W, int nrow, int padding, float pad_value, const float* range,
                 float* grid, eg_stream_t s);
size_t eg_minmax_ws_floats(void);
int eg_minmax_f32(const float* x, size_t n, float* ws, float* out2, eg_stream_t s);
int eg_quantize_u8(const float* x, int C, int H, int W, const float* range, unsigned char* out, eg_stream_t s);

/* --- device-side input pipeline: replaces the per-iteration host work of the reference loops -- DataLoader + PIL
 * RandomHorizontalFlip / ToTensor / Normalize (celebA/EAD-GAN_celebA.py:194-206, MNIST/EAD-GAN_rpqmnxy.py:235-246) and the numpy draws
 * of z / code / labels (:308-317; :351-357) -- so that a captured hipGraph feeds itself.  Philox4x32-10, counter = (element, *step,
 * stream_id), key = seed: reproducible per (seed, step), same distributions as the reference, NOT numpy's stream.
 * kind 0: uniform [a,b) fp32; 1: normal(a, b) fp32; 2: integers in [a,b) as int64; 3: Bernoulli(a) as uint8;
 * 4: epoch permutation (DataLoader(shuffle=True), celebA.py:204-206): int64 dataset indices in [0, N = a) for the stream positions
 *    *step * n + i -- a keyed bijection per epoch (position / N), every index exactly once per epoch, nothing stored; N <= 2^24 */
int eg_rng_fill(int kind, void* out, size_t n, float a, float b, unsigned long long seed, const int* step, unsigned int stream_id,
                eg_stream_t s);
/* several draws in ONE launch (the same values as one eg_rng_fill per draw).  onehot != NULL (kind 2 only): row i of onehot[n][onehot_n] is
 * the one-hot code of draw i (to_categorical fused). */
typedef struct eg_rng_seg {
    int kind;
    void* out;
    size_t n;
    float a, b;
    unsigned int stream_id;
    float* onehot;
    int onehot_n;
} eg_rng_seg;
int eg_rng_fill_multi(const eg_rng_seg* segs, int nseg, unsigned long long seed, const int* step, eg_stream_t s);
int eg_counter_add(int* counter, int v, eg_stream_t s);
/* out[b] = data[idx[b]] (uint8 NCHW dataset in HBM), mirrored along x where flip[b], * scale + shift  -> fp32 NCHW */
int eg_gather_u8_images(const unsigned char* data, const long long* idx, const unsigned char* flip, float* out, int B, int C, int H,
                        int W, float scale, float shift, eg_stream_t s);
/* the same, and step_tick[0] += 1 afterwards (NULL: no tick): the iteration's draws -- earlier launches on the stream -- have read the counter */
int eg_gather_u8_images_tick(const unsigned char* data, const long long* idx, const unsigned char* flip, float* out, int B, int C, int H,
                             int W, float scale, float shift, int* step_tick, eg_stream_t s);
int eg_onehot(const long long* labels, float* out, int B, int n, eg_stream_t s);          /* to_categorical (celebA.py:59-64) */
/* transforms.Resize (PIL bilinear, antialiased) + CenterCrop of the uint8 dataset on its way into HBM (celebA.py:194-196): ONE separable
 * pass along y (axis 0) or x (axis 1) of `planes` planar uint8 images [in_h][in_w] with PIL's 8-bit fixed-point coefficient tables
 * (bounds[2*o] = first tap, bounds[2*o+1] = tap count, kk[o*ksize + t], 22 fractional bits; the host builds them as PIL does); produces
 * outputs o0 .. o0+on-1 along the axis and copies columns / rows c0 .. c0+cn-1 of the other axis: dst is [planes][on][cn] (axis 0) or
 * [planes][cn][on] (axis 1).  Horizontal pass first, then vertical, reproduces Image.resize(..., BILINEAR) bit for bit. */
int eg_resample_u8(const unsigned char* src, unsigned char* dst, int planes, int in_h, int in_w, int axis, const int* bounds,
                   const int* kk, int ksize, int o0, int on, int c0, int cn, eg_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* EADGAN_HIP_H */
