// Affine warp (F.affine_grid + F.grid_sample bilinear/border, celebA/EAD-GAN_celebA.py:146-152), latent-code ->
// matrix kernels (celebA/utils_rpqxy.py:59-80) and the fused loss heads with their gradients
// (celebA/EAD-GAN_celebA.py:161-169,342,355-362,383-395).  All reductions are single-block and deterministic.
#include "affine_math.h"

// theta[b][2][3] from 5 latent codes (row stride ldc)
__global__ void theta_rpqxy_kernel(const float* __restrict__ code, int ldc, int B, float* __restrict__ theta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float c[5];
    for (int i = 0; i < 5; ++i) c[i] = code[(size_t)b * ldc + i];
    const Aff<float> m = matrix_rpqxy<float>(c);
    float* t = theta + (size_t)b * 6;
    t[0] = m.a; t[1] = m.b; t[2] = m.c; t[3] = m.d; t[4] = m.e; t[5] = m.f;
}

extern "C" int eg_theta_rpqxy(const float* code, int ldc, int B, float* theta, eg_stream_t s) {
    EG_REQUIRE(code && theta && ldc >= 5, "eg_theta_rpqxy: bad argument");
    hipLaunchKernelGGL(theta_rpqxy_kernel, dim3(cdiv(B, 128)), dim3(128), 0, (hipStream_t)s, code, ldc, B, theta);
    EG_LAUNCH_CHECK();
    return 0;
}

// out[b,c,y,x] = bilinear sample of img[b,c] at theta[b] * (xn, yn, 1), align_corners=False, border padding
template <bool ZEROS>      // ZEROS: grid_sample(padding_mode='zeros') (colored_dSprites/pxy_color.py:90), else 'border'
__global__ void warp_affine_kernel(const float* __restrict__ img, const float* __restrict__ theta, float* __restrict__ out, int B, int C,
                                   int H, int W) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W;
    if (idx >= total) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((size_t)W * H));
    const float* t = theta + (size_t)b * 6;
    const float xn = (2.f * x + 1.f) / W - 1.f, yn = (2.f * y + 1.f) / H - 1.f;
    const float gx = t[0] * xn + t[1] * yn + t[2];
    const float gy = t[3] * xn + t[4] * yn + t[5];
    float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
    if (!ZEROS) {
        ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
        iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    } else {                                        // keep the index arithmetic finite for far-away samples (all four taps are outside anyway)
        ix = fminf(fmaxf(ix, -2.f), (float)(W + 1));
        iy = fminf(fmaxf(iy, -2.f), (float)(H + 1));
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    for (int c = 0; c < C; ++c) {
        const float* p = img + ((size_t)b * C + c) * H * W;
        float v = 0.f;
        if (vx0 && vy0) v += p[y0 * W + x0] * (wx0 * wy0);
        if (vx1 && vy0) v += p[y0 * W + x1] * (wx1 * wy0);
        if (vx0 && vy1) v += p[y1 * W + x0] * (wx0 * wy1);
        if (vx1 && vy1) v += p[y1 * W + x1] * (wx1 * wy1);
        out[((size_t)b * C + c) * H * W + (size_t)y * W + x] = v;
    }
}

extern "C" int eg_warp_affine(const float* img, const float* theta, float* out, int B, int C, int H, int W, eg_stream_t s) {
    EG_REQUIRE(img && theta && out, "eg_warp_affine: null pointer");
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(warp_affine_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, img, theta, out, B, C, H, W);
    EG_LAUNCH_CHECK();
    return 0;
}

// theta_rpqxy + warp in one launch (the head of the CelebA iteration: A = get_matrix(code[:, :5]); scaled = trans_2D(real, A[:, 0:2]),
// celebA/EAD-GAN_celebA.py:325-327): a block's 256 pixels lie in one image (H * W a multiple of 256), its first thread forms that image's
// matrix -- the function theta_rpqxy_kernel calls -- and the rest is warp_affine_kernel<false>'s arithmetic; `zero` (optional): zero_n floats
// cleared by the launch's first thread (the iteration's loss accumulators)
__global__ __launch_bounds__(256) void warp_affine_rpqxy_kernel(const float* __restrict__ img, const float* __restrict__ code, int ldc,
                                                                float* __restrict__ theta_out, float* __restrict__ out, int B, int C, int H, int W,
                                                                float* zero, int zero_n) {
    __shared__ float th[6];
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(idx / ((size_t)W * H));
    if (blockIdx.x == 0 && threadIdx.x == 0 && zero)
        for (int i = 0; i < zero_n; ++i) zero[i] = 0.f;
    if (threadIdx.x == 0) {
        float c[5];
        for (int i = 0; i < 5; ++i) c[i] = code[(size_t)b * ldc + i];
        const Aff<float> m = matrix_rpqxy<float>(c);
        th[0] = m.a; th[1] = m.b; th[2] = m.c; th[3] = m.d; th[4] = m.e; th[5] = m.f;
        if (theta_out && (idx % ((size_t)W * H)) == 0) {
            float* t = theta_out + (size_t)b * 6;
            t[0] = m.a; t[1] = m.b; t[2] = m.c; t[3] = m.d; t[4] = m.e; t[5] = m.f;
        }
    }
    __syncthreads();
    const int x = (int)(idx % W), y = (int)((idx / W) % H);
    const float xn = (2.f * x + 1.f) / W - 1.f, yn = (2.f * y + 1.f) / H - 1.f;
    const float gx = th[0] * xn + th[1] * yn + th[2];
    const float gy = th[3] * xn + th[4] * yn + th[5];
    float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
    ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    for (int c = 0; c < C; ++c) {
        const float* p = img + ((size_t)b * C + c) * H * W;
        float v = 0.f;
        if (vx0 && vy0) v += p[y0 * W + x0] * (wx0 * wy0);
        if (vx1 && vy0) v += p[y0 * W + x1] * (wx1 * wy0);
        if (vx0 && vy1) v += p[y1 * W + x0] * (wx0 * wy1);
        if (vx1 && vy1) v += p[y1 * W + x1] * (wx1 * wy1);
        out[((size_t)b * C + c) * H * W + (size_t)y * W + x] = v;
    }
}

extern "C" int eg_warp_affine_rpqxy(const float* img, const float* code, int ldc, float* theta_out, float* out, int B, int C, int H, int W,
                                    float* zero, int zero_n, eg_stream_t s) {
    EG_REQUIRE(img && code && out && ldc >= 5 && B > 0 && ((size_t)H * W) % 256 == 0 && zero_n >= 0 && zero_n <= 64, "eg_warp_affine_rpqxy: bad argument (H * W must be a multiple of 256)");
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(warp_affine_rpqxy_kernel, dim3((unsigned)(total / 256)), dim3(256), 0, (hipStream_t)s, img, code, ldc, theta_out, out, B, C, H, W, zero, zero_n);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_warp_affine_zeros(const float* img, const float* theta, float* out, int B, int C, int H, int W, eg_stream_t s) {
    EG_REQUIRE(img && theta && out, "eg_warp_affine_zeros: null pointer");
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(warp_affine_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, img, theta, out, B, C, H, W);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- loss heads: operate on the raw head output `o` [B][ld] fp32; write d(loss)/d(o) into `dout` [B][ld] ------
// BCELoss(sigmoid(o[:,col]), target) * scale ; loss[slot] += value ; dout row zero-filled first when zero_rows!=0
__global__ void bce_sigmoid_kernel(const float* __restrict__ o, int ld, int col, int B, float target, float scale, float* loss,
                                   float* __restrict__ dout, int zero_rows) {
    __shared__ float sm[16];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float p = 1.f / (1.f + expf(-o[(size_t)b * ld + col]));
        const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
        acc += -(target * lp + (1.f - target) * l1p);
        if (dout) {
            if (zero_rows)
                for (int j = 0; j < ld; ++j) dout[(size_t)b * ld + j] = 0.f;
            // torch binary_cross_entropy_backward: (p - t) / max((1-p) p, 1e-12), then sigmoid backward p (1-p)
            const float gp = (p - target) / fmaxf((1.f - p) * p, 1e-12f) * (scale / (float)B);
            dout[(size_t)b * ld + col] = gp * p * (1.f - p);
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)B;
}

extern "C" int eg_loss_bce_sigmoid(const float* o, int ld, int col, int B, float target, float scale, float* loss, float* dout,
                                   int zero_rows, eg_stream_t s) {
    EG_REQUIRE(o && B > 0, "eg_loss_bce_sigmoid: bad argument");
    hipLaunchKernelGGL(bce_sigmoid_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, o, ld, col, B, target, scale, loss, dout, zero_rows);
    EG_LAUNCH_CHECK();
    return 0;
}

// MSE(o[:,col0:col0+n], target) * scale  (target: per-element tensor tgt[B][ldt] or constant when tgt==null)
// (the three bodies are real function calls: the fused launch and the stand-alone launches then run the SAME machine code and give the same bits)
__device__ __attribute__((noinline)) void mse_body(const float* __restrict__ o, int ld, int col0, int n, int B, const float* __restrict__ tgt, int ldt, float tconst,
                                         float scale, float* loss, float* __restrict__ dout, int zero_rows, float* sm) {
    float acc = 0.f;
    const float gs = 2.f * scale / (float)(B * n);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        if (dout && zero_rows)
            for (int j = 0; j < ld; ++j) dout[(size_t)b * ld + j] = 0.f;
        for (int j = 0; j < n; ++j) {
            const float d = o[(size_t)b * ld + col0 + j] - (tgt ? tgt[(size_t)b * ldt + j] : tconst);
            acc += d * d;
            if (dout) dout[(size_t)b * ld + col0 + j] = gs * d;
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)(B * n);
}
__global__ void mse_kernel(const float* __restrict__ o, int ld, int col0, int n, int B, const float* __restrict__ tgt, int ldt, float tconst,
                           float scale, float* loss, float* __restrict__ dout, int zero_rows) {
    __shared__ float sm[16];
    mse_body(o, ld, col0, n, B, tgt, ldt, tconst, scale, loss, dout, zero_rows, sm);
}

extern "C" int eg_loss_mse(const float* o, int ld, int col0, int n, int B, const float* tgt, int ldt, float tconst, float scale, float* loss,
                           float* dout, int zero_rows, eg_stream_t s) {
    EG_REQUIRE(o && B > 0 && n > 0, "eg_loss_mse: bad argument");
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, o, ld, col0, n, B, tgt, ldt, tconst, scale, loss, dout, zero_rows);
    EG_LAUNCH_CHECK();
    return 0;
}

// CrossEntropyLoss applied to softmax(o[:,c0:c0+n]) (the reference feeds probabilities, i.e. a double softmax:
// celebA/EAD-GAN_celebA.py:132,383 ; MNIST/EAD-GAN_rpqmnxy.py:161,427).  Adds into dout (does not zero).
#define EG_MAXCAT 16
__device__ __attribute__((noinline)) void ce_softmaxed_body(const float* __restrict__ o, int ld, int c0, int n, int B, const long long* __restrict__ labels,
                                                  float scale, float* loss, float* __restrict__ dout, float* sm) {
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float q[EG_MAXCAT], r[EG_MAXCAT];
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, o[(size_t)b * ld + c0 + j]);
        float se = 0.f;
        for (int j = 0; j < n; ++j) { q[j] = expf(o[(size_t)b * ld + c0 + j] - mx); se += q[j]; }
        for (int j = 0; j < n; ++j) q[j] /= se;
        float mq = -INFINITY;
        for (int j = 0; j < n; ++j) mq = fmaxf(mq, q[j]);
        float s2 = 0.f;
        for (int j = 0; j < n; ++j) { r[j] = expf(q[j] - mq); s2 += r[j]; }
        const int lab = (int)labels[b];
        acc += -(q[lab] - mq - logf(s2));
        if (dout) {
            float dot = 0.f;
            float gq[EG_MAXCAT];
            for (int j = 0; j < n; ++j) { gq[j] = (r[j] / s2 - (j == lab ? 1.f : 0.f)) * (scale / (float)B); dot += gq[j] * q[j]; }
            for (int j = 0; j < n; ++j) dout[(size_t)b * ld + c0 + j] += q[j] * (gq[j] - dot);
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)B;
}
__global__ void ce_softmaxed_kernel(const float* __restrict__ o, int ld, int c0, int n, int B, const long long* __restrict__ labels,
                                    float scale, float* loss, float* __restrict__ dout) {
    __shared__ float sm[16];
    ce_softmaxed_body(o, ld, c0, n, B, labels, scale, loss, dout, sm);
}

extern "C" int eg_loss_ce_softmaxed(const float* o, int ld, int c0, int n, int B, const long long* labels, float scale, float* loss,
                                    float* dout, eg_stream_t s) {
    EG_REQUIRE(o && labels && n <= EG_MAXCAT && B > 0, "eg_loss_ce_softmaxed: bad argument");
    hipLaunchKernelGGL(ce_softmaxed_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, o, ld, c0, n, B, labels, scale, loss, dout);
    EG_LAUNCH_CHECK();
    return 0;
}

// affine-consistency loss, CelebA variant:  MSE(regulariser(real_code, trans_code), code[:, :5]) * scale
// o_real / o_trans: head outputs [B][ld], codes start at column c0.  d_real / d_trans rows are zero-filled.
// nthr: threads that take samples (the stand-alone launch has 128; inside the fused info-loss launch the other waves add zeros to the
// block sum, which keeps the summation order of the stand-alone kernel)
__device__ __attribute__((noinline)) void affine_reg_rpqxy_body(const float* __restrict__ o_real, const float* __restrict__ o_trans, int ld, int c0, int B,
                                                      const float* __restrict__ code, int ldc, float scale, float* loss, float* __restrict__ d_real,
                                                      float* __restrict__ d_trans, float* __restrict__ pred_out, int nthr, float* sm) {
    float acc = 0.f;
    const float gs = 2.f * scale / (float)(B * 5);
    for (int b = threadIdx.x; b < B && (int)threadIdx.x < nthr; b += nthr) {
        Dual<10> rc[5], tc[5], out[5];
        for (int i = 0; i < 5; ++i) {
            rc[i] = dvar<10>(o_real[(size_t)b * ld + c0 + i], i);
            tc[i] = dvar<10>(o_trans[(size_t)b * ld + c0 + i], 5 + i);
        }
        regularizer_rpqxy<Dual<10>>(rc, tc, out);
        float gr[10];
        for (int i = 0; i < 10; ++i) gr[i] = 0.f;
        for (int j = 0; j < 5; ++j) {
            const float d = out[j].v - code[(size_t)b * ldc + j];
            acc += d * d;
            if (pred_out) pred_out[(size_t)b * 5 + j] = out[j].v;
            for (int i = 0; i < 10; ++i) gr[i] += gs * d * out[j].d[i];
        }
        if (d_real && d_trans) {
            for (int j = 0; j < ld; ++j) { d_real[(size_t)b * ld + j] = 0.f; d_trans[(size_t)b * ld + j] = 0.f; }
            for (int i = 0; i < 5; ++i) { d_real[(size_t)b * ld + c0 + i] = gr[i]; d_trans[(size_t)b * ld + c0 + i] = gr[5 + i]; }
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)(B * 5);
}
__global__ void affine_reg_rpqxy_kernel(const float* __restrict__ o_real, const float* __restrict__ o_trans, int ld, int c0, int B,
                                        const float* __restrict__ code, int ldc, float scale, float* loss, float* __restrict__ d_real,
                                        float* __restrict__ d_trans, float* __restrict__ pred_out) {
    __shared__ float sm[16];
    affine_reg_rpqxy_body(o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out, blockDim.x, sm);
}

// the three losses of the CelebA info step in ONE launch (celebA/EAD-GAN_celebA.py:390-396: lambda_con * MSE(cont, code) + lambda_cat *
// CE(cat, labels) on D(gen), lambda_affine * MSE(affine_regularzier(D(real), D(trans)), code[:, :5])): one block runs the three bodies one
// after the other -- they add to the same loss scalar and to the same gradient rows, in the order of the three stand-alone launches.
__global__ __launch_bounds__(256) void info_losses_rpqxy_kernel(const float* __restrict__ o_gen, const float* __restrict__ o_trans, const float* __restrict__ o_real,
                                                                int ld, int c_cont, int n_cont, int n_cat, int B, const float* __restrict__ code, int ldc,
                                                                const long long* __restrict__ labels, float lcat, float lcon, float laff, float* loss,
                                                                float* __restrict__ d_gen, float* __restrict__ d_trans, float* __restrict__ d_real) {
    __shared__ float sm[16];
    mse_body(o_gen, ld, c_cont, n_cont, B, code, ldc, 0.f, lcon, loss, d_gen, 1, sm);
    __syncthreads();
    ce_softmaxed_body(o_gen, ld, c_cont + n_cont, n_cat, B, labels, lcat, loss, d_gen, sm);
    __syncthreads();
    affine_reg_rpqxy_body(o_real, o_trans, ld, c_cont, B, code, ldc, laff, loss, d_real, d_trans, nullptr, 128, sm);
}

extern "C" int eg_loss_info_rpqxy(const float* o_gen, const float* o_trans, const float* o_real, int ld, int c_cont, int n_cont, int n_cat, int B,
                                  const float* code, int ldc, const long long* labels, float lcat, float lcon, float laff, float* loss, float* d_gen,
                                  float* d_trans, float* d_real, eg_stream_t s) {
    EG_REQUIRE(o_gen && o_trans && o_real && code && labels && loss && d_gen && d_trans && d_real && B > 0 && n_cat <= EG_MAXCAT && n_cont >= 5,
               "eg_loss_info_rpqxy: bad argument");
    hipLaunchKernelGGL(info_losses_rpqxy_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, o_gen, o_trans, o_real, ld, c_cont, n_cont, n_cat, B, code, ldc,
                       labels, lcat, lcon, laff, loss, d_gen, d_trans, d_real);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// eg_head_fused (ABI header): slice sums of the dense head -> head output -> losses -> head input gradient, one launch.
// Grid (B, KS): workgroup (b, ks) owns sample index b -- the T rows t*B + b -- and K range ks of the input gradient; every workgroup of
// a sample adds the slices and evaluates the (cheap) losses itself, the ks = 0 one stores y / dout and the sample's loss terms.
// The per-sample arithmetic is that of dense_small_combine_kernel, bce_sigmoid_kernel / mse_body / ce_softmaxed_body /
// affine_reg_rpqxy_body and dense_small_bwd_kernel (small.hip: n = 0..N-1 in order, one fma per term): the same bits as the separate
// launches.  The affine term's forward-mode Jacobian runs one derivative component per lane (Dual<1> on ten lanes instead of Dual<10> on
// one: each component's arithmetic is independent of the others, so the values are the same) -- the one-thread form was a 20 us chain.
// The batch sums are taken by the last ks = 0 workgroup to arrive, in the thread order of the stand-alone loss kernels (thread i owns
// samples i, i + 256, ...; affine: i, i + 128, ...).  Hand-off: terms are written through (agent-scope stores), vmcnt(0), relaxed counter.
// ------------------------------------------------------------------------------------------------------------------------
template <typename E, int T, int N>
__global__ __launch_bounds__(512) void head_fused_kernel(const eg_head h) {
    constexpr int VEC = Elt<E>::VEC;
    constexpr int NTH = 512;
    __shared__ float ys[T][32], dl[T][32];
    __shared__ float sm[16];
    __shared__ unsigned flag;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int B = h.B, K = h.K;
    const bool first = blockIdx.y == 0;
    const E* __restrict__ x = reinterpret_cast<const E*>(h.x);
    const E* __restrict__ wp = reinterpret_cast<const E*>(h.wp);

    // ---- 1) y[t][n] = sum of the K slices (slice order) + bias ----
    if (tid < T * 32) {
        const int t = tid >> 5, n = tid & 31;
        float tot = 0.f;
        if (n < N) {
            const size_t i = ((size_t)t * B + b) * N + n;
            for (int z = 0; z < h.nslice; ++z) tot += h.partials[(size_t)z * (T * B) * N + i];
            tot = tot + (h.bias ? h.bias[n] : 0.f);
            if (first) h.y[i] = tot;
        }
        ys[t][n] = tot;
        dl[t][n] = 0.f;
    }
    __syncthreads();

    // ---- 2) d(loss)/d(y) of the sample's rows (other columns zero) and its loss terms ----
    float* terms = h.terms;
    const int mode = h.mode;
    if (mode == 0) {
        if (tid < T) {
            const int t = tid;
            const float target = h.target[t], scale = h.scale[t];
            const float p = 1.f / (1.f + expf(-ys[t][0]));
            const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
            const float gp = (p - target) / fmaxf((1.f - p) * p, 1e-12f) * (scale / (float)B);
            dl[t][0] = gp * p * (1.f - p);
            if (first) __hip_atomic_store(terms + (size_t)t * B + b, -(target * lp + (1.f - target) * l1p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if constexpr (T == 3) {
        if (tid < 64) {
            // tape 0 (generated): MSE of the continuous codes, then CE of the soft-maxed classes, on wave 0 with lane j = code / class j.
            // The sums run over the lanes in index order (v_readlane, one add per element): the stand-alone bodies' order, the same bits;
            // as a one-thread loop over register arrays indexed by `lab` this was a 27 us chain.
            auto rl = [](float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
            const int lane = tid, c0 = h.c_cont, n = h.n_cont, k0 = c0 + n, nc = h.n_cat;
            const float gs = 2.f * h.lcon / (float)(B * n);
            const float d = lane < n ? ys[0][c0 + lane] - h.code[(size_t)b * h.ldc + lane] : 0.f;
            const float dd = d * d;
            float acc = 0.f;
            for (int j = 0; j < n; ++j) acc += rl(dd, j);
            if (lane < n) dl[0][c0 + lane] = gs * d;
            if (first && lane == 0) __hip_atomic_store(terms + b, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool on = lane < nc;
            const float o = on ? ys[0][k0 + lane] : -INFINITY;
            float mx = -INFINITY;
            for (int j = 0; j < nc; ++j) mx = fmaxf(mx, rl(o, j));
            float q = on ? expf(o - mx) : 0.f;
            float se = 0.f;
            for (int j = 0; j < nc; ++j) se += rl(q, j);
            q = q / se;
            float mq = -INFINITY;
            for (int j = 0; j < nc; ++j) mq = fmaxf(mq, rl(q, j));
            const float r = on ? expf(q - mq) : 0.f;
            float s2 = 0.f;
            for (int j = 0; j < nc; ++j) s2 += rl(r, j);
            const int lab = __builtin_amdgcn_readfirstlane((int)h.labels[b]);
            const float qlab = rl(q, lab);
            if (first && lane == 0) __hip_atomic_store(terms + (size_t)B + b, -(qlab - mq - logf(s2)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float gq = (r / s2 - (lane == lab ? 1.f : 0.f)) * (h.lcat / (float)B);
            const float pr = gq * q;
            float dot = 0.f;
            for (int j = 0; j < nc; ++j) dot += rl(pr, j);
            if (on) dl[0][k0 + lane] += q * (gq - dot);
        }
        if (tid >= 64 && tid < 74) {                    // tapes 1 (transformed) and 2 (real): the affine-consistency term, lane i = d/d(variable i)
            const int i = tid - 64, c0 = h.c_cont;
            const float gs = 2.f * h.laff / (float)(B * 5);
            Dual<1> rc[5], tc[5], out[5];
            for (int k = 0; k < 5; ++k) {
                rc[k] = k == i ? dvar<1>(ys[2][c0 + k], 0) : dconst<1>(ys[2][c0 + k]);
                tc[k] = 5 + k == i ? dvar<1>(ys[1][c0 + k], 0) : dconst<1>(ys[1][c0 + k]);
            }
            regularizer_rpqxy<Dual<1>>(rc, tc, out);
            float gr = 0.f, acc = 0.f;
            for (int j = 0; j < 5; ++j) {
                const float d = out[j].v - h.code[(size_t)b * h.ldc + j];
                acc += d * d;
                gr += gs * d * out[j].d[0];
            }
            if (i < 5) dl[2][c0 + i] = gr;
            else dl[1][c0 + i - 5] = gr;
            if (i == 0 && first) __hip_atomic_store(terms + (size_t)2 * B + b, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (first && tid < T * 32 && (tid & 31) < N) h.dout[((size_t)(tid >> 5) * B + b) * N + (tid & 31)] = dl[tid >> 5][tid & 31];

    // ---- 3) dx[t][k] = (sum_n dout[t][n] * wp[n][k]) * act'(x[t][k]) / sigma[t], k in this workgroup's range ----
    {
        float post[T];
#pragma unroll
        for (int t = 0; t < T; ++t) post[t] = h.sigma ? 1.f / h.sigma[t] : 1.f;
        E* __restrict__ dx = reinterpret_cast<E*>(h.dx);
        const int kper = (K / VEC + gridDim.y - 1) / gridDim.y * VEC;
        const int kend = min(K, ((int)blockIdx.y + 1) * kper);
        for (int k0 = blockIdx.y * kper + tid * VEC; k0 < kend; k0 += NTH * VEC) {
            uint4 xv[T], wv[N];
#pragma unroll
            for (int t = 0; t < T; ++t) xv[t] = *reinterpret_cast<const uint4*>(x + ((size_t)t * B + b) * K + k0);
#pragma unroll
            for (int n = 0; n < N; ++n) wv[n] = *reinterpret_cast<const uint4*>(wp + (size_t)n * h.Kpad + k0);
            float a[T][VEC];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int j = 0; j < VEC; ++j) a[t][j] = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const E* we = reinterpret_cast<const E*>(&wv[n]);
                float wf[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) wf[j] = Elt<E>::ld(we + j);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float g = dl[t][n];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) a[t][j] = fmaf(g, wf[j], a[t][j]);
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const E* me = reinterpret_cast<const E*>(&xv[t]);
                uint4 ov;
                E* oe = reinterpret_cast<E*>(&ov);
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    Elt<E>::st(oe + j, a[t][j] * eg_act_grad_from_out(Elt<E>::ld(me + j), h.mask_act, h.mask_slope) * post[t]);
                *reinterpret_cast<uint4*>(dx + ((size_t)t * B + b) * K + k0) = ov;
            }
        }
    }

    // ---- 4) the batch's loss: the last ks = 0 workgroup to arrive adds the terms in the stand-alone kernels' order ----
    if (!first) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) flag = __hip_atomic_fetch_add(h.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag != (unsigned)(gridDim.x - 1)) return;
    if (tid == 0) __hip_atomic_store(h.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nterm = mode == 0 ? T : 3;
    for (int q = 0; q < nterm; ++q) {
        const int nthr = (mode == 1 && q == 2) ? 128 : 256;
        float acc = 0.f;
        if (tid < nthr)
            for (int i = tid; i < B; i += nthr) acc += __hip_atomic_load(terms + (size_t)q * B + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the stand-alone kernels run 256 (affine: 128) threads: their block_sum adds the waves in order, zeros here for the extra waves
        const float tot = block_sum(acc, sm);
        if (tid == 0) {
            float add;
            if (mode == 0) add = h.scale[q] * tot / (float)B;
            else if (q == 0) add = h.lcon * tot / (float)(B * h.n_cont);
            else if (q == 1) add = h.lcat * tot / (float)B;
            else add = h.laff * tot / (float)(B * 5);
            h.loss[0] += add;
        }
        __syncthreads();
    }
}

extern "C" int eg_head_fused_ok(int dtype, int T, int K, int N) {
    return T >= 1 && T <= 3 && N == 19 && K > 0 && K % (dtype == EG_F32 ? 4 : 8) == 0;
}

template <typename E>
static int launch_head_fused(const eg_head& h, hipStream_t st) {
    // K ranges of the input gradient: at least one workgroup per CU (a sample's range streams the whole N x K panel: B = 128 alone is half a chip)
    int ks = 1;
    while (h.B * ks < 256 && ks < 8 && h.K / (ks * 2) >= 512 * Elt<E>::VEC) ks *= 2;
    const dim3 grid(h.B, ks);
    switch (h.T) {
        case 1: hipLaunchKernelGGL((head_fused_kernel<E, 1, 19>), grid, dim3(512), 0, st, h); break;
        case 2: hipLaunchKernelGGL((head_fused_kernel<E, 2, 19>), grid, dim3(512), 0, st, h); break;
        default: hipLaunchKernelGGL((head_fused_kernel<E, 3, 19>), grid, dim3(512), 0, st, h); break;
    }
    return 0;
}

extern "C" int eg_head_fused(int dtype, const eg_head* hp, eg_stream_t s) {
    EG_REQUIRE(hp, "eg_head_fused: null argument");
    const eg_head& h = *hp;
    EG_REQUIRE(h.x && h.wp && h.y && h.dout && h.dx && h.loss && h.terms && h.counter && h.partials && h.nslice > 0 && h.B > 0, "eg_head_fused: bad argument");
    EG_REQUIRE(eg_head_fused_ok(dtype, h.T, h.K, h.N) && h.Kpad >= h.K, "eg_head_fused: unsupported head (N = 19, T <= 3)");
    EG_REQUIRE(h.mode == 0 || (h.mode == 1 && h.T == 3 && h.code && h.labels && h.n_cat <= EG_MAXCAT && h.n_cont >= 5 &&
                               h.c_cont + h.n_cont + h.n_cat <= h.N), "eg_head_fused: bad loss description");
    if (dtype == EG_F32) launch_head_fused<float>(h, (hipStream_t)s);
    else if (dtype == EG_F16) launch_head_fused<f16_t>(h, (hipStream_t)s);
    else launch_head_fused<bf16_t>(h, (hipStream_t)s);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_loss_affine_rpqxy(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc, float scale,
                                    float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s) {
    EG_REQUIRE(o_real && o_trans && code && B > 0, "eg_loss_affine_rpqxy: bad argument");
    hipLaunchKernelGGL(affine_reg_rpqxy_kernel, dim3(1), dim3(128), 0, (hipStream_t)s, o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real,
                       d_trans, pred_out);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// MNIST: theta from 7 codes, and the affine-consistency loss through the frozen MLP approximator
// (MNIST/utils_rpqmnxy.py:12-43,117-134).  One block (256 threads) per sample; hidden width 256.
// ------------------------------------------------------------------------------------------------------------------------
__global__ void theta_rpqmnxy_kernel(const float* __restrict__ code, int ldc, int B, float* __restrict__ theta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float c[7];
    for (int i = 0; i < 7; ++i) c[i] = code[(size_t)b * ldc + i];
    const Aff<float> m = matrix_rpqmnxy<float>(c);
    float* t = theta + (size_t)b * 6;
    t[0] = m.a; t[1] = m.b; t[2] = m.c; t[3] = m.d; t[4] = m.e; t[5] = m.f;
}
extern "C" int eg_theta_rpqmnxy(const float* code, int ldc, int B, float* theta, eg_stream_t s) {
    EG_REQUIRE(code && theta && ldc >= 7, "eg_theta_rpqmnxy: bad argument");
    hipLaunchKernelGGL(theta_rpqmnxy_kernel, dim3(cdiv(B, 128)), dim3(128), 0, (hipStream_t)s, code, ldc, B, theta);
    EG_LAUNCH_CHECK();
    return 0;
}

#define MLP_H 256
// float offsets inside the mlp blob
#define MLP_W1 0
#define MLP_B1 (MLP_W1 + MLP_H * 6)
#define MLP_W2 (MLP_B1 + MLP_H)
#define MLP_B2 (MLP_W2 + MLP_H * MLP_H)
#define MLP_W3 (MLP_B2 + MLP_H)
#define MLP_B3 (MLP_W3 + MLP_H * MLP_H)
#define MLP_W4 (MLP_B3 + MLP_H)
#define MLP_B4 (MLP_W4 + MLP_H * MLP_H)
#define MLP_W5 (MLP_B4 + MLP_H)
#define MLP_B5 (MLP_W5 + 7 * MLP_H)
#define MLP_W2T (MLP_B5 + 8)
#define MLP_W3T (MLP_W2T + MLP_H * MLP_H)
#define MLP_W4T (MLP_W3T + MLP_H * MLP_H)
#define MLP_TOTAL (MLP_W4T + MLP_H * MLP_H)

extern "C" size_t eg_mlp_rpqmnxy_floats(void) { return MLP_TOTAL; }

__device__ __forceinline__ float lrelu01(float v) { return v > 0.f ? v : 0.01f * v; }

__global__ __launch_bounds__(256) void affine_reg_rpqmnxy_kernel(const float* __restrict__ o_real, const float* __restrict__ o_trans, int ld, int c0,
                                                                 int B, const float* __restrict__ code, int ldc, const float* __restrict__ mlp,
                                                                 float scale, float* __restrict__ sample_loss, float* __restrict__ d_real,
                                                                 float* __restrict__ d_trans, float* __restrict__ pred_out) {
    __shared__ float h[5][MLP_H];      // h[0][0..5] = input, h[1..4] = hidden activations (post LeakyReLU)
    __shared__ float dl[2][MLP_H];     // back-propagated deltas (ping-pong)
    __shared__ float jac[6][14];       // d flat6 / d (real7, trans7)
    __shared__ float out7[8], dout7[8], dx6[8];
    __shared__ float sm[16];
    const int b = blockIdx.x, j = threadIdx.x;
    // forward-mode Jacobian, one derivative component per lane (Dual<1> on 14 lanes instead of Dual<14> on one: the same arithmetic per
    // component, 1/14 of the serial chain every other thread of the workgroup waits for; 110 -> see profiles/r03_zx_mnist_affine.txt)
    if (j < 14) {
        Dual<1> rc[7], tc[7], flat[6];
        for (int i = 0; i < 7; ++i) {
            rc[i].v = o_real[(size_t)b * ld + c0 + i]; rc[i].d[0] = j == i ? 1.f : 0.f;
            tc[i].v = o_trans[(size_t)b * ld + c0 + i]; tc[i].d[0] = j == 7 + i ? 1.f : 0.f;
        }
        relative_rpqmnxy<Dual<1>>(rc, tc, flat);
        for (int i = 0; i < 6; ++i) {
            if (j == 0) h[0][i] = flat[i].v;
            jac[i][j] = flat[i].d[0];
        }
    }
    __syncthreads();
    // forward
    {
        float a = mlp[MLP_B1 + j];
        for (int i = 0; i < 6; ++i) a += mlp[MLP_W1 + j * 6 + i] * h[0][i];
        h[1][j] = lrelu01(a);
    }
    __syncthreads();
    const int wt[3] = {MLP_W2T, MLP_W3T, MLP_W4T}, bo[3] = {MLP_B2, MLP_B3, MLP_B4};
    for (int l = 0; l < 3; ++l) {
        float a = mlp[bo[l] + j];
        const float* w = mlp + wt[l];
        for (int i = 0; i < MLP_H; ++i) a += w[i * MLP_H + j] * h[l + 1][i];
        h[l + 2][j] = lrelu01(a);
        __syncthreads();
    }
    if (j < 7) {
        float a = mlp[MLP_B5 + j];
        for (int i = 0; i < MLP_H; ++i) a += mlp[MLP_W5 + j * MLP_H + i] * h[4][i];
        // affine parameters -> latent units (utils_rpqmnxy.py:66-84)
        float lat, dlat;
        if (j == 0) { lat = a * (9.f / EG_PI_F); dlat = 9.f / EG_PI_F; }
        else if (j <= 2) { lat = (a - 1.f) / 0.2f; dlat = 1.f / 0.2f; }
        else if (j <= 4) { lat = a / 0.2f; dlat = 1.f / 0.2f; }
        else { lat = a / 0.1f; dlat = 1.f / 0.1f; }
        const float d = lat - code[(size_t)b * ldc + j];
        out7[j] = d * d;
        dout7[j] = 2.f * scale / (float)(B * 7) * d * dlat;
        if (pred_out) pred_out[(size_t)b * 7 + j] = lat;
    }
    __syncthreads();
    if (j == 0) {
        float a = 0.f;
        for (int i = 0; i < 7; ++i) a += out7[i];
        sample_loss[b] = a;
    }
    if (!d_real) return;
    // backward through the frozen MLP (input gradient only)
    {
        float a = 0.f;
        for (int i = 0; i < 7; ++i) a += mlp[MLP_W5 + i * MLP_H + j] * dout7[i];
        dl[0][j] = a * (h[4][j] > 0.f ? 1.f : 0.01f);
    }
    __syncthreads();
    const int wo[3] = {MLP_W4, MLP_W3, MLP_W2};
    for (int l = 0; l < 3; ++l) {
        const float* w = mlp + wo[l];
        float a = 0.f;
        for (int i = 0; i < MLP_H; ++i) a += w[i * MLP_H + j] * dl[l & 1][i];
        dl[(l + 1) & 1][j] = a * (h[3 - l][j] > 0.f ? 1.f : 0.01f);
        __syncthreads();
    }
    // dl[1] holds delta of layer 1 (3 ping-pong steps); d input_i = sum_j W1[j][i] * delta1[j]
    for (int i = 0; i < 6; ++i) {
        const float tot = block_sum(mlp[MLP_W1 + j * 6 + i] * dl[1][j], sm);
        if (j == 0) dx6[i] = tot;
    }
    __syncthreads();
    if (j < 14) {
        float g = 0.f;
        for (int i = 0; i < 6; ++i) g += dx6[i] * jac[i][j];
        if (j < 7) d_real[(size_t)b * ld + c0 + j] = g;
        else d_trans[(size_t)b * ld + c0 + (j - 7)] = g;
    }
}

__global__ void affine_reg_finish_kernel(const float* __restrict__ sample_loss, int B, int ncode, float scale, float* loss) {
    __shared__ float sm[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) a += sample_loss[i];
    const float tot = block_sum(a, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)(B * ncode);
}

__global__ void zero_rows_kernel(float* a, float* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0.f; b[i] = 0.f; }
}

/* ws: B floats */
extern "C" int eg_loss_affine_rpqmnxy(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc,
                                      const float* mlp, float scale, float* loss, float* d_real, float* d_trans, float* pred_out, float* ws,
                                      eg_stream_t s) {
    EG_REQUIRE(o_real && o_trans && code && mlp && ws && B > 0 && (!d_real == !d_trans), "eg_loss_affine_rpqmnxy: bad argument");
    hipStream_t st = (hipStream_t)s;
    if (d_real) hipLaunchKernelGGL(zero_rows_kernel, dim3(cdiv(B * ld, 256)), dim3(256), 0, st, d_real, d_trans, B * ld);
    hipLaunchKernelGGL(affine_reg_rpqmnxy_kernel, dim3(B), dim3(256), 0, st, o_real, o_trans, ld, c0, B, code, ldc, mlp, scale, ws, d_real, d_trans, pred_out);
    hipLaunchKernelGGL(affine_reg_finish_kernel, dim3(1), dim3(256), 0, st, ws, B, 7, scale, loss);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// dSprites: theta kernels (dSprites/utils_rp.py:38-59,94-115; utils_pxy.py:69-87 with the torch.inverse of rp.py:376 folded
// in: the alignment matrix is a pure translation T(x,y), so its inverse is T(-x,-y)), closed-form affine regulariser and
// mutual_info_loss (rp.py:225-232).
// ------------------------------------------------------------------------------------------------------------------------
__global__ void theta_rp_kernel(const float* __restrict__ code, int ldc, int B, float* __restrict__ theta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float c[4];
    for (int i = 0; i < 4; ++i) c[i] = code[(size_t)b * ldc + i];
    const Aff<float> m = matrix_rp<float>(c);
    float* t = theta + (size_t)b * 6;
    t[0] = m.a; t[1] = m.b; t[2] = m.c; t[3] = m.d; t[4] = m.e; t[5] = m.f;
}
extern "C" int eg_theta_rp(const float* code, int ldc, int B, float* theta, eg_stream_t s) {
    EG_REQUIRE(code && theta && ldc >= 4, "eg_theta_rp: bad argument");
    hipLaunchKernelGGL(theta_rp_kernel, dim3(cdiv(B, 128)), dim3(128), 0, (hipStream_t)s, code, ldc, B, theta);
    EG_LAUNCH_CHECK();
    return 0;
}

// theta of inverse(get_matrix_pxy_align(code)) rows 0,1: code = (p, x, y) latent units, x,y scaled by 0.1 (p is not used by the align matrix)
__global__ void theta_pxy_align_inv_kernel(const float* __restrict__ code, int ldc, int B, float* __restrict__ theta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* t = theta + (size_t)b * 6;
    t[0] = 1.f; t[1] = 0.f; t[2] = -(code[(size_t)b * ldc + 1] * 0.1f);
    t[3] = 0.f; t[4] = 1.f; t[5] = -(code[(size_t)b * ldc + 2] * 0.1f);
}
extern "C" int eg_theta_pxy_align_inv(const float* code, int ldc, int B, float* theta, eg_stream_t s) {
    EG_REQUIRE(code && theta && ldc >= 3, "eg_theta_pxy_align_inv: bad argument");
    hipLaunchKernelGGL(theta_pxy_align_inv_kernel, dim3(cdiv(B, 128)), dim3(128), 0, (hipStream_t)s, code, ldc, B, theta);
    EG_LAUNCH_CHECK();
    return 0;
}

// One derivative component per lane: eight lanes per sample run regularizer_rp on Dual<1> (the value part eight times, one column of the
// Jacobian each) instead of one thread per sample on Dual<8> -- the same arithmetic per component, 1/5 of the serial chain (the kernel sits
// on the step's main chain between the encoder's forward and backward: 10 us at B = 128, 40 at B = 512 as one thread per sample).
__global__ __launch_bounds__(1024) void affine_reg_rp_kernel(const float* __restrict__ o_real, const float* __restrict__ o_trans, int ld, int c0, int B,
                                                             const float* __restrict__ code, int ldc, float scale, float* loss, float* __restrict__ d_real,
                                                             float* __restrict__ d_trans, float* __restrict__ pred_out) {
    __shared__ float sm[16];
    float acc = 0.f;
    const float gs = 2.f * scale / (float)(B * 4);
    const int comp = threadIdx.x & 7;                    // Jacobian column: real code 0..3, transformed code 0..3
    for (int b = threadIdx.x >> 3; b < B; b += blockDim.x >> 3) {
        Dual<1> rc[4], tc[4], out[4];
        for (int i = 0; i < 4; ++i) {
            rc[i].v = o_real[(size_t)b * ld + c0 + i]; rc[i].d[0] = comp == i ? 1.f : 0.f;
            tc[i].v = o_trans[(size_t)b * ld + c0 + i]; tc[i].d[0] = comp == 4 + i ? 1.f : 0.f;
        }
        regularizer_rp<Dual<1>>(rc, tc, out);
        float gr = 0.f;
        for (int j = 0; j < 4; ++j) {
            const float d = out[j].v - code[(size_t)b * ldc + j];
            if (comp == 0) {
                acc += d * d;
                if (pred_out) pred_out[(size_t)b * 4 + j] = out[j].v;
            }
            gr += gs * d * out[j].d[0];
        }
        if (d_real && d_trans) {
            float* dst = comp < 4 ? d_real : d_trans;
            for (int j = comp; j < ld; j += 8) { d_real[(size_t)b * ld + j] = 0.f; d_trans[(size_t)b * ld + j] = 0.f; }
            __builtin_amdgcn_wave_barrier();            // (the eight lanes of a sample sit in one wave: the zeros above are ordered before the entries below)
            dst[(size_t)b * ld + c0 + (comp & 3)] = gr;
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)(B * 4);
}
extern "C" int eg_loss_affine_rp(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc, float scale,
                                 float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s) {
    EG_REQUIRE(o_real && o_trans && code && B > 0, "eg_loss_affine_rp: bad argument");
    static const int cap = [] { const char* e = getenv("EG_AFFINE_THREADS"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 1024 ? v : 1024; }();
    const int threads = B * 8 >= cap ? cap : ((B * 8 + 63) / 64) * 64;
    hipLaunchKernelGGL(affine_reg_rp_kernel, dim3(1), dim3(threads), 0, (hipStream_t)s, o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out);
    EG_LAUNCH_CHECK();
    return 0;
}

// mutual_info_loss(softmax(o), c) = mean_b( -sum_j log(p_j + eps) c_j ) + mean_b( -sum_j log(c_j + eps) c_j ),  eps = 1e-8.
// target: probabilities tgt[b][j] (target_logits == 0) or softmax of logits tgt (target_logits != 0, treated as a constant).
// dout (+=) d loss / d o.
__global__ void mutual_info_kernel(const float* __restrict__ o, int ld, int c0, int n, int B, const float* __restrict__ tgt, int ldt, int t0,
                                   int target_logits, float scale, float* loss, float* __restrict__ dout) {
    __shared__ float sm[16];
    const float eps = 1e-8f;
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float p[EG_MAXCAT], c[EG_MAXCAT];
        float mx = -INFINITY, se = 0.f;
        for (int j = 0; j < n; ++j) mx = fmaxf(mx, o[(size_t)b * ld + c0 + j]);
        for (int j = 0; j < n; ++j) { p[j] = expf(o[(size_t)b * ld + c0 + j] - mx); se += p[j]; }
        for (int j = 0; j < n; ++j) p[j] /= se;
        if (target_logits) {
            float m2 = -INFINITY, s2 = 0.f;
            for (int j = 0; j < n; ++j) m2 = fmaxf(m2, tgt[(size_t)b * ldt + t0 + j]);
            for (int j = 0; j < n; ++j) { c[j] = expf(tgt[(size_t)b * ldt + t0 + j] - m2); s2 += c[j]; }
            for (int j = 0; j < n; ++j) c[j] /= s2;
        } else
            for (int j = 0; j < n; ++j) c[j] = tgt[(size_t)b * ldt + t0 + j];
        float dp[EG_MAXCAT], dot = 0.f;
        for (int j = 0; j < n; ++j) {
            acc += -logf(p[j] + eps) * c[j] - logf(c[j] + eps) * c[j];
            dp[j] = -c[j] / (p[j] + eps) * (scale / (float)B);
            dot += dp[j] * p[j];
        }
        if (dout)
            for (int j = 0; j < n; ++j) dout[(size_t)b * ld + c0 + j] += p[j] * (dp[j] - dot);
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)B;
}
extern "C" int eg_loss_mutual_info(const float* o, int ld, int c0, int n, int B, const float* tgt, int ldt, int t0, int target_logits, float scale,
                                   float* loss, float* dout, eg_stream_t s) {
    EG_REQUIRE(o && tgt && n <= EG_MAXCAT && B > 0, "eg_loss_mutual_info: bad argument");
    hipLaunchKernelGGL(mutual_info_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, o, ld, c0, n, B, tgt, ldt, t0, target_logits, scale, loss, dout);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// colored dSprites (colored_dSprites/rp_color.py:368-394,415-424; utils_rp_color.py:38-75,100-139; utils_pxy.py:48-57)
// ------------------------------------------------------------------------------------------------------------------------
// out[b][c][hw] = sprite_u8[b][hw] * gain[b][c]
__global__ void u8_colorize_kernel(const unsigned char* __restrict__ sp, const float* __restrict__ gain, float* __restrict__ out, int B, int C, int HW) {
    const size_t n = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        const size_t bc = i / HW;
        const int b = (int)(bc / C);
        out[i] = (float)sp[(size_t)b * HW + hw] * gain[bc];
    }
}
extern "C" int eg_u8_colorize(const unsigned char* sprites, const float* gain, float* out, int B, int C, int HW, eg_stream_t s) {
    EG_REQUIRE(sprites && gain && out, "eg_u8_colorize: null pointer");
    const size_t n = (size_t)B * C * HW;
    hipLaunchKernelGGL(u8_colorize_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, sprites, gain, out, B, C, HW);
    EG_LAUNCH_CHECK();
    return 0;
}

// out[b][c][hw] = in[b][c][hw] * g  or  / g,  g = code[b][c0 + c] * factor + 1
__global__ void color_scale_kernel(const float* __restrict__ in, const float* __restrict__ code, int ldc, int c0, float factor, int divide,
                                   float* __restrict__ out, int B, int C, int HW) {
    const size_t n = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = i / HW;
        const int b = (int)(bc / C), c = (int)(bc % C);
        const float g = code[(size_t)b * ldc + c0 + c] * factor + 1.f;
        out[i] = divide ? in[i] / g : in[i] * g;
    }
}
extern "C" int eg_color_scale(const float* in, const float* code, int ldc, int c0, float factor, int divide, float* out, int B, int C, int HW,
                              eg_stream_t s) {
    EG_REQUIRE(in && code && out, "eg_color_scale: null pointer");
    const size_t n = (size_t)B * C * HW;
    hipLaunchKernelGGL(color_scale_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, in, code, ldc, c0, factor, divide, out, B, C, HW);
    EG_LAUNCH_CHECK();
    return 0;
}

// affine (4 codes, as dSprites) + colour (3 codes): relative gain = (t*.5+1)/(r*.5+1) -> latent (g-1)/.5 ; MSE over 7 values
// (one thread per sample on Dual<14> stays: sixteen lanes per sample on Dual<1> repeat the value part -- sines, cosines, an arctangent --
//  sixteen times and need eight passes of a 1024-thread workgroup at B = 512: measured SLOWER in the step, 2.54 -> 2.57 ms,
//  profiles/r03_zzk_ab_affine_color.txt; the 8-lane form of affine_reg_rp_kernel fits B = 128 in one pass and is faster there)
__global__ void affine_reg_rp_color_kernel(const float* __restrict__ o_real, const float* __restrict__ o_trans, int ld, int c0, int B,
                                           const float* __restrict__ code, int ldc, float scale, float* loss, float* __restrict__ d_real,
                                           float* __restrict__ d_trans, float* __restrict__ pred_out) {
    __shared__ float sm[16];
    float acc = 0.f;
    const float gs = 2.f * scale / (float)(B * 7);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        Dual<14> rc[7], tc[7], out[7];
        for (int i = 0; i < 7; ++i) {
            rc[i] = dvar<14>(o_real[(size_t)b * ld + c0 + i], i);
            tc[i] = dvar<14>(o_trans[(size_t)b * ld + c0 + i], 7 + i);
        }
        regularizer_rp<Dual<14>>(rc, tc, out);
        for (int j = 0; j < 3; ++j) out[4 + j] = ((tc[4 + j] * 0.5f + 1.f) / (rc[4 + j] * 0.5f + 1.f) - 1.f) / 0.5f;
        float gr[14];
        for (int i = 0; i < 14; ++i) gr[i] = 0.f;
        for (int j = 0; j < 7; ++j) {
            const float d = out[j].v - code[(size_t)b * ldc + j];
            acc += d * d;
            if (pred_out) pred_out[(size_t)b * 7 + j] = out[j].v;
            for (int i = 0; i < 14; ++i) gr[i] += gs * d * out[j].d[i];
        }
        if (d_real && d_trans) {
            for (int j = 0; j < ld; ++j) { d_real[(size_t)b * ld + j] = 0.f; d_trans[(size_t)b * ld + j] = 0.f; }
            for (int i = 0; i < 7; ++i) { d_real[(size_t)b * ld + c0 + i] = gr[i]; d_trans[(size_t)b * ld + c0 + i] = gr[7 + i]; }
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)(B * 7);
}
extern "C" int eg_loss_affine_rp_color(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc, float scale,
                                       float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s) {
    EG_REQUIRE(o_real && o_trans && code && B > 0, "eg_loss_affine_rp_color: bad argument");
    hipLaunchKernelGGL(affine_reg_rp_color_kernel, dim3(1), dim3(128), 0, (hipStream_t)s, o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Stage-1 trainer of Encoder_pxy (dSprites/pxy.py:156-191; dSprites/utils_pxy.py:24-66,107-126): code = (p, x, y) latent units,
// p = 1 + .1 c0, x = .1 c1, y = .1 c2, A = diag(p,p,1) @ Trans(x,y) = [[p,0,p x],[0,p,p y],[0,0,1]].
// relative = A_t A_r^-1 has rel00 = rel11 = p_t / p_r, rel02 = p_t (x_t - x_r), rel12 = p_t (y_t - y_r), hence
//   rec_p = p_t / p_r,  rec_x = rel02 / rec_p = p_r (x_t - x_r),  rec_y = p_r (y_t - y_r)   (closed form, gradients by hand).
// ------------------------------------------------------------------------------------------------------------------------
__global__ void theta_pxy_kernel(const float* __restrict__ code, int ldc, int B, float* __restrict__ theta) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float p = code[(size_t)b * ldc] * 0.1f + 1.f, x = code[(size_t)b * ldc + 1] * 0.1f, y = code[(size_t)b * ldc + 2] * 0.1f;
    float* t = theta + (size_t)b * 6;
    t[0] = p; t[1] = 0.f; t[2] = p * x;
    t[3] = 0.f; t[4] = p; t[5] = p * y;
}
extern "C" int eg_theta_pxy(const float* code, int ldc, int B, float* theta, eg_stream_t s) {
    EG_REQUIRE(code && theta && ldc >= 3 && B > 0, "eg_theta_pxy: bad argument");
    hipLaunchKernelGGL(theta_pxy_kernel, dim3(cdiv(B, 128)), dim3(128), 0, (hipStream_t)s, code, ldc, B, theta);
    EG_LAUNCH_CHECK();
    return 0;
}

// loss += scale * mean((affine_regularzier_pxy(real, trans) - code)^2);  d_real / d_trans = d loss / d codes ([B][ld], zero elsewhere)
// ncol (0 or 3) further code entries are colour gains g = 1 + .1 c (colored_dSprites/utils_pxy.py:150-176): rec_j = (g_t / g_r - 1) / .1
__global__ void affine_reg_pxy_kernel(const float* __restrict__ o_real, const float* __restrict__ o_trans, int ld, int c0, int B,
                                      const float* __restrict__ code, int ldc, float scale, float* loss, float* __restrict__ d_real,
                                      float* __restrict__ d_trans, float* __restrict__ pred_out, int ncol) {
    __shared__ float sm[16];
    float acc = 0.f;
    const int nd = 3 + ncol;
    const float gs = 2.f * scale / (float)(B * nd);
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* r = o_real + (size_t)b * ld + c0;
        const float* t = o_trans + (size_t)b * ld + c0;
        const float pr = r[0] * 0.1f + 1.f, xr = r[1] * 0.1f, yr = r[2] * 0.1f;
        const float pt = t[0] * 0.1f + 1.f, xt = t[1] * 0.1f, yt = t[2] * 0.1f;
        const float rp = pt / pr;
        const float out[3] = {(rp - 1.f) / 0.1f, pr * (xt - xr) / 0.1f, pr * (yt - yr) / 0.1f};
        float d[3];
        for (int j = 0; j < 3; ++j) {
            d[j] = out[j] - code[(size_t)b * ldc + j];
            acc += d[j] * d[j];
            if (pred_out) pred_out[(size_t)b * nd + j] = out[j];
        }
        if (d_real && d_trans)
            for (int j = 0; j < ld; ++j) { d_real[(size_t)b * ld + j] = 0.f; d_trans[(size_t)b * ld + j] = 0.f; }
        for (int j = 0; j < ncol; ++j) {
            const float gr = r[3 + j] * 0.1f + 1.f, gt = t[3 + j] * 0.1f + 1.f;
            const float o = (gt / gr - 1.f) / 0.1f;
            const float dj = o - code[(size_t)b * ldc + 3 + j];
            acc += dj * dj;
            if (pred_out) pred_out[(size_t)b * nd + 3 + j] = o;
            if (d_real && d_trans) {
                d_real[(size_t)b * ld + c0 + 3 + j] = gs * dj * (-gt / (gr * gr));
                d_trans[(size_t)b * ld + c0 + 3 + j] = gs * dj / gr;
            }
        }
        if (d_real && d_trans) {
            // out0 = (pt/pr - 1)/.1 : d/dc0r = -pt/pr^2 (.1/.1), d/dc0t = 1/pr;  out1 = pr (xt - xr)/.1 : d/dc0r = .1 (xt - xr)/.1,
            // d/dc1r = -pr, d/dc1t = pr;  out2 likewise with y
            d_real[(size_t)b * ld + c0 + 0] = gs * (d[0] * (-pt / (pr * pr)) + d[1] * (xt - xr) + d[2] * (yt - yr));
            d_real[(size_t)b * ld + c0 + 1] = gs * d[1] * (-pr);
            d_real[(size_t)b * ld + c0 + 2] = gs * d[2] * (-pr);
            d_trans[(size_t)b * ld + c0 + 0] = gs * d[0] / pr;
            d_trans[(size_t)b * ld + c0 + 1] = gs * d[1] * pr;
            d_trans[(size_t)b * ld + c0 + 2] = gs * d[2] * pr;
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0 && loss) loss[0] += scale * tot / (float)(B * nd);
}
extern "C" int eg_loss_affine_pxy(const float* o_real, const float* o_trans, int ld, int c0, int B, const float* code, int ldc, int ncol, float scale,
                                  float* loss, float* d_real, float* d_trans, float* pred_out, eg_stream_t s) {
    EG_REQUIRE(o_real && o_trans && code && B > 0 && (ncol == 0 || ncol == 3) && ld >= c0 + 3 + ncol && ldc >= 3 + ncol, "eg_loss_affine_pxy: bad argument");
    hipLaunchKernelGGL(affine_reg_pxy_kernel, dim3(1), dim3(128), 0, (hipStream_t)s, o_real, o_trans, ld, c0, B, code, ldc, scale, loss, d_real, d_trans, pred_out, ncol);
    EG_LAUNCH_CHECK();
    return 0;
}

// from_latent_vector_2_affine_para of the MNIST code (MNIST/approximate_rpqmnxy.py:43-60 == utils_rpqmnxy.py:46-63): the regression
// target of the approximator fit -- theta = c0 pi/9, p,q = 1 + .2 c, m,n = .2 c, x,y = .1 c
__global__ void affine_para_rpqmnxy_kernel(const float* __restrict__ code, int ldc, int B, float* __restrict__ para) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 7) return;
    const int b = i / 7, j = i - b * 7;
    const float c = code[(size_t)b * ldc + j];
    para[i] = j == 0 ? c * (3.14159265358979323846f / 9.f) : (j <= 2 ? c * 0.2f + 1.f : (j <= 4 ? c * 0.2f : c * 0.1f));
}
extern "C" int eg_affine_para_rpqmnxy(const float* code, int ldc, int B, float* para, eg_stream_t s) {
    EG_REQUIRE(code && para && ldc >= 7 && B > 0, "eg_affine_para_rpqmnxy: bad argument");
    hipLaunchKernelGGL(affine_para_rpqmnxy_kernel, dim3(cdiv(B * 7, 256)), dim3(256), 0, (hipStream_t)s, code, ldc, B, para);
    EG_LAUNCH_CHECK();
    return 0;
}
