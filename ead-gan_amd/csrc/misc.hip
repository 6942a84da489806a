// Error plumbing, version, small utility kernels (concat/cast, elementwise activation gradients).
#include <stdarg.h>

#include "eg_common.h"

static thread_local char g_err[512] = "";

void eg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* eg_last_error(void) { return g_err; }

/* drains HIP's sticky last-error slot (e.g. hipErrorStreamCaptureInvalidated after a hipGraph capture that a collective refused);
 * returns how many errors were pending.  Callers that fall back from graph replay to eager launches call this once. */
extern "C" int eg_clear_errors(void) {
    int n = 0;
    while (hipGetLastError() != hipSuccess && n < 16) ++n;
    return n;
}
extern "C" int eg_version(void) { return 100; }

// out[b][0:wa|wa:wa+wb|..] = cast(a|b|c), zero padded to Cpad  (generator input, celebA/EAD-GAN_celebA.py:97)
template <typename T>
__global__ void concat_cast_kernel(const float* a, int wa, const float* b, int wb, const float* c, int wc, int B, int Cpad, T* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Cpad) return;
    const int r = i / Cpad, j = i % Cpad;
    float v = 0.f;
    if (j < wa) v = a[r * wa + j];
    else if (j < wa + wb) v = b[r * wb + (j - wa)];
    else if (j < wa + wb + wc) v = c[r * wc + (j - wa - wb)];
    Elt<T>::st(out + i, v);
}

extern "C" int eg_concat_cast(int dtype, const float* a, int wa, const float* b, int wb, const float* c, int wc, int B, int Cpad,
                              void* out, eg_stream_t s) {
    EG_REQUIRE(out && B > 0 && Cpad >= wa + wb + wc, "eg_concat_cast: bad argument");
    const int n = B * Cpad;
    if (dtype == EG_F32) hipLaunchKernelGGL(concat_cast_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, a, wa, b, wb, c, wc, B, Cpad, (float*)out);
    else if (dtype == EG_F16) hipLaunchKernelGGL(concat_cast_kernel<f16_t>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, a, wa, b, wb, c, wc, B, Cpad, (f16_t*)out);
    else hipLaunchKernelGGL(concat_cast_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, a, wa, b, wb, c, wc, B, Cpad, (bf16_t*)out);
    EG_LAUNCH_CHECK();
    return 0;
}

// out = g * act'(a)   (fp32 image-side tensors, e.g. tanh backward of the generator output)
__global__ void act_grad_mul_kernel(const float* g, const float* a, float* out, size_t n, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = g[i] * eg_act_grad_from_out(a[i], act, slope);
}

extern "C" int eg_act_grad_mul_f32(const float* g, const float* a, float* out, size_t n, int act, float slope, eg_stream_t s) {
    EG_REQUIRE(g && a && out, "eg_act_grad_mul_f32: null pointer");
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(act_grad_mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, g, a, out, n, act, slope);
    EG_LAUNCH_CHECK();
    return 0;
}

// layout / dtype conversion between NCHW fp32 (reference tensors) and NHWC dtype-T (internal)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int B, int C, int HW, int Cpad) {
    const size_t n = (size_t)B * HW * Cpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const size_t p = i / Cpad;
        const int hw = (int)(p % HW), b = (int)(p / HW);
        Elt<T>::st(y + i, c < C ? x[((size_t)b * C + c) * HW + hw] : 0.f);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, int B, int C, int HW, int Cpad) {
    const size_t n = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        const size_t p = i / HW;
        const int c = (int)(p % C), b = (int)(p / C);
        y[i] = Elt<T>::ld(x + ((size_t)b * HW + hw) * Cpad + c);
    }
}

extern "C" int eg_nchw_to_nhwc(int dtype, const float* x, void* y, int B, int C, int HW, int Cpad, eg_stream_t s) {
    EG_REQUIRE(x && y && Cpad >= C, "eg_nchw_to_nhwc: bad argument");
    const size_t n = (size_t)B * HW * Cpad;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == EG_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, x, (float*)y, B, C, HW, Cpad);
    else if (dtype == EG_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, x, (f16_t*)y, B, C, HW, Cpad);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, x, (bf16_t*)y, B, C, HW, Cpad);
    EG_LAUNCH_CHECK();
    return 0;
}
extern "C" int eg_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int HW, int Cpad, eg_stream_t s) {
    EG_REQUIRE(x && y && Cpad >= C, "eg_nhwc_to_nchw: bad argument");
    const size_t n = (size_t)B * C * HW;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == EG_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)x, y, B, C, HW, Cpad);
    else if (dtype == EG_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const f16_t*)x, y, B, C, HW, Cpad);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, y, B, C, HW, Cpad);
    EG_LAUNCH_CHECK();
    return 0;
}

// out[i] += src[(i / div) * s_div + (i % div) * s_mod]
__global__ void gather_add_kernel(float* __restrict__ out, const float* __restrict__ src, int n, int div, int s_div, int s_mod) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += src[(i / div) * s_div + (i % div) * s_mod];
}
extern "C" int eg_gather_add(float* out, const float* src, int n, int div, int s_div, int s_mod, eg_stream_t s) {
    EG_REQUIRE(out && src && div > 0, "eg_gather_add: bad argument");
    hipLaunchKernelGGL(gather_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, out, src, n, div, s_div, s_mod);
    EG_LAUNCH_CHECK();
    return 0;
}

// y[b][h][w][:] = sum of the 2x2 block of x[b][2h..2h+1][2w..2w+1][:]   (nearest-upsample backward), 16-byte vectors along C
template <typename T>
__global__ void sumpool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int VEC = Elt<T>::VEC;
    const int cpr = C / VEC;
    const size_t total = (size_t)B * H * W * cpr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        const size_t pix = i / cpr;
        const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const size_t o = ((((size_t)b * 2 * H + 2 * h + dy) * 2 * W) + 2 * w + dx) * C + (size_t)ch * VEC;
                const uint4 v = *reinterpret_cast<const uint4*>(x + o);
                const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += Elt<T>::ld(e + j);
            }
        uint4 ov;
        T* oe = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int j = 0; j < VEC; ++j) Elt<T>::st(oe + j, acc[j]);
        *reinterpret_cast<uint4*>(y + pix * C + (size_t)ch * VEC) = ov;
    }
}
extern "C" int eg_sumpool2x2(int dtype, const void* x, void* y, int B, int H, int W, int C, eg_stream_t s) {
    EG_REQUIRE(x && y && C % (dtype == EG_F32 ? 4 : 8) == 0, "eg_sumpool2x2: bad argument");
    const size_t total = (size_t)B * H * W * (C / (dtype == EG_F32 ? 4 : 8));
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == EG_F32) hipLaunchKernelGGL(sumpool2x2_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)x, (float*)y, B, H, W, C);
    else if (dtype == EG_F16) hipLaunchKernelGGL(sumpool2x2_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const f16_t*)x, (f16_t*)y, B, H, W, C);
    else hipLaunchKernelGGL(sumpool2x2_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
    EG_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2) [nearest] + Conv2d(Cin -> Cout, 3, 1, 1)  ==  ConvTranspose2d(Cin -> Cout, 4, 2, 1) with summed taps
// (MNIST/EAD-GAN_rpqmnxy.py:81-82, 85-86).  Output row 2y + py of the 3x3 convolution over the upsampled image reads upsampled rows
// 2y + py - 1 .. + 1 = input rows {y-1 (tap 0), y (taps 1, 2)} for py = 0 and {y (taps 0, 1), y+1 (tap 2)} for py = 1; a transposed 4x4 / stride
// 2 / pad 1 convolution sends input row y to output rows 2y - 1 + kh.  Hence W4[kh] = sum of the 3x3 taps t with E[kh][t] = 1:
//     kh 0: {2}    kh 1: {1, 2}    kh 2: {0, 1}    kh 3: {0}           (t from max(0, 2 - kh) to min(2, 3 - kh)), rows and columns alike;
// the zero padding of the 3x3 convolution is the transposed convolution's missing input row.  4 taps per output pixel instead of 9:
// 2.25 x fewer multiply-adds in the forward, the input gradient (which arrives at the LOW resolution: no sum-pool) and the weight
// gradient, whose 4x4 result goes back through the transpose of the same map.  Same values up to fp32 rounding of the tap sums.
// ------------------------------------------------------------------------------------------------
// w4t[i][o][kh][kw] = sum_{ty, tx} E[kh][ty] E[kw][tx] w3[o][i][ty][tx]      (ConvTranspose2d master layout [in][out][4][4])
__global__ void up3_expand_kernel(const float* __restrict__ w3, float* __restrict__ w4t, int Cout, int Cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cin * Cout * 16) return;
    const int kw = idx & 3, kh = (idx >> 2) & 3, o = (idx >> 4) % Cout, i = (idx >> 4) / Cout;
    const float* __restrict__ w = w3 + ((size_t)o * Cin + i) * 9;
    float acc = 0.f;
    for (int ty = max(0, 2 - kh); ty <= min(2, 3 - kh); ++ty)
        for (int tx = max(0, 2 - kw); tx <= min(2, 3 - kw); ++tx) acc = __fadd_rn(acc, w[ty * 3 + tx]);
    w4t[idx] = acc;
}
// dw3[o][i][ty][tx] (+)= sum_{kh, kw} E[kh][ty] E[kw][tx] dw4t[i][o][kh][kw]     (kh from 2 - ty to 3 - ty)
__global__ void up3_contract_kernel(const float* __restrict__ dw4t, float* __restrict__ dw3, int Cout, int Cin, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin * 9) return;
    const int t = idx % 9, ty = t / 3, tx = t - 3 * ty, i = (idx / 9) % Cin, o = (idx / 9) / Cin;
    const float* __restrict__ g = dw4t + ((size_t)i * Cout + o) * 16;
    float acc = 0.f;
    for (int kh = 2 - ty; kh <= 3 - ty; ++kh)
        for (int kw = 2 - tx; kw <= 3 - tx; ++kw) acc = __fadd_rn(acc, g[kh * 4 + kw]);
    dw3[idx] = accumulate ? __fadd_rn(dw3[idx], acc) : acc;
}
extern "C" int eg_up3_expand(const float* w3, float* w4t, int Cout, int Cin, eg_stream_t s) {
    EG_REQUIRE(w3 && w4t && Cout > 0 && Cin > 0, "eg_up3_expand: bad argument");
    hipLaunchKernelGGL(up3_expand_kernel, dim3(cdiv(Cin * Cout * 16, 256)), dim3(256), 0, (hipStream_t)s, w3, w4t, Cout, Cin);
    EG_LAUNCH_CHECK();
    return 0;
}
extern "C" int eg_up3_contract(const float* dw4t, float* dw3, int Cout, int Cin, int accumulate, eg_stream_t s) {
    EG_REQUIRE(dw4t && dw3 && Cout > 0 && Cin > 0, "eg_up3_contract: bad argument");
    hipLaunchKernelGGL(up3_contract_kernel, dim3(cdiv(Cout * Cin * 9, 256)), dim3(256), 0, (hipStream_t)s, dw4t, dw3, Cout, Cin, accumulate);
    EG_LAUNCH_CHECK();
    return 0;
}

__global__ void add_f32_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
extern "C" int eg_add_f32(float* out, const float* a, const float* b, size_t n, eg_stream_t s) {
    EG_REQUIRE(out && a && b, "eg_add_f32: null pointer");
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(add_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out, a, b, n);
    EG_LAUNCH_CHECK();
    return 0;
}
__global__ void u8_to_f32_kernel(const unsigned char* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
extern "C" int eg_u8_to_f32(const unsigned char* x, float* y, size_t n, eg_stream_t s) {
    EG_REQUIRE(x && y, "eg_u8_to_f32: null pointer");
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, x, y, n);
    EG_LAUNCH_CHECK();
    return 0;
}
