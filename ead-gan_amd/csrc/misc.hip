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
