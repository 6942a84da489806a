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
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, x, (bf16_t*)y, B, C, HW, Cpad);
    EG_LAUNCH_CHECK();
    return 0;
}
extern "C" int eg_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int HW, int Cpad, eg_stream_t s) {
    EG_REQUIRE(x && y && Cpad >= C, "eg_nhwc_to_nchw: bad argument");
    const size_t n = (size_t)B * C * HW;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == EG_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)x, y, B, C, HW, Cpad);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, y, B, C, HW, Cpad);
    EG_LAUNCH_CHECK();
    return 0;
}
