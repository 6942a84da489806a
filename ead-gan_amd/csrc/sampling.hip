// Image-grid writer of the sampling tools: what the reference gets from torchvision.utils.make_grid / save_image
// (third-party, not vendored; the release pairing torch 1.7.1 is 0.8.2) at its call sites MNIST/EAD-GAN_rpqmnxy.py:281-330,
// MNIST/generate_image.py:122-138, celebA/EAD-GAN_celebA.py:238-287, celebA/gen_imgs.py:183-199, dSprites/rp.py:299-353,
// colored_dSprites/rp_color.py:297-353.  Three byte/float passes, all HBM-trivial (a 10x10 grid of 64x64 RGB is 1.3 MB):
//   eg_make_grid     [B,C,H,W] fp32 -> [3 or C][ymaps*(H+pad)+pad][xmaps*(W+pad)+pad] fp32, single-channel images replicated to 3,
//                    optionally normalising the images (not the gaps) to a device-resident range while tiling
//   eg_minmax_f32    min and max of a tensor (the normalize=True range)
//   eg_quantize_u8   (v - lo) / (hi - lo + 1e-5) -> *255 + 0.5 -> clamp -> uint8, CHW -> HWC rows ready for the PNG encoder
// Each arithmetic step is a separately rounded fp32 operation (no fma contraction), so the bytes equal the CPU restatement's.
#include "eg_common.h"

static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// norm_ip: clamp_(min, max).add_(-min).div_(max - min + 1e-5), the divisor formed in double and rounded once
__device__ __forceinline__ float norm_ip(float v, float lo, float hi) {
    const float d = (float)((double)hi - (double)lo + 1e-5);
    return __fdiv_rn(__fadd_rn(fminf(fmaxf(v, lo), hi), -lo), d);
}

__global__ void make_grid_kernel(const float* __restrict__ img, int B, int C, int H, int W, int xmaps, int pad, float pad_value,
                                 const float* __restrict__ range, int Cg, int Hg, int Wg, float* __restrict__ grid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)Cg * Hg * Wg) return;
    const int x = (int)(i % Wg), y = (int)((i / Wg) % Hg), c = (int)(i / ((size_t)Wg * Hg));
    const int ch = H + pad, cw = W + pad;
    const int gy = y / ch, gx = x / cw, iy = y - gy * ch - pad, ix = x - gx * cw - pad;
    const int k = gy * xmaps + gx;
    float v = pad_value;
    if (iy >= 0 && ix >= 0 && gx < xmaps && k < B && iy < H && ix < W) {
        v = img[(((size_t)k * C + (C == 1 ? 0 : c)) * H + iy) * W + ix];
        if (range) v = norm_ip(v, range[0], range[1]);          // make_grid(normalize=True): images only, the gaps keep pad_value
    }
    grid[i] = v;
}

extern "C" int eg_make_grid(const float* img, int B, int C, int H, int W, int nrow, int padding, float pad_value, const float* range, float* grid, eg_stream_t s) {
    EG_REQUIRE(img && grid && B > 0 && C > 0 && H > 0 && W > 0 && nrow > 0 && padding >= 0, "eg_make_grid: bad argument");
    const int xmaps = nrow < B ? nrow : B, ymaps = (B + xmaps - 1) / xmaps;
    const int Cg = C == 1 ? 3 : C, Hg = ymaps * (H + padding) + padding, Wg = xmaps * (W + padding) + padding;
    const size_t n = (size_t)Cg * Hg * Wg;
    hipLaunchKernelGGL(make_grid_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, (hipStream_t)s, img, B, C, H, W, xmaps, padding, pad_value, range, Cg, Hg, Wg, grid);
    EG_LAUNCH_CHECK();
    return 0;
}

#define EG_MM_BLOCKS 256
__device__ __forceinline__ void block_minmax(float& lo, float& hi, float* sm /* 32 floats */) {
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = lo; sm[16 + w] = hi; }
    __syncthreads();
    lo = sm[0]; hi = sm[16];
    for (int i = 1; i < nw; ++i) { lo = fminf(lo, sm[i]); hi = fmaxf(hi, sm[16 + i]); }
    __syncthreads();
}

__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ part) {
    __shared__ float sm[32];
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    block_minmax(lo, hi, sm);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = lo; part[2 * blockIdx.x + 1] = hi; }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
    __shared__ float sm[32];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += 256) { lo = fminf(lo, part[2 * i]); hi = fmaxf(hi, part[2 * i + 1]); }
    block_minmax(lo, hi, sm);
    if (threadIdx.x == 0) { out[0] = lo; out[1] = hi; }
}

extern "C" size_t eg_minmax_ws_floats(void) { return 2 * EG_MM_BLOCKS; }
extern "C" int eg_minmax_f32(const float* x, size_t n, float* ws, float* out2, eg_stream_t s) {
    EG_REQUIRE(x && ws && out2 && n > 0, "eg_minmax_f32: bad argument");
    const int nb = (int)(cdivz(n, 256) < EG_MM_BLOCKS ? cdivz(n, 256) : EG_MM_BLOCKS);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, x, n, ws);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, ws, nb, out2);
    EG_LAUNCH_CHECK();
    return 0;
}

__global__ void quantize_u8_kernel(const float* __restrict__ x, int C, int H, int W, const float* __restrict__ range, unsigned char* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * H * W) return;
    const int c = (int)(i % C), xw = (int)((i / C) % W), y = (int)(i / ((size_t)C * W));
    float v = x[((size_t)c * H + y) * W + xw];
    if (range) v = norm_ip(v, range[0], range[1]);
    v = __fadd_rn(__fmul_rn(v, 255.f), 0.5f);       // save_image: mul(255).add_(0.5).clamp_(0, 255).to(uint8)
    v = fminf(fmaxf(v, 0.f), 255.f);
    out[i] = (unsigned char)v;
}

extern "C" int eg_quantize_u8(const float* x, int C, int H, int W, const float* range, unsigned char* out, eg_stream_t s) {
    EG_REQUIRE(x && out && C > 0 && H > 0 && W > 0, "eg_quantize_u8: bad argument");
    const size_t n = (size_t)C * H * W;
    hipLaunchKernelGGL(quantize_u8_kernel, dim3((unsigned)cdivz(n, 256)), dim3(256), 0, (hipStream_t)s, x, C, H, W, range, out);
    EG_LAUNCH_CHECK();
    return 0;
}
