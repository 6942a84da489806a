// Device-side input pipeline of the train loops: replaces the host work the reference does per iteration --
// DataLoader + PIL transforms (celebA/EAD-GAN_celebA.py:194-206: RandomHorizontalFlip, ToTensor, Normalize(0.5, 0.5); Resize /
// CenterCrop happen once when the uint8 dataset is put on the device) and the numpy draws of z, code and labels (:308-317;
// MNIST/EAD-GAN_rpqmnxy.py:351-357) -- by a counter-based generator (Philox4x32-10) keyed by (seed, stream id) and counted by
// (element, DEVICE step counter), so that a captured hipGraph draws fresh, reproducible inputs on every replay with no host work.
// These are new draws with the reference's DISTRIBUTIONS (numpy's Mersenne-Twister stream cannot be reproduced on the device; parity
// tests keep feeding the host draws through load_inputs).
#include <algorithm>

#include "eg_common.h"

struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ static void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ __forceinline__ void operator()(uint32_t (&c)[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round(c, a, b);
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
    }
};

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1), 24 bits

// Epoch-wise shuffling without replacement (DataLoader(shuffle=True), celebA/EAD-GAN_celebA.py:204-206; MNIST/EAD-GAN_rpqmnxy.py:244-246):
// position pos = step * batch + i of the sample stream belongs to epoch pos / N and takes dataset index perm_epoch(pos % N), a keyed
// bijection of [0, N) -- a 4-round Feistel network over the next even power-of-two width with cycle walking (values >= N are
// permuted again), round keys = one Philox block of (epoch, stream id) under the sampler's seed.  Every index appears exactly once per
// epoch, batches straddle epoch ends (the batch size is constant), nothing is stored.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t epoch_perm(uint32_t r, uint32_t N, const uint32_t (&k)[4]) {
    int w = 1;
    while ((1u << (2 * w)) < N) ++w;                     // half width: 2^(2w) >= N
    const uint32_t mask = (1u << w) - 1u;
    uint32_t x = r;
    do {
        uint32_t L = x >> w, R = x & mask;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t f = mix32(R * 0x9E3779B1u + k[i]) & mask;
            const uint32_t t = L ^ f;
            L = R; R = t;
        }
        x = (L << w) | R;
    } while (x >= N);
    return x;
}

// kind 0: uniform [a, b) fp32; 1: normal(mean a, std b) fp32 (Box-Muller); 2: integers in [a, b) as int64; 3: Bernoulli(a) as uint8;
// 4: epoch permutation -- dataset indices in [0, a) as int64 for positions step * n + i (see epoch_perm).
// Quad q of a draw depends on (q, step, stream id, seed) only -- not on the launch shape -- so one launch over several draws
// (rng_fill_multi_kernel) writes the very values the single launches write.  onehot (kind 2): out row i also sets onehot[i][0..onehot_n).
__device__ __forceinline__ void rng_fill_quads(int kind, void* __restrict__ out, size_t n, float a, float b, const Philox& ph, uint32_t st,
                                               uint32_t stream_id, size_t q0, size_t qstride, float* __restrict__ onehot, int onehot_n) {
    if (kind == 4) {
        const uint32_t N = (uint32_t)a;
        for (size_t i = q0; i < n; i += qstride) {
            const uint64_t pos = (uint64_t)st * n + i;
            const uint64_t epoch = pos / N;
            uint32_t k[4] = {(uint32_t)epoch, (uint32_t)(epoch >> 32), 0x5045524Du, stream_id};
            ph(k);
            reinterpret_cast<long long*>(out)[i] = (long long)epoch_perm((uint32_t)(pos - epoch * N), N, k);
        }
        return;
    }
    for (size_t q = q0; q * 4 < n; q += qstride) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), st, stream_id};
        ph(c);
        float v[4];
        if (kind == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float r = sqrtf(-2.f * logf(u01(c[2 * h]))), t = 6.28318530717958647692f * u01(c[2 * h + 1]);
                v[2 * h] = a + b * r * cosf(t);
                v[2 * h + 1] = a + b * r * sinf(t);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = u01(c[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t i = q * 4 + e;
            if (i >= n) break;
            if (kind == 0) {
                float r = a + (b - a) * v[e];
                if (r >= b) r = nextafterf(b, a);              // fp32 rounding must not reach the open end of [a, b)
                reinterpret_cast<float*>(out)[i] = r;
            }
            else if (kind == 1) reinterpret_cast<float*>(out)[i] = v[e];
            else if (kind == 2) {
                long long k = (long long)a + (long long)(v[e] * (b - a));
                if (k >= (long long)b) k = (long long)b - 1;
                reinterpret_cast<long long*>(out)[i] = k;
                if (onehot)
                    for (int j = 0; j < onehot_n; ++j) onehot[i * onehot_n + j] = k == j ? 1.f : 0.f;
            } else reinterpret_cast<unsigned char*>(out)[i] = v[e] < a ? 1 : 0;
        }
    }
}

__global__ void rng_fill_kernel(int kind, void* __restrict__ out, size_t n, float a, float b, uint64_t seed, const int* __restrict__ step,
                                uint32_t stream_id) {
    const Philox ph{(uint32_t)seed, (uint32_t)(seed >> 32)};
    const uint32_t st = step ? (uint32_t)step[0] : 0u;
    rng_fill_quads(kind, out, n, a, b, ph, st, stream_id, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, nullptr, 0);
}

// several draws of one iteration in ONE launch (blockIdx.y = draw): the five draws + the one-hot labels of the CelebA loop were six
// launches of ~4.5 us each at the head of every iteration's critical chain
struct RngSegs { eg_rng_seg s[8]; };
__global__ void rng_fill_multi_kernel(const RngSegs segs, uint64_t seed, const int* __restrict__ step) {
    const Philox ph{(uint32_t)seed, (uint32_t)(seed >> 32)};
    const uint32_t st = step ? (uint32_t)step[0] : 0u;
    const eg_rng_seg& g = segs.s[blockIdx.y];
    rng_fill_quads(g.kind, g.out, g.n, g.a, g.b, ph, st, g.stream_id, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x,
                   g.onehot, g.onehot_n);
}

extern "C" int eg_rng_fill_multi(const eg_rng_seg* segs, int nseg, unsigned long long seed, const int* step, eg_stream_t s) {
    EG_REQUIRE(segs && nseg >= 1 && nseg <= 8, "eg_rng_fill_multi: 1..8 draws");
    RngSegs p;
    memset(&p, 0, sizeof(p));
    size_t maxq = 1;
    for (int i = 0; i < nseg; ++i) {
        EG_REQUIRE(segs[i].out && segs[i].kind >= 0 && segs[i].kind <= 4 && (!segs[i].onehot || (segs[i].kind == 2 && segs[i].onehot_n > 0)), "eg_rng_fill_multi: bad draw %d", i);
        EG_REQUIRE(segs[i].kind != 4 || (segs[i].a >= 1.f && segs[i].a <= 16777216.f && segs[i].a == floorf(segs[i].a)), "eg_rng_fill_multi: epoch permutation needs 1 <= N <= 2^24 (a)");
        p.s[i] = segs[i];
        maxq = std::max(maxq, (segs[i].n + 3) / 4);
    }
    const int blocks = (int)((maxq + 255) / 256 > 256 ? 256 : (maxq + 255) / 256);
    hipLaunchKernelGGL(rng_fill_multi_kernel, dim3(blocks, nseg), dim3(256), 0, (hipStream_t)s, p, (uint64_t)seed, step);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_rng_fill(int kind, void* out, size_t n, float a, float b, unsigned long long seed, const int* step, unsigned int stream_id,
                           eg_stream_t s) {
    EG_REQUIRE(out && kind >= 0 && kind <= 4, "eg_rng_fill: bad argument");
    EG_REQUIRE(kind != 4 || (a >= 1.f && a <= 16777216.f && a == floorf(a)), "eg_rng_fill: epoch permutation needs 1 <= N <= 2^24 (a)");
    if (n == 0) return 0;
    const size_t quads = (n + 3) / 4;
    const int blocks = (int)((quads + 255) / 256 > 1024 ? 1024 : (quads + 255) / 256);
    hipLaunchKernelGGL(rng_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, kind, out, n, a, b, (uint64_t)seed, step, (uint32_t)stream_id);
    EG_LAUNCH_CHECK();
    return 0;
}

__global__ void counter_add_kernel(int* c, int v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += v;
}
extern "C" int eg_counter_add(int* counter, int v, eg_stream_t s) {
    EG_REQUIRE(counter, "eg_counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, counter, v);
    EG_LAUNCH_CHECK();
    return 0;
}

// out[b][c][y][x] = data[idx[b]][c][y][flip[b] ? W-1-x : x] * scale + shift   (uint8 NCHW dataset resident in HBM -> fp32 NCHW batch)
__global__ void gather_u8_images_kernel(const unsigned char* __restrict__ data, const long long* __restrict__ idx, const unsigned char* __restrict__ flip,
                                        float* __restrict__ out, int B, int CH, int W, float scale, float shift, int* tick) {
    if (tick && blockIdx.x == 0 && threadIdx.x == 0) tick[0] += 1;      // the draws of this iteration (earlier launches) have read the counter
    const size_t per = (size_t)CH * W;               // CH = C*H rows of W pixels per image
    const size_t total = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const size_t r = i - (size_t)b * per;
        const int x = (int)(r % W);
        const size_t row = r - x;
        const int xs = (flip && flip[b]) ? W - 1 - x : x;
        out[i] = (float)data[(size_t)idx[b] * per + row + xs] * scale + shift;
    }
}

extern "C" int eg_gather_u8_images(const unsigned char* data, const long long* idx, const unsigned char* flip, float* out, int B, int C, int H, int W,
                                   float scale, float shift, eg_stream_t s) {
    EG_REQUIRE(data && idx && out && B > 0 && C > 0 && H > 0 && W > 0, "eg_gather_u8_images: bad argument");
    const size_t total = (size_t)B * C * H * W;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(gather_u8_images_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, data, idx, flip, out, B, C * H, W, scale, shift, (int*)nullptr);
    EG_LAUNCH_CHECK();
    return 0;
}

extern "C" int eg_gather_u8_images_tick(const unsigned char* data, const long long* idx, const unsigned char* flip, float* out, int B, int C, int H,
                                        int W, float scale, float shift, int* step_tick, eg_stream_t s) {
    EG_REQUIRE(data && idx && out && B > 0 && C > 0 && H > 0 && W > 0, "eg_gather_u8_images_tick: bad argument");
    const size_t total = (size_t)B * C * H * W;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(gather_u8_images_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, data, idx, flip, out, B, C * H, W, scale, shift, step_tick);
    EG_LAUNCH_CHECK();
    return 0;
}

__global__ void onehot_kernel(const long long* __restrict__ labels, float* __restrict__ out, int B, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n) return;
    out[i] = labels[i / n] == (i % n) ? 1.f : 0.f;
}
extern "C" int eg_onehot(const long long* labels, float* out, int B, int n, eg_stream_t s) {
    EG_REQUIRE(labels && out && B > 0 && n > 0, "eg_onehot: bad argument");
    hipLaunchKernelGGL(onehot_kernel, dim3(cdiv(B * n, 256)), dim3(256), 0, (hipStream_t)s, labels, out, B, n);
    EG_LAUNCH_CHECK();
    return 0;
}

// ---- Resize + CenterCrop of the uint8 dataset on its way into HBM (celebA/EAD-GAN_celebA.py:194-196: transforms.Resize(img_size) =
// PIL bilinear with antialiasing, then CenterCrop) ------------------------------------------------------------------------------------
// One separable pass along x (axis = 1) or y (axis = 0) of planar uint8 images with the fixed-point coefficients PIL uses for 8-bit
// images (22 fractional bits, ksize taps from bounds[2*o] on, rounded with +2^21 and clamped): out = clip8((2^21 + sum src*k) >> 22).
// The host computes the tables exactly as PIL's precompute_coeffs / normalize_coeffs_8bpc do (double arithmetic), so the two passes
// (horizontal first, 8-bit intermediate, like ImagingResample) reproduce PIL's output bit for bit.  Only the output window
// [o0, o0 + on) along the axis is produced (the crop); the other axis is copied through a window [c0, c0 + cn) of the source.
__global__ void resample_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int planes, int in_h, int in_w,
                                   int axis, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int o0, int on, int c0,
                                   int cn) {
    const int out_h = axis == 0 ? on : cn, out_w = axis == 0 ? cn : on;
    const size_t total = (size_t)planes * out_h * out_w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % out_w), y = (int)((i / out_w) % out_h);
        const size_t pl = i / ((size_t)out_w * out_h);
        const int o = (axis == 0 ? y : x) + o0, c = (axis == 0 ? x : y) + c0;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = kk + (size_t)o * ksize;
        const unsigned char* p = src + pl * (size_t)in_h * in_w;
        int acc = 1 << 21;
        if (axis == 0)
            for (int t = 0; t < n; ++t) acc += (int)p[(size_t)(lo + t) * in_w + c] * k[t];
        else
            for (int t = 0; t < n; ++t) acc += (int)p[(size_t)c * in_w + lo + t] * k[t];
        acc >>= 22;
        dst[i] = (unsigned char)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}

extern "C" int eg_resample_u8(const unsigned char* src, unsigned char* dst, int planes, int in_h, int in_w, int axis, const int* bounds,
                              const int* kk, int ksize, int o0, int on, int c0, int cn, eg_stream_t s) {
    EG_REQUIRE(src && dst && bounds && kk && planes > 0 && in_h > 0 && in_w > 0 && (axis == 0 || axis == 1) && ksize > 0 && on > 0 && cn > 0 &&
                   o0 >= 0 && c0 >= 0 && c0 + cn <= (axis == 0 ? in_w : in_h),
               "eg_resample_u8: bad argument");
    const size_t total = (size_t)planes * on * cn;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(resample_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, src, dst, planes, in_h, in_w, axis, bounds, kk, ksize, o0,
                       on, c0, cn);
    EG_LAUNCH_CHECK();
    return 0;
}
