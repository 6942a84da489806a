// torch.optim.Adam's element update (celebA/EAD-GAN_celebA.py:211-217), shared by the flat-arena kernel (sn_adam.hip) and the kernels that
// update a convolution's master weights and write its packed panels in the same pass (igemm.hip): one definition, so a parameter gets the
// same bits whichever kernel updates it.
#pragma once
#include "eg_common.h"

struct AdamCoef { float step_size, bc2_sqrt, w, b2, eps; };

__device__ __forceinline__ AdamCoef adam_coef(float lr, float b1, float b2, float eps, const int* __restrict__ step) {
    const int t = step[0];
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    return {(float)((double)lr / bc1), (float)sqrt(bc2), 1.f - b1, b2, eps};
}

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamCoef& c) {
#pragma clang fp contract(off)      // the float4 and the scalar paths must round alike (slice-wise == whole-arena, bit for bit)
    // exp_avg.lerp_(grad, 1-beta1) with ATen's two-sided formula
    m = (c.w < 0.5f) ? m + c.w * (g - m) : g - (g - m) * (1.f - c.w);
    v = v * c.b2 + (1.f - c.b2) * g * g;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = p - c.step_size * (m / denom);
}

