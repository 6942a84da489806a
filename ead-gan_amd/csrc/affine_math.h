// Per-sample 3x3 affine algebra of the EAD-GAN latent codes, written once for plain floats and for
// forward-mode dual numbers (exact Jacobians for the regulariser backward, one thread per sample).
#pragma once
#include "eg_common.h"

template <int N>
struct Dual {
    float v;
    float d[N];
};
template <int N> __device__ __forceinline__ Dual<N> dconst(float c) { Dual<N> r; r.v = c; for (int i = 0; i < N; ++i) r.d[i] = 0.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> dvar(float c, int idx) { Dual<N> r = dconst<N>(c); r.d[idx] = 1.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; const float ib = 1.f / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, float c) { Dual<N> r; r.v = a.v * c; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, float c) { Dual<N> r = a; r.v += c; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, float c) { Dual<N> r = a; r.v -= c; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, float c) { return a * (1.f / c); }
template <int N> __device__ __forceinline__ Dual<N> nsin(const Dual<N>& a) { Dual<N> r; r.v = sinf(a.v); const float c = cosf(a.v); for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> ncos(const Dual<N>& a) { Dual<N> r; r.v = cosf(a.v); const float s = -sinf(a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> natan(const Dual<N>& a) { Dual<N> r; r.v = atanf(a.v); const float g = 1.f / (1.f + a.v * a.v); for (int i = 0; i < N; ++i) r.d[i] = g * a.d[i]; return r; }
__device__ __forceinline__ float nsin(float a) { return sinf(a); }
__device__ __forceinline__ float ncos(float a) { return cosf(a); }
__device__ __forceinline__ float natan(float a) { return atanf(a); }

// affine matrix [[a,b,c],[d,e,f],[0,0,1]]
template <typename Num> struct Aff { Num a, b, c, d, e, f; };

template <typename Num> __device__ __forceinline__ Aff<Num> aff_mul(const Aff<Num>& x, const Aff<Num>& y) {
    Aff<Num> r;
    r.a = x.a * y.a + x.b * y.d; r.b = x.a * y.b + x.b * y.e; r.c = x.a * y.c + x.b * y.f + x.c;
    r.d = x.d * y.a + x.e * y.d; r.e = x.d * y.b + x.e * y.e; r.f = x.d * y.c + x.e * y.f + x.f;
    return r;
}
template <typename Num> __device__ __forceinline__ Aff<Num> aff_inv(const Aff<Num>& m) {
    const Num det = m.a * m.e - m.b * m.d;
    Aff<Num> r;
    r.a = m.e / det; r.b = -m.b / det; r.c = (m.b * m.f - m.c * m.e) / det;
    r.d = -m.d / det; r.e = m.a / det; r.f = (m.c * m.d - m.a * m.f) / det;
    return r;
}

#define EG_PI_F 3.14159265358979323846f

// ---- CelebA variant, 5 codes (theta,p,q,x,y): celebA/utils_rpqxy.py:25-80 -------------------------
// A = Rot(theta) * diag(p,q,1) * Trans(x,y)
template <typename Num> __device__ __forceinline__ Aff<Num> matrix_rpqxy(const Num* c) {
    const Num th = c[0] * (EG_PI_F / 9.f);
    const Num p = c[1] * 0.2f + 1.f, q = c[2] * 0.2f + 1.f;
    const Num x = c[3] * 0.1f, y = c[4] * 0.1f;
    const Num cs = ncos(th), sn = nsin(th);
    Aff<Num> m;
    m.a = p * cs; m.b = -(q * sn); m.d = p * sn; m.e = q * cs;
    m.c = m.a * x + m.b * y;
    m.f = m.d * x + m.e * y;
    return m;
}
// closed-form parameter recovery from the relative matrix, celebA/utils_rpqxy.py:82-116; out = 5 latent units
template <typename Num> __device__ __forceinline__ void regularizer_rpqxy(const Num* real5, const Num* trans5, Num* out) {
    const Aff<Num> rel = aff_mul(matrix_rpqxy(trans5), aff_inv(matrix_rpqxy(real5)));
    const Num t1 = rel.a * rel.d - rel.b * rel.e;
    const Num t2 = rel.a * rel.a + rel.e * rel.e - rel.b * rel.b - rel.d * rel.d;
    const Num th = natan(t1 * 2.f / t2) * 0.5f;
    const Num cs = ncos(th), sn = nsin(th);
    const Num p = rel.a * cs + rel.d * sn;
    const Num q = rel.e * cs - rel.b * sn;
    const Num x = (rel.c * cs + rel.f * sn) / p;
    const Num y = (rel.f * cs - rel.c * sn) / q;
    out[0] = th * (9.f / EG_PI_F);
    out[1] = (p - 1.f) / 0.2f;
    out[2] = (q - 1.f) / 0.2f;
    out[3] = x / 0.1f;
    out[4] = y / 0.1f;
}

// ---- MNIST variant, 7 codes (theta,p,q,m,n,x,y): MNIST/utils_rpqmnxy.py:46-114 ---------------------------------------
// A = Rot(theta) * diag(p,q,1) * Skew(m,n) * Trans(x,y),  Skew = [[1,m,0],[n,1,0],[0,0,1]]
template <typename Num> __device__ __forceinline__ Aff<Num> matrix_rpqmnxy(const Num* c) {
    const Num th = c[0] * (EG_PI_F / 9.f);
    const Num p = c[1] * 0.2f + 1.f, q = c[2] * 0.2f + 1.f;
    const Num m = c[3] * 0.2f, n = c[4] * 0.2f;
    const Num x = c[5] * 0.1f, y = c[6] * 0.1f;
    const Num cs = ncos(th), sn = nsin(th);
    const Num a0 = p * cs, b0 = -(q * sn), d0 = p * sn, e0 = q * cs;   // R*Z
    Aff<Num> r;
    r.a = a0 + b0 * n; r.b = a0 * m + b0;
    r.d = d0 + e0 * n; r.e = d0 * m + e0;
    r.c = r.a * x + r.b * y;
    r.f = r.d * x + r.e * y;
    return r;
}
// relative matrix rows 0,1 flattened (a,b,c,d,e,f): the approximator's input (utils_rpqmnxy.py:125-128)
template <typename Num> __device__ __forceinline__ void relative_rpqmnxy(const Num* real7, const Num* trans7, Num* flat6) {
    const Aff<Num> rel = aff_mul(matrix_rpqmnxy(trans7), aff_inv(matrix_rpqmnxy(real7)));
    flat6[0] = rel.a; flat6[1] = rel.b; flat6[2] = rel.c; flat6[3] = rel.d; flat6[4] = rel.e; flat6[5] = rel.f;
}

// ---- dSprites variant, 4 codes (theta,p,x,y): dSprites/utils_rp.py:23-147 ----------------------------------------------
// A = Rot(theta) * diag(p,p,1) * Trans(x,y)   (get_matrix == get_matrix_D)
template <typename Num> __device__ __forceinline__ Aff<Num> matrix_rp(const Num* c) {
    const Num th = c[0] * (EG_PI_F / 9.f);
    const Num p = c[1] * 0.2f + 1.f;
    const Num x = c[2] * 0.1f, y = c[3] * 0.1f;
    const Num cs = ncos(th), sn = nsin(th);
    Aff<Num> m;
    m.a = p * cs; m.b = -(p * sn); m.d = p * sn; m.e = p * cs;
    m.c = m.a * x + m.b * y;
    m.f = m.d * x + m.e * y;
    return m;
}
// closed-form recovery, utils_rp.py:118-147; out = 4 latent units
template <typename Num> __device__ __forceinline__ void regularizer_rp(const Num* real4, const Num* trans4, Num* out) {
    const Aff<Num> rel = aff_mul(matrix_rp(trans4), aff_inv(matrix_rp(real4)));
    const Num th = natan((rel.d - rel.b) / (rel.a + rel.e));
    const Num cs = ncos(th), sn = nsin(th);
    const Num p = (cs * (rel.a + rel.e) + sn * (rel.d - rel.b)) * 0.5f;
    const Num x = (rel.c * cs + rel.f * sn) / p;
    const Num y = (rel.f * cs - rel.c * sn) / p;
    out[0] = th * (9.f / EG_PI_F);
    out[1] = (p - 1.f) / 0.2f;
    out[2] = x / 0.1f;
    out[3] = y / 0.1f;
}
