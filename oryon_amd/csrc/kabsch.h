// Weighted Kabsch (rigid_transform_3d, models/pointdsc/common.py:7-45) as a device-side building block.
//   centroids   cA = sum(w a) / (sum(w) + 1e-6),  cB likewise                      (common.py:24-25)
//   H           = sum_i w_i (a_i - cA)(b_i - cB)^T                                  (common.py:28-33)
//   U S V^T     = svd(H);   R = V diag(1, 1, det(V U^T)) U^T;   t = cB - R cA       (common.py:36-42)
// The reference ships H to the CPU for LAPACK; here a one-sided Jacobi SVD of the 3x3 runs in registers,
// in fp64 (vector fp64 is cheap on gfx950 and the solve is O(1) per problem).  R is unique whenever H has
// rank >= 2, so agreement with LAPACK is to rounding, not to SVD sign conventions.
#pragma once
#include <hip/hip_runtime.h>

namespace oryon {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// 3x3 one-sided Jacobi.  In: H (row-major).  Out: R = V diag(1,1,det(VU^T)) U^T (row-major).
__device__ inline void rotation_from_covariance(const double H[9], double R[9])
{
    double a[3][3], v[3][3];  // a[c][r]: column c of the working matrix; v likewise
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a[c][r] = H[r * 3 + c];
            v[c][r] = (r == c) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 24; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double alpha = a[p][0] * a[p][0] + a[p][1] * a[p][1] + a[p][2] * a[p][2];
            const double beta = a[q][0] * a[q][0] + a[q][1] * a[q][1] + a[q][2] * a[q][2];
            const double gamma = a[p][0] * a[q][0] + a[p][1] * a[q][1] + a[p][2] * a[q][2];
            const double lim = 1e-15 * sqrt(alpha * beta);
            if (fabs(gamma) > lim && fabs(gamma) > 1e-300) {
                off = fmax(off, fabs(gamma) / fmax(sqrt(alpha * beta), 1e-300));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double ap = a[p][r], aq = a[q][r];
                    a[p][r] = cs * ap - sn * aq;
                    a[q][r] = sn * ap + cs * aq;
                    const double vp = v[p][r], vq = v[q][r];
                    v[p][r] = cs * vp - sn * vq;
                    v[q][r] = sn * vp + cs * vq;
                }
            }
        }
        if (off < 1e-14) break;
    }
    double s[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) s[c] = sqrt(a[c][0] * a[c][0] + a[c][1] * a[c][1] + a[c][2] * a[c][2]);
    // order singular values descending (LAPACK order): the det correction must act on the smallest one.
    // Compare-exchange network on whole columns keeps every index compile-time (no scratch).
    auto cswap = [&](int i, int j) {
        if (s[i] < s[j]) {
            double t_ = s[i]; s[i] = s[j]; s[j] = t_;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                t_ = a[i][r]; a[i][r] = a[j][r]; a[j][r] = t_;
                t_ = v[i][r]; v[i][r] = v[j][r]; v[j][r] = t_;
            }
        }
    };
    cswap(0, 1);
    cswap(0, 2);
    cswap(1, 2);
    double U[3][3], V[3][3];  // [column][row]
    const double tiny = 1e-12 * fmax(s[0], 1e-300);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            V[c][r] = v[c][r];
            U[c][r] = s[c] > tiny ? a[c][r] / s[c] : 0.0;
        }
    // complete U for rank-deficient H (direction is then free up to what det() fixes below)
    if (!(s[0] > tiny)) {
        U[0][0] = 1.0; U[0][1] = 0.0; U[0][2] = 0.0;
    }
    if (!(s[1] > tiny)) {
        // any unit vector orthogonal to U[0]
        const double ax = fabs(U[0][0]), ay = fabs(U[0][1]), az = fabs(U[0][2]);
        double e[3] = {0.0, 0.0, 0.0};
        if (ax <= ay && ax <= az) e[0] = 1.0; else if (ay <= az) e[1] = 1.0; else e[2] = 1.0;
        const double d = e[0] * U[0][0] + e[1] * U[0][1] + e[2] * U[0][2];
        double n2 = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { U[1][r] = e[r] - d * U[0][r]; n2 += U[1][r] * U[1][r]; }
        const double inv = 1.0 / sqrt(n2);
#pragma unroll
        for (int r = 0; r < 3; ++r) U[1][r] *= inv;
    }
    if (!(s[2] > tiny)) {
        U[2][0] = U[0][1] * U[1][2] - U[0][2] * U[1][1];
        U[2][1] = U[0][2] * U[1][0] - U[0][0] * U[1][2];
        U[2][2] = U[0][0] * U[1][1] - U[0][1] * U[1][0];
    }
    auto det3 = [](const double M[3][3]) {
        return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
               M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    };
    const double d = det3(V) * det3(U);  // det(V U^T); both are +-1
    const double D[3] = {1.0, 1.0, d};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i * 3 + j] = V[0][i] * D[0] * U[0][j] + V[1][i] * D[1] * U[1][j] + V[2][i] * D[2] * U[2][j];
}

struct KabschAcc {
    double sw, sa[3], sb[3], sab[9];
    __device__ __forceinline__ void clear()
    {
        sw = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) sa[i] = sb[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) sab[i] = 0.0;
    }
    __device__ __forceinline__ void add(float ax, float ay, float az, float bx, float by, float bz, float w)
    {
        const double a[3] = {ax, ay, az}, b[3] = {bx, by, bz}, wd = w;
        sw += wd;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sa[i] += wd * a[i];
            sb[i] += wd * b[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) sab[i * 3 + j] += wd * a[i] * b[j];
        }
    }
    __device__ __forceinline__ void wave_reduce()
    {
        sw = wave_sum(sw);
#pragma unroll
        for (int i = 0; i < 3; ++i) { sa[i] = wave_sum(sa[i]); sb[i] = wave_sum(sb[i]); }
#pragma unroll
        for (int i = 0; i < 9; ++i) sab[i] = wave_sum(sab[i]);
    }
    // 4x4 row-major fp32 transform
    __device__ inline void solve(float T[16]) const
    {
        const double den = sw + 1e-6;
        double ca[3], cb[3], H[9], R[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) { ca[i] = sa[i] / den; cb[i] = sb[i] / den; }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) H[i * 3 + j] = sab[i * 3 + j] - ca[i] * sb[j] - sa[i] * cb[j] + sw * ca[i] * cb[j];
        rotation_from_covariance(H, R);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j];
            T[i * 4 + 3] = (float)(cb[i] - (R[i * 3] * ca[0] + R[i * 3 + 1] * ca[1] + R[i * 3 + 2] * ca[2]));
        }
        T[12] = 0.0f; T[13] = 0.0f; T[14] = 0.0f; T[15] = 1.0f;
    }
};

}  // namespace oryon
