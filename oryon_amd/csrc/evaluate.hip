// Pose-accuracy metrics on the device: ADD, ADD-S, rotation / translation error for a batch of predicted poses.
// Replaces utils/metrics.py:194-259 (compute_add, compute_adds, compute_RT_distances) + np_transform_pcd (utils/pcd.py:127-133) of
// the reference, which run per pair on the host (numpy + a KD-tree) after a .cpu() of every pose.
//
// The reference moves the model points in FLOAT16 (operands rounded to half, products accumulated in fp32 by numpy's HALF_dot, the
// result rounded to half, the half translation added and rounded again); the same roundings are applied here, point for point, so
// the transformed clouds are the reference's.  The distance statistics on top of them are taken in fp32 (the reference: half norms,
// an fp32 pairwise mean rounded to half for ADD; float64 KD-tree distances for ADD-S), which agrees with it to ~1e-3 relative on ADD
// and ~1e-6 on ADD-S - far inside the 0.1-point bar of ADD(S)-0.1d.
//
// One workgroup = 256 predicted points of one pair; the ground-truth cloud streams through LDS for the nearest-neighbour search.
#include <hip/hip_fp16.h>
#include "common.h"

namespace oryon {

__device__ __forceinline__ float h16(float x) { return __half2float(__float2half_rn(x)); }

// model point (fp32, any unit) -> the reference's float16 transform
__device__ __forceinline__ float3 move_f16(const float *P, float x, float y, float z)
{
    const float px = h16(x), py = h16(y), pz = h16(z);
    float3 o;
    // np.dot(pcd16, R16.T): fp32 accumulation in k order, one rounding to half at the end; then + t16, rounded again
    o.x = h16(h16(__fmaf_rn(pz, h16(P[2]), __fmaf_rn(py, h16(P[1]), __fmul_rn(px, h16(P[0]))))) + h16(P[3]));
    o.y = h16(h16(__fmaf_rn(pz, h16(P[6]), __fmaf_rn(py, h16(P[5]), __fmul_rn(px, h16(P[4]))))) + h16(P[7]));
    o.z = h16(h16(__fmaf_rn(pz, h16(P[10]), __fmaf_rn(py, h16(P[9]), __fmul_rn(px, h16(P[8]))))) + h16(P[11]));
    return o;
}

constexpr int EV_TILE = 1024;        // ground-truth points per LDS tile

__global__ __launch_bounds__(256) void pose_add_kernel(const float *__restrict__ pred, const float *__restrict__ gt,
                                                       const float *__restrict__ pts, const int32_t *__restrict__ pts_offset,
                                                       const int32_t *__restrict__ model_of_pair, float *__restrict__ acc /*[B,2]*/)
{
    __shared__ float sb[EV_TILE * 3];
    __shared__ float red[2][4];
    const int p = blockIdx.y;
    const int model = model_of_pair ? model_of_pair[p] : 0;
    const int m0 = pts_offset[model], M = pts_offset[model + 1] - m0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= M) return;
    const float *Pp = pred + (size_t)p * 16, *Pg = gt + (size_t)p * 16;
    const bool live = i < M;
    float3 a = make_float3(0.f, 0.f, 0.f), b = a;
    if (live) {
        const float *x = pts + (size_t)(m0 + i) * 3;
        a = move_f16(Pp, x[0], x[1], x[2]);
        b = move_f16(Pg, x[0], x[1], x[2]);
    }
    // ADD: corresponding points; the difference is taken in half like the reference's (a - b) on float16 arrays
    const float dx = h16(a.x - b.x), dy = h16(a.y - b.y), dz = h16(a.z - b.z);
    float add = live ? sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)))) : 0.0f;
    // ADD-S: nearest transformed ground-truth point
    float best = INFINITY;
    for (int j0 = 0; j0 < M; j0 += EV_TILE) {
        __syncthreads();
        for (int j = threadIdx.x; j < EV_TILE; j += 256) {
            float3 q = make_float3(1e30f, 1e30f, 1e30f);
            if (j0 + j < M) {
                const float *x = pts + (size_t)(m0 + j0 + j) * 3;
                q = move_f16(Pg, x[0], x[1], x[2]);
            }
            sb[3 * j] = q.x; sb[3 * j + 1] = q.y; sb[3 * j + 2] = q.z;
        }
        __syncthreads();
        const int lim = (M - j0) < EV_TILE ? (M - j0) : EV_TILE;
        for (int j = 0; j < lim; ++j) {
            const float ex = a.x - sb[3 * j], ey = a.y - sb[3 * j + 1], ez = a.z - sb[3 * j + 2];
            best = fminf(best, __fmaf_rn(ez, ez, __fmaf_rn(ey, ey, ex * ex)));
        }
    }
    float adds = live ? sqrtf(best) : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { add += __shfl_xor(add, off); adds += __shfl_xor(adds, off); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = add; red[1][wave] = adds; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[2 * p + 0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        atomicAdd(&acc[2 * p + 1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

// per pair: means, rotation angle (degrees) and translation distance (centimetres) as utils/metrics.py:222-259 (rotations rescaled to
// determinant 1 first, arccos of the clipped trace; NaN -> 180)
__global__ void pose_finish_kernel(int B, const float *__restrict__ pred, const float *__restrict__ gt, const int32_t *__restrict__ pts_offset,
                                   const int32_t *__restrict__ model_of_pair, const float *__restrict__ acc, float *__restrict__ out /*[B,4]*/)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const int model = model_of_pair ? model_of_pair[p] : 0;
    const int M = pts_offset[model + 1] - pts_offset[model];
    out[4 * p + 0] = M > 0 ? acc[2 * p + 0] / (float)M : 0.0f;
    out[4 * p + 1] = M > 0 ? acc[2 * p + 1] / (float)M : 0.0f;
    double R1[9], R2[9];
    const float *A = pred + (size_t)p * 16, *Bm = gt + (size_t)p * 16;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { R1[3 * r + c] = A[4 * r + c]; R2[3 * r + c] = Bm[4 * r + c]; }
    auto det = [](const double *R) {
        return R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    };
    const double s1 = cbrt(det(R1)), s2 = cbrt(det(R2));
    double tr = 0.0;                     // trace(R1 R2^T) = sum_ij R1_ij R2_ij
    for (int k = 0; k < 9; ++k) tr += (R1[k] / s1) * (R2[k] / s2);
    double c = (tr - 1.0) / 2.0;
    c = c < -1.0 + 1e-12 ? -1.0 + 1e-12 : (c > 1.0 - 1e-12 ? 1.0 - 1e-12 : c);
    double theta = acos(c) * 180.0 / 3.14159265358979323846;
    if (theta != theta) theta = 180.0;
    const double tx = (double)A[3] - Bm[3], ty = (double)A[7] - Bm[7], tz = (double)A[11] - Bm[11];
    out[4 * p + 2] = (float)theta;
    out[4 * p + 3] = (float)(sqrt(tx * tx + ty * ty + tz * tz) * 100.0);
}


// ---------------------------------------------------------------------------------------------- MSSD / MSPD (BOP, as the reference calls them)
// utils/evaluator.py:258-275 rounds both poses to FLOAT16 (translation: half(t) * 1000 in half arithmetic, i.e. millimetres on a
// ~1 mm grid) and hands them to bop_toolkit_lib/pose_error.py:370-427 (my_mssd / my_mspd), whose arithmetic then runs in float64
// (model points, symmetry set and K are float64 arrays).  Same here: the two roundings, then fp64 throughout.
//   MSSD = min over the symmetry set S of max over model points x of | P_est x - P_gt S x |            (millimetres)
//   MSPD = the same with both points projected by K first                                              (pixels)
// max_points: the reference's np_transform slices `pts[:, :3]` on the POINT axis of its [1,N,3] array (pose_error.py:345), so its
// maxima run over the first three model points only; max_points = 3 reproduces that, 0 evaluates every point (BOP's definition).
__device__ __forceinline__ double round_to_half(double d)
{
    // numpy's float64 -> float16 cast rounds once (to nearest even).  Going through float would round twice, so the intermediate
    // float is made by ROUND-TO-ODD (truncate, set the last bit if inexact): a following round-to-nearest to 11 bits is then exact.
    float f = (float)d;
    const double back = (double)f;
    if (back != d) {
        unsigned u = __float_as_uint(f);
        if (fabs(back) > fabs(d)) u -= 1u;              // undo a rounding away from zero (sign-magnitude: one step towards zero)
        u |= 1u;
        f = __uint_as_float(u);
    }
    return (double)__half2float(__float2half_rn(f));
}

struct Pose34 { double m[12]; };

__device__ __forceinline__ Pose34 pose_f16_mm(const double *P)          // [4,4] row-major, metres -> R16 | half(t16 * 1000)
{
    Pose34 o;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) o.m[4 * r + c] = round_to_half(P[4 * r + c]);
        const float t16 = (float)round_to_half(P[4 * r + 3]);
        o.m[4 * r + 3] = (double)__half2float(__float2half_rn(__fmul_rn(t16, 1000.0f)));
    }
    return o;
}

__device__ __forceinline__ void apply34(const Pose34 &T, double x, double y, double z, double &ox, double &oy, double &oz)
{
    ox = (x * T.m[0] + y * T.m[1] + z * T.m[2]) + T.m[3];
    oy = (x * T.m[4] + y * T.m[5] + z * T.m[6]) + T.m[7];
    oz = (x * T.m[8] + y * T.m[9] + z * T.m[10]) + T.m[11];
}

// grid (symmetry index, pair); workspace [B, max_syms, 2] = per symmetry (max 3-D distance, max projected distance)
// max that PROPAGATES NaN like numpy's max (the reference's my_mssd / my_mspd, bop_toolkit_lib/pose_error.py:339-427, take `.max()` of the
// per-point errors: a degenerate projection - w = 0 - makes the whole error NaN there); fmax alone would drop it
__device__ __forceinline__ double nmax(double a, double b) { return (a != a) ? a : (b != b) ? b : fmax(a, b); }

__global__ __launch_bounds__(256) void pose_bop_kernel(const double *__restrict__ pred, const double *__restrict__ gt, const double *__restrict__ Kc,
                                                       const double *__restrict__ pts, const int32_t *__restrict__ pts_offset,
                                                       const double *__restrict__ syms, const int32_t *__restrict__ sym_offset,
                                                       const int32_t *__restrict__ model_of_pair, int max_syms, int max_points,
                                                       double *__restrict__ ws)
{
    __shared__ double red[2][4];
    const int p = blockIdx.y, si = blockIdx.x;
    const int model = model_of_pair ? model_of_pair[p] : 0;
    const int m0 = pts_offset[model];
    int M = pts_offset[model + 1] - m0;
    if (max_points > 0 && M > max_points) M = max_points;
    const int s0 = sym_offset[model], S = sym_offset[model + 1] - s0;
    if (si >= S) return;
    const Pose34 E = pose_f16_mm(pred + (size_t)p * 16), G = pose_f16_mm(gt + (size_t)p * 16);
    const double *Sy = syms + (size_t)(s0 + si) * 12;
    Pose34 GS;                                                          // P_gt * S: R_gt S_R | R_gt S_t + t_gt
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) GS.m[4 * r + c] = G.m[4 * r] * Sy[c] + G.m[4 * r + 1] * Sy[4 + c] + G.m[4 * r + 2] * Sy[8 + c];
        GS.m[4 * r + 3] = (G.m[4 * r] * Sy[3] + G.m[4 * r + 1] * Sy[7] + G.m[4 * r + 2] * Sy[11]) + G.m[4 * r + 3];
    }
    const double *K = Kc + (size_t)p * 9;
    double d3 = 0.0, d2 = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) {
        const double *x = pts + (size_t)(m0 + i) * 3;
        double ax, ay, az, bx, by, bz;
        apply34(E, x[0], x[1], x[2], ax, ay, az);
        apply34(GS, x[0], x[1], x[2], bx, by, bz);
        const double ex = ax - bx, ey = ay - by, ez = az - bz;
        d3 = nmax(d3, sqrt(ex * ex + ey * ey + ez * ez));
        // my_project_pts: (R x + t) K^T, divided by its third component
        const double aw = ax * K[6] + ay * K[7] + az * K[8], bw = bx * K[6] + by * K[7] + bz * K[8];
        const double au = (ax * K[0] + ay * K[1] + az * K[2]) / aw, av = (ax * K[3] + ay * K[4] + az * K[5]) / aw;
        const double bu = (bx * K[0] + by * K[1] + bz * K[2]) / bw, bv = (bx * K[3] + by * K[4] + bz * K[5]) / bw;
        const double du = au - bu, dv = av - bv;
        d2 = nmax(d2, sqrt(du * du + dv * dv));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d3 = nmax(d3, __shfl_xor(d3, off)); d2 = nmax(d2, __shfl_xor(d2, off)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = d3; red[1][wave] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[((size_t)p * max_syms + si) * 2 + 0] = nmax(nmax(red[0][0], red[0][1]), nmax(red[0][2], red[0][3]));
        ws[((size_t)p * max_syms + si) * 2 + 1] = nmax(nmax(red[1][0], red[1][1]), nmax(red[1][2], red[1][3]));
    }
}

__global__ void pose_bop_finish_kernel(int B, const int32_t *__restrict__ sym_offset, const int32_t *__restrict__ model_of_pair, int max_syms,
                                       const double *__restrict__ ws, double *__restrict__ out /*[B,2]*/)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const int model = model_of_pair ? model_of_pair[p] : 0;
    const int S = sym_offset[model + 1] - sym_offset[model];
    double a = INFINITY, b = INFINITY;
    // numpy's .min() over the symmetries propagates NaN as well (dist.min(), pose_error.py:398,427)
    auto nmin = [](double x, double y) { return (x != x) ? x : (y != y) ? y : fmin(x, y); };
    for (int s = 0; s < S; ++s) { a = nmin(a, ws[((size_t)p * max_syms + s) * 2]); b = nmin(b, ws[((size_t)p * max_syms + s) * 2 + 1]); }
    out[2 * p] = a;
    out[2 * p + 1] = b;
}

}  // namespace oryon

using namespace oryon;

extern "C" int oryon_pose_metrics(const float *pred_pose, const float *gt_pose, int B, const float *model_pts, const int32_t *pts_offset,
                                  int n_models, int max_pts, const int32_t *model_of_pair, float *workspace /*[B,2]*/, float *out /*[B,4]*/,
                                  void *stream)
{
    ORYON_CHECK_ARG(pred_pose && gt_pose && model_pts && pts_offset && workspace && out && B >= 0 && n_models >= 1 && max_pts >= 1);
    if (B == 0) return ORYON_OK;
    hipStream_t st = as_stream(stream);
    ORYON_CHECK_HIP(hipMemsetAsync(workspace, 0, (size_t)B * 2 * sizeof(float), st));
    hipLaunchKernelGGL(pose_add_kernel, dim3((max_pts + 255) / 256, B), dim3(256), 0, st, pred_pose, gt_pose, model_pts, pts_offset,
                       model_of_pair, workspace);
    hipLaunchKernelGGL(pose_finish_kernel, dim3((B + 127) / 128), dim3(128), 0, st, B, pred_pose, gt_pose, pts_offset, model_of_pair,
                       workspace, out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}

extern "C" size_t oryon_pose_bop_workspace_bytes(int B, int max_syms) { return B > 0 && max_syms > 0 ? (size_t)B * max_syms * 2 * sizeof(double) : 0; }

extern "C" int oryon_pose_bop_errors(const double *pred_pose, const double *gt_pose, const double *K, int B, const double *model_pts_mm,
                                     const int32_t *pts_offset, const double *syms, const int32_t *sym_offset, int n_models, int max_syms,
                                     const int32_t *model_of_pair, int max_points, double *workspace, double *out, void *stream)
{
    ORYON_CHECK_ARG(pred_pose && gt_pose && K && model_pts_mm && pts_offset && syms && sym_offset && workspace && out);
    ORYON_CHECK_ARG(B >= 0 && n_models >= 1 && max_syms >= 1 && max_points >= 0);
    if (B == 0) return ORYON_OK;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(pose_bop_kernel, dim3(max_syms, B), dim3(256), 0, st, pred_pose, gt_pose, K, model_pts_mm, pts_offset, syms, sym_offset,
                       model_of_pair, max_syms, max_points, workspace);
    hipLaunchKernelGGL(pose_bop_finish_kernel, dim3((B + 127) / 128), dim3(128), 0, st, B, sym_offset, model_of_pair, max_syms, workspace, out);
    ORYON_CHECK_LAUNCH();
    return ORYON_OK;
}
