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
