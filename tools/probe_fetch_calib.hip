// Calibration of rocprofv3's FETCH_SIZE for the access widths K0v3 uses (MI355X_MICROARCH.md: only 16 B/lane streams are calibrated:
// FETCH_SIZE reports half their bytes).  Reads a 1 GiB buffer once with 4-byte-per-lane loads (256-byte runs per wave instruction, the
// NCHW gather's pattern) and once with 16-byte-per-lane loads; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare the
// counter with the known byte count.   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_fetch tools/probe_fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void read_b32(const float *p, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) *out = acc;
}
__global__ void read_b128(const float4 *p, size_t n, float *out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) *out = acc;
}
int main()
{
    const size_t bytes = 1ull << 30;
    float *p, *o;
    hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 0, bytes);
    hipLaunchKernelGGL(read_b32, dim3(4096), dim3(256), 0, 0, p, bytes / 4, o);
    hipLaunchKernelGGL(read_b128, dim3(4096), dim3(256), 0, 0, (const float4 *)p, bytes / 16, o);
    hipDeviceSynchronize();
    printf("read %zu bytes twice\n", bytes);
    return 0;
}
