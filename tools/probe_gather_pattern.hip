// Probe (round 4): what does the NCHW gather pattern of K0v3 deliver by itself?  Every variant reads the same bytes - B maps x C planes x N
// consecutive pixels, fp32 - and writes almost nothing; only the shape of the loads differs.
//   V0  K0v3's pattern: a wave owns 64 pixels, 256 dword loads (one 256-byte run per plane), all in flight, one wave per SIMD (256 registers)
//   V1  two waves per SIMD: a wave owns 64 pixels x 128 planes (128 registers)
//   V2  dwordx4: a wave owns 256 pixels x 64 planes (64 loads of 1 KB), one wave per SIMD
//   V3  dwordx4, 256 pixels x 32 planes (128 registers, two waves per SIMD)
//   V4  V0 followed by ~3 us of dependent VALU work per tile (the norm chain + conversions of K0v3), to see what the phases cost
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pg tools/probe_gather_pattern.hip && /tmp/pg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NCH, int VALU>
__global__ __launch_bounds__(256, (NCH > 128 ? 1 : 2)) void v_dword(const float *__restrict__ feat, int C, int HW, int N, int tiles_per_map, float *out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = C / NCH;                                  // waves per 64-pixel tile
    const int gw = blockIdx.x * 4 + wave;
    const int m = gw / (tiles_per_map * chunks), r = gw % (tiles_per_map * chunks);
    const int tile = r / chunks, ch0 = (r % chunks) * NCH;
    const int pix = tile * 64 + lane;
    if (tile * 64 >= N) return;
    const float *fb = feat + ((size_t)m * C + ch0) * HW + (pix < N ? pix : N - 1);
    float v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = fb[(size_t)i * HW];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s = fmaf(v[i], v[i], s);
    if (VALU) {
        // a dependent chain roughly as long as K0v3's per-tile compute (~1500 VALU instructions)
#pragma unroll 1
        for (int k = 0; k < VALU; ++k)
#pragma unroll
            for (int i = 0; i < NCH; i += 4) s = fmaf(v[i], s, v[i + 1]);
    }
    if (s == 123.456f) out[gw] = s;
}

template <int NCH>
__global__ __launch_bounds__(256, (NCH > 32 ? 1 : 2)) void v_x4(const float *__restrict__ feat, int C, int HW, int N, int tiles_per_map, float *out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = C / NCH;
    const int gw = blockIdx.x * 4 + wave;
    const int m = gw / (tiles_per_map * chunks), r = gw % (tiles_per_map * chunks);
    const int tile = r / chunks, ch0 = (r % chunks) * NCH;       // tile = 256 pixels
    if (tile * 256 >= N) return;
    const int pix = tile * 256 + lane * 4;
    const float4 *fb = reinterpret_cast<const float4 *>(feat + ((size_t)m * C + ch0) * HW + (pix + 3 < N ? pix : N - 4));
    float4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = fb[(size_t)i * (HW / 4)];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s = fmaf(v[i].x, v[i].y, fmaf(v[i].z, v[i].w, s));
    if (s == 123.456f) out[gw] = s;
}

int main()
{
    const int B = 64, C = 256, HW = 224 * 224, N = 34304;       // ~ the cfg2 query ROI (134 tiles of 256 pixels)
    float *feat, *out;
    hipMalloc(&feat, (size_t)B * C * HW * 4);
    hipMalloc(&out, 1 << 24);
    hipMemset(feat, 0, (size_t)B * C * HW * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb = (double)B * C * N * 4 / 1e9;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-52s %.3f ms  %.2f TB/s read\n", name, ms, gb / ms);
    };
    const int t64 = (N + 63) / 64, t256 = (N + 255) / 256;
    run("V0 dword, 64 px x 256 planes per wave, 1 wave/SIMD", [&] { hipLaunchKernelGGL((v_dword<256, 0>), dim3(B * t64 * 1 / 4), dim3(256), 0, 0, feat, C, HW, N, t64, out); });
    run("V1 dword, 64 px x 128 planes per wave, 2 waves/SIMD", [&] { hipLaunchKernelGGL((v_dword<128, 0>), dim3(B * t64 * 2 / 4), dim3(256), 0, 0, feat, C, HW, N, t64, out); });
    run("V1b dword, 64 px x 64 planes per wave", [&] { hipLaunchKernelGGL((v_dword<64, 0>), dim3(B * t64 * 4 / 4), dim3(256), 0, 0, feat, C, HW, N, t64, out); });
    run("V2 dwordx4, 256 px x 64 planes per wave, 1 wave/SIMD", [&] { hipLaunchKernelGGL((v_x4<64>), dim3(B * t256 * 4 / 4), dim3(256), 0, 0, feat, C, HW, N, t256, out); });
    run("V3 dwordx4, 256 px x 32 planes per wave, 2 waves/SIMD", [&] { hipLaunchKernelGGL((v_x4<32>), dim3(B * t256 * 8 / 4), dim3(256), 0, 0, feat, C, HW, N, t256, out); });
    run("V4 V0 + ~1500 dependent VALU per tile", [&] { hipLaunchKernelGGL((v_dword<256, 24>), dim3(B * t64 * 1 / 4), dim3(256), 0, 0, feat, C, HW, N, t64, out); });
    run("V4b V0 + ~3000 dependent VALU per tile", [&] { hipLaunchKernelGGL((v_dword<256, 48>), dim3(B * t64 * 1 / 4), dim3(256), 0, 0, feat, C, HW, N, t64, out); });
    return 0;
}
