// ds_read_b128 throughput of the LDS fragment-read address patterns used by the kernels, against a lane-linear read:
//   hipcc --offload-arch=gfx950 -O3 tools/probe_lds_pattern.hip -o /tmp/probe_lds && /tmp/probe_lds
// (a pattern that costs more cycles per instruction than the linear one has bank conflicts inside the hardware's lane groups)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void probe(int *out, int iters)
{
    __shared__ __attribute__((aligned(256))) char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<int *>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
    unsigned off[8];
    for (int c = 0; c < 8; ++c) {
        if (PAT == 0) off[c] = (unsigned)(lane * 16 + c * 1024);                                             // lane-linear
        if (PAT == 1) off[c] = (unsigned)(l31 * 256 + (((hi ^ (l31 & 15)) ^ (2 * c)) << 4));                 // screen: 256-B rows, chunk ^ (row & 15)
        if (PAT == 2) off[c] = (unsigned)((l31 + 32 * (c >> 1)) * 64 + ((((c & 1) * 2 + hi) ^ ((l31 >> 2) & 3)) << 4));   // GEMM: 64-B rows, slot ^ ((row >> 2) & 3)
        if (PAT == 3) off[c] = (unsigned)(l31 * 256 + (((2 * c + hi) ^ ((l31 & 7) * 2 + ((l31 >> 3) & 1))) << 4));       // 256-B rows, alternative swizzle
        if (PAT == 4) off[c] = (unsigned)(l31 * 272 + ((2 * c + hi) << 4));                                   // 256-B rows padded by 16 bytes
        off[c] += (unsigned)(wave * 8192 * (PAT == 4 ? 0 : 1)) & 32767u;
    }
    i32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            asm volatile("" : "+v"(off[c]));                      // keep the read inside the loop
            const i32x4 v = *reinterpret_cast<const i32x4 *>(lds + off[c]);
            acc += v;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int PAT>
static void run(const char *name)
{
    int *out;
    (void)hipMalloc(&out, sizeof(int) * 256 * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe<PAT>, dim3(256), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double insts = 4.0 * iters * 8;             // wave-instructions per CU
    printf("%-52s %.3f ms  -> %.2f ns per ds_read_b128 per CU\n", name, ms, ms * 1e6 / insts);
    (void)hipFree(out);
}

int main()
{
    run<0>("lane-linear");
    run<1>("screen: 256-B rows, chunk ^ (row & 15)");
    run<2>("GEMM: 64-B rows, slot ^ ((row >> 2) & 3)");
    run<3>("256-B rows, alternative swizzle");
    run<4>("272-B rows (16-byte pad)");
    return 0;
}
