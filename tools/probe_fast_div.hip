// Is the three-instruction division  q0 = x*y; r = fma(-q0, d, x); q = fma(r, y, q0)  with y = RN(1/d)  bit-identical to the IEEE
// division x / d on gfx950?  (Markstein's theorem says yes while q and r stay in the normal range; K0 divides every element of a row by
// the same norm, so y costs one real division per row.)  Brute force over random and adversarial operands.
//   hipcc --offload-arch=gfx950 -O3 -fhip-fp32-correctly-rounded-divide-sqrt tools/probe_fast_div.hip -o /tmp/probe_fast_div && /tmp/probe_fast_div
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

__global__ void probe(int mode, uint64_t seed, unsigned long long *n_bad, unsigned long long *n_guarded, float *ex)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad = 0, guarded = 0;
    for (int it = 0; it < 4096; ++it) {
        const uint64_t h = mix(seed ^ (tid * 4096 + it));
        uint32_t xb = (uint32_t)h, db = (uint32_t)(h >> 32);
        float x, d;
        if (mode == 0) {            // unit-norm-like: d in [2^-8, 2^8), |x| <= d, any mantissas
            d = __uint_as_float((db & 0x007fffffu) | ((119u + (db >> 28)) << 23));
            x = __uint_as_float((xb & 0x807fffffu) | ((uint32_t)(__float_as_uint(d) >> 23) - (xb >> 27 & 15u)) << 23);
        } else if (mode == 1) {     // mantissa patterns near all-ones / all-zeros, wide exponents
            const uint32_t pat[8] = {0x7fffffu, 0x7ffffeu, 0x000000u, 0x000001u, 0x400000u, 0x3fffffu, 0x555555u, 0x2aaaabu};
            d = __uint_as_float(pat[db & 7] | ((60u + (db >> 8) % 130u) << 23));
            x = __uint_as_float((xb & 0x80000000u) | pat[(xb >> 3) & 7] | ((40u + (xb >> 8) % 170u) << 23));
        } else {                    // anything: all exponents including denormals, zeros, inf, nan
            d = __uint_as_float(db & 0x7fffffffu);
            x = __uint_as_float(xb);
        }
        d = fmaxf(d, 1e-8f);
        const float y = 1.0f / d;
        const float ref = x / d;
        // guard: the fast path is taken for +0 and for 2^-100 * max(1, d) <= |x|, with d <= 2^60
        const float lo = 7.888609052210118e-31f * fmaxf(1.0f, d);
        const uint32_t ax = __float_as_uint(x) & 0x7fffffffu;
        const bool ok = (d <= 1.152921504606847e18f) && ((ax >= __float_as_uint(lo) && ax < 0x7f800000u && fabsf(x) <= d * 4.0f) || __float_as_uint(x) == 0u);
        if (!ok) { ++guarded; continue; }
        const float q0 = x * y;
        const float r = __builtin_fmaf(-q0, d, x);
        const float q = __builtin_fmaf(r, y, q0);
        if (__float_as_uint(q) != __float_as_uint(ref)) {
            if (bad == 0 && atomicAdd(n_bad, 0ull) == 0ull) { ex[0] = x; ex[1] = d; ex[2] = q; ex[3] = ref; }
            ++bad;
        }
    }
    if (bad) atomicAdd(n_bad, bad);
    atomicAdd(n_guarded, guarded);
}

int main()
{
    unsigned long long *cnt; float *ex;
    hipMalloc(&cnt, 16); hipMalloc(&ex, 16);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(cnt, 0, 16);
        for (int rep = 0; rep < 8; ++rep) hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, mode, 1234ull + 77ull * rep + 1000ull * mode, cnt, cnt + 1, ex);
        unsigned long long h[2]; float he[4];
        hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost);
        printf("mode %d: %.3g divisions, %llu guarded (slow path), %llu mismatches", mode, 8.0 * 4096 * 256 * 4096, h[1], h[0]);
        if (h[0]) printf("  e.g. x=%a d=%a fast=%a ieee=%a", he[0], he[1], he[2], he[3]);
        printf("\n");
    }
    return 0;
}
