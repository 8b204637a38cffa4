// Sustained shader clock under a VALU-saturating load: s_memtime (shader clock cycles) against s_memrealtime (100 MHz reference) per wave,
// with every SIMD holding 7 busy waves for ~20 ms.  bench.py's VALU issue peak assumes 2.4 GHz; this says what the chip actually runs at
// while a kernel like k_render_ctr_simple is resident.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o clock clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
// MODE 0: 4 chains of v_mul_f32 + v_add_f32 (the renderer's mix; no contraction)   1: 8 independent chains of the same
//      2: v_fma_f32 only, 8 chains   3: v_add_u32 only, 8 chains                       (8 VALU instructions per group in every mode)
template <int MODE> __global__ void __launch_bounds__(256) k(float* out, unsigned long long* t, int iters) {
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    float a = threadIdx.x * 1e-3f + 1.0f, b = blockIdx.x * 1e-4f + 0.5f, c = a * 0.25f, d = b + 0.125f, e = a + 2.f, f = b + 3.f, g = c + 4.f, h = d + 5.f;
    unsigned ua = threadIdx.x, ub = blockIdx.x, uc = ua * 3u, ud = ub * 5u, ue = 1u, uf = 2u, ug = 3u, uh = 4u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) { a = a * 1.0000001f + b; b = b * 0.9999999f + c; c = c * 1.0000001f + d; d = d * 0.9999999f + a; }
            if (MODE == 1) { a = a * 1.0000001f; b = b * 0.9999999f; c = c + 1.0000001f; d = d + 0.9999999f; e = e * 1.0000001f; f = f * 0.9999999f; g = g + 1.0000001f; h = h + 0.9999999f; }
            if (MODE == 2) { a = __builtin_fmaf(a, 1.0000001f, 0.5f); b = __builtin_fmaf(b, 0.9999999f, 0.5f); c = __builtin_fmaf(c, 1.0000001f, 0.5f); d = __builtin_fmaf(d, 0.9999999f, 0.5f);
                             e = __builtin_fmaf(e, 1.0000001f, 0.5f); f = __builtin_fmaf(f, 0.9999999f, 0.5f); g = __builtin_fmaf(g, 1.0000001f, 0.5f); h = __builtin_fmaf(h, 0.9999999f, 0.5f); }
            if (MODE == 3) { ua += 0x9E3779B9u; ub += 0x7F4A7C15u; uc += 0x85EBCA6Bu; ud += 0xC2B2AE35u; ue += 0x27D4EB2Fu; uf += 0x165667B1u; ug += 0xD3A2646Cu; uh += 0xFD7046C5u; }
        }
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(ua ^ ub ^ uc ^ ud ^ ue ^ uf ^ ug ^ uh);
    if ((threadIdx.x & 63) == 0) { const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; t[2 * w] = c1 - c0; t[2 * w + 1] = r1 - r0; }
}
int main() {
    const int blocks = 256 * 7, threads = 256; int waves = 256 * 8 * threads / 64;
    float* d; hipMalloc(&d, (size_t)256 * 8 * threads * sizeof(float));
    unsigned long long* t; hipMalloc(&t, (size_t)waves * 16);
    std::vector<unsigned long long> h(2 * (size_t)waves);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"warm-up", "4 chains mul+add", "8 chains mul/add", "8 chains v_fma_f32", "8 chains v_add_u32", "4 chains mul+add, 4 blocks per CU", "4 chains mul+add, 8 blocks per CU"};
    for (int rep = 0; rep < 7; ++rep) {
        const int iters = rep == 0 ? 2000 : 60000;
        const int blocks_now = rep == 5 ? 256 * 4 : rep == 6 ? 256 * 8 : blocks;
        hipEventRecord(e0);
        if (rep <= 1 || rep >= 5) hipLaunchKernelGGL(k<0>, dim3(blocks_now), dim3(threads), 0, 0, d, t, iters);
        if (rep == 2) hipLaunchKernelGGL(k<1>, dim3(blocks_now), dim3(threads), 0, 0, d, t, iters);
        if (rep == 3) hipLaunchKernelGGL(k<2>, dim3(blocks_now), dim3(threads), 0, 0, d, t, iters);
        if (rep == 4) hipLaunchKernelGGL(k<3>, dim3(blocks_now), dim3(threads), 0, 0, d, t, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        const int waves_now = blocks_now * threads / 64;
        for (int w = 0; w < waves_now; ++w) if (h[2 * w + 1]) mhz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] / 100.0));   // cycles per microsecond of the 100 MHz reference
        std::sort(mhz.begin(), mhz.end());
        const double insts = (double)waves_now * iters * 16 * 8;                  // 4 mul + 4 add per group (no contraction)
        printf("%-36s %.2f ms, shader clock min %.0f / median %.0f / max %.0f MHz; %.1f G wave-instructions/s = %.3f of 1228.8\n", names[rep], ms,
               mhz.front(), mhz[mhz.size() / 2], mhz.back(), insts / ms / 1e6, insts / ms / 1e6 / 1228.8);
    }
    return 0;
}
