// a / b, correctly rounded: the compiler's expansion (v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup) against
//   r = RN(1/b) by v_rcp + one Newton step (exact: recip.hip);  q0 = a*r;  e = fma(-b, q0, a);  q = fma(e, r, q0)        [V = 1]
//   ... and with a second correction  e2 = fma(-b, q, a);  q2 = fma(e2, r, q)                                            [V = 2]
// A function of two floats cannot be compared over 2^64 arguments, but rounding commutes with scaling by powers of two as long as nothing
// leaves the normal range, so the outcome depends only on the two SIGNIFICANDS: all 2^23 x 2^23 = 7.0e13 pairs a, b in [1, 2) are
// enumerated here (the quotient covers both binades (0.5, 1) and [1, 2)).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -o div div.hip
// usage: ./div [V] [first b chunk] [chunks of 2^17 b values]   (64 chunks = everything)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int V> __device__ __forceinline__ float div_short(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    { const float e = __builtin_fmaf(-b, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    float q = a * r;
    if (V != 3) { const float e = __builtin_fmaf(-b, q, a); q = __builtin_fmaf(e, r, q); }        // V == 3: no correction (checks the harness: must differ)
    if (V == 2) { const float e = __builtin_fmaf(-b, q, a); q = __builtin_fmaf(e, r, q); }
    return q;
}
template <int V> __global__ void __launch_bounds__(256) k_enum(unsigned long long* bad, uint32_t* first, uint32_t b_base) {
    const uint32_t bm = b_base + blockIdx.x * blockDim.x + threadIdx.x;             // significand of b
    const float b = __uint_as_float(0x3F800000u | bm);
    unsigned long long n_bad = 0;
    for (uint32_t am = 0; am < (1u << 23); ++am) {
        const float a = __uint_as_float(0x3F800000u | am);
        const float want = a / b, got = div_short<V>(a, b);
        if (__float_as_uint(want) != __float_as_uint(got)) { if (n_bad == 0) { first[0] = am; first[1] = bm; } ++n_bad; }
    }
    if (n_bad) atomicAdd(bad, n_bad);
}
template <int V> __global__ void __launch_bounds__(256) k_time(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.1f, b = blockIdx.x * 1e-4f + 0.7f, c = a * 0.25f + 1.f, d = b + 1.125f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (V == 0) { a = a / c + 0.5f; b = b / d + 0.5f; c = c / a + 0.5f; d = d / b + 0.5f; }
            else { a = div_short<V == 0 ? 1 : V>(a, c) + 0.5f; b = div_short<V == 0 ? 1 : V>(b, d) + 0.5f; c = div_short<V == 0 ? 1 : V>(c, a) + 0.5f; d = div_short<V == 0 ? 1 : V>(d, b) + 0.5f; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int V> void timeit(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time<V>, dim3(256 * 7), dim3(256), 0, 0, d, 1000);
    hipEventRecord(e0); hipLaunchKernelGGL(k_time<V>, dim3(256 * 7), dim3(256), 0, 0, d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-44s %8.3f ms\n", name, ms);
}
int main(int argc, char** argv) {
    const int V = argc > 1 ? atoi(argv[1]) : 1, c0 = argc > 2 ? atoi(argv[2]) : 0, nc = argc > 3 ? atoi(argv[3]) : 64;
    unsigned long long* bad; uint32_t* first; hipMalloc(&bad, 8); hipMalloc(&first, 8); hipMemset(bad, 0, 8); hipMemset(first, 0, 8);
    for (int c = c0; c < c0 + nc && c < 64; ++c) {                                   // 2^17 values of b per launch, every a for each
        if (V == 3) hipLaunchKernelGGL(k_enum<3>, dim3(512), dim3(256), 0, 0, bad, first, (uint32_t)c << 17);
        else if (V == 1) hipLaunchKernelGGL(k_enum<1>, dim3(512), dim3(256), 0, 0, bad, first, (uint32_t)c << 17);
        else hipLaunchKernelGGL(k_enum<2>, dim3(512), dim3(256), 0, 0, bad, first, (uint32_t)c << 17);
        hipDeviceSynchronize();
        unsigned long long hb; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        printf("chunk %d of 64 done, %llu pairs differ so far\n", c, hb); fflush(stdout);
    }
    unsigned long long hb; uint32_t hf[2]; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
    printf("V=%d, b significands [%d, %d) x 2^17, every a significand: %llu pairs differ from a / b", V, c0, c0 + nc, hb);
    if (hb) printf(" (one of them: a = 0x%08x, b = 0x%08x)", 0x3F800000u | hf[0], 0x3F800000u | hf[1]);
    printf("\n");
    float* d; hipMalloc(&d, 256 * 7 * 256 * 4);
    timeit<0>("a / b (compiler, correctly rounded)", d); timeit<1>("exact reciprocal + 1 correction", d); timeit<2>("exact reciprocal + 2 corrections", d);
    return 0;
}
