// 1.0f / x, correctly rounded: the compiler's expansion (v_div_scale x2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup) against a short form
// (v_rcp + one Newton step + fixup), compared EXHAUSTIVELY over all 2^32 bit patterns of x, and timed.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -o recip recip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float recip_full(float x) { return 1.0f / x; }
template <int V> __device__ __forceinline__ float recip_short(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    if (V >= 1) { const float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    if (V >= 2) { const float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    return __builtin_amdgcn_div_fixupf(r, x, 1.0f);
}
template <int V> __global__ void k_check(unsigned long long* bad, uint32_t* first_bad, uint32_t* hist) {
    const uint64_t n = 1ull << 32;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const uint32_t a = __float_as_uint(recip_full(x)), b = __float_as_uint(recip_short<V>(x));
        const bool nan_both = ((a & 0x7FFFFFFFu) > 0x7F800000u) && ((b & 0x7FFFFFFFu) > 0x7F800000u);
        if (a != b && !nan_both) { atomicAdd(bad, 1ull); atomicMin(first_bad, (uint32_t)i); atomicAdd(&hist[((uint32_t)i >> 23) & 0xFFu], 1u); }
    }
}
template <int V> __global__ void __launch_bounds__(256) k_time(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.1f, b = blockIdx.x * 1e-4f + 0.7f, c = a * 0.25f, d = b + 0.125f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (V < 0) { a = recip_full(a) + 0.5f; b = recip_full(b) + 0.5f; c = recip_full(c) + 0.5f; d = recip_full(d) + 0.5f; }
            else { a = recip_short<V < 0 ? 0 : V>(a) + 0.5f; b = recip_short<V < 0 ? 0 : V>(b) + 0.5f; c = recip_short<V < 0 ? 0 : V>(c) + 0.5f; d = recip_short<V < 0 ? 0 : V>(d) + 0.5f; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int V> void check(const char* name) {
    unsigned long long* bad; uint32_t* first; uint32_t* hist;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMalloc(&hist, 1024);
    hipMemset(bad, 0, 8); hipMemset(first, 0xFF, 4); hipMemset(hist, 0, 1024);
    hipLaunchKernelGGL(k_check<V>, dim3(256 * 8), dim3(256), 0, 0, bad, first, hist);
    unsigned long long hb; uint32_t hf, hh[256];
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 1024, hipMemcpyDeviceToHost);
    printf("%s: %llu of 2^32 inputs differ from 1.0f/x", name, hb);
    if (hb) { printf(" (first 0x%08x); by biased exponent of x:", hf); for (int e = 0; e < 256; ++e) if (hh[e]) printf(" %d:%u", e, hh[e]); }
    printf("\n");
}
template <int V> void timeit(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time<V>, dim3(256 * 7), dim3(256), 0, 0, d, 1000);
    hipEventRecord(e0); hipLaunchKernelGGL(k_time<V>, dim3(256 * 7), dim3(256), 0, 0, d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-40s %8.3f ms\n", name, ms);
}
__global__ void k_zeros(uint32_t* out) {
    const uint32_t pat[6] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x00200000u, 0x80200000u};
    for (int i = 0; i < 6; ++i) { const float x = __uint_as_float(pat[i]); out[2 * i] = __float_as_uint(recip_full(x)); out[2 * i + 1] = __float_as_uint(recip_short<1>(x)); }
}
int main() {
    { uint32_t* o; hipMalloc(&o, 48); hipLaunchKernelGGL(k_zeros, dim3(1), dim3(1), 0, 0, o); uint32_t h[12]; hipMemcpy(h, o, 48, hipMemcpyDeviceToHost);
      const char* n[6] = {"+0", "-0", "+min denormal", "-min denormal", "+2^-128", "-2^-128"};
      for (int i = 0; i < 6; ++i) printf("1/%-14s compiler 0x%08x  short 0x%08x  %s\n", n[i], h[2 * i], h[2 * i + 1], h[2 * i] == h[2 * i + 1] ? "equal" : "DIFFERENT"); }
    check<0>("v_rcp + fixup"); check<1>("v_rcp + 1 Newton step + fixup"); check<2>("v_rcp + 2 Newton steps + fixup");
    float* d; hipMalloc(&d, 256 * 7 * 256 * 4);
    timeit<-1>("1.0f / x (compiler, correctly rounded)", d); timeit<0>("v_rcp + fixup", d); timeit<1>("v_rcp + 1 Newton step + fixup", d); timeit<2>("v_rcp + 2 Newton steps + fixup", d);
    return 0;
}
