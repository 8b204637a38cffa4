// sqrtf(x), correctly rounded: the compiler's expansion (denormal scaling, v_sqrt_f32, the two neighbours tested with fma, selects) against
// shorter forms, compared EXHAUSTIVELY over all 2^32 bit patterns of x, and timed.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -o sqrt sqrt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float sqrt_full(float x) { return sqrtf(x); }
template <int V> __device__ __forceinline__ float sqrt_short(float x) {
    if (V == 0) {                                           // rsq, one coupled Newton step on (g ~ sqrt x, h ~ 1 / (2 sqrt x)), residual correction
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y, h = 0.5f * y;
        const float r = __builtin_fmaf(-h, g, 0.5f);
        g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
        const float d = __builtin_fmaf(-g, g, x);
        const float s = __builtin_fmaf(d, h, g);
        return (x == 0.0f || x == __builtin_inff()) ? x : s;
    }
    if (V == 1) {                                           // the same without refining h
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y; const float h = 0.5f * y;
        const float r = __builtin_fmaf(-h, g, 0.5f);
        g = __builtin_fmaf(g, r, g);
        const float d = __builtin_fmaf(-g, g, x);
        const float s = __builtin_fmaf(d, h, g);
        return (x == 0.0f || x == __builtin_inff()) ? x : s;
    }
    if (V == 2) {                                           // hardware sqrt + one residual correction with the hardware rsq as 1 / (2 s)
        const float g = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
        const float d = __builtin_fmaf(-g, g, x);
        const float s = __builtin_fmaf(d, h, g);
        return (x == 0.0f || x == __builtin_inff()) ? x : s;
    }
    {                                                       // V == 3: rsq, residual correction only
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y, h = 0.5f * y;
        const float d = __builtin_fmaf(-g, g, x);
        const float s = __builtin_fmaf(d, h, g);
        return (x == 0.0f || x == __builtin_inff()) ? x : s;
    }
}
// The form rt_math.h ships (length_for_normalize): exact where it matters, "below 1e-4" where only that is asked.
__device__ __forceinline__ float sqrt_norm(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    float s = __builtin_fmaf(d, h, g);
    s = (x < 0x1p-100f) ? 0.0f : s;
    return (x == __builtin_inff()) ? x : s;
}
// Contract check over every x >= +0 (and NaN): bits equal to sqrtf(x) for x >= 2^-100, +inf and NaN; for 0 <= x < 2^-100 a value below 1e-4.
__global__ void k_contract(unsigned long long* bad) {
    const uint64_t n = 1ull << 32;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)i;
        const bool is_nan = (u & 0x7FFFFFFFu) > 0x7F800000u;
        if ((u >> 31) && !is_nan) continue;                                  // negative numbers (and -0): a sum of squares is never one
        const float x = __uint_as_float(u);
        const float want = sqrt_full(x), got = sqrt_norm(x);
        bool ok;
        if (is_nan) ok = got != got;
        else if (x >= 0x1p-100f) ok = __float_as_uint(want) == __float_as_uint(got);
        else ok = (got < 1e-4f) && (want < 1e-4f);
        if (!ok) atomicAdd(bad, 1ull);
    }
}
template <int V> __global__ void k_check(unsigned long long* bad, uint32_t* hist) {
    const uint64_t n = 1ull << 32;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const uint32_t a = __float_as_uint(sqrt_full(x)), b = __float_as_uint(sqrt_short<V>(x));
        const bool nan_both = ((a & 0x7FFFFFFFu) > 0x7F800000u) && ((b & 0x7FFFFFFFu) > 0x7F800000u);
        if (a != b && !nan_both) { atomicAdd(bad, 1ull); atomicAdd(&hist[(uint32_t)i >> 23], 1u); }   // sign + biased exponent
    }
}
template <int V> __global__ void __launch_bounds__(256) k_time(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.1f, b = blockIdx.x * 1e-4f + 0.7f, c = a * 0.25f, d = b + 0.125f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (V < 0) { a = sqrt_full(a) + 0.5f; b = sqrt_full(b) + 0.5f; c = sqrt_full(c) + 0.5f; d = sqrt_full(d) + 0.5f; }
            else { a = sqrt_short<V < 0 ? 0 : V>(a) + 0.5f; b = sqrt_short<V < 0 ? 0 : V>(b) + 0.5f; c = sqrt_short<V < 0 ? 0 : V>(c) + 0.5f; d = sqrt_short<V < 0 ? 0 : V>(d) + 0.5f; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int V> void check(const char* name) {
    unsigned long long* bad; uint32_t* hist;
    hipMalloc(&bad, 8); hipMalloc(&hist, 2048); hipMemset(bad, 0, 8); hipMemset(hist, 0, 2048);
    hipLaunchKernelGGL(k_check<V>, dim3(256 * 8), dim3(256), 0, 0, bad, hist);
    unsigned long long hb; uint32_t hh[512];
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hh, hist, 2048, hipMemcpyDeviceToHost);
    printf("%s: %llu of 2^32 inputs differ from sqrtf(x)", name, hb);
    if (hb) { printf("; by sign/biased exponent of x (ranges):"); int e = 0; while (e < 512) { if (!hh[e]) { ++e; continue; } int f = e; unsigned long long sum = 0; while (f < 512 && hh[f]) { sum += hh[f]; ++f; } printf(" [%d..%d]:%llu", e, f - 1, sum); e = f; } }
    printf("\n");
}
template <int V> void timeit(const char* name, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time<V>, dim3(256 * 7), dim3(256), 0, 0, d, 1000);
    hipEventRecord(e0); hipLaunchKernelGGL(k_time<V>, dim3(256 * 7), dim3(256), 0, 0, d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-52s %8.3f ms\n", name, ms);
}
int main() {
    check<0>("rsq + coupled step + residual"); check<1>("rsq + g step + residual"); check<2>("sqrt + residual (h from rsq)"); check<3>("rsq + residual");
    { unsigned long long* bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8); hipLaunchKernelGGL(k_contract, dim3(256 * 8), dim3(256), 0, 0, bad);
      unsigned long long hb; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
      printf("length_for_normalize contract (x >= +0, NaN): %llu violations of 2^31 + NaNs\n", hb); }
    float* d; hipMalloc(&d, 256 * 7 * 256 * 4);
    timeit<-1>("sqrtf(x) (compiler, correctly rounded)", d); timeit<0>("rsq + coupled step + residual", d); timeit<1>("rsq + g step + residual", d);
    timeit<2>("sqrt + residual (h from rsq)", d); timeit<3>("rsq + residual", d);
    return 0;
}
