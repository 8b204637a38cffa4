// The proofs behind rt_math.h's short arithmetic, run on the SHIPPED functions (this file includes the product's headers and is built with the
// product's flags): every one is compared with the compiler's correctly rounded form over its whole argument space.
//   recip_normal_range(x) == 1.0f / x      for all 2^32 x except denormals and |x| >= 2^126 (and is checked to differ ONLY there)
//   recip3<true>(x, 1.5, -3)               == 1.0f / x for ALL 2^32 x (its ballot sends the others to the compiler's division)
//   length_for_normalize(x)                == sqrtf(x) for x >= 2^-100, +inf, NaN; below 1e-4 (and not NaN) for 0 <= x < 2^-100
//   div_bounded(a, b), div_by_rn(a, b, 1/b) == a / b for the pairs of significands enumerated (argv[1] chunks of 2^17 b x all 2^23 a; 64 = all 2^46)
//   u32_to_range11(w)                      == (value1_2 - 1) * 2 - 1 for all 2^23 mantissas
// usage: short_arithmetic [division chunks, default 2]     exit code 0 = every check passed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../raytracer-rust_amd/csrc/device/rt_math.h"
#include "../../raytracer-rust_amd/csrc/device/rt_rng.h"
using namespace mi355rt;
__device__ bool same(float a, float b) { const uint32_t x = __float_as_uint(a), y = __float_as_uint(b); return x == y || (((x & 0x7FFFFFFFu) > 0x7F800000u) && ((y & 0x7FFFFFFFu) > 0x7F800000u)); }
__global__ void k_recip(unsigned long long* out) {          // out[0]: differences inside the proven range, out[1]: outside it, out[2]: recip3 differences
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)i, e = (u >> 23) & 0xFFu;
        const float x = __uint_as_float(u), want = 1.0f / x;
        const bool in_range = (e >= 1u && e <= 252u) || e == 255u || (u & 0x7FFFFFFFu) == 0u;
        if (!same(want, recip_normal_range(x))) atomicAdd(&out[in_range ? 0 : 1], 1ull);
        float ix, iy, iz; recip3<true>(x, 1.5f, -3.0f, ix, iy, iz);
        if (!same(want, ix) || !same(iy, 1.0f / 1.5f) || !same(iz, 1.0f / -3.0f)) atomicAdd(&out[2], 1ull);
    }
}
__global__ void k_sqrt(unsigned long long* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)i; const bool is_nan = (u & 0x7FFFFFFFu) > 0x7F800000u;
        if ((u >> 31) && !is_nan) continue;                                  // a sum of squares is never negative, never -0
        const float x = __uint_as_float(u), want = sqrtf(x), got = length_for_normalize(x);
        const bool ok = is_nan ? (got != got) : (x >= 0x1p-100f ? __float_as_uint(want) == __float_as_uint(got) : (got < 1e-4f && want < 1e-4f));
        if (!ok) atomicAdd(&out[0], 1ull);
    }
}
__global__ void __launch_bounds__(256) k_div(unsigned long long* out, uint32_t b_base) {
    const float b = __uint_as_float(0x3F800000u | (b_base + blockIdx.x * blockDim.x + threadIdx.x));
    unsigned long long bad = 0;
    const float rb = 1.0f / b;                                           // what the host hands div_by_rn(): the correctly rounded reciprocal
    for (uint32_t am = 0; am < (1u << 23); ++am) { const float a = __uint_as_float(0x3F800000u | am); const uint32_t want = __float_as_uint(a / b);
                                                   bad += (want != __float_as_uint(div_bounded(a, b))) + (want != __float_as_uint(div_by_rn(a, b, rb))); }
    if (bad) atomicAdd(&out[0], bad);
}
__global__ void k_div_specials(unsigned long long* out) {   // zeros, infinities, NaN, the ends of the stated range, negative operands
    const float as[] = {0.0f, -0.0f, 0x1p-100f, -0x1p-100f, 0x1.fffffep99f, 1.0f, -3.0f, 0x1p-24f, 16777215.0f, __builtin_inff(), -__builtin_inff(), __builtin_nanf("")};
    const float bs[] = {0x1p-25f, -0x1p-25f, 0x1.fffffep24f, 1.0f, -1.0f, 3.0f, 800.0f, 600.0f, 1e-4f, 0.7071068f, __builtin_inff(), __builtin_nanf("")};
    for (float a : as) for (float b : bs) if (!same(a / b, div_bounded(a, b)) || !same(a / b, div_by_rn(a, b, 1.0f / b))) atomicAdd(&out[0], 1ull);
}
__global__ void k_range11(unsigned long long* out) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < (1u << 23); k += gridDim.x * blockDim.x) {
        const float v12 = __uint_as_float(k | 0x3F800000u), want = (v12 - 1.0f) * 2.0f + -1.0f;
        if (__float_as_uint(want) != __float_as_uint(u32_to_range11(k << 9)) || __float_as_uint(want) != __float_as_uint(u32_to_range11((k << 9) | 0x1FFu))) atomicAdd(&out[0], 1ull);
    }
}
int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 2;
    unsigned long long* d; if (hipMalloc(&d, 64) != hipSuccess) { printf("no device\n"); return 2; }
    unsigned long long h[8]; int fails = 0;
    auto fetch = [&]() { hipDeviceSynchronize(); hipMemcpy(h, d, 64, hipMemcpyDeviceToHost); hipMemset(d, 0, 64); };
    hipMemset(d, 0, 64);
    hipLaunchKernelGGL(k_recip, dim3(2048), dim3(256), 0, 0, d); fetch();
    printf("recip_normal_range: %llu differences inside the proven range (want 0), %llu outside it (denormals, |x| >= 2^126: the reason for the range)\n", h[0], h[1]);
    printf("recip3<true>: %llu differences over all 2^32 x (want 0)\n", h[2]); fails += h[0] != 0 || h[2] != 0 || h[1] == 0;
    hipLaunchKernelGGL(k_sqrt, dim3(2048), dim3(256), 0, 0, d); fetch();
    printf("length_for_normalize: %llu contract violations over every x >= +0 and every NaN (want 0)\n", h[0]); fails += h[0] != 0;
    for (int c = 0; c < chunks && c < 64; ++c) hipLaunchKernelGGL(k_div, dim3(512), dim3(256), 0, 0, d, (uint32_t)(c * (64 / (chunks < 64 ? chunks : 64))) << 17);
    fetch();
    printf("div_bounded, div_by_rn: %llu differences over %d x 2^17 b significands x all 2^23 a significands (want 0)\n", h[0], chunks); fails += h[0] != 0;
    hipLaunchKernelGGL(k_div_specials, dim3(1), dim3(1), 0, 0, d); fetch();
    printf("div_bounded: %llu differences on zeros / infinities / NaN / range ends (want 0)\n", h[0]); fails += h[0] != 0;
    hipLaunchKernelGGL(k_range11, dim3(512), dim3(256), 0, 0, d); fetch();
    printf("u32_to_range11: %llu differences over all 2^23 mantissas (want 0)\n", h[0]); fails += h[0] != 0;
    printf(fails ? "FAILED\n" : "all checks passed\n");
    return fails ? 1 : 0;
}
