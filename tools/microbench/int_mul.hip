// Issue rate of 32-bit integer multiplies vs adds on gfx950 (Philox4x32 uses 20 32x32->64 multiplies per call).
// Build: hipcc --offload-arch=gfx950 -O3 -o int_mul int_mul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(unsigned* out, int iters) {
    unsigned a = threadIdx.x * 2654435761u + 1u, b = blockIdx.x * 40503u + 7u, c = a ^ b, d = a + b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) { a = a + b; b = b + c; c = c + d; d = d + a; }                                    // v_add_u32
            if (MODE == 1) { a = a * 0xD2511F53u; b = b * 0xCD9E8D57u; c = c * 0x9E3779B9u; d = d * 0xBB67AE85u; }   // v_mul_lo_u32
            if (MODE == 2) { a = __umulhi(a, 0xD2511F53u) ^ b; b = __umulhi(b, 0xCD9E8D57u) ^ c; c = __umulhi(c, 0x9E3779B9u) ^ d; d = __umulhi(d, 0xBB67AE85u) ^ a; }   // v_mul_hi_u32 + xor
            if (MODE == 4) { unsigned long long p, q; unsigned long long cy;                                  // the same round on v_mad_u64_u32 (one instruction per 32x32->64)
                             asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(cy) : "v"(a), "v"(0xD2511F53u));
                             asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(q), "=s"(cy) : "v"(c), "v"(0xCD9E8D57u));
                             a = (unsigned)(q >> 32) ^ b; b = (unsigned)q; c = (unsigned)(p >> 32) ^ d; d = (unsigned)p; }
            if (MODE == 3) { unsigned long long p = (unsigned long long)a * 0xD2511F53u, q = (unsigned long long)c * 0xCD9E8D57u;
                             a = (unsigned)(q >> 32) ^ b; b = (unsigned)q; c = (unsigned)(p >> 32) ^ d; d = (unsigned)p; }        // one Philox round shape
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
int main() {
    unsigned* d; hipMalloc(&d, 2048 * 256 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"4 x v_add_u32", "4 x v_mul_lo_u32", "4 x (v_mul_hi_u32 + xor)", "Philox round (2 x 32x32->64 + 2 xor)", "Philox round on 2 x v_mad_u64_u32"};
    for (int rep = 0; rep < 2; ++rep) for (int m = 0; m < 5; ++m) {
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 4) hipLaunchKernelGGL(k<4>, dim3(2048), dim3(256), 0, 0, d, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-40s %8.3f ms per 64 groups\n", names[m], ms);
    }
    return 0;
}
