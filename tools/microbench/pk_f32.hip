// Issue rate of packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two f32 per lane per instruction) against the plain
// one-per-lane forms on gfx950.  Question: does a packed instruction cost the issue slot of ONE plain instruction (then pairing the x/y
// halves of the 3-vector arithmetic halves those instructions) or of two?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o pk_f32 pk_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = blockIdx.x * 1e-4f + 0.5f, c = a * 0.25f, d = b + 0.125f;
    f2 p = {a, b}, q = {c, d}, r = {d, a}, s = {b, c};
    const float m0 = 1.0000001f, m1 = 0.9999999f;
    const f2 mm = {m0, m1};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) { a = a * m0; b = b * m1; c = c * m0; d = d * m1; }                         // 4 x v_mul_f32: 4 results
            if (MODE == 1) { p = p * mm; q = q * mm; r = r * mm; s = s * mm; }                         // 4 x v_pk_mul_f32: 8 results
            if (MODE == 2) { a = a + m0; b = b + m1; c = c + m0; d = d + m1; }                         // 4 x v_add_f32
            if (MODE == 3) { p = p + mm; q = q + mm; r = r + mm; s = s + mm; }                         // 4 x v_pk_add_f32
            if (MODE == 4) { a = a * m0; b = b * m1; c = c * m0; d = d * m1; p = p * mm; q = q * mm; }   // 4 plain + 2 packed: 8 results in 6 instructions
            if (MODE == 5) { p = p * q.x; r = r * s.y; q = q + mm; s = s + mm; }                       // packed with a broadcast half (op_sel)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + p.x + p.y + q.x + q.y + r.x + r.y + s.x + s.y;
}
int main() {
    float* d; hipMalloc(&d, 2048 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"4 x v_mul_f32 (4 results)", "4 x v_pk_mul_f32 (8 results)", "4 x v_add_f32", "4 x v_pk_add_f32", "4 x v_mul_f32 + 2 x v_pk_mul_f32", "2 x v_pk_mul (broadcast) + 2 x v_pk_add"};
    for (int rep = 0; rep < 2; ++rep) for (int m = 0; m < 6; ++m) {
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 4) hipLaunchKernelGGL(k<4>, dim3(2048), dim3(256), 0, 0, d, 2000);
        if (m == 5) hipLaunchKernelGGL(k<5>, dim3(2048), dim3(256), 0, 0, d, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-44s %8.3f ms\n", names[m], ms);
    }
    return 0;
}
