// Issue cost of the VALU instructions the render kernels are made of, in units of one v_mul_f32 / v_add_f32 (wave64, gfx950), measured with
// 7 waves per SIMD: each mode runs 8 independent copies of one instruction per group, 16 groups per loop iteration.
// Build: hipcc --offload-arch=gfx950 -O3 -o inst_cost inst_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(OP) OP(a) OP(b) OP(c) OP(d) OP(e) OP(f) OP(g) OP(h)
template <int MODE> __global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = blockIdx.x * 1e-4f + 0.5f, c = a * 0.25f, d = b + 0.125f, e = a + 2.f, f = b + 3.f, g = c + 4.f, h = d + 5.f;
    const float m = 1.0000001f, n = 0.5f;
    unsigned long long p = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#define MUL(x)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define FMA(x)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
#define FMAC(x)  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
#define RCP(x)   asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define SQRT(x)  asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define DSCALE(x) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(m) : "vcc");
#define DFMAS(x) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n) : "vcc");
#define DFIX(x)  asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
#define CND(x)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : "vcc");
#define MIN3(x)  asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
#define MAX(x)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(m));
#define CMP(x)   asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(m) : "vcc");
#define MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(p) : "v"(x), "v"(m) : "vcc");
#define CVT(x)   asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define XOR(x)   asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(m));
#define MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(m));
#define FREXP(x) asm volatile("v_frexp_exp_i32_f32 %0, %0" : "+v"(x));
#define MED3(x)  asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(n));
            if (MODE == 0) { R8(MUL) }   if (MODE == 1) { R8(FMA) }    if (MODE == 2) { R8(FMAC) }  if (MODE == 3) { R8(RCP) }
            if (MODE == 4) { R8(SQRT) }  if (MODE == 5) { R8(DSCALE) } if (MODE == 6) { R8(DFMAS) } if (MODE == 7) { R8(DFIX) }
            if (MODE == 8) { R8(CND) }   if (MODE == 9) { R8(MIN3) }   if (MODE == 10) { R8(MAX) }  if (MODE == 11) { R8(CMP) }
            if (MODE == 12) { R8(MAD64) } if (MODE == 13) { R8(CVT) }  if (MODE == 14) { R8(XOR) }  if (MODE == 15) { R8(MULLO) }
            if (MODE == 16) { R8(FREXP) } if (MODE == 17) { R8(MED3) }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)p;
}
template <int MODE> float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 7), dim3(256), 0, 0, d, iters / 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 7), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 7 * 256 * sizeof(float));
    const int it = 20000;
    const char* names[] = {"v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_rcp_f32", "v_sqrt_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_cndmask_b32", "v_min3_f32",
                           "v_max_f32", "v_cmp_lt_f32", "v_mad_u64_u32", "v_cvt_f32_u32", "v_xor_b32", "v_mul_lo_u32", "v_frexp_exp_i32_f32", "v_med3_f32"};
    float ms[18];
    ms[0] = run<0>(d, it); ms[1] = run<1>(d, it); ms[2] = run<2>(d, it); ms[3] = run<3>(d, it); ms[4] = run<4>(d, it); ms[5] = run<5>(d, it);
    ms[6] = run<6>(d, it); ms[7] = run<7>(d, it); ms[8] = run<8>(d, it); ms[9] = run<9>(d, it); ms[10] = run<10>(d, it); ms[11] = run<11>(d, it);
    ms[12] = run<12>(d, it); ms[13] = run<13>(d, it); ms[14] = run<14>(d, it); ms[15] = run<15>(d, it); ms[16] = run<16>(d, it); ms[17] = run<17>(d, it);
    for (int i = 0; i < 18; ++i) printf("%-22s %8.3f ms  = %.2f x v_mul_f32\n", names[i], ms[i], ms[i] / ms[0]);
    return 0;
}
