// Does gfx950 skip the second 32-lane pass of a wave64 VALU instruction when that half of EXEC is all zero?
// Times a long dependent-free VALU loop executed by (a) all 64 lanes, (b) lanes 0-31 only, (c) even lanes only,
// (d) lanes 0-15 only.  Build: hipcc --offload-arch=gfx950 -O3 -o half_wave half_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, int mode, int iters) {
    const unsigned lane = threadIdx.x & 63u;
    bool on = mode == 0 ? true : mode == 1 ? lane < 32u : mode == 2 ? (lane & 1u) == 0u : lane < 16u;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    if (on) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.9999f + 1e-4f; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"all 64 lanes", "lanes 0-31", "even lanes", "lanes 0-15"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d, mode, 4000); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-14s %8.3f ms\n", names[mode], ms);
        }
    return 0;
}
