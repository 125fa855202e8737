// VALU issue-rate microbenchmark for gfx950: how many wave64 v_mul_f32 / v_add_f32 per second?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITERS 4096
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float s0, float s1) {
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], s0, s1);
            if (MODE == 1) { float t = a[i] * s0; asm volatile("" : "+v"(t)); a[i] = t + s1; }
            if (MODE == 2) { float t = a[(i + 1) & 15] * s0; asm volatile("" : "+v"(t)); a[i] = a[i] + t; }
        }
    }
    float r = 0; for (int i = 0; i < 16; i++) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void kpk(float *out, float s0, float s1) {
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = (f2){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    f2 m = {s0, s0}, c = {s1, s1};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) { f2 t = a[i] * m; asm volatile("" : "+v"(t)); a[i] = t + c; }
    }
    float r = 0; for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; mode++) for (int blocks : {256, 512, 1024, 2048, 4096}) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
            if (mode == 3) hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        double waves = blocks * 4.0, ops = waves * ITERS * 16.0;   // element-ops per lane-group
        double instr = (mode == 0) ? ops : (mode == 3 ? ops : 2 * ops);   // wave-instructions issued
        const char *nm[] = {"fma", "mul+add(dep)", "mul+add(indep mul)", "pk_mul+pk_add"};
        printf("%-20s blocks %5d: %.3f ms  %.3f T wave-instr/s  (%.1f T elem-ops/s/lane-group -> %.1f TFLOP/s)\n", nm[mode], blocks, ms,
               instr / ms / 1e9, ops / ms / 1e9, ops * 64 * (mode == 0 ? 2 : 2) / ms / 1e9);
    }
    return 0;
}
