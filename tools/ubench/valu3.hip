// VALU issue rate vs waves per SIMD on gfx950 (wave64): packed f32 mul+add, scalar f32 mul+add, with shader-clock readout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define ITERS 2048
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE, int CH> __global__ __launch_bounds__(64) void k(float *out, float s0, float s1, unsigned long long *clk) {
    f2 a[CH];
    for (int i = 0; i < CH; i++) a[i] = (f2){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    f2 m = {s0, s0}, c = {s1, s1};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CH; i++) {
            if (MODE == 0) { f2 t = a[i] * m; asm volatile("" : "+v"(t)); a[i] = t + c; }                         // pk_mul + pk_add
            else { float t = a[i].x * s0; asm volatile("" : "+v"(t)); a[i].x = t + s1; }                           // v_mul + v_add
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float r = 0; for (int i = 0; i < CH; i++) r += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
    float *d; hipMalloc(&d, 1 << 24);
    unsigned long long *clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    for (int mode = 0; mode < 2; mode++) for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 1024 * wps;      // one-wave blocks: wps waves per SIMD on 256 CUs x 4 SIMDs
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((k<0, 4>), dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, clk);
            else hipLaunchKernelGGL((k<1, 4>), dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f, clk);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double instr = (double)blocks * ITERS * 4 * 2;
        const double shader_mhz = (double)h[0] / ((double)h[1] / (rate * 1e3)) / 1e6;
        printf("%-22s %d waves/SIMD: %.3f ms  %.3f T wave-instr/s  | wave 0: %.2f cycles per instr, shader clock %.0f MHz\n", mode == 0 ? "pk_mul+pk_add (4 chains)" : "v_mul+v_add (4 chains)",
               wps, ms, instr / ms / 1e9, (double)h[0] / (ITERS * 8.0), shader_mhz);
    }
    return 0;
}
