// VALU issue-rate microbenchmark #2 for gfx950: integer SAD, f64 and conversion ops (wave64 instr/s, whole chip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define ITERS 4096
template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t s0, double d0, double d1) {
    uint32_t a[16]; double f[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 2654435761u + i; f[i] = threadIdx.x * 0.001 + i; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) a[i] = __builtin_amdgcn_sad_u8(a[i], s0, a[(i + 1) & 15]);
            if (MODE == 1) a[i] = __builtin_amdgcn_sad_u16(a[i], s0, a[(i + 1) & 15]);
            if (MODE == 2) a[i] = a[i] + (a[(i + 1) & 15] ^ s0);          // xor + add: 2 simple int ops
            if (MODE == 3) f[i] = __builtin_fma(f[i], d0, d1);
            if (MODE == 4) { double t = f[i] * d0; asm volatile("" : "+v"(t)); f[i] = t + d1; }
            if (MODE == 5) a[i] = __builtin_amdgcn_udot4(a[i], s0, a[(i + 1) & 15], false);
        }
    }
    uint32_t r = 0; double rf = 0; for (int i = 0; i < 16; i++) { r += a[i]; rf += f[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + (uint32_t)rf;
}
int main() {
    uint32_t *d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *nm[] = {"v_sad_u8", "v_sad_u16", "xor+add (2 instr)", "v_fma_f64", "mul_f64+add_f64 (2 instr)", "v_dot4_u32_u8"};
    for (int mode = 0; mode < 6; mode++) for (int blocks : {1024, 4096}) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 0x01020304u, 1.0000001, 0.5);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 0x01020304u, 1.0000001, 0.5);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, 0x01020304u, 1.0000001, 0.5);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, 0x01020304u, 1.0000001, 0.5);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, 0x01020304u, 1.0000001, 0.5);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, 0x01020304u, 1.0000001, 0.5);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        double waves = blocks * 4.0, ops = waves * ITERS * 16.0;
        double instr = (mode == 2 || mode == 4) ? 2 * ops : ops;
        printf("%-28s blocks %5d: %8.3f ms  %.3f T wave-instr/s\n", nm[mode], blocks, ms, instr / ms / 1e9);
    }
    return 0;
}
