// dev: achievable read-only HBM bandwidth on one MI355X for a few load widths / grid sizes (sum reduction over 402 MB,
// the size of the six planes the extrema kernel reads)   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_bw tools/ubench/read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename T, int U> __global__ __launch_bounds__(256) void rd(const T *__restrict__ p, size_t n, float *out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { const float *f = reinterpret_cast<const float *>(&v[u]); for (int k = 0; k < (int)(sizeof(T) / 4); k++) acc += f[k]; }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <typename T, int U> int run(const char *name, const void *buf, size_t bytes, float *out, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t n = bytes / sizeof(T);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((rd<T, U>), dim3(blocks), dim3(256), 0, 0, (const T *)buf, n, out);
    CK(hipEventRecord(a, 0));
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL((rd<T, U>), dim3(blocks), dim3(256), 0, 0, (const T *)buf, n, out);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-14s unroll %d blocks %5d: %7.1f us  %5.2f TB/s\n", name, U, blocks, 100.0 * ms, bytes / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
int main() {
    const size_t bytes = (size_t)6 * 4096 * 4096 * 4;
    void *buf; float *out; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 0, bytes));
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        run<float, 4>("dword", buf, bytes, out, blocks);
        run<float2, 4>("dwordx2", buf, bytes, out, blocks);
        run<float4, 4>("dwordx4", buf, bytes, out, blocks);
        run<float4, 8>("dwordx4", buf, bytes, out, blocks);
    }
    return 0;
}
