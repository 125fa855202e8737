// ubench: what does a ds_add_f32 cost, by how many lanes share an address (and by how many waves per CU do it at once)?
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/lds_fadd_rate tools/ubench/lds_fadd_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) unsigned lds_u32;

template <int OP>   // 0: ds_add_f32, 1: ds_or_b32, 2: plain ds_write_b32 (the floor)
__global__ __launch_bounds__(256) void rate_kernel(int share, int stride, int iters, unsigned long long *cycles, float *sink) {
    __shared__ float acc[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = lane; k < 1024; k += 64) acc[wave][k] = 0.0f;
    __syncthreads();
    // `share` lanes per address; the groups' addresses `stride` words apart
    const int idx = ((lane / share) * stride) & 1023;
    float *p = &acc[wave][idx];
    const float v = 1.0f + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 1) (void)__hip_atomic_fetch_or(reinterpret_cast<unsigned *>(p), (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *reinterpret_cast<volatile float *>(p) = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    if (sink && acc[wave][lane] == 123.456f) sink[0] = 1.0f;
}

int main() {
    unsigned long long *cyc; CHK(hipMalloc(&cyc, 8 * 4 * 4096));
    const int iters = 200;
    printf("cycles per instruction (s_memtime domain of __builtin_readcyclecounter), one wave's view, 8 instructions back to back x %d\n", iters);
    for (int wgs_per_cu : {1, 3}) for (int op = 0; op < 3; op++) for (int share : {1, 2, 4, 8, 16, 32, 64}) for (int stride : {1, 8}) {
        if (stride == 8 && (share == 64)) continue;
        const int grid = 256 * wgs_per_cu;
        if (op == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(256), 0, 0, share, stride, iters, cyc, (float *)nullptr);
        else if (op == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(256), 0, 0, share, stride, iters, cyc, (float *)nullptr);
        else hipLaunchKernelGGL(rate_kernel<2>, dim3(grid), dim3(256), 0, 0, share, stride, iters, cyc, (float *)nullptr);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(grid * 4);
        CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto c : h) s += (double)c;
        printf("%d WG/CU (%2d waves/CU)  %-12s  %2d lanes per address, groups %d words apart: %7.1f cycles per instruction per wave\n", wgs_per_cu, 4 * wgs_per_cu,
               op == 0 ? "ds_add_f32" : (op == 1 ? "ds_or_b32" : "ds_write_b32"), share, stride, s / h.size() / (8.0 * iters));
    }
    return 0;
}
