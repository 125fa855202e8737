// ubench: hipEventElapsedTime between events BOUND to launches (hipExtLaunchKernelGGL stopEvent), against the kernels' own stamps
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/ext_event2 tools/ubench/ext_event2.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin_kernel(unsigned long long *stamp, int slot, int ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = t0;
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot + 1] = wall_clock64();
}
int main() {
    hipStream_t A; CHK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    unsigned long long *stamp; CHK(hipHostMalloc((void **)&stamp, 64 * 16));
    hipEvent_t e0, e1, r0, r1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); CHK(hipEventCreate(&r0)); CHK(hipEventCreate(&r1));
    for (int rep = 0; rep < 5; rep++) {
        // k0 (stop = e0)  k1  k2  k3 (stop = e1):   elapsed(e0, e1) should be end(k0) -> end(k3) ~ 3 kernels + gaps
        hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, nullptr, e0, 0, stamp, 0, 1000);
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, stamp, 1, 2000);
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, stamp, 2, 3000);
        hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, nullptr, e1, 0, stamp, 3, 4000);
        CHK(hipStreamSynchronize(A));
        float ms = -1; hipError_t q = hipEventElapsedTime(&ms, e0, e1);
        printf("bound -> bound : %s  %.2f us   (stamps: end k0 -> end k3 %.2f us, start k1 -> end k3 %.2f us)\n", hipGetErrorString(q), ms * 1e3,
               (stamp[7] - stamp[1]) / 100.0, (stamp[7] - stamp[2]) / 100.0);
        // recorded start, bound stop
        CHK(hipEventRecord(r0, A));
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, stamp, 1, 2000);
        hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, nullptr, e1, 0, stamp, 3, 4000);
        CHK(hipStreamSynchronize(A));
        q = hipEventElapsedTime(&ms, r0, e1);
        printf("record -> bound: %s  %.2f us   (stamps: start k1 -> end k3 %.2f us)\n", hipGetErrorString(q), ms * 1e3, (stamp[7] - stamp[2]) / 100.0);
        // both recorded (the present bracket)
        CHK(hipEventRecord(r0, A));
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, stamp, 1, 2000);
        hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, stamp, 3, 4000);
        CHK(hipEventRecord(r1, A));
        CHK(hipStreamSynchronize(A));
        q = hipEventElapsedTime(&ms, r0, r1);
        printf("record -> record: %s  %.2f us  (stamps: start k1 -> end k3 %.2f us)\n", hipGetErrorString(q), ms * 1e3, (stamp[7] - stamp[2]) / 100.0);
        // query semantics of a bound event
        hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, A, nullptr, e0, 0, stamp, 0, 100000);
        hipError_t q0 = hipEventQuery(e0);
        CHK(hipEventSynchronize(e0));
        hipError_t q1 = hipEventQuery(e0);
        printf("query while running: %s, after synchronize: %s\n", hipGetErrorString(q0), hipGetErrorString(q1));
    }
    return 0;
}
