// ubench: what does a cross-stream dependency cost on the producer's stream and on the consumer's?
//   (a) hipEventRecord between two kernels of stream A + hipStreamWaitEvent on stream B
//   (b) the event attached to the producing launch itself (hipExtLaunchKernelGGL stopEvent): no packet of its own
// every kernel stamps wall_clock64() (100 MHz) at its start and end; printed: A's gap around the event, B's start latency.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ext_event tools/ubench/ext_event.hip && /tmp/ext_event
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long *stamp, int slot, int ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = t0;
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot + 1] = wall_clock64();
}

int main() {
    hipStream_t A, B;
    CHK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    unsigned long long *stamp; CHK(hipHostMalloc((void **)&stamp, 64 * 16));
    const unsigned flagsets[3] = {hipEventDefault, hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence};
    const char *fname[3] = {"default", "no timing", "no timing, no system fence"};
    const int T = 2000;   // 20 us per kernel
    for (int grid : {1, 1024}) for (int mode = 0; mode < 3; mode++) for (int fs = 0; fs < 3; fs++) {
        hipEvent_t ev; CHK(hipEventCreateWithFlags(&ev, flagsets[fs]));
        std::vector<double> gapA, latB, gap01;
        for (int rep = 0; rep < 40; rep++) {
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, A, stamp, 0, T);
            if (mode == 0) {            // no dependency at all: the floor
                hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, A, stamp, 1, T);
            } else if (mode == 1) {     // event record between the kernels
                hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, A, stamp, 1, T);
                CHK(hipEventRecord(ev, A));
                CHK(hipStreamWaitEvent(B, ev, 0));
            } else {                    // event rides on the launch
                hipExtLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, A, nullptr, ev, 0, stamp, 1, T);
                CHK(hipStreamWaitEvent(B, ev, 0));
            }
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, A, stamp, 2, T);
            if (mode) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, B, stamp, 3, T);
            CHK(hipStreamSynchronize(A)); CHK(hipStreamSynchronize(B));
            if (rep < 8) continue;
            gap01.push_back((double)(stamp[2] - stamp[1]) / 100.0);
            gapA.push_back((double)(stamp[4] - stamp[3]) / 100.0);
            if (mode) latB.push_back((double)(stamp[6] - stamp[3]) / 100.0);
        }
        auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("grid %4d  %-22s %-28s: A gap k0->k1 %5.2f us, k1->[event]->k2 %5.2f us, B starts %5.2f us after k1\n", grid,
               mode == 0 ? "no event" : (mode == 1 ? "hipEventRecord" : "ext launch stopEvent"), fname[fs], med(gap01), med(gapA), med(latB));
        CHK(hipEventDestroy(ev));
        if (mode == 0) break;
    }
    return 0;
}
