// dev tool: is the marching blur throughput-bound or phase/latency-bound at 4096^2?  Run 1, 2 and 3 independent
// launches concurrently on separate streams (different planes) and compare with back-to-back execution.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"
using namespace siftk;
template <int N> void run(int W, int H) {
    using G = MarchGeom<N, 128>;
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = 1.0f / N;
    const int gx = (W + G::TX - 1) / G::TX;
    int want_segments = (1024 + gx - 1) / gx;
    int rows = (H + want_segments - 1) / want_segments;
    int nblocks = (rows + (N - 1) + N - 1) / N;
    if (nblocks < 3) nblocks = 3;
    const int rows_out = nblocks * N - (N - 1);
    dim3 grid((unsigned)gx, (unsigned)((H + rows_out - 1) / rows_out));
    const int K = 3;
    float *in[K], *out[K]; hipStream_t st[K];
    for (int k = 0; k < K; k++) {
        hipMalloc(&in[k], (size_t)W * H * 4); hipMalloc(&out[k], (size_t)W * H * 4);
        hipMemset(in[k], 0, (size_t)W * H * 4);
        hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int conc = 1; conc <= K; conc++) {
        float best = 1e9;
        for (int rep = 0; rep < 6; rep++) {
            hipDeviceSynchronize();
            hipEventRecord(e0, st[0]);
            for (int k = 1; k < conc; k++) hipStreamWaitEvent(st[k], e0, 0);
            for (int k = 0; k < conc; k++)
                hipLaunchKernelGGL((blur_march_kernel<N, false, 128, 0>), grid, dim3(128), (size_t)G::LDS_BYTES, st[k], (const void *)in[k], out[k], W, H, nblocks, ta, (const uint32_t *)nullptr);
            hipEvent_t j[K];
            for (int k = 1; k < conc; k++) { hipEventCreateWithFlags(&j[k], hipEventDisableTiming); hipEventRecord(j[k], st[k]); hipStreamWaitEvent(st[0], j[k], 0); }
            hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
            for (int k = 1; k < conc; k++) hipEventDestroy(j[k]);
        }
        printf("N %2d  %d concurrent launches: %.1f us total = %.1f us per plane\n", N, conc, best * 1e3, best * 1e3 / conc);
    }
    for (int k = 0; k < K; k++) { hipFree(in[k]); hipFree(out[k]); }
}
int main() { run<11>(4096, 4096); run<15>(4096, 4096); run<27>(4096, 4096); return 0; }
