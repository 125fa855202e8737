// dev tool: time blur_march_kernel<N,false,128> on a 4096^2 plane with parts of the kernel switched off
// (compile with -DBLUR_ABL=<mask>: 1 no H pass, 2 no V accumulation, 4 no global loads, 8 no stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"
using namespace siftk;
template <int N> float run(const float *in, float *out, int W, int H, int nblocks_override) {
    using G = MarchGeom<N, 128>;
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = 1.0f / N;
    const int gx = (W + G::TX - 1) / G::TX;
    int want_segments = (1024 + gx - 1) / gx;
    int rows = (H + want_segments - 1) / want_segments;
    int nblocks = (rows + (N - 1) + N - 1) / N;
    if (nblocks < 3) nblocks = 3;
    if (nblocks_override) nblocks = nblocks_override;
    const int rows_out = nblocks * N - (N - 1);
    dim3 grid((unsigned)gx, (unsigned)((H + rows_out - 1) / rows_out));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((blur_march_kernel<N, false, 128, 0>), grid, dim3(128), (size_t)G::LDS_BYTES, 0, (const void *)in, out, W, H, nblocks, ta, (const uint32_t *)nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("ABL %2d  N %2d  grid %u x %u  nblocks %d  LDS %d B : %.1f us\n", BLUR_ABL, N, grid.x, grid.y, nblocks, G::LDS_BYTES, best * 1e3);
    return best;
}
int main(int argc, char **argv) {
    const int W = 4096, H = 4096;
    float *in, *out;
    hipMalloc(&in, (size_t)W * H * 4); hipMalloc(&out, (size_t)W * H * 4);
    std::vector<float> h((size_t)W * H);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 977) * 0.001f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int nb = argc > 1 ? atoi(argv[1]) : 0;
    run<11>(in, out, W, H, nb); run<15>(in, out, W, H, nb); run<27>(in, out, W, H, nb);
    return 0;
}
