// dev tool: the product's team-form marching blur (blur_team_kernel: 2 H waves + 2 V waves per workgroup, triple-buffered
// sub-blocks) against its one-block form (blur_march_kernel): bitwise equality + time on a 4096^2 plane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"
using namespace siftk;
static int g_wgs = 1024;
template <int N> int nblocks_for(int W, int H, int TX, int wgs) {
    const int gx = (W + TX - 1) / TX;
    int want_segments = (wgs + gx - 1) / gx;
    int rows = (H + want_segments - 1) / want_segments;
    int nblocks = (rows + (N - 1) + N - 1) / N;
    return nblocks < 3 ? 3 : nblocks;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    return best * 1e3f;
}
template <int N, int S, int HW = 2> void run(const float *in, float *o1, float *o2, int W, int H, const float *taps) {
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = taps[i];
    using G1 = MarchGeom<N, 128>; using G2 = March2Geom<N, 128, S>;
    const int nb = nblocks_for<N>(W, H, G1::TX, 1024);
    const int rows_out = nb * N - (N - 1);
    dim3 grid((unsigned)((W + G1::TX - 1) / G1::TX), (unsigned)((H + rows_out - 1) / rows_out));
    const int nb2 = nblocks_for<N>(W, H, G2::TX, g_wgs);
    const int rows_out2 = nb2 * N - (N - 1);
    dim3 grid2((unsigned)((W + G2::TX - 1) / G2::TX), (unsigned)((H + rows_out2 - 1) / rows_out2));
    const size_t lds3 = (size_t)3 * G2::LDS_BYTES;
    hipMemset(o1, 0, (size_t)W * H * 4); hipMemset(o2, 0xff, (size_t)W * H * 4);
    float t1 = timeit([&] { hipLaunchKernelGGL((blur_march_kernel<N, false, 128, 0>), grid, dim3(128), (size_t)G1::LDS_BYTES, 0, (const void *)in, o1, W, H, nb, ta, (const uint32_t *)nullptr); });
    float t2 = timeit([&] { hipLaunchKernelGGL((blur_team_kernel<N, false, S, 0, HW>), grid2, dim3(64 * HW + 128), lds3, 0, (const void *)in, o2, W, H, nb2, ta, (const uint32_t *)nullptr); });
    std::vector<float> a((size_t)W * H), b((size_t)W * H);
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    const bool same = memcmp(a.data(), b.data(), a.size() * 4) == 0;
    size_t bad = 0; for (size_t i = 0; i < a.size() && !same; i++) bad += memcmp(&a[i], &b[i], 4) != 0;
    printf("N %2d S %d HW %d  %dx%d  product: grid %ux%u nb %d %.1f us | teams: grid %ux%u nb %d LDS %zu B %.1f us  %s (%zu differ)\n", N, S, HW, W, H, grid.x, grid.y, nb, t1,
           grid2.x, grid2.y, nb2, lds3, t2, same ? "BITWISE EQUAL" : "MISMATCH", bad);
}
int main(int argc, char **argv) {
    int W = argc > 1 ? atoi(argv[1]) : 4096, H = argc > 2 ? atoi(argv[2]) : 4096;
    if (argc > 3) g_wgs = atoi(argv[3]);
    float *in, *o1, *o2;
    hipMalloc(&in, (size_t)W * H * 4); hipMalloc(&o1, (size_t)W * H * 4); hipMalloc(&o2, (size_t)W * H * 4);
    std::vector<float> h((size_t)W * H);
    uint32_t st = 12345;
    for (size_t i = 0; i < h.size(); i++) { st = st * 1664525u + 1013904223u; h[i] = (float)(st >> 8) * (255.0f / 16777216.0f); }
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float taps[64];
    auto mk = [&](int n) { double sum = 0; for (int i = 0; i < n; i++) { double x = (i - (n - 1) / 2.0) / (n / 8.0); taps[i] = (float)exp(-x * x / 2); sum += taps[i]; }
                           for (int i = 0; i < n; i++) taps[i] = (float)(taps[i] / sum); for (int i = 0; i < n / 2; i++) taps[n - 1 - i] = taps[i]; };
    mk(11); run<11, 2>(in, o1, o2, W, H, taps);
    mk(15); run<15, 2>(in, o1, o2, W, H, taps); run<15, 3>(in, o1, o2, W, H, taps);
    mk(17); run<17, 3>(in, o1, o2, W, H, taps);
    mk(21); run<21, 3>(in, o1, o2, W, H, taps);
    mk(27); run<27, 3>(in, o1, o2, W, H, taps); run<27, 4>(in, o1, o2, W, H, taps);
    mk(15); run<15, 2, 3>(in, o1, o2, W, H, taps); mk(17); run<17, 3, 3>(in, o1, o2, W, H, taps); mk(21); run<21, 3, 3>(in, o1, o2, W, H, taps);
    mk(27); run<27, 4, 3>(in, o1, o2, W, H, taps); run<27, 4, 4>(in, o1, o2, W, H, taps); mk(21); run<21, 3, 4>(in, o1, o2, W, H, taps);
    return 0;
}
