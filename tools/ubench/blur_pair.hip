// dev tool: two scales per march (blur_pair_kernel, k_pair.hpp) against two launches of the product's team kernel:
// bitwise equality of both planes + time.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   -fhip-fp32-correctly-rounded-divide-sqrt tools/ubench/blur_pair.hip -o tools/ubench/blur_pair_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "blur_pair_kernel.hpp"
using namespace siftk;

static std::vector<float> gauss(int n, float sigma) {
    std::vector<float> t(n);
    float s = 0;
    for (int i = 0; i < n; i++) { float x = (i - (n - 1) / 2.0f) / sigma; t[i] = expf(-x * x / 2); s += t[i]; }
    for (int i = 0; i < n; i++) t[i] /= s;
    for (int i = 0; i < n / 2; i++) t[n - 1 - i] = t[i];
    return t;
}
template <int N, int S> void launch_team_ref(const float *in, float *out, int W, int H, const float *taps, int wgs, float *half, hipStream_t st = nullptr) {
    using G = March2Geom<N, 128, S>;
    using SS = SubSplit<N, S>;
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = taps[i];
    const int gx = (W + G::TX - 1) / G::TX;
    int gy = wgs / gx; if (gy < 1) gy = 1; if (gy > H) gy = H;
    int rows_out = (H + gy - 1) / gy;
    if (rows_out < 2 * N + 1) rows_out = 2 * N + 1;
    gy = (H + rows_out - 1) / rows_out;
    auto covered = [](int b, int m) { int r = b * N; for (int q = 0; q < m; q++) r += SS::rows(q); return r; };
    const int need = rows_out + N - 1;
    const int b = need / N;
    int m = 0;
    while (covered(b, m) < need) m++;
    const int nblocks = b + (m > 0 ? 1 : 0), last_subs = m > 0 ? m : S;
    hipLaunchKernelGGL((blur_team_kernel<N, false, S, 0>), dim3(gx, gy), dim3(256), (size_t)3 * G::LDS_BYTES, st, (const void *)in, out, W, H, nblocks, last_subs,
                       rows_out, ta, (const uint32_t *)nullptr, half);
}
template <class F> float timeit(F f, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / reps < best) best = ms / reps;
    }
    return best * 1e3f;
}
static int g_wgs = 512;
template <int NA, int SA, int NB, int SB> void run(const float *in, float *q1, float *r1, float *q2, float *r2, int W, int H, int *fault) {
    using G = PairGeom<NA, SA, NB, SB>;
    auto ta = gauss(NA, 0.17f * NA), tb = gauss(NB, 0.17f * NB);
    TapsArg<NA> A; TapsArg<NB> B;
    for (int i = 0; i < NA; i++) A.t[i] = ta[i];
    for (int i = 0; i < NB; i++) B.t[i] = tb[i];
    const int gx = (W + G::TXR - 1) / G::TXR;
    int gy = g_wgs / gx; if (gy < 1) gy = 1;
    int rows_out = (H + gy - 1) / gy;
    if (rows_out < 2 * NB + 1) rows_out = 2 * NB + 1;
    gy = (H + rows_out - 1) / rows_out;
    hipMemset(q1, 0, (size_t)W * H * 4); hipMemset(r1, 0, (size_t)W * H * 4);
    hipMemset(q2, 0xff, (size_t)W * H * 4); hipMemset(r2, 0xff, (size_t)W * H * 4);
    hipMemset(fault, 0, 4);
    hipFuncSetAttribute((const void *)blur_pair_kernel<NA, SA, NB, SB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    const float t_ref = timeit([&] {
        launch_team_ref<NA, SA>(in, q1, W, H, ta.data(), NA >= 27 ? 768 : 1024, nullptr);
        launch_team_ref<NB, SB>(q1, r1, W, H, tb.data(), NB >= 27 ? 768 : 1024, nullptr);
    });
    const float t_pair = timeit([&] {
        hipLaunchKernelGGL((blur_pair_kernel<NA, SA, NB, SB, false>), dim3(gx, gy), dim3(512), (size_t)G::LDS_BYTES, 0, in, q2, r2, W, H, rows_out, A, B,
                           (const uint32_t *)nullptr, (float *)nullptr, (float *)nullptr, fault);
    });
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> a((size_t)W * H), b((size_t)W * H);
    int hf = 0; hipMemcpy(&hf, fault, 4, hipMemcpyDeviceToHost);
    size_t badq = 0, badr = 0; long fq = -1, fr = -1;
    hipMemcpy(a.data(), q1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), q2, b.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < a.size(); i++) if (memcmp(&a[i], &b[i], 4)) { if (fq < 0) fq = (long)i; badq++; }
    hipMemcpy(a.data(), r1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r2, b.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < a.size(); i++) if (memcmp(&a[i], &b[i], 4)) { if (fr < 0) fr = (long)i; badr++; }
    printf("pair %2d(S%d)+%2d(S%d) %dx%d grid %dx%d rows_out %d LDS %d B ring %d rows: two launches %.1f us | one march %.1f us | Q %s (%zu differ, first y %ld x %ld) R %s (%zu differ, first y %ld x %ld) fault %d err %d\n",
           NA, SA, NB, SB, W, H, gx, gy, rows_out, G::LDS_BYTES, G::RING_ROWS, t_ref, t_pair, badq ? "MISMATCH" : "EQUAL", badq, fq / W, fq % W,
           badr ? "MISMATCH" : "EQUAL", badr, fr / W, fr % W, hf, (int)e);
}
int main(int argc, char **argv) {
    int W = argc > 1 ? atoi(argv[1]) : 4096, H = argc > 2 ? atoi(argv[2]) : 4096;
    if (argc > 3) g_wgs = atoi(argv[3]);
    const int which = argc > 4 ? atoi(argv[4]) : 0;
    const size_t n = (size_t)W * H;
    std::vector<float> h(n);
    srand(1);
    for (size_t i = 0; i < n; i++) h[i] = (float)(rand() & 0xffff) / 257.0f;
    float *in, *q1, *r1, *q2, *r2; int *fault;
    hipMalloc(&in, n * 4); hipMalloc(&q1, n * 4); hipMalloc(&r1, n * 4); hipMalloc(&q2, n * 4); hipMalloc(&r2, n * 4); hipMalloc(&fault, 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    if (which == 0 || which == 1) run<11, 2, 15, 2>(in, q1, r1, q2, r2, W, H, fault);
    if (which == 0 || which == 2) run<15, 2, 17, 3>(in, q1, r1, q2, r2, W, H, fault);
    if (which == 0 || which == 3) run<17, 3, 21, 3>(in, q1, r1, q2, r2, W, H, fault);
    if (which == 0 || which == 4) run<21, 3, 27, 4>(in, q1, r1, q2, r2, W, H, fault);
    if (which == 0 || which == 5) run<15, 2, 11, 2>(in, q1, r1, q2, r2, W, H, fault);
    return 0;
}
