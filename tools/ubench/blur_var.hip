// dev tool: variants of the marching team blur (blur_var_kernel.hpp) against the product kernel (k_pyramid.hpp:
// blur_team_kernel) on one plane: bitwise equality of the output + time per launch, per tap count of a pyramid.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         tools/ubench/blur_var.hip -o tools/ubench/blur_var_bench
//   ./tools/ubench/blur_var_bench [W H] [which]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "blur_var_kernel.hpp"
#include "blur_team_x.hpp"
using namespace siftk;

static std::vector<float> gauss(int n, float sigma) {
    std::vector<float> t(n);
    float s = 0;
    for (int i = 0; i < n; i++) { float x = (i - (n - 1) / 2.0f) / sigma; t[i] = expf(-x * x / 2); s += t[i]; }
    for (int i = 0; i < n; i++) t[i] /= s;
    for (int i = 0; i < n / 2; i++) t[n - 1 - i] = t[i];
    return t;
}
struct Geo { int gx, gy, nblocks, last_subs, rows_out; };
template <int N, int S> Geo geometry(int W, int H, int wgs) {       // siftmi.hip: launch_team
    using G = March2Geom<N, 128, S>;
    using SS = SubSplit<N, S>;
    Geo g;
    g.gx = (W + G::TX - 1) / G::TX;
    g.gy = wgs / g.gx; if (g.gy < 1) g.gy = 1; if (g.gy > H) g.gy = H;
    g.rows_out = (H + g.gy - 1) / g.gy;
    if (g.rows_out < 2 * N + 1) g.rows_out = 2 * N + 1;
    g.gy = (H + g.rows_out - 1) / g.rows_out;
    auto covered = [](int b, int m) { int r = b * N; for (int q = 0; q < m; q++) r += SS::rows(q); return r; };
    const int need = g.rows_out + N - 1;
    const int b = need / N;
    int m = 0;
    while (covered(b, m) < need) m++;
    g.nblocks = b + (m > 0 ? 1 : 0); g.last_subs = m > 0 ? m : S;
    return g;
}
// time of one launch inside a train of `reps` back-to-back launches (as in a pyramid), best of 5 trains; clocks warmed by the caller
template <class F> float timeit(F f, int reps = 10) {
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
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best * 1e3f;
}
static unsigned long long *g_clk;
static int g_ref_prio = 0;      // priority feedback of the product kernel in the reference launches
static std::vector<float> g_ref, g_got;
static bool same_as_ref(const float *dev, size_t n, size_t *bad) {
    hipMemcpy(g_got.data(), dev, n * 4, hipMemcpyDeviceToHost);
    *bad = 0;
    if (memcmp(g_got.data(), g_ref.data(), n * 4) == 0) return true;
    for (size_t i = 0; i < n; i++) *bad += memcmp(&g_got[i], &g_ref[i], 4) != 0;
    return false;
}

template <int N, int S, int D, int VAR> void variant(const char *name, const float *in, float *o2, int W, int H, const TapsArg<N> &ta, int wgs, float t_ref) {
    using G = March2Geom<N, 128, S>;
    const Geo g = geometry<N, S>(W, H, wgs);
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    hipMemset(o2, 0xff, (size_t)W * H * 4);
    auto launch = [&] { hipLaunchKernelGGL((blur_front_kernel<N, false, S, 0, D, VAR>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, g_clk); };
    const float t = timeit(launch);
    size_t bad;
    const bool same = same_as_ref(o2, (size_t)W * H, &bad);
    printf("  N %2d  %-34s wgs %4d (%2dx%3d, %3d rows)  %7.2f us  (%+5.1f %%)  %s", N, name, g.gx * g.gy, g.gx, g.gy, g.rows_out, t, 100.0 * (t - t_ref) / t_ref,
           same ? "bitwise equal" : "MISMATCH");
    if (!same) printf(" (%zu differ)", bad);
    if (VAR & 2) {
        hipMemset(g_clk, 0, 64);
        launch(); hipDeviceSynchronize();
        unsigned long long c[8];
        hipMemcpy(c, g_clk, 64, hipMemcpyDeviceToHost);
        const double fw = 2.0 * g.gx * g.gy, front = (double)(c[0] + c[1] + c[2] + c[3]), back = (double)(c[4] + c[5]);
        printf("\n        clock, cycles per wave: front stage(+load wait) %.0f  prefetch %.0f  hpass %.0f  barrier %.0f (%.0f %%) | back vpass %.0f  barrier %.0f (%.0f %%)",
               c[0] / fw, c[1] / fw, c[2] / fw, c[3] / fw, 100.0 * c[3] / front, c[4] / fw, c[5] / fw, 100.0 * c[5] / back);
    }
    printf("\n");
}

template <int N, int S, int TV> void tapsv(const float *in, float *o2, int W, int H, const TapsArg<N> &ta, int wgs, float t_ref) {
    using G = March2Geom<N, 128, S>;
    const Geo g = geometry<N, S>(W, H, wgs);
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    hipMemset(o2, 0xff, (size_t)W * H * 4);
    auto launch = [&] { hipLaunchKernelGGL((blur_team_x<N, false, S, 0, 2, TV>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, 0, 0, 0, (unsigned long long *)nullptr, 0); };
    const float t = timeit(launch);
    size_t bad;
    const bool same = same_as_ref(o2, (size_t)W * H, &bad);
    printf("  N %2d  product, taps in VGPR pairs: %s%s  wgs %4d  %7.2f us  (%+5.1f %%)  %s\n", N, (TV & 1) ? "H " : "", (TV & 2) ? "V" : "", g.gx * g.gy, t,
           100.0 * (t - t_ref) / t_ref, same ? "bitwise equal" : "MISMATCH");
}

template <int N, int S> void prio(const float *in, float *o2, int W, int H, const TapsArg<N> &ta, int wgs, float t_ref, int mode) {
    using G = March2Geom<N, 128, S>;
    const Geo g = geometry<N, S>(W, H, wgs);
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    hipMemset(o2, 0xff, (size_t)W * H * 4);
    auto launch = [&] { hipLaunchKernelGGL((blur_team_x<N, false, S, 0, 2>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, 0, 0, 0, (unsigned long long *)nullptr, mode); };
    const float t = timeit(launch);
    size_t bad;
    const bool same = same_as_ref(o2, (size_t)W * H, &bad);
    printf("  N %2d  product + priority feedback mode %d   wgs %4d  %7.2f us  (%+5.1f %%)  %s\n", N, mode, g.gx * g.gy, t, 100.0 * (t - t_ref) / t_ref,
           same ? "bitwise equal" : "MISMATCH");
}

template <int N, int S, int LP> void ldspitch(const float *in, float *o2, int W, int H, const TapsArg<N> &ta, int wgs, float t_ref, int prio) {
    using G = March2Geom<N, 128, S>;
    const Geo g = geometry<N, S>(W, H, wgs);
    const size_t lds = (size_t)3 * G::NPS * (G::PITCH + (LP ? 2 : 0)) * 2 * 4;
    hipMemset(o2, 0xff, (size_t)W * H * 4);
    auto launch = [&] { hipLaunchKernelGGL((blur_team_x<N, false, S, 0, 2, 0, 0, LP>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, prio, 0, 0, (unsigned long long *)nullptr, 0); };
    const float t = timeit(launch);
    size_t bad;
    const bool same = same_as_ref(o2, (size_t)W * H, &bad);
    printf("  N %2d  product (prio %d)%s   wgs %4d  %7.2f us  (%+5.1f %%)  %s\n", N, prio, LP ? " + conflict-free H reads (pitch 16 mod 32, two row pairs per wave)" : "", g.gx * g.gy, t,
           100.0 * (t - t_ref) / t_ref, same ? "bitwise equal" : "MISMATCH");
}

template <int N, int S, int PL> void peel(const float *in, float *o2, int W, int H, const TapsArg<N> &ta, int wgs, float t_ref, int prio) {
    using G = March2Geom<N, 128, S>;
    const Geo g = geometry<N, S>(W, H, wgs);
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    hipMemset(o2, 0xff, (size_t)W * H * 4);
    auto launch = [&] { hipLaunchKernelGGL((blur_team_x<N, false, S, 0, 2, 0, 0, 0, PL>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, prio, 0, 0, (unsigned long long *)nullptr, 0); };
    const float t = timeit(launch);
    size_t bad;
    const bool same = same_as_ref(o2, (size_t)W * H, &bad);
    printf("  N %2d  product (prio %d)%s   wgs %4d  %7.2f us  (%+5.1f %%)  %s\n", N, prio, PL ? " + first period peeled" : "", g.gx * g.gy, t,
           100.0 * (t - t_ref) / t_ref, same ? "bitwise equal" : "MISMATCH");
}

template <int N, int S, int X2> void wideload(const float *in, float *o2, int W, int H, const TapsArg<N> &ta, int wgs, float t_ref, int prio) {
    using G = March2Geom<N, 128, S>;
    const Geo g = geometry<N, S>(W, H, wgs);
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    hipMemset(o2, 0xff, (size_t)W * H * 4);
    auto launch = [&] { hipLaunchKernelGGL((blur_team_x<N, false, S, 0, 2, 0, 0, 0, 0, X2>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, prio, 0, 0, (unsigned long long *)nullptr, 0); };
    const float t = timeit(launch);
    size_t bad;
    const bool same = same_as_ref(o2, (size_t)W * H, &bad);
    printf("  N %2d  product (prio %d)%s   wgs %4d  %7.2f us  (%+5.1f %%)  %s\n", N, prio, X2 ? " + 8-byte main loads, 16-byte staging" : "", g.gx * g.gy, t,
           100.0 * (t - t_ref) / t_ref, same ? "bitwise equal" : "MISMATCH");
}

template <int N, int S> void stagger(const float *in, float *o2, int W, int H, const TapsArg<N> &ta, int wgs, float t_ref, int mode, int units) {
    using G = March2Geom<N, 128, S>;
    const Geo g = geometry<N, S>(W, H, wgs);
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    hipMemset(o2, 0xff, (size_t)W * H * 4);
    auto launch = [&] { hipLaunchKernelGGL((blur_team_x<N, false, S, 0, 2>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, 0, mode, units, (unsigned long long *)nullptr, 0); };
    const float t = timeit(launch);
    size_t bad;
    const bool same = same_as_ref(o2, (size_t)W * H, &bad);
    printf("  N %2d  product + stagger mode %d, %d x 1024 cycles   wgs %4d  %7.2f us  (%+5.1f %%)  %s\n", N, mode, units, g.gx * g.gy, t, 100.0 * (t - t_ref) / t_ref,
           same ? "bitwise equal" : "MISMATCH");
}

// per-wave timeline of the product kernel (blur_team_x<..., TR = 1>): the workgroups that share ONE CU, step by step
#include <map>
#include <algorithm>
template <int N, int S> void timeline(const float *in, float *o2, int W, int H, int prio = 0) {
    using G = March2Geom<N, 128, S>;
    constexpr int TRN = 160;
    auto tv = gauss(N, 0.125f * N);
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = tv[i];
    const Geo g = geometry<N, S>(W, H, N >= 27 ? 768 : 1024);
    const size_t lds = (size_t)3 * G::LDS_BYTES + 4 * TRN * 8;
    const int nwg = g.gx * g.gy;
    unsigned long long *dtr;
    hipMalloc(&dtr, (size_t)nwg * 4 * TRN * 8);
    hipMemset(dtr, 0, (size_t)nwg * 4 * TRN * 8);
    auto launch = [&] { hipLaunchKernelGGL((blur_team_x<N, false, S, 0, 2, 0, 1>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o2, W, H, g.nblocks,
                                           g.last_subs, g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, 0, 0, 0, dtr, prio); };
    const float t = timeit(launch);
    std::vector<unsigned long long> tr((size_t)nwg * 4 * TRN);
    hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost);
    hipFree(dtr);
    printf("N %2d  traced launch, prio_mode %d: %.2f us (%d workgroups); shader-clock cycles\n", N, prio, t, nwg);
    // group the workgroups by CU: XCD = linear id % 8 (the dispatcher's round robin), then HW_ID se / sh / cu
    std::map<int, std::vector<int>> by_cu;
    for (int wg = 0; wg < nwg; wg++) {
        const unsigned hw = (unsigned)tr[((size_t)wg * 4) * TRN];
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by_cu[((wg & 7) << 8) | (se << 5) | (sh << 4) | cu].push_back(wg);
    }
    std::map<int, int> hist;
    for (auto &kv : by_cu) hist[(int)kv.second.size()]++;
    printf("  workgroups per CU (by HW_ID): ");
    for (auto &kv : hist) printf("%d CUs with %d;  ", kv.second, kv.first);
    printf("\n");
    // launch-wide, on the chip-wide 100 MHz clock (s_memrealtime): when do the waves start and end, XCD by XCD
    {
        unsigned long long z = ~0ull, zend = 0;
        for (int wg = 0; wg < nwg; wg++) for (int w = 0; w < 4; w++) { const unsigned long long *e = &tr[((size_t)wg * 4 + w) * TRN]; if (e[1] > 3) { z = std::min(z, e[TRN - 2]); zend = std::max(zend, e[TRN - 1]); } }
        printf("  chip: first wave start -> last wave end %.2f us (launch period in the train %.2f us)\n", (zend - z) / 100.0, t);
        {   // shader clock seen by the waves: cycles between a wave's first and last mark over its 100 MHz time from start to end
            double cyc = 0, us = 0;
            for (int wg = 0; wg < nwg; wg += 7) { const unsigned long long *e = &tr[((size_t)wg * 4 + 2) * TRN]; const int n = (int)std::min<unsigned long long>(e[1], TRN);
                if (n > 3) { cyc += (double)(e[n - 1] - e[2]); us += (e[TRN - 1] - e[TRN - 2]) / 100.0; } }
            printf("  shader clock while the launch runs: %.0f MHz (V waves: cycles first -> last mark / their 100 MHz start -> end time)\n", cyc / us);
        }
        for (int xcd = 0; xcd < 8; xcd++) {
            std::vector<double> st, en;
            for (int wg = xcd; wg < nwg; wg += 8) for (int w = 0; w < 4; w++) {
                const unsigned long long *e = &tr[((size_t)wg * 4 + w) * TRN];
                if (e[1] > 3) { st.push_back((e[TRN - 2] - z) / 100.0); en.push_back((e[TRN - 1] - z) / 100.0); }
            }
            std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
            printf("    XCD %d: starts %.2f / %.2f / %.2f us (min / median / max) | ends %.2f / %.2f / %.2f / %.2f / %.2f us (min / 10 %% / median / 90 %% / max)\n", xcd,
                   st[0], st[st.size() / 2], st.back(), en[0], en[en.size() / 10], en[en.size() / 2], en[en.size() * 9 / 10], en.back());
        }
    }
    auto cu_it = by_cu.begin();
    for (auto it = by_cu.begin(); it != by_cu.end(); ++it) if (it->second.size() == (N >= 27 ? 3u : 4u)) { cu_it = it; break; }
    const std::vector<int> &wgs = cu_it->second;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int wg : wgs) for (int w = 0; w < 4; w++) {
        const unsigned long long *e = &tr[((size_t)wg * 4 + w) * TRN];
        const int n = (int)std::min<unsigned long long>(e[1], TRN);
        for (int i = 2; i < n; i++) { t0 = std::min(t0, e[i]); t1 = std::max(t1, e[i]); }
    }
    printf("  one CU (key %#x): workgroups", cu_it->first);
    for (int wg : wgs) printf(" %d", wg);
    printf("; first mark -> last mark %llu cycles\n", t1 - t0);
    for (int wg : wgs) for (int w = 0; w < 4; w++) {
        const unsigned long long *e = &tr[((size_t)wg * 4 + w) * TRN];
        const unsigned hw = (unsigned)e[0];
        const int role = (int)((e[0] >> 32) & 1), n = (int)std::min<unsigned long long>(e[1], TRN);
        printf("  wg %4d wave %d %s simd %d slot %d: ", wg, w, role ? "V" : "H", (hw >> 4) & 3, hw & 15);
        if (!role) {         // marks: [start, end of hpass] per step, then the exit mark
            unsigned long long work = 0, wait = 0;
            printf("start@%llu | per step hpass/barrier:", e[2] - t0);
            for (int i = 2; i + 2 < n; i += 2) { printf(" %llu/%llu", e[i + 1] - e[i], e[i + 2] - e[i + 1]); work += e[i + 1] - e[i]; wait += e[i + 2] - e[i + 1]; }
            printf(" | total hpass %llu barrier %llu end@%llu\n", work, wait, e[n - 1] - t0);
        } else {             // marks: [top, after vpass, after stage (incl. the wait for its loads), after prefetch] per step
            unsigned long long a = 0, b = 0, c = 0, d = 0;
            printf("start@%llu | per step vpass/stage+loadwait/prefetch/barrier:", e[2] - t0);
            int i = 2;
            for (; i + 4 < n; i += 4) {
                printf(" %llu/%llu/%llu/%llu", e[i + 1] - e[i], e[i + 2] - e[i + 1], e[i + 3] - e[i + 2], e[i + 4] - e[i + 3]);
                a += e[i + 1] - e[i]; b += e[i + 2] - e[i + 1]; c += e[i + 3] - e[i + 2]; d += e[i + 4] - e[i + 3];
            }
            printf(" | totals %llu/%llu/%llu/%llu, tail marks %d, end@%llu\n", a, b, c, d, n - i, e[n - 1] - t0);
        }
    }
}

// the product kernel at 1, 2, 3, 4 workgroups per CU (segments of the launch's own height, a plane just tall enough): how much of
// one workgroup's step latency do the others fill?
template <int N, int S> void occupancy(const float *in, float *o1, int W, int Hmax) {
    using G = March2Geom<N, 128, S>;
    auto tv = gauss(N, 0.125f * N);
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = tv[i];
    const int rows = N >= 27 ? 86 : 64;
    float t1 = 0;
    for (int per_cu = 1; per_cu <= 5; per_cu++) {
        const int gy = 16 * per_cu, H = rows * gy;
        if (H > Hmax) break;
        const Geo g = geometry<N, S>(W, H, 16 * gy);
        const size_t lds = (size_t)3 * G::LDS_BYTES;
        auto ref = [&] { hipLaunchKernelGGL((blur_team_kernel<N, false, S, 0, 2>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o1, W, H, g.nblocks, g.last_subs,
                                            g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, g_ref_prio); };
        const float t = timeit(ref);
        if (per_cu == 1) t1 = t;
        printf("  N %2d  %d workgroup(s) per CU: grid %2dx%3d, %3d rows each, plane %dx%d  %7.2f us  = %.2f x the time of one per CU, %.0f GB/s algorithmic\n", N, per_cu, g.gx, g.gy,
               g.rows_out, W, H, t, t / t1, 8.0 * W * H / t / 1e3);
    }
}

template <int N, int S> void run(const float *in, float *o1, float *o2, int W, int H, int which) {
    using G = March2Geom<N, 128, S>;
    auto tv = gauss(N, 0.125f * N);
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = tv[i];
    const int wgs0 = N >= 27 ? 768 : 1024;
    const Geo g = geometry<N, S>(W, H, wgs0);
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    hipMemset(o1, 0, (size_t)W * H * 4);
    auto ref = [&] { hipLaunchKernelGGL((blur_team_kernel<N, false, S, 0, 2>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o1, W, H, g.nblocks, g.last_subs,
                                        g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, g_ref_prio); };
    const float t_ref = timeit(ref);
    hipMemcpy(g_ref.data(), o1, (size_t)W * H * 4, hipMemcpyDeviceToHost);
    printf("N %2d  product blur_team_kernel<S=%d>          wgs %4d (%2dx%3d, %3d rows)  %7.2f us   %.0f GB/s algorithmic\n", N, S, g.gx * g.gy, g.gx, g.gy, g.rows_out, t_ref,
           8.0 * W * H / t_ref / 1e3);
    if (which & 1) {
        variant<N, S, 1, 0>("front loads, D=1", in, o2, W, H, ta, wgs0, t_ref);
        variant<N, S, 2, 0>("front loads, D=2", in, o2, W, H, ta, wgs0, t_ref);
        variant<N, S, 3, 0>("front loads, D=3", in, o2, W, H, ta, wgs0, t_ref);
    }
    if (which & 2) {
        variant<N, S, 2, 4>("front loads, D=2, feed late", in, o2, W, H, ta, wgs0, t_ref);
        variant<N, S, 2, 1>("front loads, D=2, peel", in, o2, W, H, ta, wgs0, t_ref);
    }
    if (which & 4) {
        variant<N, S, 2, 2>("front loads, D=2, clock", in, o2, W, H, ta, wgs0, t_ref);
    }
    if (which & 4096) { for (int rep = 0; rep < 3; rep++) { wideload<N, S, 0>(in, o2, W, H, ta, wgs0, t_ref, 1); wideload<N, S, 1>(in, o2, W, H, ta, wgs0, t_ref, 1); } }
    if (which & 1024) { for (int rep = 0; rep < 2; rep++) { peel<N, S, 0>(in, o2, W, H, ta, wgs0, t_ref, 1); peel<N, S, 1>(in, o2, W, H, ta, wgs0, t_ref, 1); } }
    if (which & 512) { for (int rep = 0; rep < 2; rep++) { ldspitch<N, S, 0>(in, o2, W, H, ta, wgs0, t_ref, 1); ldspitch<N, S, 1>(in, o2, W, H, ta, wgs0, t_ref, 1); } }
    if (which & 256) { for (int mode : {0, 1, 3, 4, 9, 11, 1, 0}) prio<N, S>(in, o2, W, H, ta, wgs0, t_ref, mode); }
    if (which & 128) { timeline<N, S>(in, o2, W, H); if (which & 256) timeline<N, S>(in, o2, W, H, 1); }
    if (which & 64) occupancy<N, S>(in, o1, W, H);
    if (which & 32) {
        tapsv<N, S, 0>(in, o2, W, H, ta, wgs0, t_ref);
        tapsv<N, S, 1>(in, o2, W, H, ta, wgs0, t_ref);
        tapsv<N, S, 2>(in, o2, W, H, ta, wgs0, t_ref);
        tapsv<N, S, 3>(in, o2, W, H, ta, wgs0, t_ref);
    }
    if (which & 16) {
        stagger<N, S>(in, o2, W, H, ta, wgs0, t_ref, 0, 0);
        for (int mode : {1, 2}) for (int units : {1, 2, 3, 5, 8}) stagger<N, S>(in, o2, W, H, ta, wgs0, t_ref, mode, units);
    }
    if (which & 8) {
        for (int wgs : {512, 640, 768, 896, 1024, 1280}) variant<N, S, 2, 0>("front loads, D=2", in, o2, W, H, ta, wgs, t_ref);
    }
}

// the product kernel with other sub-block counts S (rows per step), priority feedback on: has the best S moved?
template <int N, int S> void subsplit(const float *in, float *o1, int W, int H) {
    using G = March2Geom<N, 128, S>;
    using SS = SubSplit<N, S>;
    auto tv = gauss(N, 0.125f * N);
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = tv[i];
    const size_t lds = (size_t)3 * G::LDS_BYTES;
    for (int wgs : {640, 768, 896, 1024}) {
        const Geo g = geometry<N, S>(W, H, wgs);
        if (lds > 64 * 1024) { printf("  N %2d S %d: %zu bytes of LDS per workgroup: skipped\n", N, S, lds); return; }
        for (int prio = 1; prio < 2; prio++) {
            auto ref = [&] { hipLaunchKernelGGL((blur_team_kernel<N, false, S, 0, 2>), dim3(g.gx, g.gy), dim3(256), lds, 0, (const void *)in, o1, W, H, g.nblocks, g.last_subs,
                                                g.rows_out, ta, (const uint32_t *)nullptr, (float *)nullptr, 1, prio); };
            const float t = timeit(ref);
            printf("  N %2d S %d (%d rows per step, %d row pairs, LDS %5zu B) wgs %4d prio %d  %7.2f us\n", N, S, SS::RB, G::NPS, lds, g.gx * g.gy, prio, t);
        }
    }
}

int main(int argc, char **argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 4096, H = argc > 2 ? atoi(argv[2]) : 4096;
    const int which = argc > 3 ? atoi(argv[3]) : 7;
    float *in, *o1, *o2;
    hipMalloc(&in, (size_t)W * H * 4); hipMalloc(&o1, (size_t)W * H * 4); hipMalloc(&o2, (size_t)W * H * 4);
    hipMalloc(&g_clk, 64);
    std::vector<float> h((size_t)W * H);
    g_ref.resize(h.size()); g_got.resize(h.size());
    uint32_t st = 12345;
    for (size_t i = 0; i < h.size(); i++) { st = st * 1664525u + 1013904223u; h[i] = (float)(st >> 8) * (255.0f / 16777216.0f); }
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("plane %d x %d, xcd_map 1; times: one launch inside a train of 10, best of 5 trains\n", W, H);
    if (which & 2048) {
        for (int pass = 0; pass < 3; pass++) {
            subsplit<11, 1>(in, o1, W, H); subsplit<11, 2>(in, o1, W, H);
            subsplit<15, 2>(in, o1, W, H); subsplit<15, 3>(in, o1, W, H);
            subsplit<17, 2>(in, o1, W, H); subsplit<17, 3>(in, o1, W, H);
            subsplit<21, 2>(in, o1, W, H); subsplit<21, 3>(in, o1, W, H); subsplit<21, 4>(in, o1, W, H);
            subsplit<27, 2>(in, o1, W, H); subsplit<27, 4>(in, o1, W, H);
            printf("---- second pass\n");
        }
        return 0;
    }
    for (int pass = 0; pass < 2; pass++) {            // the first pass warms the clocks (and is printed: compare)
        run<11, 2>(in, o1, o2, W, H, which);
        run<15, 2>(in, o1, o2, W, H, which);
        run<17, 3>(in, o1, o2, W, H, which);
        run<21, 3>(in, o1, o2, W, H, which);
        run<27, 4>(in, o1, o2, W, H, which);
        printf("---- second pass\n");
    }
    return 0;
}
