// dev tool / EXPERIMENT: unpacked one-column-per-thread marching blur (256 threads per 256-column strip) against the
// product blur_march_kernel (128 threads, packed f32x2): bitwise equality + time on a 4096^2 plane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"
namespace siftk {
template <int N> struct M1Geom {
    static constexpr int NT = 256, TX = 256;
    static constexpr int C = (N & 1) ? N / 2 : N / 2 - 1;
    static constexpr int COLS = TX + N - 1;
    static constexpr int PITCH = (COLS + 3) & ~3;
    static constexpr int NW = (N + 3 + 3) & ~3;
    static constexpr int LDS_BYTES = N * PITCH * 4;
    static constexpr int HALO = N - 1;
    static constexpr int NB = (N * HALO + NT - 1) / NT;
};
template <int N, bool NORM>
__global__ __launch_bounds__(256) void blur_march1_kernel(const float *__restrict__ in, float *__restrict__ out, int W, int H, int nblocks,
                                                          TapsArg<N> taps, const uint32_t *__restrict__ mm) {
    using G = M1Geom<N>;
    extern __shared__ float4 smem4[];
    float *s = reinterpret_cast<float *>(smem4);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * G::TX;
    const int rows_out = nblocks * N - (N - 1);
    const int ys = blockIdx.y * rows_out;
    const int yend = min(ys + rows_out, H);
    float mn = 0.f, range = 1.f;
    if (NORM) { mn = ord2f(mm[0]); range = ord2f(mm[1]) - mn; }
    const int gx_a = reflect_index(x0 - G::C + tid, W);
    int hb_r[G::NB], hb_col[G::NB], hb_gx[G::NB];
#pragma unroll
    for (int u = 0; u < G::NB; u++) {
        const int e = tid + G::NT * u;
        hb_r[u] = (e < N * G::HALO) ? e / G::HALO : -1;
        hb_col[u] = G::TX + e % G::HALO;
        hb_gx[u] = reflect_index(x0 - G::C + hb_col[u], W);
    }
    auto ld = [&](unsigned byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(in) + byte_off); };
    const unsigned W4 = (unsigned)W * 4u;
    auto norm1 = [&](float v) { if (NORM) v = 255.0f * (v - mn) / range; return v; };
    float acc[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = 0.f;
    const int gxo = x0 + tid;
    float pa[N], ph[G::NB];
    auto prefetch = [&](int blk) {
        const int v0 = ys - G::C + blk * N;
        if (v0 >= 0 && v0 + N <= H) {
            unsigned oa = ((unsigned)v0 * (unsigned)W + (unsigned)gx_a) * 4u;
#pragma unroll
            for (int r = 0; r < N; r++) { pa[r] = ld(oa); oa += W4; }
        } else {
#pragma unroll
            for (int r = 0; r < N; r++) pa[r] = ld((unsigned)reflect_index(v0 + r, H) * W4 + 4u * gx_a);
        }
#pragma unroll
        for (int u = 0; u < G::NB; u++) {
            ph[u] = 0.f;
            if (hb_r[u] >= 0) ph[u] = ld((unsigned)reflect_index(v0 + hb_r[u], H) * W4 + 4u * hb_gx[u]);
        }
    };
    prefetch(0);
    for (int blk = 0; blk < nblocks; blk++) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < N; r++) s[r * G::PITCH + tid] = norm1(pa[r]);
#pragma unroll
        for (int u = 0; u < G::NB; u++)
            if (hb_r[u] >= 0) s[hb_r[u] * G::PITCH + hb_col[u]] = norm1(ph[u]);
        __syncthreads();
        if (blk + 1 < nblocks) prefetch(blk + 1);
        // H pass: task = 4 consecutive columns of one row; the 64 tasks of a row are one wave
        for (int task = tid; task < N * 64; task += 256) {
            const int r = task >> 6, t4 = task & 63;
            float *rowp = s + r * G::PITCH + 4 * t4;
            float w[G::NW];
#pragma unroll
            for (int k = 0; k < G::NW / 4; k++) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(rowp + 4 * k);
                w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
            }
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int q = 0; q < N; q++) {
                const float tp = taps.t[N - 1 - q];
                a0 = a0 + w[q] * tp; a1 = a1 + w[q + 1] * tp; a2 = a2 + w[q + 2] * tp; a3 = a3 + w[q + 3] * tp;
            }
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<f32x4 *>(rowp) = (f32x4){a0, a1, a2, a3};
        }
        __syncthreads();
        const int ybase = ys + blk * N - (N - 1);
        float *optr = out + ((ptrdiff_t)ybase * W + gxo);
#pragma unroll
        for (int kk = 0; kk < N; kk++) {
            const float h = s[kk * G::PITCH + tid];
#pragma unroll
            for (int k = 0; k < (N + 1) / 2; k++) {
                const float prod = h * taps.t[k];
                const int slot_a = (kk - k + N) % N, slot_b = (kk - (N - 1 - k) + N) % N;
                if (k == 0) acc[slot_a] = 0.f + prod; else acc[slot_a] = acc[slot_a] + prod;
                asm volatile("" : "+v"(acc[slot_a]));
                if (k != N - 1 - k) { acc[slot_b] = acc[slot_b] + prod; asm volatile("" : "+v"(acc[slot_b])); }
            }
            const int done = (kk + 1) % N;
            const int y = ybase + kk;
            if (y >= ys && y < yend && gxo < W) optr[0] = acc[done];
            optr += W;
        }
    }
}
}  // namespace siftk
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
template <int N> void run(const float *in, float *o1, float *o2, int W, int H, const float *taps) {
    TapsArg<N> ta;
    for (int i = 0; i < N; i++) ta.t[i] = taps[i];
    using G1 = MarchGeom<N, 128>; using G2 = M1Geom<N>;
    const int nb = nblocks_for<N>(W, H, G1::TX, 1024);
    const int rows_out = nb * N - (N - 1);
    dim3 grid((unsigned)((W + G1::TX - 1) / G1::TX), (unsigned)((H + rows_out - 1) / rows_out));
    const int nb2 = nblocks_for<N>(W, H, G2::TX, g_wgs);
    const int rows_out2 = nb2 * N - (N - 1);
    dim3 grid2((unsigned)((W + G2::TX - 1) / G2::TX), (unsigned)((H + rows_out2 - 1) / rows_out2));
    hipMemset(o1, 0, (size_t)W * H * 4); hipMemset(o2, 0xff, (size_t)W * H * 4);
    float t1 = timeit([&] { hipLaunchKernelGGL((blur_march_kernel<N, false, 128, 0>), grid, dim3(128), (size_t)G1::LDS_BYTES, 0, (const void *)in, o1, W, H, nb, ta, (const uint32_t *)nullptr); });
    float t2 = timeit([&] { hipLaunchKernelGGL((blur_march1_kernel<N, false>), grid2, dim3(256), (size_t)G2::LDS_BYTES, 0, in, o2, W, H, nb2, ta, (const uint32_t *)nullptr); });
    std::vector<float> a((size_t)W * H), b((size_t)W * H);
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    const bool same = memcmp(a.data(), b.data(), a.size() * 4) == 0;
    printf("N %2d  %dx%d  packed: grid %ux%u nb %d %.1f us | unpacked-256: grid %ux%u nb %d LDS %d B %.1f us  %s\n", N, W, H, grid.x, grid.y, nb, t1,
           grid2.x, grid2.y, nb2, G2::LDS_BYTES, t2, same ? "BITWISE EQUAL" : "MISMATCH");
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
    mk(11); run<11>(in, o1, o2, W, H, taps);
    mk(15); run<15>(in, o1, o2, W, H, taps);
    mk(17); run<17>(in, o1, o2, W, H, taps);
    mk(21); run<21>(in, o1, o2, W, H, taps);
    mk(27); run<27>(in, o1, o2, W, H, taps);
    return 0;
}
