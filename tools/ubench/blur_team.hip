// dev tool / EXPERIMENT: team-specialised marching blur (2 H waves + 2 V waves per workgroup, triple-buffered sub-blocks)
// against the product blur_march_kernel: bitwise equality + time on a 4096^2 plane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"
namespace siftk {
template <int N, int S> struct SubSplit {
    static constexpr int RB = (((N + S - 1) / S) + 1) & ~1;
    static constexpr int rows(int s) { return (N - s * RB) < RB ? (N - s * RB) : RB; }
    static constexpr int pairs(int s) { return (rows(s) + 1) / 2; }
    static constexpr int NPS = RB / 2;
    static_assert(N - (S - 1) * RB > 0, "empty last sub-block");
};

template <int N, int NT, int S> struct March2Geom {
    using SS = SubSplit<N, S>;
    static constexpr int TX = 2 * NT;
    static constexpr int C = (N & 1) ? N / 2 : N / 2 - 1;
    static constexpr int NPS = SS::NPS;
    static constexpr int COLS = TX + N - 1;
    static constexpr int PITCH = (COLS + 3) & ~3;
    static constexpr int NW = (N + 3 + 1) & ~1;
    static constexpr int LDS_BYTES = NPS * PITCH * 2 * 4;
    static constexpr int HALO = N - 1;
    static constexpr int NB = (NPS * HALO + NT - 1) / NT;
};


// One workgroup = 256 threads = an H team (waves 0-1) and a V team (waves 2-3) on one 256-column strip.
// Step g = (block, sub-block): the H team filters sub-block g horizontally in LDS buffer g % 3 while the V team
// marches sub-block g-1 vertically out of buffer (g-1) % 3 (accumulators, global stores) and then stages sub-block
// g+1 (prefetched registers -> buffer (g+1) % 3) and issues the loads of g+2.  One barrier per step.
template <int N, bool NORM, int S, int DT = 0>
__global__ __launch_bounds__(256) void blur_march3_kernel(const void *__restrict__ in, float *__restrict__ out,
                                                          int W, int H, int nblocks, TapsArg<N> taps,
                                                          const uint32_t *__restrict__ mm) {
    constexpr int NT = 128;                       // threads per team
    using G = March2Geom<N, NT, S>;
    using SS = SubSplit<N, S>;
    static_assert(N & 1, "marching blur needs an odd tap count");
    constexpr int BUF = G::NPS * G::PITCH * 2;    // floats per LDS buffer
    extern __shared__ float4 smem4[];
    float *sbase = reinterpret_cast<float *>(smem4);
    const int role = threadIdx.x >> 7;            // 0: H team, 1: V team (wave-uniform)
    const int tid = threadIdx.x & 127;
    const int x0 = blockIdx.x * G::TX;
    const int rows_out = nblocks * N - (N - 1);
    const int ys = blockIdx.y * rows_out;
    const int yend = min(ys + rows_out, H);
    float mn = 0.f, range = 1.f;
    if (NORM) { mn = ord2f(mm[0]); range = ord2f(mm[1]) - mn; }

    // ---- V-team state: staging duty, accumulators, look-ahead registers
    const int gx_a = reflect_index(x0 - G::C + tid, W);
    const int gx_b = reflect_index(x0 - G::C + NT + tid, W);
    int hb_rp[G::NB], hb_col[G::NB], hb_gx[G::NB];
#pragma unroll
    for (int u = 0; u < G::NB; u++) {
        const int e = tid + NT * u;
        hb_rp[u] = (e < G::NPS * G::HALO) ? e / G::HALO : 1 << 20;
        hb_col[u] = G::TX + e % G::HALO;
        hb_gx[u] = reflect_index(x0 - G::C + hb_col[u], W);
    }
    auto ld = [&](unsigned byte_off) {
        if constexpr (DT == 0) return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(in) + byte_off);
        else return load_px<DT>(in, (size_t)(byte_off >> 2));
    };
    const unsigned W4 = (unsigned)W * 4u;
    auto norm2 = [&](f32x2 v) {
        if (NORM) { v.x = 255.0f * (v.x - mn) / range; v.y = 255.0f * (v.y - mn) / range; }
        return v;
    };
    f32x2 acc[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = (f32x2){0.f, 0.f};
    const int gxo = x0 + 2 * tid;
    const bool vec_store = ((W & 1) == 0) && (gxo + 1 < W);
    f32x2 pa[G::NPS], pb[G::NPS], ph[G::NB];

    // both teams stage: the H team row pairs [0, np/2), the V team [np/2, np) and the halo columns
    auto prefetch = [&](int blk, int sub, int np) {
        const int v0 = ys - G::C + blk * N + sub * SS::RB;
        const int lo = role == 0 ? 0 : np / 2, hi = role == 0 ? np / 2 : np;
        if (v0 >= 0 && v0 + 2 * np <= H) {
            unsigned oa = ((unsigned)(v0 + 2 * lo) * (unsigned)W + (unsigned)gx_a) * 4u;
            unsigned ob = ((unsigned)(v0 + 2 * lo) * (unsigned)W + (unsigned)gx_b) * 4u;
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp >= lo && rp < hi) {
                    pa[rp].x = ld(oa); pa[rp].y = ld(oa + W4);
                    pb[rp].x = ld(ob); pb[rp].y = ld(ob + W4);
                    oa += 2u * W4; ob += 2u * W4;
                }
        } else {
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp >= lo && rp < hi) {
                    const unsigned r0 = (unsigned)reflect_index(v0 + 2 * rp, H) * W4, r1 = (unsigned)reflect_index(v0 + 2 * rp + 1, H) * W4;
                    pa[rp].x = ld(r0 + 4u * gx_a); pa[rp].y = ld(r1 + 4u * gx_a);
                    pb[rp].x = ld(r0 + 4u * gx_b); pb[rp].y = ld(r1 + 4u * gx_b);
                }
        }
#pragma unroll
        for (int u = 0; u < G::NB; u++) {
            ph[u] = (f32x2){0.f, 0.f};
            if (role == 1 && hb_rp[u] < np) {
                ph[u].x = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u], H) * W4 + 4u * hb_gx[u]);
                ph[u].y = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u] + 1, H) * W4 + 4u * hb_gx[u]);
            }
        }
    };
    auto stage = [&](float *s, int np) {
        const int lo = role == 0 ? 0 : np / 2, hi = role == 0 ? np / 2 : np;
#pragma unroll
        for (int rp = 0; rp < G::NPS; rp++)
            if (rp >= lo && rp < hi) {
                *reinterpret_cast<f32x2 *>(s + (rp * G::PITCH + tid) * 2) = norm2(pa[rp]);
                *reinterpret_cast<f32x2 *>(s + (rp * G::PITCH + NT + tid) * 2) = norm2(pb[rp]);
            }
#pragma unroll
        for (int u = 0; u < G::NB; u++)
            if (role == 1 && hb_rp[u] < np) *reinterpret_cast<f32x2 *>(s + (hb_rp[u] * G::PITCH + hb_col[u]) * 2) = norm2(ph[u]);
    };
    // horizontal pass of one sub-block (np row pairs) in place, by the 128 threads of the H team
    auto hpass = [&](float *s, int np) {
        for (int task = tid; task < np * (NT / 2); task += NT) {
            const int rp = task / (NT / 2), t4 = task % (NT / 2);
            float *rowp = s + (rp * G::PITCH + 4 * t4) * 2;
            f32x2 w[G::NW];
            constexpr int PRE = 4;
#pragma unroll
            for (int k = 0; k < PRE && k < G::NW / 2; k++) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(rowp + 4 * k);
                w[2 * k] = v.xy; w[2 * k + 1] = v.zw;
            }
            f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < N; q++) {
                if ((q & 1) == 0) {
                    const int k = q / 2 + PRE;
                    if (k < G::NW / 2) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(rowp + 4 * k);
                        w[2 * k] = v.xy; w[2 * k + 1] = v.zw;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float tp = taps.t[N - 1 - q];
                const f32x2 tp2 = {tp, tp};
                a0 = a0 + w[q] * tp2;
                a1 = a1 + w[q + 1] * tp2;
                a2 = a2 + w[q + 2] * tp2;
                a3 = a3 + w[q + 3] * tp2;
            }
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<f32x4 *>(rowp) = (f32x4){a0.x, a1.x, a0.y, a1.y};
            *reinterpret_cast<f32x4 *>(rowp + 4) = (f32x4){a2.x, a3.x, a2.y, a3.y};
        }
    };

    // vertical march over the rows of sub-block `sub` (compile-time after unrolling) of block blk
#define VPASS(sbuf, blk_, sub_)                                                                              \
    {                                                                                                        \
        const int np_ = SS::pairs(sub_), nrows_ = SS::rows(sub_);                                            \
        const int ybase_ = ys + (blk_) * N - (N - 1);                                                        \
        float *optr = out + ((ptrdiff_t)(ybase_ + (sub_) * SS::RB) * W + gxo);                              \
        f32x4 hv_next = *reinterpret_cast<const f32x4 *>((sbuf) + (2 * tid) * 2);                           \
        _Pragma("unroll") for (int rp = 0; rp < G::NPS; rp++) {                                              \
            if (rp < np_) {                                                                                  \
                const f32x4 hv = hv_next;                                                                    \
                if (rp + 1 < np_) hv_next = *reinterpret_cast<const f32x4 *>((sbuf) + ((rp + 1) * G::PITCH + 2 * tid) * 2); \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                _Pragma("unroll") for (int half = 0; half < 2; half++) {                                     \
                    if (2 * rp + half < nrows_) {                                                            \
                        const int kk = (sub_) * SS::RB + 2 * rp + half;                                      \
                        const f32x2 h = half ? hv.zw : hv.xy;                                                \
                        _Pragma("unroll") for (int k = 0; k < (N + 1) / 2; k++) {                            \
                            const f32x2 t2 = {taps.t[k], taps.t[k]};                                         \
                            const f32x2 prod = h * t2;                                                       \
                            const int slot_a = (kk - k + N) % N, slot_b = (kk - (N - 1 - k) + N) % N;        \
                            if (k == 0) acc[slot_a] = (f32x2){0.f, 0.f} + prod;                              \
                            else acc[slot_a] = acc[slot_a] + prod;                                           \
                            asm volatile("" : "+v"(acc[slot_a]));                                            \
                            if (k != N - 1 - k) { acc[slot_b] = acc[slot_b] + prod; asm volatile("" : "+v"(acc[slot_b])); } \
                        }                                                                                    \
                        const int done = (kk + 1) % N;                                                       \
                        const int y = ybase_ + kk;                                                           \
                        if (y >= ys && y < yend) {                                                           \
                            if (vec_store) *reinterpret_cast<f32x2 *>(optr) = acc[done];                     \
                            else { if (gxo < W) optr[0] = acc[done].x; if (gxo + 1 < W) optr[1] = acc[done].y; } \
                        }                                                                                    \
                        optr += W;                                                                           \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }

    // ---- prologue: the V team stages step 0 and looks ahead to step 1
    {
        prefetch(0, 0, SS::pairs(0));
        stage(sbase, SS::pairs(0));
        if (S > 1) prefetch(0, 1 % S, SS::pairs(1 % S));
        else if (nblocks > 1) prefetch(1, 0, SS::pairs(0));
    }
    __syncthreads();
    int g = 0;                                     // step counter: buffer of step g is g % 3
    for (int blk = 0; blk < nblocks; blk++) {
#pragma unroll
        for (int sub = 0; sub < S; sub++, g++) {
            float *cur = sbase + (g % 3) * BUF;
            if (role == 0) {
                hpass(cur, SS::pairs(sub));
            } else if (g > 0) {
                // vertical march of the previous step
                float *prev = sbase + ((g + 2) % 3) * BUF;
                if (sub == 0) { VPASS(prev, blk - 1, S - 1) } else { VPASS(prev, blk, (sub + S - 1) % S) }
            }
            // both teams: stage their half of step g+1 (already in registers) and look ahead to g+2
            const bool has1 = (sub + 1 < S) || (blk + 1 < nblocks);
            if (has1) {
                float *nxt = sbase + ((g + 1) % 3) * BUF;
                stage(nxt, SS::pairs((sub + 1) % S));
                const int sub2 = (sub + 2) % S;
                const int blk2 = blk + (sub + 2) / S;
                if (blk2 < nblocks) prefetch(blk2, sub2, SS::pairs(sub2));
            }
            __syncthreads();
        }
    }
    // ---- epilogue: vertical march of the last step
    if (role == 1) {
        float *prev = sbase + ((g + 2) % 3) * BUF;
        VPASS(prev, nblocks - 1, S - 1)
    }
#undef VPASS
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
template <int N, int S> void run(const float *in, float *o1, float *o2, int W, int H, const float *taps) {
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
    float t2 = timeit([&] { hipLaunchKernelGGL((blur_march3_kernel<N, false, S, 0>), grid2, dim3(256), lds3, 0, (const void *)in, o2, W, H, nb2, ta, (const uint32_t *)nullptr); });
    std::vector<float> a((size_t)W * H), b((size_t)W * H);
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    const bool same = memcmp(a.data(), b.data(), a.size() * 4) == 0;
    size_t bad = 0; for (size_t i = 0; i < a.size() && !same; i++) bad += memcmp(&a[i], &b[i], 4) != 0;
    printf("N %2d S %d  %dx%d  product: grid %ux%u nb %d %.1f us | teams: grid %ux%u nb %d LDS %zu B %.1f us  %s (%zu differ)\n", N, S, W, H, grid.x, grid.y, nb, t1,
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
    return 0;
}
