// dev tool: blur_march2_kernel (sub-block form) against blur_march_kernel on a 4096^2 plane: bitwise equality + time
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"
namespace siftk {
// EXPERIMENT (not part of the product): result on MI355X, 4096^2 -- bitwise equal to blur_march_kernel, LDS per
// workgroup / 3.5, VGPRs 196 -> 132 at 27 taps, and NO speed-up (72.5 -> 70.3 us): the launch is grid-limited
// (800 workgroups of 2 waves = 1.5 waves per SIMD), not resource-limited, and more workgroups mean more warm-up rows.
// ------------------------------------------------------------------------------------------
// Marching blur, sub-block form.  Same arithmetic and the same rotating-accumulator vertical march as
// blur_march_kernel, but the N rows of one accumulator period are staged and filtered in S sub-blocks of RB rows
// (RB even, the last one shorter).  LDS per workgroup and the register look-ahead shrink by ~S, which is what lets
// 4-5 waves share a SIMD instead of 2.5: with the one-block form the load/stage/H/V phases of the few resident
// waves line up and the memory skeleton (32 us at 27 taps, 4096^2) and the arithmetic (41 us) add instead of
// overlapping (tools/ubench/blur_abl.hip).
template <int N, int S> struct XSubSplit {
    static constexpr int RB = (((N + S - 1) / S) + 1) & ~1;
    static constexpr int rows(int s) { return (N - s * RB) < RB ? (N - s * RB) : RB; }
    static constexpr int pairs(int s) { return (rows(s) + 1) / 2; }
    static constexpr int NPS = RB / 2;
    static_assert(N - (S - 1) * RB > 0, "empty last sub-block");
};

template <int N, int NT, int S> struct XMarch2Geom {
    using SS = XSubSplit<N, S>;
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

template <int N, bool NORM, int NT, int S, int DT = 0>
__global__ __launch_bounds__(NT) void blur_march2_kernel(const void *__restrict__ in, float *__restrict__ out,
                                                         int W, int H, int nblocks, TapsArg<N> taps,
                                                         const uint32_t *__restrict__ mm) {
    using G = XMarch2Geom<N, NT, S>;
    using SS = XSubSplit<N, S>;
    static_assert(N & 1, "marching blur needs an odd tap count");
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
        if (NORM) { v.x = 255.0f * (v.x - mn) / range; v.y = 255.0f * (v.y - mn) / range; }   // preprocess.cl:250
        return v;
    };

    f32x2 acc[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = (f32x2){0.f, 0.f};
    const int gxo = x0 + 2 * tid;
    const bool vec_store = ((W & 1) == 0) && (gxo + 1 < W);

    f32x2 pa[G::NPS], pb[G::NPS], ph[G::NB];
    // look-ahead for sub-block `sub` (np row pairs) of block blk
    auto prefetch = [&](int blk, int sub, int np) {
        const int v0 = ys - G::C + blk * N + sub * SS::RB;
        if (v0 >= 0 && v0 + 2 * np <= H) {
            unsigned oa = ((unsigned)v0 * (unsigned)W + (unsigned)gx_a) * 4u;
            unsigned ob = ((unsigned)v0 * (unsigned)W + (unsigned)gx_b) * 4u;
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp < np) {
                    pa[rp].x = ld(oa); pa[rp].y = ld(oa + W4);
                    pb[rp].x = ld(ob); pb[rp].y = ld(ob + W4);
                    oa += 2u * W4; ob += 2u * W4;
                }
        } else {
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp < np) {
                    const unsigned r0 = (unsigned)reflect_index(v0 + 2 * rp, H) * W4, r1 = (unsigned)reflect_index(v0 + 2 * rp + 1, H) * W4;
                    pa[rp].x = ld(r0 + 4u * gx_a); pa[rp].y = ld(r1 + 4u * gx_a);
                    pb[rp].x = ld(r0 + 4u * gx_b); pb[rp].y = ld(r1 + 4u * gx_b);
                }
        }
#pragma unroll
        for (int u = 0; u < G::NB; u++) {
            ph[u] = (f32x2){0.f, 0.f};
            if (hb_rp[u] < np) {
                ph[u].x = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u], H) * W4 + 4u * hb_gx[u]);
                ph[u].y = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u] + 1, H) * W4 + 4u * hb_gx[u]);
            }
        }
    };
    prefetch(0, 0, SS::pairs(0));

    for (int blk = 0; blk < nblocks; blk++) {
        const int ybase = ys + blk * N - (N - 1);     // output row completed by period step kk is ybase + kk
#pragma unroll
        for (int sub = 0; sub < S; sub++) {
            constexpr int dummy = 0; (void)dummy;
            const int np = SS::pairs(sub), nrows = SS::rows(sub);
            __syncthreads();                          // previous sub-block's vertical reads are done
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp < np) {
                    *reinterpret_cast<f32x2 *>(s + (rp * G::PITCH + tid) * 2) = norm2(pa[rp]);
                    *reinterpret_cast<f32x2 *>(s + (rp * G::PITCH + NT + tid) * 2) = norm2(pb[rp]);
                }
#pragma unroll
            for (int u = 0; u < G::NB; u++)
                if (hb_rp[u] < np) *reinterpret_cast<f32x2 *>(s + (hb_rp[u] * G::PITCH + hb_col[u]) * 2) = norm2(ph[u]);
            __syncthreads();
            if (sub + 1 < S) prefetch(blk, sub + 1, SS::pairs((sub + 1) % S));
            else if (blk + 1 < nblocks) prefetch(blk + 1, 0, SS::pairs(0));
            // ---- horizontal pass in place (see blur_march_kernel)
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
            __syncthreads();
            // ---- vertical march over the rows of this sub-block (static accumulator slots)
            float *optr = out + ((ptrdiff_t)(ybase + sub * SS::RB) * W + gxo);
            f32x4 hv_next = *reinterpret_cast<const f32x4 *>(s + (2 * tid) * 2);
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++) {
                if (rp < np) {
                    const f32x4 hv = hv_next;
                    if (rp + 1 < np) hv_next = *reinterpret_cast<const f32x4 *>(s + ((rp + 1) * G::PITCH + 2 * tid) * 2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        if (2 * rp + half < nrows) {
                            const int kk = sub * SS::RB + 2 * rp + half;
                            const f32x2 h = half ? hv.zw : hv.xy;
#pragma unroll
                            for (int k = 0; k < (N + 1) / 2; k++) {
                                const f32x2 t2 = {taps.t[k], taps.t[k]};
                                const f32x2 prod = h * t2;
                                const int slot_a = (kk - k + N) % N, slot_b = (kk - (N - 1 - k) + N) % N;
                                if (k == 0) acc[slot_a] = (f32x2){0.f, 0.f} + prod;
                                else acc[slot_a] = acc[slot_a] + prod;
                                asm volatile("" : "+v"(acc[slot_a]));
                                if (k != N - 1 - k) {
                                    acc[slot_b] = acc[slot_b] + prod;
                                    asm volatile("" : "+v"(acc[slot_b]));
                                }
                            }
                            const int done = (kk + 1) % N;
                            const int y = ybase + kk;
                            if (y >= ys && y < yend) {
                                if (vec_store) *reinterpret_cast<f32x2 *>(optr) = acc[done];
                                else {
                                    if (gxo < W) optr[0] = acc[done].x;
                                    if (gxo + 1 < W) optr[1] = acc[done].y;
                                }
                            }
                            optr += W;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace siftk
using namespace siftk;
static int g_wgs = 1024;
template <int N> int nblocks_for(int W, int H, int TX) {
    const int gx = (W + TX - 1) / TX;
    int want_segments = (g_wgs + gx - 1) / gx;
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
    using G1 = MarchGeom<N, 128>; using G2 = XMarch2Geom<N, 128, S>;
    const int nb = nblocks_for<N>(W, H, G1::TX);
    const int rows_out = nb * N - (N - 1);
    dim3 grid((unsigned)((W + G1::TX - 1) / G1::TX), (unsigned)((H + rows_out - 1) / rows_out));
    hipMemset(o1, 0, (size_t)W * H * 4); hipMemset(o2, 0xff, (size_t)W * H * 4);
    float t1 = timeit([&] { hipLaunchKernelGGL((blur_march_kernel<N, false, 128, 0>), grid, dim3(128), (size_t)G1::LDS_BYTES, 0, (const void *)in, o1, W, H, nb, ta, (const uint32_t *)nullptr); });
    float t2 = timeit([&] { hipLaunchKernelGGL((blur_march2_kernel<N, false, 128, S, 0>), grid, dim3(128), (size_t)G2::LDS_BYTES, 0, (const void *)in, o2, W, H, nb, ta, (const uint32_t *)nullptr); });
    std::vector<float> a((size_t)W * H), b((size_t)W * H);
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    const bool same = memcmp(a.data(), b.data(), a.size() * 4) == 0;
    printf("N %2d S %d  %dx%d grid %ux%u nb %d  LDS %5d -> %5d B : march %.1f us, sub-block %.1f us  %s\n", N, S, W, H, grid.x, grid.y, nb,
           G1::LDS_BYTES, G2::LDS_BYTES, t1, t2, same ? "BITWISE EQUAL" : "MISMATCH");
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
    mk(17); run<17, 2>(in, o1, o2, W, H, taps); run<17, 3>(in, o1, o2, W, H, taps);
    mk(21); run<21, 2>(in, o1, o2, W, H, taps); run<21, 3>(in, o1, o2, W, H, taps);
    mk(27); run<27, 2>(in, o1, o2, W, H, taps); run<27, 3>(in, o1, o2, W, H, taps); run<27, 4>(in, o1, o2, W, H, taps);
    return 0;
}
