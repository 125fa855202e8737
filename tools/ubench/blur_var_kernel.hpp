// dev: variants of the marching team blur (sift_pyocl_amd/csrc/k_pyramid.hpp: blur_team_kernel) measured against it by
// tools/ubench/blur_var.hip.  Same arithmetic, same order, same bits; what changes is who does what and when.
//
// blur_front_kernel<N, NORM, S, DT, D, VAR>:
//   * the FRONT team (waves 0-1, the H team of the product kernel) also owns the global loads and the LDS staging, with D
//     register sets of look-ahead (sub-block q is loaded D steps before it is staged); the BACK team (waves 2-3) marches
//     vertically and stores, and never waits for a load.  In the product kernel the V team loads, stages, marches and stores:
//     its `s_waitcnt vmcnt(0)` in front of the staging also waits for the row stores it has just issued, its loads have one
//     VPASS of lead, and the H team sits at the step barrier while it waits (profiles/r06/pmc_blur_team_kernel.txt:
//     20 % of all wave-cycles in s_waitcnt, another 22-35 % in the barrier).
//   VAR bits: 1 = first accumulator period peeled (a warm-up row only feeds the accumulators whose window starts inside the
//                 segment: row r of the march adds taps j <= r only);
//             2 = phase clock (cycles per role and phase into `clk`);
//             4 = stage + prefetch AFTER the H pass of a step instead of before it.
#pragma once
#include "../../sift_pyocl_amd/csrc/k_pyramid.hpp"

namespace siftk {

template <int N, bool NORM, int S, int DT = 0, int D = 2, int VAR = 0>
__global__ __launch_bounds__(256) void blur_front_kernel(const void *__restrict__ in, float *__restrict__ out,
                                                         int W, int H, int nblocks, int last_subs, int rows_out,
                                                         TapsArg<N> taps, const uint32_t *__restrict__ mm,
                                                         float *__restrict__ next0, int xcd_map, unsigned long long *clk) {
    constexpr int NT = 128;
    using G = March2Geom<N, NT, S>;
    using SS = SubSplit<N, S>;
    static_assert(N & 1, "marching blur needs an odd tap count");
    static_assert(D >= 1 && D <= 3, "1..3 look-ahead register sets");
    constexpr int BUF = G::NPS * G::PITCH * 2;
    constexpr bool PEEL = (VAR & 1) != 0, CLK = (VAR & 2) != 0, LATE = (VAR & 4) != 0;
    extern __shared__ float4 smem4[];
    float *sbase = reinterpret_cast<float *>(smem4);
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 128 ? 1 : 0;
    const int tid = role ? (int)threadIdx.x - 128 : (int)threadIdx.x;
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (xcd_map) {
        const int gx = (int)gridDim.x;
        const int M = xcd_contiguous(bx + gx * by, gx * (int)gridDim.y);
        by = M / gx; bx = M - by * gx;
    }
    const int x0 = bx * G::TX;
    const int ys = by * rows_out;
    const int yend = min(ys + rows_out, H);
    float mn = 0.f, range = 1.f;
    if (NORM) { mn = ord2f(mm[0]); range = ord2f(mm[1]) - mn; }
    unsigned long long tmark = 0, tacc[6] = {0, 0, 0, 0, 0, 0};
    auto mark = [&](int k) {
        if (CLK) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tacc[k] += now - tmark; tmark = now; }
    };
    if (CLK) tmark = __builtin_amdgcn_s_memtime();

    if (role == 0) {
        // ------------------------------------------------------------------ FRONT: loads, staging, horizontal pass
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
        struct Set { f32x2 pa[G::NPS], pb[G::NPS], ph[G::NB]; };
        Set set0, set1, set2;
        auto prefetch = [&](Set &s, int blk, int sub, int np) {
            const int v0 = ys - G::C + blk * N + sub * SS::RB;
            if (v0 >= 0 && v0 + 2 * np <= H) {
                unsigned oa = ((unsigned)v0 * (unsigned)W + (unsigned)gx_a) * 4u;
                unsigned ob = ((unsigned)v0 * (unsigned)W + (unsigned)gx_b) * 4u;
#pragma unroll
                for (int rp = 0; rp < G::NPS; rp++)
                    if (rp < np) {
                        s.pa[rp].x = ld(oa); s.pa[rp].y = ld(oa + W4);
                        s.pb[rp].x = ld(ob); s.pb[rp].y = ld(ob + W4);
                        oa += 2u * W4; ob += 2u * W4;
                    }
            } else {
#pragma unroll
                for (int rp = 0; rp < G::NPS; rp++)
                    if (rp < np) {
                        const unsigned r0 = (unsigned)reflect_index(v0 + 2 * rp, H) * W4, r1 = (unsigned)reflect_index(v0 + 2 * rp + 1, H) * W4;
                        s.pa[rp].x = ld(r0 + 4u * gx_a); s.pa[rp].y = ld(r1 + 4u * gx_a);
                        s.pb[rp].x = ld(r0 + 4u * gx_b); s.pb[rp].y = ld(r1 + 4u * gx_b);
                    }
            }
#pragma unroll
            for (int u = 0; u < G::NB; u++) {
                s.ph[u] = (f32x2){0.f, 0.f};
                if (hb_rp[u] < np) {
                    s.ph[u].x = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u], H) * W4 + 4u * hb_gx[u]);
                    s.ph[u].y = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u] + 1, H) * W4 + 4u * hb_gx[u]);
                }
            }
        };
        auto stage = [&](const Set &s, float *dst, int np) {
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp < np) {
                    *reinterpret_cast<f32x2 *>(dst + (rp * G::PITCH + tid) * 2) = norm2(s.pa[rp]);
                    *reinterpret_cast<f32x2 *>(dst + (rp * G::PITCH + NT + tid) * 2) = norm2(s.pb[rp]);
                }
#pragma unroll
            for (int u = 0; u < G::NB; u++)
                if (hb_rp[u] < np) *reinterpret_cast<f32x2 *>(dst + (hb_rp[u] * G::PITCH + hb_col[u]) * 2) = norm2(s.ph[u]);
        };
        auto with_set = [&](int slot, auto f) {          // slot is wave uniform
            if (D == 1 || slot == 0) f(set0);
            else if (D == 2 || slot == 1) f(set1);
            else f(set2);
        };
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
        // linear sub-block index q = blk * S + sub; sub-block q lives in register set q % D and LDS buffer q % 3
        const int T = (nblocks - 1) * S + last_subs;
        auto pairs_of = [&](int q) { const int sub = q % S; int np = SS::pairs(0); for (int k = 1; k < S; k++) if (sub == k) np = SS::pairs(k); return np; };
        // prologue: the first D sub-blocks requested, the first one staged, its set refilled
        for (int q = 0; q < D && q < T; q++) with_set(q % D, [&](Set &s) { prefetch(s, q / S, q % S, pairs_of(q)); });
        with_set(0, [&](Set &s) { stage(s, sbase, pairs_of(0)); if (D < T) prefetch(s, D / S, D % S, pairs_of(D)); });
        mark(0);
        __syncthreads();
        mark(3);
        int g = 0;
        for (int blk = 0; blk < nblocks; blk++) {
#pragma unroll
            for (int sub = 0; sub < S; sub++) {
                if (blk == nblocks - 1 && sub >= last_subs) break;
                auto feed = [&]() {
                    if (g + 1 < T) {
                        const int q1 = g + 1;
                        with_set(q1 % D, [&](Set &s) {
                            stage(s, sbase + (q1 % 3) * BUF, pairs_of(q1));
                            mark(0);
                            const int q2 = q1 + D;
                            if (q2 < T) prefetch(s, q2 / S, q2 % S, pairs_of(q2));
                            mark(1);
                        });
                    }
                };
                if (!LATE) feed();
                hpass(sbase + (g % 3) * BUF, SS::pairs(sub));
                mark(2);
                if (LATE) feed();
                __syncthreads();
                mark(3);
                g++;
            }
        }
        if (CLK && (threadIdx.x & 63) == 0) for (int k = 0; k < 4; k++) atomicAdd(&clk[k], tacc[k]);
        return;
    }

    // ---------------------------------------------------------------------- BACK: vertical march and stores
    f32x2 acc[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = (f32x2){0.f, 0.f};
    const int gxo = x0 + 2 * tid;
    const bool vec_store = ((W & 1) == 0) && (gxo + 1 < W);

    // vertical march over the rows of sub-block `sub_` (compile time) of block blk_.  FIRST_ (compile time): the block is the
    // segment's first accumulator period -- march row r = kk adds to the windows that start at rows r - j >= 0 only.
#define VPASSX(sbuf, blk_, sub_, FIRST_)                                                                     \
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
                            const bool use_a = !(FIRST_) || k <= kk, use_b = (k != N - 1 - k) && (!(FIRST_) || (N - 1 - k) <= kk); \
                            if (use_a || use_b) {                                                            \
                                const f32x2 t2 = {taps.t[k], taps.t[k]};                                     \
                                const f32x2 prod = h * t2;                                                   \
                                const int slot_a = (kk - k + N) % N, slot_b = (kk - (N - 1 - k) + N) % N;    \
                                if (use_a) {                                                                 \
                                    if (k == 0) acc[slot_a] = (f32x2){0.f, 0.f} + prod;                      \
                                    else acc[slot_a] = acc[slot_a] + prod;                                   \
                                    asm volatile("" : "+v"(acc[slot_a]));                                    \
                                }                                                                            \
                                if (use_b) { acc[slot_b] = acc[slot_b] + prod; asm volatile("" : "+v"(acc[slot_b])); } \
                            }                                                                                \
                        }                                                                                    \
                        const int done = (kk + 1) % N;                                                       \
                        const int y = ybase_ + kk;                                                           \
                        if (y >= ys && y < yend) {                                                           \
                            if (vec_store) *reinterpret_cast<f32x2 *>(optr) = acc[done];                     \
                            else { if (gxo < W) optr[0] = acc[done].x; if (gxo + 1 < W) optr[1] = acc[done].y; } \
                            if (next0 && !(y & 1) && (y >> 1) < (H >> 1) && (gxo >> 1) < (W >> 1))           \
                                next0[(size_t)(y >> 1) * (W >> 1) + (gxo >> 1)] = acc[done].x;              \
                        }                                                                                    \
                        optr += W;                                                                           \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }
    // (k == 0 with FIRST_: tap 0 is the one a window receives FIRST -- j = 0, the row where it starts -- so a window that
    // starts inside the segment always gets its `0 + prod`; the windows that started above row 0 are never stored.)

    mark(5);
    __syncthreads();                               // the front team has staged sub-block 0
    mark(5);
    int g = 0;
    for (int blk = 0; blk < nblocks; blk++) {
#pragma unroll
        for (int sub = 0; sub < S; sub++) {
            if (blk == nblocks - 1 && sub >= last_subs) break;      // workgroup uniform
            if (g > 0) {
                float *prev = sbase + ((g + 2) % 3) * BUF;
                // the sub-block marched here is the PREVIOUS step's; it belongs to block 0 (the first accumulator period) in the
                // steps (0, 1..S-1) and (1, 0)
                if (sub == 0) {
                    if (PEEL && blk == 1) { VPASSX(prev, 0, S - 1, true) } else { VPASSX(prev, blk - 1, S - 1, false) }
                } else {
                    if (PEEL && blk == 0) { VPASSX(prev, 0, (sub + S - 1) % S, true) } else { VPASSX(prev, blk, (sub + S - 1) % S, false) }
                }
            }
            mark(4);
            __syncthreads();
            mark(5);
            g++;
        }
    }
    {
        float *prev = sbase + ((g + 2) % 3) * BUF;
        if (last_subs >= S) { VPASSX(prev, nblocks - 1, S - 1, false) }
        if constexpr (S > 1) { if (last_subs == 1) { VPASSX(prev, nblocks - 1, 0, false) } }
        if constexpr (S > 2) { if (last_subs == 2) { VPASSX(prev, nblocks - 1, 1, false) } }
        if constexpr (S > 3) { if (last_subs == 3) { VPASSX(prev, nblocks - 1, 2, false) } }
    }
    mark(4);
#undef VPASSX
    if (CLK && (threadIdx.x & 63) == 0) for (int k = 4; k < 6; k++) atomicAdd(&clk[k], tacc[k]);
}

}  // namespace siftk
