// k_extrema.hpp -- DoG extrema detection for the three detection scales in ONE pass over the six
// blur planes, and sub-pixel refinement with compaction.
//
// Replaces combine x5 (algebra.cl:18-37), local_maxmin x3 (image.cl:119-213), interp_keypoint x3
// (image.cl:235-369), compact x3 (algebra.cl:57-84) and the memsets between them.  DoG planes are
// never stored: DoG[s] = blur[s] - blur[s+1] is recomputed where needed, which is the same single
// IEEE subtraction the reference's combine() performs ((-1*b)+(1*a) == a-b exactly).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "siftmath.hpp"
#include "k_xcd.hpp"
#include <type_traits>

namespace siftk {

struct BlurPlanes { const float *p[6]; };

// 3x3x3 extremum test restated: v is kept as a maximum iff v > 0 and no sample of the 27 is
// strictly greater, i.e. v >= max27 (image.cl:156-167); likewise for minima.  max27 is separable:
// max over the 3 scales, then 3 columns (neighbour lanes), then 3 rows (rolling registers).
// One wave scans a strip 62 columns wide (lanes 0 and 63 are halo) and `rows` rows high.  A strip is a serial march
// (one row of loads in flight ahead of the row being tested), so its height is the latency of the launch: 32 rows on
// large planes, fewer where 32-row strips would leave most SIMDs without a wave (extrema_strip_rows).
#define SIFT_EXT_ROWS 64
inline int extrema_strip_rows(int W, int H, int border, int min_strips = 2000) {
    const int nx = (W - 2 * border + 61) / 62;
    int rows = SIFT_EXT_ROWS;
    while (rows > 4 && (int64_t)nx * ((H - 2 * border + rows - 1) / rows) < min_strips) rows >>= 1;   // 2000: 4096^2 -> 64, 2048^2 -> 32, 1024^2 -> 8, below -> 4
    return rows;
}

__device__ __forceinline__ float dog_at(const BlurPlanes &b, int s, size_t pos) { return b.p[s][pos] - b.p[s + 1][pos]; }

// 80 VGPRs (12 B of scratch) = 6 waves per SIMD instead of the natural 81 = 5: the kernel is latency bound (PMC: 70 % of
// the wave time in s_waitcnt), 8 waves (64 VGPRs, 64 B of scratch) is twice as slow.
//
// Candidates are appended to ONE list through one device-scope counter.  Same-address atomics are served one at a time by
// the L2 (measured 5.2 ns each: 20 k candidates = 104 us of a 118 us kernel, 200 k = 1 ms), so a lane never touches the
// counter itself: a wave parks its candidates in LDS (order-preserving ballot compaction), reserves slots for a whole
// buffer at once, and the four waves of a workgroup share a single atomicAdd for what is left at the end of their strips.
#ifndef SIFT_EXT_WAVES
#define SIFT_EXT_WAVES 5      // 92 VGPRs without scratch; forcing 6 waves (80 VGPRs) now spills 11 registers in the row loop: 0.18 ms instead of 0.10
#endif
#define SIFT_EXT_BUF 128          // candidates a wave parks before reserving slots (a row adds at most 3 x 62)
template <int BUF> struct ExtWaveLdsT { float4 buf[BUF + 192]; };
using ExtWaveLds = ExtWaveLdsT<SIFT_EXT_BUF>;

// what the refinement of a candidate needs besides the planes (refine_candidates below)
struct RefineArgs {
    float peak_thresh, init_sigma;
    float4 *kp;
    int *kp_aux, *n_kp;
    int kp_capacity, oct;
    int *c_scale;     // three counters of this octave: candidates per detection scale (Counters::c_scale), or null
};
__device__ __forceinline__ void refine_candidates(const BlurPlanes &b, int W, int H, const float4 *__restrict__ cand, int n,
                                                  float peak_thresh, float init_sigma, float4 *__restrict__ kp,
                                                  int *__restrict__ kp_aux, int *__restrict__ n_kp, int kp_capacity, int oct,
                                                  int first, int stride, int *__restrict__ c_scale);

// parked candidates [0, count) of a wave -> cand[slot ...]
template <int BUF>
__device__ __forceinline__ void ext_store_pending(const ExtWaveLdsT<BUF> &L, float4 *__restrict__ cand, int capacity, int slot, int count, int lane) {
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < count; e += 64)
        if (slot + e < capacity) cand[slot + e] = L.buf[e];
    __builtin_amdgcn_wave_barrier();
}

// Edge test of the parked extrema [from, pending), compacted in place (image.cl:176-196).  The 2-D Hessian needs sixteen
// more samples; taking them inside the row march made nearly every row of a textured frame run that divergent path for a
// lane or two (white noise: 1.5 % of the samples are 27-neighbour extrema), so the march only parks the extrema and the
// test runs here, one parked entry per lane.
template <int BUF>
__device__ __forceinline__ void ext_edge_filter(const BlurPlanes &b, int W, float edth, ExtWaveLdsT<BUF> &L, int from, int &pending, int lane) {
    int out = from;
    __builtin_amdgcn_wave_barrier();                                  // the entries were parked by other lanes of this wave
    for (int base = from; base < pending; base += 64) {               // wave uniform
        const int e = base + lane;
        bool keep = false;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < pending) {
            c = L.buf[e];
            const float val = c.x;
            const int s = (int)c.w;
            const size_t pc = (size_t)(int)c.y * W + (int)c.z;
            const float *pa = b.p[1], *pb = b.p[2];
            if (s == 2) { pa = b.p[2]; pb = b.p[3]; } else if (s == 3) { pa = b.p[3]; pb = b.p[4]; }
#define DOG_AT(o) (pa[pc + (o)] - pb[pc + (o)])
            // 2-D Hessian; "2.0" and "4.0" are double literals in image.cl:180-184
            const float up = DOG_AT(-(ptrdiff_t)W), dn = DOG_AT(W);
            const float lf = DOG_AT(-1), rt = DOG_AT(1);
            const float H00 = (float)(((double)up - 2.0 * (double)val) + (double)dn);
            const float H11 = (float)(((double)lf - 2.0 * (double)val) + (double)rt);
            const float dd = (DOG_AT(W + 1) - DOG_AT(W - 1)) - (DOG_AT(-(ptrdiff_t)W + 1) - DOG_AT(-(ptrdiff_t)W - 1));
#undef DOG_AT
            const float H01 = (float)((double)dd / 4.0);
            const float det = H00 * H11 - H01 * H01;
            const float tr = H00 + H11;
            keep = !(det < edth * tr * tr) && val != 0.0f;
        }
        __builtin_amdgcn_wave_barrier();                              // every lane holds its entry before slots are rewritten
        const unsigned long long m = __ballot(keep);
        if (keep) L.buf[out + __popcll(m & ((1ull << lane) - 1ull))] = c;
        out += __popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    pending = out;
}

// One wave marches strip (sx, sy) (all 64 lanes enter; an inactive wave does nothing but keeps the wave-wide operations
// convergent).  Extrema are parked in L.buf; `pending` (wave uniform) counts them; a full buffer goes through the edge
// test and is flushed through one atomicAdd.  What is still parked on return has passed the edge test and is the
// caller's to flush.
// REFINE: the survivors of the edge test are refined and appended to the keypoint list right here, one parked entry per
// lane (no candidate list, no refinement launch); nothing is left parked on return.
template <int BUF, bool REFINE = false>
__device__ __forceinline__ void extrema_strip(const BlurPlanes &b, int W, int H, int border, int rows, bool active, int sx, int sy,
                                              double contrast, float edth, float4 *__restrict__ cand, int *__restrict__ counter,
                                              int capacity, ExtWaveLdsT<BUF> &L, int &pending, const RefineArgs *ra = nullptr,
                                              int y_lo = -1, int y_hi = -1) {
    if (y_lo < 0) { y_lo = border; y_hi = H - border; }
    auto refine_parked = [&]() {
        __builtin_amdgcn_wave_barrier();
        refine_candidates(b, W, H, L.buf, pending, ra->peak_thresh, ra->init_sigma, ra->kp, ra->kp_aux, ra->n_kp, ra->kp_capacity,
                          ra->oct, threadIdx.x & 63, 64, ra->c_scale);
        __builtin_amdgcn_wave_barrier();
        pending = 0;
    };
    const int lane = threadIdx.x & 63;
    const int x = border + sx * 62 + lane - 1;
    const int xc = min(max(x, 0), W - 1);
    const bool col_ok = (lane >= 1) && (lane <= 62) && (x < W - border);
    const int ya = y_lo + sy * rows;                     // (y_lo, y_hi) = (border, H - border), or one band of it
    const int yb = active ? min(ya + rows, y_hi) : ya - 2;
    int tested = pending;                                // entries parked by earlier strips have had their edge test

    // Row loop: loads address a plane as SGPR base + one 32-bit byte offset shared by the six planes, advanced by a row
    // pitch per iteration (planes are < 4 GB); neighbour columns come through DPP wave shifts (lanes 0 / 63 receive 0:
    // they are the halo columns, their results are never used); the contrast test (double)|v| > contrast (image.cl:152)
    // is the float test |v| >= cf with cf the smallest float above `contrast` -- the same predicate without f64 work.
    // (Measured on a 4096^2 plane: 0.098 ms whatever the instruction count -- 200 or 110 per row, ds_bpermute or DPP,
    // 4 / 5 / 6 waves per SIMD, one or two rows of loads in flight, a hand-unrolled window without register moves was
    // even slower: the launch moves 403 MB at 4.1 TB/s and that is its bound.)
    float cf = (float)contrast;
    if (!((double)cf > contrast)) cf = __uint_as_float(__float_as_uint(cf) + 1u);   // contrast >= 0: next float up
    auto ld = [&](int k, unsigned off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(b.p[k]) + off); };
    auto from_left = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); };    // wave_shr:1
    auto from_right = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); };   // wave_shl:1

    float hM[3][3], hm[3][3];   // [row slot][scale]: horizontal+scale max / min for rows y-2, y-1, y
    float ctr[3] = {0.f, 0.f, 0.f}, ctr_next[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 3; k++) { hM[r][k] = 0.f; hm[r][k] = 0.f; }

    const unsigned pitch = (unsigned)W * 4u;
    unsigned off = ((unsigned)max(ya - 1, 0) * (unsigned)W + (unsigned)xc) * 4u;   // byte offset of the row being loaded
    float vn[6];                                     // next row's samples, loaded one iteration ahead
    if (active) {
#pragma unroll
        for (int k = 0; k < 6; k++) vn[k] = ld(k, off);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) vn[k] = 0.f;
    }
    for (int y = ya - 1; y <= yb; y++) {
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = vn[k];
        if (y < yb) {
            off += pitch;
#pragma unroll
            for (int k = 0; k < 6; k++) vn[k] = ld(k, off);
        }
        float d[5];
#pragma unroll
        for (int k = 0; k < 5; k++) d[k] = v[k] - v[k + 1];
        // shift rolling window
#pragma unroll
        for (int k = 0; k < 3; k++) { hM[0][k] = hM[1][k]; hM[1][k] = hM[2][k]; hm[0][k] = hm[1][k]; hm[1][k] = hm[2][k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float M = fmaxf(fmaxf(d[k], d[k + 1]), d[k + 2]);
            const float m = fminf(fminf(d[k], d[k + 1]), d[k + 2]);
            hM[2][k] = fmaxf(fmaxf(from_left(M), M), from_right(M));
            hm[2][k] = fminf(fminf(from_left(m), m), from_right(m));
            ctr_next[k] = d[k + 1];
        }
        // centre row is y-1; it is complete once rows y-2, y-1, y have been seen
        bool found[3] = {false, false, false};
        const int yc = y - 1;
        if (y >= ya + 1 && col_ok) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float val = ctr[k];
                if (fabsf(val) >= cf) {
                    const float M27 = fmaxf(fmaxf(hM[0][k], hM[1][k]), hM[2][k]);
                    const float m27 = fminf(fminf(hm[0][k], hm[1][k]), hm[2][k]);
                    found[k] = (val > 0.0f) ? (val >= M27) : (val <= m27);
                }
            }
        }
        // park this row's candidates (ballot compaction: no atomics)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const unsigned long long m = __ballot(found[k]);
            if (m) {                                                  // wave uniform
                if (found[k]) L.buf[pending + __popcll(m & ((1ull << lane) - 1ull))] = make_float4(ctr[k], (float)yc, (float)x, (float)(k + 1));
                pending += __popcll(m);
            }
        }
        if (pending > BUF) {                                 // wave uniform: edge test, then slots for the whole buffer
            ext_edge_filter(b, W, edth, L, tested, pending, lane);
            tested = 0;
            if (REFINE) refine_parked();
            else {
                int slot = 0;
                if (lane == 0) slot = atomicAdd(counter, pending);
                ext_store_pending(L, cand, capacity, __shfl(slot, 0), pending, lane);
                pending = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) ctr[k] = ctr_next[k];
    }
    ext_edge_filter(b, W, edth, L, tested, pending, lane);
    if (REFINE) refine_parked();
}

// (the fused-refinement form is the one of small planes, a latency chain of few workgroups: it takes the registers it
// needs -- at the five-wave budget it spilled four to scratch)
template <bool REFINE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(REFINE ? 4 : SIFT_EXT_WAVES, 8))) void extrema_kernel(BlurPlanes b, int W, int H, int border, int rows, double contrast,
                                                      float edth, float4 *__restrict__ cand,
                                                      int *__restrict__ counter, int capacity, RefineArgs ra,
                                                      int y_lo, int y_hi,        // rows [y_lo, y_hi) of the detection area (a band), or -1: all of it
                                                      int xcd_map) {             // strips in an order that gives every XCD one contiguous range (k_pyramid.hpp)
    __shared__ ExtWaveLds lds_all[4];
    __shared__ int s_pending[4], s_base;
    const int lane = threadIdx.x & 63;
    ExtWaveLds &L = lds_all[threadIdx.x >> 6];
    if (y_lo < 0) { y_lo = border; y_hi = H - border; }
    const int nx = (W - 2 * border + 61) / 62;
    const int ny = (y_hi - y_lo + rows - 1) / rows;
    const int wg = xcd_map ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int wid = wg * 4 + (threadIdx.x >> 6);
    const bool active = wid < nx * ny;                   // no early exit: the workgroup meets at the end
    int pending = 0;                                     // candidates parked in L.buf (wave uniform)
    extrema_strip<SIFT_EXT_BUF, REFINE>(b, W, H, border, rows, active, active ? wid % nx : 0, active ? wid / nx : 0, contrast, edth, cand,
                                        counter, capacity, L, pending, &ra, y_lo, y_hi);
    if (REFINE) return;                                  // every survivor is already in the keypoint list
    // ---- what is left leaves with one atomicAdd per workgroup
    if (lane == 0) s_pending[threadIdx.x >> 6] = pending;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_pending[0] + s_pending[1] + s_pending[2] + s_pending[3];
        s_base = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    {
        const int w = threadIdx.x >> 6;
        int slot = s_base;
        for (int q = 0; q < w; q++) slot += s_pending[q];
        ext_store_pending(L, cand, capacity, slot, pending, lane);
    }
}

// Sub-pixel refinement of every candidate + compaction of the survivors (image.cl:235-369,
// algebra.cl:57-84).  One thread per candidate, grid-stride over the device-side count.
// Output: (peak, row, col, sigma) and the integer detection scale (the reference keeps the
// scale implicitly as the loop variable of plan.py:626).
// c_scale (may be null): the octave's three counters of candidates per detection scale -- what the reference's keypoint
// counter holds after local_maxmin of that scale (plan.py:626-642); the host evaluates the reference's capacity rule from
// them (siftmi.hip: reference_overflow).  One atomic per wave and scale with a candidate.
__device__ __forceinline__ void refine_candidates(const BlurPlanes &b, int W, int H, const float4 *__restrict__ cand, int n,
                                                  float peak_thresh, float init_sigma, float4 *__restrict__ kp,
                                                  int *__restrict__ kp_aux, int *__restrict__ n_kp, int kp_capacity, int oct,
                                                  int first, int stride, int *__restrict__ c_scale) {
    int seen[3] = {0, 0, 0};              // candidates of scales 1 / 2 / 3 this lane has read
    for (int i = first; i < n; i += stride) {
        const float4 k = cand[i];
        int r = (int)k.y, c = (int)k.z;
        const int scale = (int)k.w;
        if (r == -1) continue;
        seen[0] += scale <= 1; seen[1] += scale == 2; seen[2] += scale >= 3;
        // (selects, not b.p[scale + k]: a BlurPlanes built at run time -- the tail kernel's -- indexed by a per-lane value
        // would have to live in scratch memory)
        const float *Pa = b.p[0], *Pb = b.p[1], *Pc = b.p[2], *Pd = b.p[3];
        if (scale == 2) { Pa = b.p[1]; Pb = b.p[2]; Pc = b.p[3]; Pd = b.p[4]; }
        else if (scale == 3) { Pa = b.p[2]; Pb = b.p[3]; Pc = b.p[4]; Pd = b.p[5]; }
        // P = DoG[scale-1] = Pa-Pb, D = DoG[scale] = Pb-Pc, N = DoG[scale+1] = Pc-Pd
        int newr = r, newc = c, moves = 5;
        bool again = true;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, peak = 0.f;
        while (again) {
            r = newr; c = newc;
            const size_t pos = (size_t)r * W + c;
#define DOGP(o) (Pa[pos + (o)] - Pb[pos + (o)])
#define DOGD(o) (Pb[pos + (o)] - Pc[pos + (o)])
#define DOGN(o) (Pc[pos + (o)] - Pd[pos + (o)])
            const float P0 = DOGP(0), D0 = DOGD(0), N0 = DOGN(0);
            const float Dd = DOGD(W), Du = DOGD(-W), Dr = DOGD(1), Dl = DOGD(-1);
            const float Pd_ = DOGP(W), Pu = DOGP(-W), Pr = DOGP(1), Pl = DOGP(-1);
            const float Nd = DOGN(W), Nu = DOGN(-W), Nr = DOGN(1), Nl = DOGN(-1);
            const float Ddr = DOGD(W + 1), Ddl = DOGD(W - 1), Dur = DOGD(-W + 1), Dul = DOGD(-W - 1);
#undef DOGP
#undef DOGD
#undef DOGN
            const float g0 = (N0 - P0) / 2.0f;
            const float g1 = (Dd - Du) / 2.0f;
            const float g2 = (Dr - Dl) / 2.0f;
            const float H00 = P0 - 2.0f * D0 + N0;
            const float H11 = Du - 2.0f * D0 + Dd;
            const float H22 = Dl - 2.0f * D0 + Dr;
            const float H01 = ((Nd - Nu) - (Pd_ - Pu)) / 4.0f;
            const float H02 = ((Nr - Nl) - (Pr - Pl)) / 4.0f;
            const float H12 = ((Ddr - Ddl) - (Dur - Dul)) / 4.0f;
            const float H10 = H01, H20 = H02, H21 = H12;
            const float det = -(H02 * H11 * H20) + H01 * H12 * H20 + H02 * H10 * H21
                              - H00 * H12 * H21 - H01 * H10 * H22 + H00 * H11 * H22;
            const float K00 = H11 * H22 - H12 * H21;
            const float K01 = H02 * H21 - H01 * H22;
            const float K02 = H01 * H12 - H02 * H11;
            const float K10 = H12 * H20 - H10 * H22;
            const float K11 = H00 * H22 - H02 * H20;
            const float K12 = H02 * H10 - H00 * H12;
            const float K20 = H10 * H21 - H11 * H20;
            const float K21 = H01 * H20 - H00 * H21;
            const float K22 = H00 * H11 - H01 * H10;
            s0 = -(g0 * K00 + g1 * K01 + g2 * K02) / det;
            s1 = -(g0 * K10 + g1 * K11 + g2 * K12) / det;
            s2 = -(g0 * K20 + g1 * K21 + g2 * K22) / det;
            peak = D0 + 0.5f * (s0 * g0 + s1 * g1 + s2 * g2);
            if (s1 > 0.6f && newr < H - 3) newr++;
            else if (s1 < -0.6f && newr > 3) newr--;
            if (s2 > 0.6f && newc < W - 3) newc++;
            else if (s2 < -0.6f && newc > 3) newc--;
            if (moves > 0 && (newr != r || newc != c)) moves--;
            else again = false;
        }
        if (fabsf(s0) <= 1.5f && fabsf(s1) <= 1.5f && fabsf(s2) <= 1.5f && fabsf(peak) >= peak_thresh) {
            const float sig = init_sigma * siftmath::exp2f_(((float)scale + s0) / 3.0f);
            const int slot = atomicAdd(n_kp, 1);
            if (slot < kp_capacity) {
                kp[slot] = make_float4(peak, (float)r + s1, (float)c + s2, sig);
                kp_aux[slot] = scale | (oct << 8);
            }
        }
    }
    if (c_scale) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            int t = seen[q];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
            if (t && (threadIdx.x & 63) == q) atomicAdd(c_scale + q, t);
        }
    }
}

__global__ __launch_bounds__(256) void refine_kernel(BlurPlanes b, int W, int H, const float4 *__restrict__ cand,
                                                     const int *__restrict__ n_cand, int cand_capacity,
                                                     float peak_thresh, float init_sigma,
                                                     float4 *__restrict__ kp, int *__restrict__ kp_aux,
                                                     int *__restrict__ n_kp, int kp_capacity, int oct,
                                                     int *__restrict__ c_scale) {
    // (a list cut at its capacity is seen by the host in the counters: it grows the list and runs the image again)
    const int n = min(*n_cand, cand_capacity);
    refine_candidates(b, W, H, cand, n, peak_thresh, init_sigma, kp, kp_aux, n_kp, kp_capacity, oct,
                      blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, c_scale);
}

// Stand-alone compaction (stage replay of algebra.cl:57-84 as called at plan.py:758-795): entries [start, end) of `in`
// whose row (s1) is not -1 are appended to `out` from *counter on, in no particular order -- the reference does not
// guarantee one either.  The hot path never runs this: its refinement appends survivors directly (refine_kernel).
// The append is convergent, so one atomic per wave reaches the counter (the compiler's atomic optimizer).
__global__ void compact_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int *__restrict__ counter,
                               int start, int end, int capacity) {
    for (int i = start + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        const float4 k = in[i];
        if (k.y != -1.0f) {
            const int slot = atomicAdd(counter, 1);
            if (slot < capacity) out[slot] = k;
        }
    }
}

// DoG plane (stage replay of algebra.cl:18-37 as called at plan.py:619-623)
__global__ void dog_kernel(const float *__restrict__ a, const float *__restrict__ bnext, float *__restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = -1.0f * bnext[i] + 1.0f * a[i];
}

}  // namespace siftk
