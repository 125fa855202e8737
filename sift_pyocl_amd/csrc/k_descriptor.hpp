// k_descriptor.hpp -- the 128-D descriptor, row-interval form: ONE WAVEFRONT per oriented keypoint, four keypoints per
// 256-thread workgroup, no workgroup barriers inside a keypoint.
//
// Same arithmetic and the same per-bin order of float additions as the reference's CPU kernel
// (keypoints_cpu.cl:36-161): every in-window sample's eight trilinear contributions are added to their bins one by
// one in the raster order of the (2R+1)^2 window.
//
//  1. ROW INTERVALS instead of a scan of the window.  For a fixed window row ii the reference's `inside` predicate
//     (keypoints_cpu.cl:68-72) holds on a contiguous run of jj: rx(jj), cx(jj) are compositions of monotone IEEE
//     operations (a product with a constant, two subtractions, a division by the positive `spacing`, an addition), so
//     each of the four comparisons flips once along the row; the image bounds are intervals too.  The division is taken
//     out of the search: g(u) = u / spacing + 1.5f is non-decreasing in u, hence g(u) < 4 <=> u < t_hi and
//     g(u) > -1 <=> u > t_lo for two float thresholds found once per keypoint by stepping to the exact flip points of g
//     (verified; the search falls back to evaluating g when the verification fails, e.g. non-finite spacing).  One lane
//     per row locates the four flips on the exact float expression of u (a closed-form bracket of four evaluations around
//     the real crossing; bisection where the bracket finds no step), a wave prefix sum turns the run lengths into the raster
//     rank of every in-window sample; a row's (first rank, first column) is one packed word of the wave's row table, and a
//     lane finds the row of its rank among four words requested a batch ahead.  (The streaming form tests all (2R+1)^2
//     positions, half of which lie outside the rotated window, and re-packs its sample list every 64 samples: a quarter
//     of its time.)
//  2. EVALUATION.  64 consecutive ranks at a time, one sample per lane: gradient magnitude / orientation from
//     blur[scale] (image.cl:58-77), Gaussian weight, the eight (bin, value) contributions.  atan2 / exp go through the
//     Ziv fast paths of siftmath.hpp (bit-identical to the defining functions).
//  3. ORDERED ACCUMULATION.  Lane l owns bins l and l + 64.
//     a. A sample lane sets its bit in S[half of the wave][cell * 8 + o] for each of its (up to four) cells, o being the
//        lower of its two orientation bins: one 32-bit LDS atomic OR per cell (order independent).
//     b. The owner of bin (cell, ob) forms the bin's 64-bit contributor mask S[ob] | S[ob - 1] (a sample reaches ob as
//        its lower or as its upper orientation bin), counts it, and a wave prefix sum of the counts (DPP, no LDS round
//        trip) gives every bin a 16-byte aligned segment of a value pool; the owner publishes a 16-byte ENTRY: the mask
//        and the LDS address of the segment.
//     c. Each sample lane reads the entries of its eight bins and stores each value at segment + 4 * (number of lower
//        lanes in the mask): ascending lane == raster order.
//     d. The owner adds its segments front to back, four values per LDS read.
//     A cell that does not exist (outside the 4 x 4 grid, or no sample in the lane) is aimed at a private dummy word
//     (a.) and at ONE shared dummy entry whose mask is all ones (c.: rank == lane, so every lane lands in its own dump
//     slot): the routing code has no validity tests or exec-masked blocks, only the address selects.
//     (PMC, round 3, profiles/r03: 70 % of the kernel's VALU instructions were integer / select / move instructions of
//     this bookkeeping and binary64 6 %; with those cut the LDS became the pole -- 16-byte entries put every cell's
//     bins on the same eight bank groups -- hence word arrays for the atomics and one atomic per cell, not per bin.)
//  4. Normalise / clamp 0.2 / renormalise / quantise with the reference's sequential 128-term sums.
//
// Windows with more than 2 * SIFT_DESC_MAXRAD + 1 rows are left to descriptor_stream_kernel (a plan cannot produce them:
// the 64-tap limit of the blur schedule caps init_sigma at 4.08, i.e. 247 rows; the stage entry point can).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "k_keypoint.hpp"

namespace siftk {

#define SIFT_DESC_MAXRAD 127
#ifndef SIFT_DESC_WAVES
#define SIFT_DESC_WAVES 4      // 128 VGPRs, no scratch: 4.03 ms against 4.44 ms at 5 waves (96 VGPRs, 76 B of scratch) on 154 k keypoints
#endif

typedef float desc_f2 __attribute__((ext_vector_type(2)));
#ifdef SIFT_PHASE_CLOCK
#define PH_PARAM , PhaseClock *php
#define PH_PASS , &ph
#define PH_NONE , nullptr
#define PHP_MARK(k) do { if (php) php->mark((k), lane); } while (0)
#else
#define PH_PARAM
#define PH_PASS
#define PH_NONE
#define PHP_MARK(k)
#endif

struct alignas(16) DescEntry { unsigned mlo, mhi, seg, spare; };     // contributor mask (lanes 0-31 / 32-63), LDS byte address of the first contributor's pool slot
#define SIFT_DESC_DUMMY 128                    // entry [128]: the shared dummy {~0, ~0, dump slots}: a lane's rank in it is its lane number
struct alignas(16) DescPool {
    float pool[1024];                          // [0, 896): <= 512 values, every bin's segment zero-padded to a multiple of 4;
                                               // [896, 900): zeros, read by an owner past its segment; [960 + lane]: dump slots
    DescEntry ent[SIFT_DESC_DUMMY + 1];        // published by the owners every batch
    unsigned S[2][128];                        // [half][cell * 8 + o]: lanes of that half with a sample in `cell`, lower orientation bin o
    unsigned Sdummy[64];                       // lane l's private word for the cells that do not exist (never read)
};
struct alignas(16) DescRowLds {
    DescPool P;                                // P.pool doubles as the 128 squares of the normalisation (step 4)
    // per window row: (exclusive prefix of the per-row run lengths) << 8 | (first in-window jj + 128); rows [S, S + 4): ~0
    // (the look-up reads four rows ahead; a row start beyond every rank never counts as passed)
    unsigned row_pack[2 * SIFT_DESC_MAXRAD + 8];
#ifdef SIFT_PHASE_CLOCK
    unsigned long long ph[16];
#endif
};

__device__ __forceinline__ unsigned desc_lds_addr(const void *p) { return (unsigned)(uintptr_t)p; }   // LDS byte address of a __shared__ object
__device__ __forceinline__ void desc_pool_init(DescPool &P, int lane) {
    P.S[0][lane] = 0u; P.S[0][lane + 64] = 0u; P.S[1][lane] = 0u; P.S[1][lane + 64] = 0u;
    P.Sdummy[lane] = 0u;
    // (the workgroup-per-keypoint form reads the entries of every wave's area each round: an area whose wave has had no
    // batch yet -- a window of fewer than 256 samples -- must read as "no contributors")
    *reinterpret_cast<uint4 *>(&P.ent[lane]) = make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4 *>(&P.ent[lane + 64]) = make_uint4(0u, 0u, 0u, 0u);
    if (lane == 0) *reinterpret_cast<uint4 *>(&P.ent[SIFT_DESC_DUMMY]) = make_uint4(~0u, ~0u, desc_lds_addr(&P.pool[960]), 0u);
    if (lane < 4) P.pool[896 + lane] = 0.0f;
}

// LDS objects are addressed by their 32-bit byte address (the low half of the generic address of a __shared__ object):
// a pointer kept in registers or selected against another one would be a 64-bit generic pointer and every access a
// flat_* instruction.  The constant part of an address (which of the four cells) rides in the instruction's offset field.
typedef unsigned desc_u4v __attribute__((ext_vector_type(4)));
typedef float desc_f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned desc_lds_u32;
typedef __attribute__((address_space(3))) const desc_u4v desc_lds_u4;
typedef __attribute__((address_space(3))) float desc_lds_f32;
typedef __attribute__((address_space(3))) desc_f4v desc_lds_f4;
template <int K> __device__ __forceinline__ void desc_or32(unsigned addr, unsigned bits) {
    (void)__hip_atomic_fetch_or(reinterpret_cast<desc_lds_u32 *>(addr + K), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int K> __device__ __forceinline__ uint4 desc_entry(unsigned addr) {
    const desc_u4v v = *reinterpret_cast<desc_lds_u4 *>(addr + K);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// offsets of the four cells (rb, cb), (rb, cb + 1), (rb + 1, cb), (rb + 1, cb + 1) from the first, in bins (8 per cell)
#define SIFT_DESC_C0 0
#define SIFT_DESC_C1 8
#define SIFT_DESC_C2 32
#define SIFT_DESC_C3 40

// what a lane needs to address its wave's routing tables
struct DescRoute {
    unsigned ent;       // LDS address of entry 0
    unsigned dummy;     // ... of the shared dummy entry
    unsigned S;         // ... of S[this lane's half][0]
    unsigned Sdummy;    // ... of the lane's private word
    unsigned bit;       // 1 << (lane & 31)
};
__device__ __forceinline__ DescRoute desc_route_of(DescPool &P, int lane) {
    DescRoute r;
    r.ent = desc_lds_addr(&P.ent[0]); r.dummy = desc_lds_addr(&P.ent[SIFT_DESC_DUMMY]);
    r.S = desc_lds_addr(&P.S[lane >> 5][0]); r.Sdummy = desc_lds_addr(&P.Sdummy[lane]);
    r.bit = 1u << (lane & 31);
    return r;
}

// next representable float towards +inf (up) or -inf, x finite and non-zero
__device__ __forceinline__ float f32_neighbour(float x, bool up) {
    const int b = __float_as_int(x);
    return __int_as_float(((b >= 0) == up) ? b + 1 : b - 1);
}


// Row interval of window row ii (keypoints_cpu.cl:64-72): the first in-window jj and the number of in-window samples.
// The four flips of the row's predicates are monotone false -> true in jj (see 1. above).  Each is located by a CLOSED-FORM
// BRACKET: the real crossing of the linear expression, x* = (ci_ - drow - T) / sine (resp. (T + dcol - si_) / cosine), lies
// within a fraction of a step of the float expression's flip, so the exact predicate is evaluated at four consecutive
// integers around floor(x*) and the flip is read off; positions outside [-R, R] count as false below and true above.
// A lane whose four evaluations do not contain a false -> true step (an estimate that is off: sine == 0, NaN) reports
// failure and the caller runs the bisection for the whole wave -- the result is the same either way, the bracket only
// replaces ~6 bisection rounds of four evaluations each (round 4: ~700 of a keypoint's ~2400 set-up instructions).
struct DescRowSearch {
    float sine, cosine, drow, dcol, t_hi, t_lo, rsine, rcosine;
    int R;
    bool rdec, cdec;
};
// flip q of 0..3 (0 / 1: entry into / exit from the row band, 2 / 3: the column band), all four through ONE copy of the code:
// u = (A + B * jj) - D with (A, B, D) = (ci_, -sine, drow) or (si_, cosine, dcol) -- a - s * j == a + (-s) * j exactly --,
// the threshold and the direction of the comparison picked by wave-uniform selects (a loop the compiler must not unroll:
// four inlined copies held enough registers to push loop-invariant values of the kernel into scratch)
__device__ __forceinline__ int desc_row_flip(const DescRowSearch &q, int which, float ci_, float si_, bool &ok) {
    const bool col = which >= 2, odd = which & 1;
    const bool dec = col ? q.cdec : q.rdec;
    const float A = col ? si_ : ci_, B = col ? q.cosine : -q.sine, D = col ? q.dcol : q.drow;
    // even: first jj inside the band (dec: u < t_hi, else u > t_lo); odd: first jj beyond it (dec: !(u > t_lo), else !(u < t_hi))
    const bool use_hi = (dec != odd);                        // which threshold the comparison is against
    const float T = use_hi ? q.t_hi : q.t_lo;
    const float xs = col ? ((T + q.dcol) - si_) * q.rcosine : ((ci_ - q.drow) - T) * q.rsine;
    // (a NaN / infinite estimate lands anywhere in the clamp range: the step test below decides)
    const int k0 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(xs) - 1.0f, (float)(-q.R - 2)), (float)(q.R - 2));
    auto at = [&](int e) {
        const int k = k0 + e;
        const float u = (A + B * (float)k) - D;
        // u < t_hi (even, dec) | u > t_lo (even, !dec) | !(u > t_lo) (odd, dec) | !(u < t_hi) (odd, !dec)
        const bool cmp = use_hi ? (u < T) : (u > T);
        return (k > q.R) || (k >= -q.R && (cmp != odd));
    };
    // monotone false -> true: the flip is k0 + 1 + (number of false among positions 1 and 2), provided position 0 is false and
    // position 3 true (otherwise the flip lies outside the four positions: no claim)
    int f = k0 + 1;
    ok = ok && !at(0);
    f += at(1) ? 0 : 1;
    f += at(2) ? 0 : 1;
    ok = ok && at(3);
    return f;
}

// The bisection the bracket replaces: exact for any row, used where the thresholds could not be verified (thr_ok false: the
// predicates then evaluate g(u) = u / spacing + 1.5f itself) or a lane's bracket found no step.  Not inlined: it runs for
// a handful of keypoints per image and would otherwise hold its registers across the common path.
__device__ __attribute__((noinline)) int4 desc_row_bisect(const DescRowSearch q, float spacing, bool thr_ok, int iters, float ci_, float si_) {
    auto g = [&](float u) { return u / spacing + 1.5f; };
    auto below_hi = [&](float u) { return thr_ok ? (u < q.t_hi) : (g(u) < 4.0f); };
    auto above_lo = [&](float u) { return thr_ok ? (u > q.t_lo) : (g(u) > -1.0f); };
    int lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { lo[k] = -q.R; hi[k] = q.R + 1; }
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int mid = (lo[k] + hi[k]) >> 1;
            const float fj = (float)mid;
            const float u = (k < 2) ? ((ci_ - q.sine * fj) - q.drow) : ((si_ + q.cosine * fj) - q.dcol);
            const bool dec = (k < 2) ? q.rdec : q.cdec;
            // k even: first jj inside the band; k odd: first jj beyond it (monotone false ... true predicates)
            bool pred;
            if ((k & 1) == 0) pred = dec ? below_hi(u) : above_lo(u);
            else pred = dec ? !above_lo(u) : !below_hi(u);
            if (lo[k] < hi[k]) { if (pred) hi[k] = mid; else lo[k] = mid + 1; }
        }
    }
    return make_int4(lo[0], lo[1], lo[2], lo[3]);
}

// what is the same for every sample of a keypoint's window (wave uniform)
struct DescWindow {
    const float *I;            // blur[scale] of the keypoint's octave
    const float *G, *O;        // MAPS forms: its gradient magnitude / orientation maps (OctaveTable::gmap / omap)
    unsigned W4;               // row pitch in bytes
    int W, H, irow, icol;
    float sine, cosine, drow, dcol, spacing, rspacing, angle;
    bool fast_div, interior;
};

// The four neighbours a sample's gradient needs (image.cl:58-77), fetched one batch ahead of their use: the loads of
// batch b + 1 are in flight while batch b is evaluated and accumulated (a wave runs its batches one after the other and
// only four waves share a SIMD: what a wave does not overlap itself is not overlapped).
struct DescSample { int ii, jj; float right, left, up, down; };
template <bool INTERIOR, bool MAPS>
__device__ __forceinline__ void desc_fetch(const DescWindow &w, int ii, int jj, DescSample &q) {
    // loads address the plane as scalar base + 32-bit byte offset
    const int x = w.icol + jj, y = w.irow + ii;
    const unsigned off = __umul24((unsigned)y, w.W4) + ((unsigned)x << 2);   // planes hold <= 2^30 pixels, rows <= 2^22 bytes
    auto ld = [&](unsigned o) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(w.I) + o); };
    q.ii = ii; q.jj = jj;
    if (MAPS) {                 // (magnitude, orientation) of the sample, as compute_gradient_orientation left them
        q.right = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(w.G) + off);
        q.left = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(w.O) + off);
        q.up = 0.0f; q.down = 0.0f;
    } else if (INTERIOR) {
        q.right = ld(off + 4u); q.left = ld(off - 4u); q.up = ld(off - w.W4); q.down = ld(off + w.W4);
    } else {
        q.right = ld(x == w.W - 1 ? off : off + 4u); q.left = ld(x == 0 ? off : off - 4u);
        q.up = ld(y == 0 ? off : off - w.W4); q.down = ld(y == w.H - 1 ? off : off + w.W4);
    }
}

// One sample of the descriptor window: everything keypoints_cpu.cl:74-117 does for it, except the additions -- in four
// parts, so that the software-pipelined batch loop (desc_one_wave) can put one of them behind each of the LDS round trips
// of the PREVIOUS batch's routing; desc_eval below runs them back to back.
//   1. window coordinates, gradient, argument of the weight        2. orientation (Ziv candidate)
//   3. Gaussian weight (Ziv candidate) + the rare fall-backs        4. bins, the eight values, routing addresses
// INTERIOR: the whole window lies at least one pixel inside the plane (no one-sided differences, image.cl:58-77).
struct DescEvalState {
    desc_f2 RC;               // (rx, cx)
    float gx, gy, g, earg, o, ew;
    bool ok_a, ok_e;
    siftmath::Atan2Try at;    // the orientation's candidate between parts 2 and 3
};
template <bool INTERIOR, bool MAPS>
__device__ __forceinline__ void desc_eval1(const DescWindow &w, const DescSample &q, DescEvalState &st) {
    // ---- window coordinates (keypoints_cpu.cl:64-67): rx = ((cos*i - sin*j) - drow) / spacing + 1.5, cx likewise.
    //      Both at once in packed-f32 instructions: -(cos*j) == (-cos)*j exactly, so uc = sin*i - (-cos)*j.
    const int ii = q.ii, jj = q.jj;
    const float fi = (float)ii, fj = (float)jj;
    const desc_f2 A = (desc_f2){w.cosine, w.sine} * (desc_f2){fi, fi};
    const desc_f2 B = (desc_f2){w.sine, -w.cosine} * (desc_f2){fj, fj};
    const desc_f2 U = (A - B) - (desc_f2){w.drow, w.dcol};
    desc_f2 Q;
    if (w.fast_div) {          // Markstein: the correctly rounded quotient from the correctly rounded reciprocal (siftmath.hpp)
        const desc_f2 rb = {w.rspacing, w.rspacing}, nb = {-w.spacing, -w.spacing};
        Q = U * rb;
        desc_f2 r = __builtin_elementwise_fma(nb, Q, U);
        Q = __builtin_elementwise_fma(r, rb, Q);
        r = __builtin_elementwise_fma(nb, Q, U);
        Q = __builtin_elementwise_fma(r, rb, Q);
    } else {
        Q = (desc_f2){U.x / w.spacing, U.y / w.spacing};
    }
    st.RC = Q + (desc_f2){1.5f, 1.5f};
    // ---- gradient of blur[scale] at the sample (image.cl:58-77)
    float gx = q.right - q.left, gy = q.up - q.down;
    if (!INTERIOR) {
        const int x = w.icol + jj, y = w.irow + ii;
        if ((x == 0) || (x == w.W - 1)) gx = 2.0f * gx;
        if ((y == 0) || (y == w.H - 1)) gy = 2.0f * gy;
    }
    st.gx = gx; st.gy = gy;
    st.g = MAPS ? q.right : sqrtf(gx * gx + gy * gy);
    st.o = MAPS ? q.left : 0.0f;
    const desc_f2 E = st.RC - (desc_f2){1.5f, 1.5f};
    const desc_f2 E2 = E * E;
    st.earg = -0.125f * (E2.x + E2.y);
    st.ok_a = true;
}
template <bool MAPS>
__device__ __forceinline__ void desc_eval2(DescEvalState &st, const double *fold) {
    if (!MAPS) siftmath::atan2f_fast_begin(-st.gy, st.gx, fold, st.at);      // (its table value is on its way while part 3 starts)
}
template <bool MAPS>
__device__ __forceinline__ void desc_eval3(DescEvalState &st) {
    st.ew = siftmath::expf_fast_try(st.earg, st.ok_e);
    if (!MAPS) st.o = siftmath::atan2f_fast_end(st.at, st.ok_a);
    if (!(st.ok_a && st.ok_e)) {                  // 2^-14 of the samples: the defining functions
        if (!MAPS && !st.ok_a) st.o = siftmath::atan2f_(-st.gy, st.gx);
        if (!st.ok_e) st.ew = siftmath::expf_(st.earg);
    }
}
// tgs[c] (+ the cell's constant): LDS address of the S word of cell c; tgt[n] (+ the cell's constant): LDS address of
// the entry of the bin contribution n goes to, cval[n] its value; the dummies where the cell does not exist or the lane
// is not `live`.
__device__ __forceinline__ void desc_eval4(const DescWindow &w, const DescEvalState &st, bool live, const DescRoute &rt,
                                           unsigned (&tgs)[4], unsigned (&tgt)[8], float (&cval)[8]) {
    const desc_f2 RC = st.RC;
    const float rx = RC.x, cx = RC.y;
    const float mag = st.g * st.ew;
    float o = st.o - w.angle;
    // keypoints_cpu.cl:85-87: while (o > 2 pi) o -= 2 pi; while (o < 0) o += 2 pi.  o and angle lie in [-pi, pi]: the first
    // loop never runs and the second at most once -- one select; the loops themselves only where a lane is still out of
    // range after it (wave uniform test; same result: a lane that was negative has had its first addition, the
    // subtracting loop does not apply to it, and the adding loop continues where the reference's would)
    if (o < 0.0f) o += 2.0f * SM_PI_F;
    if (__builtin_expect(__ballot(o > 2.0f * SM_PI_F || o < 0.0f) != 0ull, 0)) {
        while (o > 2.0f * SM_PI_F) o -= 2.0f * SM_PI_F;
        while (o < 0.0f) o += 2.0f * SM_PI_F;
    }
    const float oval = 4.0f * o * SM_1_PI_F;
    const int ri = (int)((rx >= 0.0f) ? rx : rx - 1.0f);
    const int ci = (int)((cx >= 0.0f) ? cx : cx - 1.0f);
    const int oi = (int)((oval >= 0.0f) ? oval : oval - 1.0f);
    const desc_f2 F = RC - (desc_f2){(float)ri, (float)ci};
    const float rf = F.x, cf = F.y, of = oval - (float)oi;
    const desc_f2 G = (desc_f2){1.0f, 1.0f} - F;
    const bool contributes = live && ri >= -1 && ri < 4 && oi >= 0 && oi <= 8 && rf >= 0.0f && rf <= 1.0f;
    // the eight values, products in the reference's order: ((mag * row weight) * column weight) * orientation weight
    const desc_f2 RW = (desc_f2){mag, mag} * (desc_f2){G.x, rf};                 // a = 0, 1
    const desc_f2 CWa = (desc_f2){RW.x, RW.x} * (desc_f2){G.y, cf};              // a = 0: bb = 0, 1
    const desc_f2 CWb = (desc_f2){RW.y, RW.y} * (desc_f2){G.y, cf};              // a = 1
    const desc_f2 OW = {1.0f - of, of};
    const desc_f2 v00 = (desc_f2){CWa.x, CWa.x} * OW, v01 = (desc_f2){CWa.y, CWa.y} * OW;
    const desc_f2 v10 = (desc_f2){CWb.x, CWb.x} * OW, v11 = (desc_f2){CWb.y, CWb.y} * OW;
    // the eight bins (rb * 4 + cb) * 8 + ob.  Orientation bin oi + 1 wraps to 0; oi == 8 happens only for oval == 8.0f
    // exactly (o == 2*pi_f), where of == 0: e = 0 adds cw*1 to bin 0, e = 1 would add cw*0 == +0 to bin 0 again (no
    // effect on a non-negative sum); here such a sample counts as one of orientation bin 0 whose upper value is +0.
    const bool nd = oi != 8;
    cval[0] = v00.x; cval[1] = nd ? v00.y : 0.0f; cval[2] = v01.x; cval[3] = nd ? v01.y : 0.0f;
    cval[4] = v10.x; cval[5] = nd ? v10.y : 0.0f; cval[6] = v11.x; cval[7] = nd ? v11.y : 0.0f;
    const unsigned c00 = (unsigned)((ri * 4 + ci) * 8);                          // first bin of the cell (used where it exists)
    const unsigned o0 = (unsigned)(oi & 7), o1 = (unsigned)((oi + 1) & 7);
    const unsigned s0 = rt.S + 4u * (c00 + o0);
    const unsigned t0 = rt.ent + 16u * (c00 + o0), t1 = rt.ent + 16u * (c00 + o1);
    const bool r0 = contributes && ri >= 0, r1 = contributes && ri < 3;          // rb = ri, ri + 1 in 0..3
    const bool k0 = ci >= 0 && ci < 4, k1 = ci >= -1 && ci < 3;                  // cb = ci, ci + 1 in 0..3
    const bool v0 = r0 && k0, v1 = r0 && k1, v2 = r1 && k0, v3 = r1 && k1;
    tgs[0] = v0 ? s0 : rt.Sdummy - 4u * SIFT_DESC_C0;   tgs[1] = v1 ? s0 : rt.Sdummy - 4u * SIFT_DESC_C1;
    tgs[2] = v2 ? s0 : rt.Sdummy - 4u * SIFT_DESC_C2;   tgs[3] = v3 ? s0 : rt.Sdummy - 4u * SIFT_DESC_C3;
    tgt[0] = v0 ? t0 : rt.dummy - 16u * SIFT_DESC_C0;   tgt[1] = v0 ? t1 : rt.dummy - 16u * SIFT_DESC_C0;
    tgt[2] = v1 ? t0 : rt.dummy - 16u * SIFT_DESC_C1;   tgt[3] = v1 ? t1 : rt.dummy - 16u * SIFT_DESC_C1;
    tgt[4] = v2 ? t0 : rt.dummy - 16u * SIFT_DESC_C2;   tgt[5] = v2 ? t1 : rt.dummy - 16u * SIFT_DESC_C2;
    tgt[6] = v3 ? t0 : rt.dummy - 16u * SIFT_DESC_C3;   tgt[7] = v3 ? t1 : rt.dummy - 16u * SIFT_DESC_C3;
}
// the four parts back to back (workgroup-per-keypoint form; the two Ziv candidates share a basic block: two independent
// binary64 chains side by side, one branch for both fall-backs)
template <bool INTERIOR, bool MAPS>
__device__ __forceinline__ void desc_eval(const DescWindow &w, const DescSample &q, bool live, const double *fold, const DescRoute &rt,
                                          unsigned (&tgs)[4], unsigned (&tgt)[8], float (&cval)[8]) {
    DescEvalState st;
    desc_eval1<INTERIOR, MAPS>(w, q, st);
    desc_eval2<MAPS>(st, fold);
    desc_eval3<MAPS>(st);
    desc_eval4(w, st, live, rt, tgs, tgt, cval);
}

// Steps 3a-3c for one batch of a wave.  On return the pool holds every bin's values in lane order; (base, padded counts)
// of the caller's two bins come back for the owners' sums.
__device__ __forceinline__ void desc_route(DescPool &P, const DescRoute &rt, const unsigned (&tgs)[4], const unsigned (&tgt)[8],
                                           const float (&cval)[8], int lane, int &base_a, int &pa, int &pb PH_PARAM) {
    // ---- 3a. one 32-bit LDS atomic per cell
    desc_or32<4 * SIFT_DESC_C0>(tgs[0], rt.bit); desc_or32<4 * SIFT_DESC_C1>(tgs[1], rt.bit);
    desc_or32<4 * SIFT_DESC_C2>(tgs[2], rt.bit); desc_or32<4 * SIFT_DESC_C3>(tgs[3], rt.bit);
    __builtin_amdgcn_wave_barrier();
    // ---- 3b. bin owners: mask = S[ob] | S[ob - 1] of the cell, counts, 16-byte aligned pool segments from a wave prefix sum
    const int prev = (lane & ~7) | ((lane + 7) & 7);           // same cell, orientation bin - 1 (bins lane and lane + 64 alike)
    const unsigned alo = P.S[0][lane] | P.S[0][prev], ahi = P.S[1][lane] | P.S[1][prev];
    const unsigned blo = P.S[0][lane + 64] | P.S[0][prev + 64], bhi = P.S[1][lane + 64] | P.S[1][prev + 64];
    const int cnta = __popc(alo) + __popc(ahi), cntb = __popc(blo) + __popc(bhi);
    pa = (cnta + 3) & ~3; pb = (cntb + 3) & ~3;
    base_a = wave_prefix_incl(pa + pb) - (pa + pb);
    const int base_b = base_a + pa;
    const unsigned pool0 = desc_lds_addr(&P.pool[0]);
    *reinterpret_cast<uint4 *>(&P.ent[lane]) = make_uint4(alo, ahi, pool0 + 4u * (unsigned)base_a, 0u);
    *reinterpret_cast<uint4 *>(&P.ent[lane + 64]) = make_uint4(blo, bhi, pool0 + 4u * (unsigned)base_b, 0u);
    // the last group of four of every segment starts as +0: its padding then adds +0 (an exact no-op on these
    // non-negative sums), so the owners' loops need no per-element masks
    if (cnta) *reinterpret_cast<desc_lds_f4 *>(pool0 - 16u + 4u * (unsigned)base_b) = (desc_f4v){0.f, 0.f, 0.f, 0.f};
    if (cntb) *reinterpret_cast<desc_lds_f4 *>(pool0 - 16u + 4u * (unsigned)(base_b + pb)) = (desc_f4v){0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_wave_barrier();
    PHP_MARK(6);
    // ---- 3c. every value to segment start + 4 * rank among the contributors of its bin (mbcnt: set bits of the mask
    //          below this lane); the eight 16-byte reads are in flight together
    // (two rounds of four: eight 12-byte entries at once cost 24 registers at the kernel's register peak)
    auto place = [&](const uint4 &en, float v) {
        *reinterpret_cast<desc_lds_f32 *>(en.z + 4u * __builtin_amdgcn_mbcnt_hi(en.y, __builtin_amdgcn_mbcnt_lo(en.x, 0u))) = v;
    };
    {
        const uint4 e0 = desc_entry<16 * SIFT_DESC_C0>(tgt[0]), e1 = desc_entry<16 * SIFT_DESC_C0>(tgt[1]);
        const uint4 e2 = desc_entry<16 * SIFT_DESC_C1>(tgt[2]), e3 = desc_entry<16 * SIFT_DESC_C1>(tgt[3]);
        place(e0, cval[0]); place(e1, cval[1]); place(e2, cval[2]); place(e3, cval[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint4 e4 = desc_entry<16 * SIFT_DESC_C2>(tgt[4]), e5 = desc_entry<16 * SIFT_DESC_C2>(tgt[5]);
        const uint4 e6 = desc_entry<16 * SIFT_DESC_C3>(tgt[6]), e7 = desc_entry<16 * SIFT_DESC_C3>(tgt[7]);
        place(e4, cval[4]); place(e5, cval[5]); place(e6, cval[6]); place(e7, cval[7]);
    }
    __builtin_amdgcn_wave_barrier();
    PHP_MARK(7);
}

// after the owners' sums: the S words of the caller's two bins are cleared for the next batch
__device__ __forceinline__ void desc_route_reset(DescPool &P, int lane) {
    P.S[0][lane] = 0u; P.S[1][lane] = 0u; P.S[0][lane + 64] = 0u; P.S[1][lane + 64] = 0u;
}

// the owner's ordered sum of one segment (n4 groups of four, the first at pool4[q]); reads run one group ahead -- the
// group after the last one of a segment is another segment's or the zeros at [896], in bounds either way
__device__ __forceinline__ float desc_sum_segment(const float4 *pool4, int q, int n4, float acc) {
    if (n4 > 0) {
        float4 v = pool4[q];
        for (int g4 = 0; g4 < n4; g4++) {
            const float4 c = v;
            v = pool4[q + g4 + 1];
            __builtin_amdgcn_sched_barrier(0);
            acc = acc + c.x; acc = acc + c.y; acc = acc + c.z; acc = acc + c.w;
        }
    }
    return acc;
}

// acc + v as one v_add_f32, written out: the compiler otherwise pairs the two chains into v_pk_add_f32 and assembles
// every operand pair with two v_mov_b32 (8 packed adds + 12 moves per batch; a packed op holds the SIMD for 4.2 cycles,
// a plain f32 add for 2.4 and lets an integer / LDS-address instruction issue beside it: profiles/r04/valu_issue_rate.txt)
__device__ __forceinline__ float desc_add(float acc, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(v));
    return acc;
#else
    return acc + v;
#endif
}

// The ordered sums of a lane's two bins (adjacent segments: a at group qa, then b).  A bin receives 2.6 values per batch
// on average and rarely more than eight: the first two groups of both chains are read together (one LDS round trip for
// both sums of nearly every lane; a lane past its segment reads the four zeros at group 224: +0 is an exact no-op on
// these non-negative sums), the rest -- if any lane of the wave has one -- in a loop with one read in flight per chain.
__device__ __forceinline__ void desc_sum_pair(const float4 *pool4, int qa, int ea, int eb, float &acc0, float &acc1) {
    const int qb = qa + ea;
    const float4 a0 = pool4[ea > 0 ? qa : 224], a1 = pool4[ea > 1 ? qa + 1 : 224];
    const float4 b0 = pool4[eb > 0 ? qb : 224], b1 = pool4[eb > 1 ? qb + 1 : 224];
    __builtin_amdgcn_sched_barrier(0);
    acc0 = desc_add(acc0, a0.x); acc1 = desc_add(acc1, b0.x);
    acc0 = desc_add(acc0, a0.y); acc1 = desc_add(acc1, b0.y);
    acc0 = desc_add(acc0, a0.z); acc1 = desc_add(acc1, b0.z);
    acc0 = desc_add(acc0, a0.w); acc1 = desc_add(acc1, b0.w);
    acc0 = desc_add(acc0, a1.x); acc1 = desc_add(acc1, b1.x);
    acc0 = desc_add(acc0, a1.y); acc1 = desc_add(acc1, b1.y);
    acc0 = desc_add(acc0, a1.z); acc1 = desc_add(acc1, b1.z);
    acc0 = desc_add(acc0, a1.w); acc1 = desc_add(acc1, b1.w);
    if (__ballot(ea > 2 || eb > 2)) {                 // wave uniform
        const int nmax = max(ea, eb);
        float4 va = pool4[ea > 2 ? qa + 2 : 224], vb = pool4[eb > 2 ? qb + 2 : 224];
        for (int g4 = 2; g4 < nmax; g4++) {
            const float4 ca = va, cb = vb;
            va = pool4[(g4 + 1 < ea) ? qa + g4 + 1 : 224];
            vb = pool4[(g4 + 1 < eb) ? qb + g4 + 1 : 224];
            __builtin_amdgcn_sched_barrier(0);
            acc0 = desc_add(acc0, ca.x); acc1 = desc_add(acc1, cb.x);
            acc0 = desc_add(acc0, ca.y); acc1 = desc_add(acc1, cb.y);
            acc0 = desc_add(acc0, ca.z); acc1 = desc_add(acc1, cb.z);
            acc0 = desc_add(acc0, ca.w); acc1 = desc_add(acc1, cb.w);
        }
    }
}

// per-keypoint set-up shared by the two forms: the window's constants from the oriented keypoint
template <bool MAPS>
__device__ __forceinline__ void desc_window(const OctaveTable &tab, const float4 kq, int aux, DescWindow &w, int &R) {
    const int scale = aux & 0xff, oct = aux >> 8;
    w.W = tab.W[oct]; w.H = tab.H[oct];
    w.W4 = (unsigned)w.W * 4u;
    w.I = tab.base + tab.off[oct] + (size_t)scale * w.W * w.H;
    w.G = MAPS ? tab.gmap + map_offset(tab, oct, scale) : nullptr;
    w.O = MAPS ? tab.omap + map_offset(tab, oct, scale) : nullptr;
    const float foct = (float)(1 << oct);
    const float row = kq.y / foct, col = kq.x / foct;
    w.angle = kq.w;
    w.irow = (int)(row + 0.5f); w.icol = (int)(col + 0.5f);
    siftmath::sincosf_(w.angle, &w.sine, &w.cosine);
    w.spacing = kq.z / foct * 3.0f;
    R = (int)((1.414f * w.spacing * 2.5f) + 0.5f);
    w.drow = row - (float)w.irow; w.dcol = col - (float)w.icol;
    w.rspacing = 1.0f / w.spacing;
    w.interior = w.irow - R >= 1 && w.irow + R <= w.H - 2 && w.icol - R >= 1 && w.icol + R <= w.W - 2;
}

// One oriented keypoint (x, y, sigma*oct, angle), detection scale | octave << 8 in `aux`, described by the calling WAVE:
// steps 1-4 of the file header; the record is number `rec_i` of the group's block (store_record).  Everything about the keypoint is wave uniform.
template <bool MAPS>
__device__ __forceinline__ void desc_one_wave(const OctaveTable &tab, const float4 kq, int aux, const RecordSink *sink, int rec_i,
                                              DescRowLds &L, const double *fold, const DescRoute &rt, const float4 *pool4, int lane PH_PARAM) {
    {
#ifdef SIFT_PHASE_CLOCK
        PhaseClock &ph = *php;
#endif
        DescWindow w;
        int R;
        desc_window<MAPS>(tab, kq, aux, w, R);
        const int W = w.W, H = w.H, irow = w.irow, icol = w.icol;
        const float sine = w.sine, cosine = w.cosine, spacing = w.spacing, drow = w.drow, dcol = w.dcol;
        const int S = 2 * R + 1;
        if (R > SIFT_DESC_MAXRAD) __builtin_trap();   // the host launches descriptor_stream_kernel for such plans

        // ---- 1a. thresholds: g(u) < 4 <=> u < t_hi, g(u) > -1 <=> u > t_lo.  Lane l tries the float l - 32 steps away from
        //          2.5 * spacing (resp. its negative): g is monotone, so the candidates that satisfy g >= 4 (g <= -1) are
        //          the upper lanes; the first of them is the flip point, provided the lane below it exists (verification).
        auto g = [&](float u) { return u / spacing + 1.5f; };
        float t_hi = 0.0f, t_lo = 0.0f;
        bool thr_ok = spacing > 1e-30f && spacing < 1e30f;
        if (thr_ok) {
            const float cand = __int_as_float(__float_as_int(2.5f * spacing) + lane - 32);
            const unsigned long long m_hi = __ballot(g(cand) >= 4.0f), m_lo = __ballot(g(-cand) <= -1.0f);
            const int i_hi = m_hi ? __ffsll(m_hi) - 1 : 0, i_lo = m_lo ? __ffsll(m_lo) - 1 : 0;
            thr_ok = i_hi > 0 && i_lo > 0 && (m_hi >> i_hi) == (~0ull >> i_hi) && (m_lo >> i_lo) == (~0ull >> i_lo);
            t_hi = __shfl(cand, i_hi);
            t_lo = -__shfl(cand, i_lo);
        }
        w.fast_div = thr_ok && spacing >= 0.1f && spacing <= 128.0f;      // no under / overflow in div_by_reciprocal

        PH_COUNT(13);
        PH_MARK(1);
        // ---- 1b. row intervals: lane l owns the window rows l, l + 64, ...
        const bool rdec = sine >= 0.0f;          // u_r non-increasing in jj
        const bool cdec = !(cosine >= 0.0f);     // u_c non-increasing in jj
        const int iters = 32 - __clz(S);         // 2R + 2 candidate positions
        int carry = 0;
        for (int r0 = 0; r0 < S; r0 += 64) {
            const DescRowSearch rs = {sine, cosine, drow, dcol, t_hi, t_lo, __builtin_amdgcn_rcpf(sine), __builtin_amdgcn_rcpf(cosine), R, rdec, cdec};
            const int r = r0 + lane;
            const int ii = r - R;
            const float fi = (float)ii;
            const float ci_ = cosine * fi, si_ = sine * fi;
            int lo[4] = {0, 0, 0, 0};
            bool bracket_ok = thr_ok;
            if (thr_ok) {                                // wave uniform
#pragma unroll 1
                for (int which = 0; which < 4; which++) {            // (no lo[which]: a run-time register index)
                    const int f = desc_row_flip(rs, which, ci_, si_, bracket_ok);
                    lo[0] = which == 0 ? f : lo[0]; lo[1] = which == 1 ? f : lo[1];
                    lo[2] = which == 2 ? f : lo[2]; lo[3] = which == 3 ? f : lo[3];
                }
            }
            if (__ballot(!bracket_ok)) {                 // wave uniform, rare: the bisection (exact for every lane)
                const int4 b4 = desc_row_bisect(rs, spacing, thr_ok, iters, ci_, si_);
                lo[0] = b4.x; lo[1] = b4.y; lo[2] = b4.z; lo[3] = b4.w;
            }
            const int jlo = max(max(lo[0], lo[2]), max(-R, -icol));
            const int jhi = min(min(lo[1], lo[3]) - 1, min(R, W - 1 - icol));
            const int yy = irow + ii;
            int c = jhi - jlo + 1;
            if (r >= S || yy < 0 || yy >= H || c < 0) c = 0;
            const int incl = wave_prefix_incl(c);
            // (an empty row may report a first jj of R + 1 = 128: clamped, it is never the row of a rank)
            if (r < S) L.row_pack[r] = ((unsigned)(carry + incl - c) << 8) | (unsigned)(min(jlo, 127) + 128);
            carry += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane < 4) L.row_pack[S + lane] = ~0u;
        const int total = carry;
        __builtin_amdgcn_wave_barrier();
        PH_MARK(2);

        // ---- 2 + 3. 64 ranks at a time.  Two forms of the loop (SIFT_DESC_PIPE); the product build uses the plain one.
        //      SOFTWARE-PIPELINED form (built in round 4 on the hypothesis that the launch is bound by the LDS round trips of
        //      a batch's routing -- contributor atomics -> masks -> entries -> pool slots -> sums --: alone it scales with the
        //      number of resident waves, 288 / 576 / 960 workgroups 438 / 288 / 242 us).  Every LDS request of batch b's routing
        //      is followed by a quarter of the EVALUATION of batch b + 1 (which touches no LDS but the 8-byte fold entry)
        //      before its result is used (checked in the disassembly: no wait is left in front of a stage but the one for the
        //      words that stage consumes, an evaluation part earlier):
        //          R1  atomics of b, S words requested, row table of b + 2 requested       | E1  coordinates, gradient of b + 1
        //          R2  masks, counts, prefix sum, entries published, entries requested,    | E2  orientation of b + 1
        //              rows of b + 2 resolved, its four neighbours requested (HBM)         |
        //          R3  ranks, pool stores, the owners' segments requested                  | E3  weight of b + 1 (+ rare fall-backs)
        //          R4  ordered sums, S words cleared                                       | E4  bins, values, addresses of b + 1
        //      LDS operations of one wave execute in program order, so the stages need no waits between them -- only the
        //      compiler must not reorder: every stage boundary is a scheduling barrier.  The same additions in the same
        //      order as the plain loop.  RESULT: no faster at the same number of waves, and it needs 152-168 VGPRs (three waves
        //      per SIMD; at 128 it spills 29).  What the hypothesis missed: a wave issues one instruction per ~5 cycles
        //      whatever it is (profiles/r04/valu_issue_rate.txt, one wave per SIMD), a batch is ~600 instructions (418 VALU,
        //      62 SALU, 36 LDS, ~85 waits / nops / branches) = ~3000 cycles of the wave's own issue against ~700 of exposed
        //      LDS latency, and the pipelined loop does not have fewer instructions.  More waves or fewer instructions move
        //      this launch; overlap inside a wave does not.
        float acc0 = 0.0f, acc1 = 0.0f;  // bins lane and lane + 64
        int rcur = 0;                    // row of this lane's current rank (ranks only grow)
        unsigned rword = L.row_pack[0];  // ... and its packed (start, first jj)
        DescSample nxt;
        // The row of a rank: ranks grow by 64 per batch, rows hold 20-70 samples, so a lane moves on by one to three rows per
        // batch.  The packed words of the FOUR rows after the current one are requested a whole batch before they are
        // needed (a search that read a row, compared and read the next was a chain of dependent LDS round trips at the
        // head of every batch: a quarter of a wave's time, tools/dev/phase_clock.py); a lane whose rank lies further on --
        // the short rows at a window corner -- looks again (wave uniform, rare).
        // A lane beyond the last rank evaluates the last sample again, aimed at its dummy entry: no divergence, no
        // default values to materialise.
        unsigned wd[4];
        auto row_request = [&]() {
            wd[0] = L.row_pack[rcur + 1]; wd[1] = L.row_pack[rcur + 2]; wd[2] = L.row_pack[rcur + 3]; wd[3] = L.row_pack[rcur + 4];
        };
        auto fetch = [&](int s0) {
            const int sc = min(s0 + lane, total - 1);
            const unsigned lim = (unsigned)(sc + 1) << 8;
            bool more = true;
            for (;;) {
                const bool a0 = wd[0] < lim, a1 = wd[1] < lim, a2 = wd[2] < lim, a3 = wd[3] < lim;
                if (more) {
                    rword = a3 ? wd[3] : (a2 ? wd[2] : (a1 ? wd[1] : (a0 ? wd[0] : rword)));
                    rcur += (int)a0 + (int)a1 + (int)a2 + (int)a3;
                }
                more = more && a3;
                if (!__ballot(more)) break;
                row_request();
            }
            const int ii = rcur - R, jj = (int)(rword & 0xffu) - 128 + (sc - (int)(rword >> 8));
            if (w.interior) desc_fetch<true, MAPS>(w, ii, jj, nxt);
            else desc_fetch<false, MAPS>(w, ii, jj, nxt);
            row_request();               // for the batch after this one
        };
        row_request();
        if (total > 0) fetch(0);
        PH_MARK(3);
        for (int s0 = 0; s0 < total; s0 += 64) {
            const DescSample cur = nxt;
            if (s0 + 64 < total) fetch(s0 + 64);     // wave uniform: the next batch's neighbours, in flight during this one
#ifdef SIFT_PHASE_CLOCK
            PH_COUNT(14);
            PH_MARK(3);
            if (s0 + 64 < total) { if (MAPS) PH_WAIT_VM(2); else PH_WAIT_VM(4); } else PH_WAIT_VM(0);
            PH_MARK(4);
#endif
            unsigned tgs[4], tgt[8];
            float cval[8];
            if (w.interior) desc_eval<true, MAPS>(w, cur, s0 + lane < total, fold, rt, tgs, tgt, cval);
            else desc_eval<false, MAPS>(w, cur, s0 + lane < total, fold, rt, tgs, tgt, cval);
            PH_MARK(5);
            int base_a, pa, pb;
            desc_route(L.P, rt, tgs, tgt, cval, lane, base_a, pa, pb PH_PASS);
            // ---- 3d. ordered sums of this lane's two bins
            desc_sum_pair(pool4, base_a >> 2, pa >> 2, pb >> 2, acc0, acc1);
            __builtin_amdgcn_wave_barrier();
            PH_MARK(8);
            desc_route_reset(L.P, lane);
            __builtin_amdgcn_wave_barrier();
        }

        // ---- 4. normalise, clamp at 0.2, renormalise, quantise (keypoints_cpu.cl:125-160): the reference sums the 128
        //         squares sequentially in index order; every lane repeats that sum from LDS (eight 16-byte reads in
        //         flight per batch of 32 terms).
        auto sum_squares = [&]() {
            float t = 0.0f;
#pragma unroll 1
            for (int k8 = 0; k8 < 4; k8++) {
                float4 q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) q[u] = reinterpret_cast<const float4 *>(L.P.pool)[8 * k8 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { t = t + q[u].x; t = t + q[u].y; t = t + q[u].z; t = t + q[u].w; }
            }
            return t;
        };
        L.P.pool[lane] = acc0 * acc0; L.P.pool[lane + 64] = acc1 * acc1;
        __builtin_amdgcn_wave_barrier();
        float norm = 1.0f / sqrtf(sum_squares());       // rsqrt
        acc0 = acc0 * norm; acc1 = acc1 * norm;
        const bool ch = (acc0 > 0.2f) || (acc1 > 0.2f);
        if (acc0 > 0.2f) acc0 = 0.2f;
        if (acc1 > 0.2f) acc1 = 0.2f;
        __builtin_amdgcn_wave_barrier();
        if (__ballot(ch)) {
            L.P.pool[lane] = acc0 * acc0; L.P.pool[lane + 64] = acc1 * acc1;
            __builtin_amdgcn_wave_barrier();
            const float n2 = 1.0f / sqrtf(sum_squares());
            acc0 = acc0 * n2; acc1 = acc1 * n2;
        }
        __builtin_amdgcn_wave_barrier();
        // (int)(512.0*v) in double, MIN(255, .), NaN -> 0 (see the oracle's note)
        const int i0 = (acc0 == acc0) ? (int)(512.0 * (double)acc0) : 0;
        const int i1 = (acc1 == acc1) ? (int)(512.0 * (double)acc1) : 0;
        store_record(sink, rec_i, kq, min(255, i0), min(255, i1), lane, reinterpret_cast<unsigned char *>(L.P.pool));
        PH_MARK(9);
    }
}

// `next`: device counter for dynamic hand-out (null: static stride).  Every wave takes keypoint `start + its index` first;
// after that it asks the counter, so that a wave with a small window does not idle while another still has two large
// ones to go (windows differ by 4x in samples within an octave).
template <bool MAPS>
__device__ __forceinline__ void descriptor_waves(const OctaveTable &tab, const float4 *__restrict__ okp, const int *__restrict__ oaux,
                                                 int start, int end, const RecordSink *sink, DescRowLds *lds_all, double *fold, int *next, int nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    DescRowLds &L = lds_all[wave];
#ifdef SIFT_PHASE_CLOCK
    PhaseClock ph;
    ph.start(L.ph, lane);
#endif
    siftmath::load_atan_fold(fold);
    desc_pool_init(L.P, lane);
    __syncthreads();                     // the only workgroup barrier: the fold table
    PH_MARK(0);
    const int gwave = blockIdx.x * 4 + wave, nwaves = nblocks * 4;
    const DescRoute rt = desc_route_of(L.P, lane);
    const float4 *pool4 = reinterpret_cast<const float4 *>(L.P.pool);

    auto advance = [&](int i) {
        if (!next) return i + nwaves;
        int t = 0;
        if (lane == 0) t = atomicAdd(next, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        PH_MARK(10);
        return start + nwaves + t;
    };
    for (int t = start + gwave; t < end; t = advance(t)) {
        const int i = t;                 // list order (see the note on hand-out orders above descriptor_kernel)
        // the keypoint is the same in every lane: keep its integer attributes in scalar registers
        float4 kq = okp[i];              // (x, y, sigma*oct, angle)
        kq.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.x)));
        kq.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.y)));
        kq.z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.z)));
        kq.w = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.w)));
        const int aux = __builtin_amdgcn_readfirstlane(oaux[i]);         // detection scale | octave << 8
        if (!(kq.y >= 0.0f)) {           // hole of an oriented list (stage replay only)
            store_record(sink, i, kq, 0, 0, lane, reinterpret_cast<unsigned char *>(L.P.pool));
            continue;
        }
        desc_one_wave<MAPS>(tab, kq, aux, sink, i, L, fold, rt, pool4, lane PH_PASS);
    }
#ifdef SIFT_PHASE_CLOCK
    ph.flush(16, lane);
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// The same descriptor, ONE WORKGROUP (four waves) per keypoint: for sparse groups.  With a wave per keypoint a launch
// lasts as long as its slowest keypoint (60-120 us: ~40-75 batches of 64 samples, one after the other, on a SIMD that
// has nothing else to issue), however few keypoints there are.  Here the four waves evaluate four consecutive batches
// at once, each into its own entries / pool, and after a workgroup barrier the bin owners (threads 0-127, one bin
// each) add the four areas in batch order -- the same additions in the same order, so the same bits.  One launch holds
// both forms; the group's count, known on the device only, picks one (descriptor_kernel, team_below).
struct alignas(16) DescTeamLds {
    DescPool w[4];
    float V[128];
    int Q[128];
    int row_start[2 * SIFT_DESC_MAXRAD + 4];
    short row_jlo[2 * SIFT_DESC_MAXRAD + 4];
    int blk_total[4];
};

template <bool MAPS>
__device__ __forceinline__ void descriptor_team(const OctaveTable &tab, const float4 *__restrict__ okp, const int *__restrict__ oaux,
                                                int start, int end, const RecordSink *sink, DescTeamLds &T, double *fold) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DescPool &P = T.w[wave];
    siftmath::load_atan_fold(fold);
    desc_pool_init(P, lane);
    __syncthreads();
    const DescRoute rt = desc_route_of(P, lane);

    for (int i = start + blockIdx.x; i < end; i += gridDim.x) {      // workgroup uniform
        float4 kq = okp[i];              // (x, y, sigma*oct, angle)
        kq.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.x)));
        kq.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.y)));
        kq.z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.z)));
        kq.w = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(kq.w)));
        const int aux = __builtin_amdgcn_readfirstlane(oaux[i]);         // detection scale | octave << 8
        if (!(kq.y >= 0.0f)) {           // hole of an oriented list (stage replay only)
            __syncthreads();             // the previous keypoint's record may still be leaving through T.V
            if (wave == 0) store_record(sink, i, kq, 0, 0, lane, reinterpret_cast<unsigned char *>(T.V));
            continue;
        }
        DescWindow w;
        int R;
        desc_window<MAPS>(tab, kq, aux, w, R);
        const int W = w.W, H = w.H, irow = w.irow, icol = w.icol;
        const float sine = w.sine, cosine = w.cosine, spacing = w.spacing, drow = w.drow, dcol = w.dcol;
        const int S = 2 * R + 1;
        if (R > SIFT_DESC_MAXRAD) __builtin_trap();   // the host launches descriptor_stream_kernel for such plans

        // ---- 1a. thresholds (every wave for itself: 64 candidates in one ballot), as in descriptor_waves
        auto g = [&](float u) { return u / spacing + 1.5f; };
        float t_hi = 0.0f, t_lo = 0.0f;
        bool thr_ok = spacing > 1e-30f && spacing < 1e30f;
        if (thr_ok) {
            const float cand = __int_as_float(__float_as_int(2.5f * spacing) + lane - 32);
            const unsigned long long m_hi = __ballot(g(cand) >= 4.0f), m_lo = __ballot(g(-cand) <= -1.0f);
            const int i_hi = m_hi ? __ffsll(m_hi) - 1 : 0, i_lo = m_lo ? __ffsll(m_lo) - 1 : 0;
            thr_ok = i_hi > 0 && i_lo > 0 && (m_hi >> i_hi) == (~0ull >> i_hi) && (m_lo >> i_lo) == (~0ull >> i_lo);
            t_hi = __shfl(cand, i_hi);
            t_lo = -__shfl(cand, i_lo);
        }
        w.fast_div = thr_ok && spacing >= 0.1f && spacing <= 128.0f;

        // ---- 1b. row intervals: thread t owns window row t (S <= 255)
        {
            const bool rdec = sine >= 0.0f, cdec = !(cosine >= 0.0f);
            const int iters = 32 - __clz(S);
            const int r = tid;
            const int ii = r - R;
            const float fi = (float)ii;
            const float ci_ = cosine * fi, si_ = sine * fi;
            int lo[4] = {-R, -R, -R, -R};
            if (64 * wave < S) {                                      // wave uniform
                const DescRowSearch rs = {sine, cosine, drow, dcol, t_hi, t_lo, __builtin_amdgcn_rcpf(sine), __builtin_amdgcn_rcpf(cosine), R, rdec, cdec};
                bool bracket_ok = thr_ok;
                if (thr_ok) {
#pragma unroll 1
                    for (int which = 0; which < 4; which++) {
                        const int f = desc_row_flip(rs, which, ci_, si_, bracket_ok);
                        lo[0] = which == 0 ? f : lo[0]; lo[1] = which == 1 ? f : lo[1];
                        lo[2] = which == 2 ? f : lo[2]; lo[3] = which == 3 ? f : lo[3];
                    }
                }
                if (__ballot(!bracket_ok)) {                          // wave uniform, rare: the bisection
                    const int4 b4 = desc_row_bisect(rs, spacing, thr_ok, iters, ci_, si_);
                    lo[0] = b4.x; lo[1] = b4.y; lo[2] = b4.z; lo[3] = b4.w;
                }
            }
            const int jlo = max(max(lo[0], lo[2]), max(-R, -icol));
            const int jhi = min(min(lo[1], lo[3]) - 1, min(R, W - 1 - icol));
            const int yy = irow + ii;
            int c = jhi - jlo + 1;
            if (r >= S || yy < 0 || yy >= H || c < 0) c = 0;
            const int incl = wave_prefix_incl(c);
            if (lane == 63) T.blk_total[wave] = incl;
            __syncthreads();                                          // (also: the previous keypoint's epilogue is over)
            int offset = 0;
            for (int q = 0; q < wave; q++) offset += T.blk_total[q];
            if (r < S) { T.row_start[r] = offset + incl - c; T.row_jlo[r] = (short)jlo; }
            if (tid == 0) T.row_start[S] = T.blk_total[0] + T.blk_total[1] + T.blk_total[2] + T.blk_total[3];
            __syncthreads();
        }
        const int total = T.row_start[S];

        // ---- 2 + 3. four batches of 64 ranks at a time, one per wave
        float acc = 0.0f;                // bin tid (threads 0-127)
        int rcur = 0;
        for (int t0 = 0; t0 < total; t0 += 256) {                     // workgroup uniform
            const int s = t0 + 64 * wave + lane;
            if (t0 + 64 * wave < total) {                             // wave uniform: this wave has a batch
                unsigned tgs[4], tgt[8];
                float cval[8];
                {
                    const int sc = min(s, total - 1);
                    while (sc >= T.row_start[rcur + 1]) rcur++;
                    const int ii = rcur - R, jj = (int)T.row_jlo[rcur] + (sc - T.row_start[rcur]);
                    DescSample q;
                    if (w.interior) { desc_fetch<true, MAPS>(w, ii, jj, q); desc_eval<true, MAPS>(w, q, s < total, fold, rt, tgs, tgt, cval); }
                    else { desc_fetch<false, MAPS>(w, ii, jj, q); desc_eval<false, MAPS>(w, q, s < total, fold, rt, tgs, tgt, cval); }
                }
                int base_a, pa, pb;
                desc_route(P, rt, tgs, tgt, cval, lane, base_a, pa, pb PH_NONE);
                desc_route_reset(P, lane);           // (the owners below read the published entries, not S)
            }
            __syncthreads();
            // 3d. owners: bin tid, the four areas in batch order (an area without a batch this round has empty masks)
            if (tid < 128) {
                uint4 e[4];
#pragma unroll
                for (int q = 0; q < 4; q++) e[q] = *reinterpret_cast<const uint4 *>(&T.w[q].ent[tid]);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n4 = (__popc(e[q].x) + __popc(e[q].y) + 3) >> 2;
                    if (n4) {
                        acc = desc_sum_segment(reinterpret_cast<const float4 *>(T.w[q].pool), (int)((e[q].z - desc_lds_addr(&T.w[q].pool[0])) >> 4), n4, acc);
                        *reinterpret_cast<uint2 *>(&T.w[q].ent[tid]) = make_uint2(0u, 0u);     // an area may have no batch next round
                    }
                }
            }
            __syncthreads();
        }

        // ---- 4. normalise, clamp at 0.2, renormalise, quantise (keypoints_cpu.cl:125-160), as in descriptor_waves
        auto sum_squares = [&]() {
            float t = 0.0f;
#pragma unroll 1
            for (int k8 = 0; k8 < 4; k8++) {
                float4 q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) q[u] = reinterpret_cast<const float4 *>(T.V)[8 * k8 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { t = t + q[u].x; t = t + q[u].y; t = t + q[u].z; t = t + q[u].w; }
            }
            return t;
        };
        if (tid < 128) T.V[tid] = acc * acc;
        __syncthreads();
        const float norm = 1.0f / sqrtf(sum_squares());
        acc = acc * norm;
        const bool ch = tid < 128 && acc > 0.2f;
        if (acc > 0.2f) acc = 0.2f;
        if (__syncthreads_or(ch)) {                                   // (a barrier: every thread has read T.V)
            if (tid < 128) T.V[tid] = acc * acc;
            __syncthreads();
            const float n2 = 1.0f / sqrtf(sum_squares());
            acc = acc * n2;
        }
        // (int)(512.0*v) in double, MIN(255, .), NaN -> 0 (see the oracle's note)
        if (tid < 128) T.Q[tid] = min(255, (acc == acc) ? (int)(512.0 * (double)acc) : 0);
        __syncthreads();                                              // T.Q complete, T.V free
        if (wave == 0) store_record(sink, i, kq, T.Q[lane], T.Q[lane + 64], lane, reinterpret_cast<unsigned char *>(T.V));
        // the next keypoint's first barrier (1b) orders this store_record before anything rewrites T.V / T.Q
    }
}

// Hand-out order of the wave form: list order.  Largest windows first shortens the launch (a window has 17 to 75 batches of
// samples; in list order the longest wave runs 2.4x the mean) but never the frame: round 3's counting sort over 16 size classes
// (launch of the headline frame 336 -> 285 us alone) sat on the critical path; round 5 had the orientation launch append every
// keypoint to one of three per-scale lists for free and walked them scale 3 first -- launch alone 278 -> 250 us, whole call
// 0.789 / 0.795 ms with / without on the headline frame, +1.3 % on 4096^2 with every octave, +1.4 ... 3.5 % on keypoint-rich
// frames (neighbours in the list are neighbours in the image and share their window pixels in the caches).  The later
// octaves' chain ends the frame, and what slows that chain is the resources this launch holds, not how long it holds them.
//
// The launch: both forms share the grid (workgroups of four waves), the LDS block and the fold table; the count of the
// group decides -- fewer than `team_below` oriented keypoints: a workgroup per keypoint, else a wave per keypoint.
// Measured cross-over 1000-1800 keypoints (a 256-CU device holds 1024 workgroups of this kernel at once).
union DescLds {
    DescRowLds rows[4];
    DescTeamLds team;
    __device__ DescLds() {}
};

template <bool MAPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SIFT_DESC_WAVES, 8)))
void descriptor_kernel(OctaveTable tab, const float4 *__restrict__ okp, const int *__restrict__ oaux, Counters *cnt,
                       int group, int range_start, int range_end,   // range used when cnt == nullptr
                       int out_capacity, KpRecord *__restrict__ records, int rec_capacity, KpRecord *host_records, int host_capacity,
                       int team_below, int dynamic, int dense_blocks, int small_blocks) {
    __shared__ DescLds lds;
    __shared__ double fold[36];
    __shared__ RecordSink sink;
    int start = range_start, end = range_end;
    if (cnt) { start = 0; end = min(cnt->g_out[group], out_capacity); }
    descriptor_open(cnt, group, end, records, rec_capacity, host_records, host_capacity, &sink);   // (before any workgroup leaves)
    if (end - start < team_below) descriptor_team<MAPS>(tab, okp, oaux, start, end, &sink, lds.team, fold);
    else {
        // three workgroups per CU instead of four on a dense group: 154 k keypoints 5.56 -> 5.45 ms per call (the 9 k
        // keypoints of the headline frame prefer the full set: 0.903 against 0.927 ms)
        // ... and 576 on a group of fewer than 16384: such a group never is the whole frame's work, the later octaves' chain
        // runs beside it and ends the image (round 3, headline frame: 960 workgroups 0.857 ms, 576: 0.836, 448: 0.863; a
        // 39 k-keypoint group wants all 960: 1.55 against 1.72 ms at 640)
        const int count = end - start;
        const int nblocks = count >= 65536 ? min((int)gridDim.x, dense_blocks) : (count < 16384 ? min((int)gridDim.x, small_blocks) : (int)gridDim.x);
        if ((int)blockIdx.x >= nblocks) return;
        descriptor_waves<MAPS>(tab, okp, oaux, start, end, &sink, lds.rows, fold, (cnt && dynamic) ? &cnt->desc_next[group] : nullptr, nblocks);
    }
}

}  // namespace siftk
