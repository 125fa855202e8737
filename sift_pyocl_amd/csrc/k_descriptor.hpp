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
//     per row bisects the four flips on the exact float expression of u, a wave prefix sum turns the run lengths into
//     the raster rank of every in-window sample.  (The streaming form tests all (2R+1)^2 positions, half of which lie
//     outside the rotated window, and re-packs its sample list every 64 samples: a quarter of its time.)
//  2. EVALUATION.  64 consecutive ranks at a time, one sample per lane: gradient magnitude / orientation from
//     blur[scale] (image.cl:58-77), Gaussian weight, the eight (bin, value) contributions.  atan2 / exp go through the
//     Ziv fast paths of siftmath.hpp (bit-identical to the defining functions).
//  3. ORDERED ACCUMULATION.  Lane l owns bins l and l + 64.  Every sample lane sets its bit in the 64-bit "who
//     contributes" mask of each bin it touches (LDS atomic OR: order independent); a wave prefix sum of the per-bin
//     counts (DPP, no LDS round trip) gives every bin a 16-byte aligned segment of a value pool; each sample lane stores
//     each value at segment base + (number of lower lanes contributing to the same bin); the owner adds its segment
//     front to back (four values per LDS read): ascending lane == raster order.
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

struct alignas(16) DescRowLds {
    float pool[1024];                          // [0, 896): <= 512 values, every bin's segment zero-padded to a multiple of 4;
                                               // [896, 900): zeros, read by an owner past the end of its segment; [960 + lane]: dump slots
    float V[128];
    uint2 mask[128];                           // per bin: 64-bit mask of contributing lanes (x: lanes 0-31, y: lanes 32-63)
    unsigned mbase[128];                       // per bin: pool position of the first contributor
    int row_start[2 * SIFT_DESC_MAXRAD + 4];   // exclusive prefix of the per-row run lengths; [S] = total
    short row_jlo[2 * SIFT_DESC_MAXRAD + 4];   // first in-window jj of every row
};

// next representable float towards +inf (up) or -inf, x finite and non-zero
__device__ __forceinline__ float f32_neighbour(float x, bool up) {
    const int b = __float_as_int(x);
    return __int_as_float(((b >= 0) == up) ? b + 1 : b - 1);
}

// One sample of the descriptor window: everything keypoints_cpu.cl:74-117 does for it, except the additions.
// cbin[n] < 0: no contribution.  INTERIOR: the whole window lies at least one pixel inside the plane (no one-sided
// differences, image.cl:58-77).
template <bool INTERIOR>
__device__ __forceinline__ void descriptor_sample(const float *__restrict__ I, int W, int H, int x, int y, float rx, float cx,
                                                  float angle, const double *fold, int (&cbin)[8], float (&cval)[8]) {
    float gx, gy;
    const unsigned pos = (unsigned)y * (unsigned)W + (unsigned)x;     // planes hold <= 2^30 pixels (siftmi_plan_create)
    if (INTERIOR) {
        gx = I[pos + 1u] - I[pos - 1u];
        gy = I[pos - (unsigned)W] - I[pos + (unsigned)W];
    } else {
        const bool bx = (x == 0) || (x == W - 1), by = (y == 0) || (y == H - 1);
        gx = I[x == W - 1 ? pos : pos + 1u] - I[x == 0 ? pos : pos - 1u];
        gy = I[y == 0 ? pos : pos - (unsigned)W] - I[y == H - 1 ? pos : pos + (unsigned)W];
        if (bx) gx = 2.0f * gx;
        if (by) gy = 2.0f * gy;
    }
    const float g = sqrtf(gx * gx + gy * gy);
    float o = siftmath::atan2f_fast(-gy, gx, fold);
    const float er = rx - 1.5f, ec = cx - 1.5f;
    const float mag = g * siftmath::expf_fast(-0.125f * (er * er + ec * ec));
    o = o - angle;
    while (o > 2.0f * SM_PI_F) o -= 2.0f * SM_PI_F;
    while (o < 0.0f) o += 2.0f * SM_PI_F;
    const float oval = 4.0f * o * SM_1_PI_F;
    const int ri = (int)((rx >= 0.0f) ? rx : rx - 1.0f);
    const int ci = (int)((cx >= 0.0f) ? cx : cx - 1.0f);
    const int oi = (int)((oval >= 0.0f) ? oval : oval - 1.0f);
    const float rf = rx - (float)ri, cf = cx - (float)ci, of = oval - (float)oi;
    const bool contributes = ri >= -1 && ri < 4 && oi >= 0 && oi <= 8 && rf >= 0.0f && rf <= 1.0f;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const int rb = ri + a;
        const float rw = mag * (a == 0 ? 1.0f - rf : rf);
#pragma unroll
        for (int bb = 0; bb < 2; bb++) {
            const int cb = ci + bb;
            const float cw = rw * (bb == 0 ? 1.0f - cf : cf);
            const bool ok = contributes && rb >= 0 && rb < 4 && cb >= 0 && cb < 4;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                int ob = oi + e;
                // oi == 8 only for oval == 8.0f exactly (o == 2*pi_f), where of == 0: e = 0 adds cw*1 to bin 0,
                // e = 1 would add cw*0 == +0 to bin 0 again (no effect on a non-negative sum) -> skipped.
                const bool dup = (e == 1 && oi == 8);
                if (ob >= 8) ob = 0;
                const int n8 = a * 4 + bb * 2 + e;
                cbin[n8] = (ok && !dup) ? (rb * 4 + cb) * 8 + ob : -1;
                cval[n8] = cw * (e == 0 ? 1.0f - of : of);
            }
        }
    }
}

// `next`: device counter for dynamic hand-out (null: static stride).  Every wave takes keypoint `start + its index` first;
// after that it asks the counter, so that a wave with a small window does not idle while another still has two large
// ones to go (windows differ by 4x in samples within an octave).
__device__ __forceinline__ void descriptor_waves(const OctaveTable &tab, const float4 *__restrict__ okp, const int *__restrict__ oaux,
                                                 int start, int end, KpRecord *__restrict__ records, KpRecord *host_records,
                                                 int host_capacity, DescRowLds *lds_all, double *fold, int *next, int nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    DescRowLds &L = lds_all[wave];
    siftmath::load_atan_fold(fold);
    L.mask[lane] = make_uint2(0u, 0u); L.mask[lane + 64] = make_uint2(0u, 0u);
    if (lane < 4) L.pool[896 + lane] = 0.0f;
    __syncthreads();                     // the only workgroup barrier: the fold table
    const int gwave = blockIdx.x * 4 + wave, nwaves = nblocks * 4;

    auto advance = [&](int i) {
        if (!next) return i + nwaves;
        int t = 0;
        if (lane == 0) t = atomicAdd(next, 1);
        return start + nwaves + __builtin_amdgcn_readfirstlane(t);
    };
    for (int i = start + gwave; i < end; i = advance(i)) {
        const float4 kq = okp[i];        // (x, y, sigma*oct, angle)
        const int aux = oaux[i];         // detection scale | octave << 8
        const int scale = aux & 0xff, oct = aux >> 8;
        const int W = tab.W[oct], H = tab.H[oct], octsize = 1 << oct;
        KpRecord *rec = records + i;
        KpRecord *hrec = (host_records && i < host_capacity) ? host_records + i : nullptr;
        if (!(kq.y >= 0.0f)) {           // hole of an oriented list (stage replay only)
            store_record(rec, hrec, kq, 0, 0, lane, reinterpret_cast<unsigned char *>(L.V));
            continue;
        }
        const float *I = tab.base + tab.off[oct] + (size_t)scale * W * H;
        const float foct = (float)octsize;
        const float row = kq.y / foct, col = kq.x / foct, angle = kq.w;
        const int irow = (int)(row + 0.5f), icol = (int)(col + 0.5f);
        float sine, cosine;
        siftmath::sincosf_(angle, &sine, &cosine);
        const float spacing = kq.z / foct * 3.0f;
        const int R = (int)((1.414f * spacing * 2.5f) + 0.5f);
        const float drow = row - (float)irow, dcol = col - (float)icol;
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
        auto below_hi = [&](float u) { return thr_ok ? (u < t_hi) : (g(u) < 4.0f); };
        auto above_lo = [&](float u) { return thr_ok ? (u > t_lo) : (g(u) > -1.0f); };

        // ---- 1b. row intervals: lane l owns the window rows l, l + 64, ...
        const bool rdec = sine >= 0.0f;          // u_r non-increasing in jj
        const bool cdec = !(cosine >= 0.0f);     // u_c non-increasing in jj
        const int iters = 32 - __clz(S);         // 2R + 2 candidate positions
        int carry = 0;
        for (int r0 = 0; r0 < S; r0 += 64) {
            const int r = r0 + lane;
            const int ii = r - R;
            const float fi = (float)ii;
            const float ci_ = cosine * fi, si_ = sine * fi;
            int lo[4], hi[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { lo[q] = -R; hi[q] = R + 1; }
#pragma unroll 1
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int mid = (lo[q] + hi[q]) >> 1;
                    const float fj = (float)mid;
                    const float u = (q < 2) ? ((ci_ - sine * fj) - drow) : ((si_ + cosine * fj) - dcol);
                    const bool dec = (q < 2) ? rdec : cdec;
                    // q even: first jj inside the band; q odd: first jj beyond it (monotone false ... true predicates)
                    bool pred;
                    if ((q & 1) == 0) pred = dec ? below_hi(u) : above_lo(u);
                    else pred = dec ? !above_lo(u) : !below_hi(u);
                    if (lo[q] < hi[q]) { if (pred) hi[q] = mid; else lo[q] = mid + 1; }
                }
            }
            const int jlo = max(max(lo[0], lo[2]), max(-R, -icol));
            const int jhi = min(min(lo[1], lo[3]) - 1, min(R, W - 1 - icol));
            const int yy = irow + ii;
            int c = jhi - jlo + 1;
            if (r >= S || yy < 0 || yy >= H || c < 0) c = 0;
            const int incl = wave_prefix_incl(c);
            if (r < S) { L.row_start[r] = carry + incl - c; L.row_jlo[r] = (short)jlo; }
            carry += __builtin_amdgcn_readlane(incl, 63);
        }
        if (lane == 0) L.row_start[S] = carry;
        const int total = carry;
        __builtin_amdgcn_wave_barrier();

        // ---- 2 + 3. 64 ranks at a time
        const float rspacing = 1.0f / spacing;
        const bool fast_div = thr_ok && spacing >= 0.1f && spacing <= 128.0f;      // no under / overflow in div_by_reciprocal
        const bool interior = irow - R >= 1 && irow + R <= H - 2 && icol - R >= 1 && icol + R <= W - 2;   // wave uniform
        float4 *pool4 = reinterpret_cast<float4 *>(L.pool);
        float acc0 = 0.0f, acc1 = 0.0f;  // bins lane and lane + 64
        int rcur = 0;                    // row of this lane's current rank (ranks only grow)
        for (int s0 = 0; s0 < total; s0 += 64) {
            const int s = s0 + lane;
            int cbin[8];
            float cval[8];
#pragma unroll
            for (int n8 = 0; n8 < 8; n8++) { cbin[n8] = -1; cval[n8] = 0.0f; }
            if (s < total && !ABL(13)) {
                while (s >= L.row_start[rcur + 1]) rcur++;
                const int ii = rcur - R, jj = (int)L.row_jlo[rcur] + (s - L.row_start[rcur]);
                const float ur = (cosine * (float)ii - sine * (float)jj) - drow, uc = (sine * (float)ii + cosine * (float)jj) - dcol;
                const float rx = (fast_div ? siftmath::div_by_reciprocal(ur, spacing, rspacing) : ur / spacing) + 1.5f;
                const float cx = (fast_div ? siftmath::div_by_reciprocal(uc, spacing, rspacing) : uc / spacing) + 1.5f;
                if (ABL(12)) { cbin[0] = (ii * 7 + jj) & 127; cval[0] = rx + cx; }
                else if (interior) descriptor_sample<true>(I, W, H, icol + jj, irow + ii, rx, cx, angle, fold, cbin, cval);
                else descriptor_sample<false>(I, W, H, icol + jj, irow + ii, rx, cx, angle, fold, cbin, cval);
                if (ABL(11)) { acc0 += cval[0] + cval[7] + (float)cbin[3]; continue; }
            }
            if (ABL(11) || ABL(13)) continue;
            // ---- 3a. contributor masks
            {
                const unsigned bit = 1u << (lane & 31);
                unsigned *mw = reinterpret_cast<unsigned *>(L.mask) + (lane >> 5);
#pragma unroll
                for (int n8 = 0; n8 < 8; n8++)
                    if (cbin[n8] >= 0) atomicOr(mw + 2 * cbin[n8], bit);
            }
            __builtin_amdgcn_wave_barrier();
            // ---- 3b. bin owners: counts, 16-byte aligned pool segments from a wave prefix sum
            const uint2 ia = L.mask[lane], ib = L.mask[lane + 64];
            const int cnta = __popc(ia.x) + __popc(ia.y), cntb = __popc(ib.x) + __popc(ib.y);
            const int pa = (cnta + 3) & ~3, pb = (cntb + 3) & ~3;
            const int base_a = wave_prefix_incl(pa + pb) - (pa + pb);
            const int base_b = base_a + pa;
            L.mbase[lane] = (unsigned)base_a;
            L.mbase[lane + 64] = (unsigned)base_b;
            // the last group of four of every segment starts as +0: its padding then adds +0 (an exact no-op on these
            // non-negative sums), so the owners' loop needs no per-element masks
            if (cnta) pool4[(base_a + pa - 4) >> 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cntb) pool4[(base_b + pb - 4) >> 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            __builtin_amdgcn_wave_barrier();
            // ---- 3c. every value to segment base + rank among the contributors of its bin.  Branch free, so that the 16
            //          LDS reads are in flight together (exec-masked blocks would serialise them); a lane without a
            //          contribution reads bin 0 and writes to its own dump slot.
            if (!ABL(15)) {
                uint2 mk[8];
                unsigned mb[8];
#pragma unroll
                for (int n8 = 0; n8 < 8; n8++) {
                    const int b = max(cbin[n8], 0);
                    mk[n8] = L.mask[b];
                    mb[n8] = L.mbase[b];
                }
#pragma unroll
                for (int n8 = 0; n8 < 8; n8++) {
                    // mbcnt: number of set bits of the mask below this lane
                    const int pos = mb[n8] + __builtin_amdgcn_mbcnt_hi(mk[n8].y, __builtin_amdgcn_mbcnt_lo(mk[n8].x, 0u));
                    L.pool[(cbin[n8] >= 0) ? pos : 960 + lane] = cval[n8];
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- 3d. ordered sums, four values per read, the next read in flight while the current four are added; past
            //          the end of its own segment a lane reads the four zeros at [896]
            const int nmax = ABL(14) ? 0 : max(pa, pb);
            const int qa = base_a >> 2, qb = base_b >> 2, ea = pa >> 2, eb = pb >> 2;
            float4 va = pool4[ea ? qa : 224], vb = pool4[eb ? qb : 224];
            for (int g4 = 0; g4 < (nmax >> 2); g4++) {
                const float4 ca = va, cb = vb;
                va = pool4[(g4 + 1 < ea) ? qa + g4 + 1 : 224];
                vb = pool4[(g4 + 1 < eb) ? qb + g4 + 1 : 224];
                acc0 = acc0 + ca.x; acc1 = acc1 + cb.x;
                acc0 = acc0 + ca.y; acc1 = acc1 + cb.y;
                acc0 = acc0 + ca.z; acc1 = acc1 + cb.z;
                acc0 = acc0 + ca.w; acc1 = acc1 + cb.w;
            }
            __builtin_amdgcn_wave_barrier();
            if (cnta) L.mask[lane] = make_uint2(0u, 0u);
            if (cntb) L.mask[lane + 64] = make_uint2(0u, 0u);
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
                for (int u = 0; u < 8; u++) q[u] = reinterpret_cast<const float4 *>(L.V)[8 * k8 + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { t = t + q[u].x; t = t + q[u].y; t = t + q[u].z; t = t + q[u].w; }
            }
            return t;
        };
        L.V[lane] = acc0 * acc0; L.V[lane + 64] = acc1 * acc1;
        __builtin_amdgcn_wave_barrier();
        float norm = ABL(16) ? L.V[5] : 1.0f / sqrtf(sum_squares());       // rsqrt
        acc0 = acc0 * norm; acc1 = acc1 * norm;
        const bool ch = (acc0 > 0.2f) || (acc1 > 0.2f);
        if (acc0 > 0.2f) acc0 = 0.2f;
        if (acc1 > 0.2f) acc1 = 0.2f;
        __builtin_amdgcn_wave_barrier();
        if (__ballot(ch)) {
            L.V[lane] = acc0 * acc0; L.V[lane + 64] = acc1 * acc1;
            __builtin_amdgcn_wave_barrier();
            const float n2 = 1.0f / sqrtf(sum_squares());
            acc0 = acc0 * n2; acc1 = acc1 * n2;
        }
        __builtin_amdgcn_wave_barrier();
        // (int)(512.0*v) in double, MIN(255, .), NaN -> 0 (see the oracle's note)
        const int i0 = (acc0 == acc0) ? (int)(512.0 * (double)acc0) : 0;
        const int i1 = (acc1 == acc1) ? (int)(512.0 * (double)acc1) : 0;
        store_record(rec, hrec, kq, min(255, i0), min(255, i1), lane, reinterpret_cast<unsigned char *>(L.V));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same descriptor, ONE WORKGROUP (four waves) per keypoint: for sparse groups.  With a wave per keypoint a launch
// lasts as long as its slowest keypoint (60-120 us: ~40-75 batches of 64 samples, one after the other, on a SIMD that
// has nothing else to issue), however few keypoints there are.  Here the four waves evaluate four consecutive batches
// at once, each into its own mask / pool area, and after a workgroup barrier the bin owners (threads 0-127, one bin
// each) add the four areas in batch order -- the same additions in the same order, so the same bits.  One launch holds
// both forms; the group's count, known on the device only, picks one (descriptor_kernel, team_below).
struct alignas(16) DescTeamWaveLds {
    float pool[1024];                          // as DescRowLds::pool
    uint2 mask[128];
    unsigned mbase[128];
};
struct alignas(16) DescTeamLds {
    DescTeamWaveLds w[4];
    float V[128];
    int Q[128];
    int row_start[2 * SIFT_DESC_MAXRAD + 4];
    short row_jlo[2 * SIFT_DESC_MAXRAD + 4];
    int blk_total[4];
};

__device__ __forceinline__ void descriptor_team(const OctaveTable &tab, const float4 *__restrict__ okp, const int *__restrict__ oaux,
                                                int start, int end, KpRecord *__restrict__ records, KpRecord *host_records,
                                                int host_capacity, DescTeamLds &T, double *fold) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DescTeamWaveLds &L = T.w[wave];
    siftmath::load_atan_fold(fold);
    L.mask[lane] = make_uint2(0u, 0u); L.mask[lane + 64] = make_uint2(0u, 0u);
    if (lane < 4) L.pool[896 + lane] = 0.0f;
    __syncthreads();

    for (int i = start + blockIdx.x; i < end; i += gridDim.x) {      // workgroup uniform
        const float4 kq = okp[i];        // (x, y, sigma*oct, angle)
        const int aux = oaux[i];         // detection scale | octave << 8
        const int scale = aux & 0xff, oct = aux >> 8;
        const int W = tab.W[oct], H = tab.H[oct], octsize = 1 << oct;
        KpRecord *rec = records + i;
        KpRecord *hrec = (host_records && i < host_capacity) ? host_records + i : nullptr;
        if (!(kq.y >= 0.0f)) {           // hole of an oriented list (stage replay only)
            __syncthreads();             // the previous keypoint's record may still be leaving through T.V
            if (wave == 0) store_record(rec, hrec, kq, 0, 0, lane, reinterpret_cast<unsigned char *>(T.V));
            continue;
        }
        const float *I = tab.base + tab.off[oct] + (size_t)scale * W * H;
        const float foct = (float)octsize;
        const float row = kq.y / foct, col = kq.x / foct, angle = kq.w;
        const int irow = (int)(row + 0.5f), icol = (int)(col + 0.5f);
        float sine, cosine;
        siftmath::sincosf_(angle, &sine, &cosine);
        const float spacing = kq.z / foct * 3.0f;
        const int R = (int)((1.414f * spacing * 2.5f) + 0.5f);
        const float drow = row - (float)irow, dcol = col - (float)icol;
        const int S = 2 * R + 1;
        if (R > SIFT_DESC_MAXRAD) __builtin_trap();   // the host launches descriptor_stream_kernel for such plans

        // ---- 1a. thresholds (every wave for itself: 64 candidates in one ballot), as in descriptor_kernel
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
        auto below_hi = [&](float u) { return thr_ok ? (u < t_hi) : (g(u) < 4.0f); };
        auto above_lo = [&](float u) { return thr_ok ? (u > t_lo) : (g(u) > -1.0f); };

        // ---- 1b. row intervals: thread t owns window row t (S <= 255)
        {
            const bool rdec = sine >= 0.0f, cdec = !(cosine >= 0.0f);
            const int iters = 32 - __clz(S);
            const int r = tid;
            const int ii = r - R;
            const float fi = (float)ii;
            const float ci_ = cosine * fi, si_ = sine * fi;
            int lo[4], hi[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { lo[q] = -R; hi[q] = R + 1; }
            if (64 * wave < S) {                                      // wave uniform
#pragma unroll 1
                for (int it = 0; it < iters; it++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int mid = (lo[q] + hi[q]) >> 1;
                        const float fj = (float)mid;
                        const float u = (q < 2) ? ((ci_ - sine * fj) - drow) : ((si_ + cosine * fj) - dcol);
                        const bool dec = (q < 2) ? rdec : cdec;
                        bool pred;
                        if ((q & 1) == 0) pred = dec ? below_hi(u) : above_lo(u);
                        else pred = dec ? !above_lo(u) : !below_hi(u);
                        if (lo[q] < hi[q]) { if (pred) hi[q] = mid; else lo[q] = mid + 1; }
                    }
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
        const float rspacing = 1.0f / spacing;
        const bool fast_div = thr_ok && spacing >= 0.1f && spacing <= 128.0f;
        const bool interior = irow - R >= 1 && irow + R <= H - 2 && icol - R >= 1 && icol + R <= W - 2;
        float4 *pool4 = reinterpret_cast<float4 *>(L.pool);
        float acc = 0.0f;                // bin tid (threads 0-127)
        int rcur = 0;
        for (int t0 = 0; t0 < total; t0 += 256) {                     // workgroup uniform
            const int s = t0 + 64 * wave + lane;
            if (t0 + 64 * wave < total) {                             // wave uniform: this wave has a batch
                int cbin[8];
                float cval[8];
#pragma unroll
                for (int n8 = 0; n8 < 8; n8++) { cbin[n8] = -1; cval[n8] = 0.0f; }
                if (s < total) {
                    while (s >= T.row_start[rcur + 1]) rcur++;
                    const int ii = rcur - R, jj = (int)T.row_jlo[rcur] + (s - T.row_start[rcur]);
                    const float ur = (cosine * (float)ii - sine * (float)jj) - drow, uc = (sine * (float)ii + cosine * (float)jj) - dcol;
                    const float rx = (fast_div ? siftmath::div_by_reciprocal(ur, spacing, rspacing) : ur / spacing) + 1.5f;
                    const float cx = (fast_div ? siftmath::div_by_reciprocal(uc, spacing, rspacing) : uc / spacing) + 1.5f;
                    if (interior) descriptor_sample<true>(I, W, H, icol + jj, irow + ii, rx, cx, angle, fold, cbin, cval);
                    else descriptor_sample<false>(I, W, H, icol + jj, irow + ii, rx, cx, angle, fold, cbin, cval);
                }
                // 3a. contributor masks of this wave's batch
                {
                    const unsigned bit = 1u << (lane & 31);
                    unsigned *mw = reinterpret_cast<unsigned *>(L.mask) + (lane >> 5);
#pragma unroll
                    for (int n8 = 0; n8 < 8; n8++)
                        if (cbin[n8] >= 0) atomicOr(mw + 2 * cbin[n8], bit);
                }
                __builtin_amdgcn_wave_barrier();
                // 3b. segments of this wave's pool
                const uint2 ia = L.mask[lane], ib = L.mask[lane + 64];
                const int cnta = __popc(ia.x) + __popc(ia.y), cntb = __popc(ib.x) + __popc(ib.y);
                const int pa = (cnta + 3) & ~3, pb = (cntb + 3) & ~3;
                const int base_a = wave_prefix_incl(pa + pb) - (pa + pb);
                const int base_b = base_a + pa;
                L.mbase[lane] = (unsigned)base_a;
                L.mbase[lane + 64] = (unsigned)base_b;
                if (cnta) pool4[(base_a + pa - 4) >> 2] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cntb) pool4[(base_b + pb - 4) >> 2] = make_float4(0.f, 0.f, 0.f, 0.f);
                __builtin_amdgcn_wave_barrier();
                // 3c. values to their ranks (branch free, see descriptor_kernel)
                uint2 mk[8];
                unsigned mb[8];
#pragma unroll
                for (int n8 = 0; n8 < 8; n8++) {
                    const int b = max(cbin[n8], 0);
                    mk[n8] = L.mask[b];
                    mb[n8] = L.mbase[b];
                }
#pragma unroll
                for (int n8 = 0; n8 < 8; n8++) {
                    const int pos = mb[n8] + __builtin_amdgcn_mbcnt_hi(mk[n8].y, __builtin_amdgcn_mbcnt_lo(mk[n8].x, 0u));
                    L.pool[(cbin[n8] >= 0) ? pos : 960 + lane] = cval[n8];
                }
            }
            __syncthreads();
            // 3d. owners: bin tid, the four areas in batch order (an area without a batch this round has empty masks)
            if (tid < 128) {
                uint2 m[4];
                unsigned mb[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { m[q] = T.w[q].mask[tid]; mb[q] = T.w[q].mbase[tid]; }
                int n4[4];
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    n4[q] = (__popc(m[q].x) + __popc(m[q].y) + 3) >> 2;
                    v[q] = reinterpret_cast<const float4 *>(T.w[q].pool)[n4[q] ? (mb[q] >> 2) : 224];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (n4[q]) {
                        float4 cur = v[q];
                        for (int g4 = 0; g4 < n4[q]; g4++) {
                            const float4 c4 = cur;
                            if (g4 + 1 < n4[q]) cur = reinterpret_cast<const float4 *>(T.w[q].pool)[(mb[q] >> 2) + g4 + 1];
                            acc = acc + c4.x; acc = acc + c4.y; acc = acc + c4.z; acc = acc + c4.w;
                        }
                        T.w[q].mask[tid] = make_uint2(0u, 0u);
                    }
                }
            }
            __syncthreads();
        }

        // ---- 4. normalise, clamp at 0.2, renormalise, quantise (keypoints_cpu.cl:125-160), as in descriptor_kernel
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
        if (wave == 0) store_record(rec, hrec, kq, T.Q[lane], T.Q[lane + 64], lane, reinterpret_cast<unsigned char *>(T.V));
        // the next keypoint's first barrier (1b) orders this store_record before anything rewrites T.V / T.Q
    }
}

// The launch: both forms share the grid (workgroups of four waves), the LDS block and the fold table; the count of the
// group decides -- fewer than `team_below` oriented keypoints: a workgroup per keypoint, else a wave per keypoint.
// Measured cross-over 1000-1800 keypoints (a 256-CU device holds 1024 workgroups of this kernel at once).
union DescLds {
    DescRowLds rows[4];
    DescTeamLds team;
    __device__ DescLds() {}
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SIFT_DESC_WAVES, 8)))
void descriptor_kernel(OctaveTable tab, const float4 *__restrict__ okp, const int *__restrict__ oaux, const Counters *cnt,
                       int group, int range_start, int range_end,   // range used when cnt == nullptr
                       int out_capacity, KpRecord *__restrict__ records, KpRecord *host_records, int host_capacity,
                       int team_below, int dynamic, int dense_blocks) {
    __shared__ DescLds lds;
    __shared__ double fold[36];
    int start = range_start, end = range_end;
    if (cnt) { start = cnt->grp_out_start[group]; end = min(cnt->grp_out_end[group], out_capacity); }
    if (end - start < team_below) descriptor_team(tab, okp, oaux, start, end, records, host_records, host_capacity, lds.team, fold);
    else {
        // three workgroups per CU instead of four on a dense group: 154 k keypoints 5.56 -> 5.45 ms per call (the 9 k
        // keypoints of the headline frame prefer the full set: 0.903 against 0.927 ms)
        const int nblocks = (end - start >= 65536) ? min((int)gridDim.x, dense_blocks) : (int)gridDim.x;
        if ((int)blockIdx.x >= nblocks) return;
        descriptor_waves(tab, okp, oaux, start, end, records, host_records, host_capacity, lds.rows, fold,
                         (cnt && dynamic) ? const_cast<int *>(&cnt->desc_next[group]) : nullptr, nblocks);
    }
}

}  // namespace siftk
