// k_keypoint.hpp -- per-keypoint kernels: orientation assignment and the 128-D descriptor.
//
// Parity target is the reference's CPU variants (orientation_cpu.cl:41-174, keypoints_cpu.cl:36-161)
// because those are what its "CPU devicetype" path runs.  Both accumulate float histograms in the
// raster order of the sample window, and float addition is not associative, so bit-exact bins need
// the same per-bin order.  Here one wavefront owns one keypoint: its 64 lanes evaluate 64 consecutive
// samples of the raster scan at a time, and the histogram bins are owned by fixed lanes that add the
// chunk's contributions in lane (= raster) order.
//
// Gradient magnitude / orientation (compute_gradient_orientation, image.cl:47-80) are not
// materialised as full maps: they are evaluated from blur[scale] inside the window with the same
// expressions, which yields the same values and saves 9 plane transfers per octave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "k_extrema.hpp"
#include "siftmath.hpp"

namespace siftk {

struct KpRecord { float x, y, scale, angle; uint8_t desc[128]; };   // == siftmi_keypoint
static_assert(sizeof(KpRecord) == 144, "record layout");

// image.cl:58-77, split in two so that the four pixel loads of the NEXT batch of samples can be in flight while the
// current batch is evaluated (the per-keypoint kernels spent half their time in s_waitcnt on exactly these loads).
// gx = sx * (xa - xb), gy = sy * (ya - yb) with sx, sy = 2 on the image border (one-sided difference), else 1.
struct GradTaps { float xa, xb, ya, yb; bool bx, by; };

__device__ __forceinline__ GradTaps gradient_fetch(const float *__restrict__ I, int x, int y, int W, int H) {
    const size_t pos = (size_t)y * W + x;
    GradTaps t;
    t.bx = (x == 0) || (x == W - 1);
    t.by = (y == 0) || (y == H - 1);
    t.xa = I[x == W - 1 ? pos : pos + 1];
    t.xb = I[x == 0 ? pos : pos - 1];
    t.ya = I[y == 0 ? pos : pos - W];
    t.yb = I[y == H - 1 ? pos : pos + W];
    return t;
}

__device__ __forceinline__ void gradient_eval(const GradTaps &t, float &mag, float &ori) {
    float gx = t.xa - t.xb, gy = t.ya - t.yb;
    if (t.bx) gx = 2.0f * gx;
    if (t.by) gy = 2.0f * gy;
    mag = sqrtf(gx * gx + gy * gy);
    ori = siftmath::atan2f_(-gy, gx);
}

__device__ __forceinline__ void gradient_at(const float *__restrict__ I, int x, int y, int W, int H,
                                            float &mag, float &ori) {
    gradient_eval(gradient_fetch(I, x, y, W, H), mag, ori);
}

// exact idx / d for 0 <= idx < 2^22, 1 <= d < 2^12 (inv = 1.0f / d)
__device__ __forceinline__ int div_exact(int idx, int d, float inv, int &rem) {
    int q = (int)((float)idx * inv);
    rem = idx - q * d;
    if (rem < 0) { q--; rem += d; }
    else if (rem >= d) { q++; rem -= d; }
    return q;
}

// Device-side bookkeeping.  The octaves of an image form up to three GROUPS, each with its own refined and oriented lists
// and counters, so that nothing has to be frozen or ordered between them: the groups run on different streams and meet
// only in the record list, where each reserves ONE block when its descriptor launch starts (descriptor_reserve below).
//   group 0: octave 0 -- or, on a large frame whose octave 0 is split by scale (siftmi.hip: Options::split0), its detection
//            scale 1, which needs planes 0-3 only and starts while the last two blurs of the octave still run;
//   group 1: detection scales 2 and 3 of a split octave 0 (unused otherwise);
//   group 2: all later octaves (their few keypoints would otherwise pay one latency-bound launch chain per octave).
#define SIFT_MAX_OCTAVES 24
#define SIFT_GROUPS 3
struct Counters {
    int n_rec;                          // record slots reserved so far (descriptor launches, one block per group)
    int tail_timeout;                   // octave_tail_kernel: a workgroup stopped waiting for the octave above (k_tail.hpp)
    unsigned rec_word[SIFT_GROUPS];     // 0x80000000 | first record of the group's block, once the block is reserved (0 before)
    int g_kp[SIFT_GROUPS];              // refined keypoints appended to the group's list (may exceed its capacity: the host grows the list)
    int g_out[SIFT_GROUPS];             // oriented keypoints appended to the group's list (likewise)
    int desc_next[SIFT_GROUPS];         // descriptor_kernel: keypoints of the group handed out beyond every wave's first one
    int n_cand[SIFT_MAX_OCTAVES + 1];   // candidates per octave ([SIFT_MAX_OCTAVES]: scales 2-3 of a split octave 0)
    // What the reference's per-octave keypoint counter would have held (plan.py:626-731, one counter per octave, reset by
    // _reset_keypoints): candidates that passed local_maxmin, and oriented keypoints (NaN rows included: the reference drops
    // them on the host), per octave and detection scale.  The host evaluates the reference's capacity rule from these.
    int c_scale[SIFT_MAX_OCTAVES][3];
    int o_scale[SIFT_MAX_OCTAVES][3];
    uint32_t mm[2];                     // order-encoded min / max of the input (k_pyramid.hpp), read back with the counters
    int tail_ready[8];                  // octave_tail_kernel: plane 3 of tail octave k is in HBM (k_tail.hpp)
};

// where the six planes of every octave live: plane(o, s) = base + off[o] + s * W[o] * H[o]
struct OctaveTable {
    const float *base;
    long long off[SIFT_MAX_OCTAVES];
    int W[SIFT_MAX_OCTAVES], H[SIFT_MAX_OCTAVES];
    // Full gradient maps of the detection scales (planes 1..3) of every octave, for the MAPS forms of the per-keypoint
    // kernels: magnitude / orientation of (octave o, plane s) start at off[o] / 2 + (s - 1) * W * H (three planes per
    // octave where the pyramid has six).  Written by gradient_maps_kernel when the host expects a keypoint-rich frame.
    const float *gmap, *omap;
};
__device__ __forceinline__ size_t map_offset(const OctaveTable &tab, int oct, int scale) {
    return (size_t)(tab.off[oct] / 2) + (size_t)(scale - 1) * tab.W[oct] * tab.H[oct];
}

// Where the records of a group go: filled once per workgroup of a descriptor launch
// (descriptor_open) and kept in LDS -- a record's addresses are formed from here, in vector registers, at the moment the
// record leaves (nothing of it lives in the scalar registers a descriptor needs for its window).
// Keypoint i of the group's oriented list is record i of the group's block; a record beyond a list's capacity is not
// written (the host sees the counts, grows the list and runs the image again; the caller's array may be smaller than the
// device list: it then fetches from the device).
struct alignas(16) RecordSink {
    KpRecord *dev, *host;          // the group's block in the device list / in the caller's pinned array (or null)
    int dev_limit, host_limit;     // keypoints of the group that fit each
};

// The record block of a group: workgroup 0 of the group's descriptor launch reserves [base, base + n) of the image's
// record list with ONE atomic and publishes the base; the other workgroups wait for it (workgroup 0 is dispatched first and
// waits for nobody; they poll at ~1 us intervals -- a tight poll from a thousand workgroups held the publishing store up
// for tens of microseconds).  Groups on different streams thus share one compact record list -- and the caller's pinned
// result array -- without an event between them and without an atomic per keypoint.
// Call with every thread of the workgroup (it ends with a workgroup barrier); cnt == null: no block, base 0.
__device__ __forceinline__ void descriptor_open(Counters *cnt, int group, int n, KpRecord *records, int rec_capacity, KpRecord *host_records,
                                                int host_capacity, RecordSink *S) {
    if (threadIdx.x == 0) {
        int base = 0;
        if (cnt) {
            // (flag and base travel in ONE word, written and read by relaxed device-scope atomics: nothing else has to be ordered,
            // so no release / acquire -- on this chip each of those writes back or invalidates the L2 of an XCD)
            if (blockIdx.x == 0) {
                base = atomicAdd(&cnt->n_rec, n);
                __hip_atomic_store(&cnt->rec_word[group], 0x80000000u | (unsigned)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                unsigned w;
                while (!((w = __hip_atomic_load(&cnt->rec_word[group], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x80000000u)) __builtin_amdgcn_s_sleep(16);
                base = (int)(w & 0x7fffffffu);
            }
        }
        S->dev = records + base; S->dev_limit = rec_capacity - base;
        S->host = host_records ? host_records + base : nullptr; S->host_limit = host_records ? host_capacity - base : 0;
    }
    __syncthreads();
}

// The 144-byte record of keypoint i of the group leaves the wave as 36 coalesced dwords (lanes 0-3: x, y, scale, angle;
// lanes 4-35: four descriptor bytes each, packed through `bytes`, 128 bytes of the wave's LDS) -- to the device list and,
// when the call has a pinned result array, also straight into it (zero-copy over PCIe: no device-to-host copy after the
// last kernel).
__device__ __forceinline__ void store_record(const RecordSink *S, int i, const float4 kq, int b0, int b1, int lane, unsigned char *bytes) {
    bytes[lane] = (unsigned char)b0; bytes[lane + 64] = (unsigned char)b1;
    __builtin_amdgcn_wave_barrier();
    unsigned w = 0u;
    if (lane < 4) w = __float_as_uint(lane == 0 ? kq.x : (lane == 1 ? kq.y : (lane == 2 ? kq.z : kq.w)));
    else if (lane < 36) w = reinterpret_cast<const unsigned *>(bytes)[lane - 4];
    if (lane < 36) {
        const volatile RecordSink *v = S;
        if (i < v->dev_limit) reinterpret_cast<unsigned *>(v->dev + i)[lane] = w;
        if (i < v->host_limit) reinterpret_cast<unsigned *>(v->host + i)[lane] = w;
    }
    __builtin_amdgcn_wave_barrier();
}

// records[idx[j]] -> out[j], 36 dwords per record (siftmi.hip: cap_octaves, the rare path of an image beyond the
// reference's per-octave capacity)
__global__ void gather_records_kernel(const KpRecord *__restrict__ records, const int *__restrict__ idx, KpRecord *__restrict__ out, int n) {
    const long long total = (long long)n * 36;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(t / 36), w = (int)(t - (long long)j * 36);
        reinterpret_cast<unsigned *>(out + j)[w] = reinterpret_cast<const unsigned *>(records + idx[j])[w];
    }
}

// DEVELOPMENT INSTRUMENT (-DSIFT_PHASE_CLOCK, tools/dev/phase_clock.py): where a wave of the per-keypoint kernels spends
// its time.  Lane 0 adds the shader-clock cycles since the previous mark to the phase's accumulator in the wave's LDS;
// reading the clock waits for the wave's outstanding LDS / scalar-memory operations, so a phase is charged with the
// latencies it started.  Accumulators [0, 12): cycles per phase, [12]: wave time, [13]: keypoints, [14]: batches; the
// waves add theirs to g_phase (+16 per kernel: 0 orientation, 16 descriptor) and keep the longest wave in [15].
#ifdef SIFT_PHASE_CLOCK
__device__ unsigned long long g_phase[32];
struct PhaseClock {
    unsigned long long t, t0;
    unsigned long long *acc;
    __device__ __forceinline__ void start(unsigned long long *lds16, int lane) {
        acc = lds16;
        if (lane < 16) lds16[lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        t0 = t = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int k, int lane) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        // (an LDS add without return: nothing of the instrument is waited for but the clock itself)
        if (lane == 0) (void)__hip_atomic_fetch_add(acc + k, now - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = now;
    }
    __device__ __forceinline__ void count(int k, int lane) { if (lane == 0) (void)__hip_atomic_fetch_add(acc + k, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __device__ __forceinline__ void flush(int base, int lane) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (lane == 0) acc[12] = now - t0;
        __builtin_amdgcn_wave_barrier();
        if (lane < 15) atomicAdd(&g_phase[base + lane], acc[lane]);
        if (lane == 15) atomicMax(&g_phase[base + 15], acc[12]);
    }
};
#define PH_MARK(k) ph.mark((k), lane)
#define PH_COUNT(k) ph.count((k), lane)
#define PH_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#else
#define PH_MARK(k)
#define PH_COUNT(k)
#define PH_WAIT_VM(n)
#endif

#define ABL(x) false

// ------------------------------------------------------------------------------------------
// Orientation assignment: one wave per refined keypoint (orientation_cpu.cl:41-174).
// Output goes straight to the image-wide oriented list (x, y, sigma*oct, angle) + detection scale.
//
// The reference adds the window samples into hist[bin] in raster order.  64 consecutive samples at a time, one per lane:
// every voting lane sets its bit in its bin's 64-bit mask (LDS atomic OR, order independent); lane b < 36 owns hist[b]
// and gets a 16-byte aligned segment of a small value pool from a wave prefix sum of the vote counts; every voter stores
// its value at segment base + (number of lower lanes voting for the same bin); the owner adds its segment front to back
// (ascending lane == raster order), four values per LDS read.  atan2 / exp: Ziv fast paths of siftmath.hpp.
#define SIFT_ORI_OBUF 96          // a keypoint yields at most 36 entries (1 + 35 further peaks)
struct alignas(16) OriWaveLds {
    float pool[64 + 36 * 3 + 4];   // 64 values, every bin's segment padded to a multiple of 4
    uint2 mask[36];
    unsigned mbase[36];
    float4 obuf[SIFT_ORI_OBUF];    // oriented keypoints waiting for their slots in the global list
    int oaux[SIFT_ORI_OBUF];
#ifdef SIFT_PHASE_CLOCK
    unsigned long long ph[16];
#endif
};

__device__ __forceinline__ int wave_prefix_incl(int x) {      // inclusive prefix sum over the 64 lanes (DPP, no LDS)
    int t = x;
    t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, false);   // row_shr:1
    t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, false);   // row_shr:2
    t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, false);   // row_shr:4
    t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xf, false);   // row_shr:8
    t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
    t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
    return t;
}

// Slots in the global oriented list come from ONE device-scope counter.  Same-address atomics are served one at a time
// by the L2 (measured: 5.2 ns each), so an atomicAdd per keypoint made the counter the bottleneck of the kernel (half of
// its time at 120 k keypoints).  A wave therefore parks its results in LDS and reserves slots for many keypoints at once;
// at the end the four waves of a workgroup share a single atomicAdd.
// MAPS: gradient magnitude / orientation come from the full maps (tab.gmap / tab.omap, same values: the same functions
// wrote them) instead of four pixel loads, a square root and an arc tangent per sample -- a window sample of a dense frame
// is evaluated by a dozen keypoints.
template <bool MAPS>
__global__ __launch_bounds__(256) void orientation_kernel(OctaveTable tab, float ori_sigma,
                                                          const float4 *__restrict__ kp,
                                                          const int *__restrict__ kp_aux, Counters *cnt, int group,
                                                          int kp_capacity, float4 *__restrict__ okp,
                                                          int *__restrict__ oaux, int out_capacity, int team_below, int small_blocks) {
    __shared__ OriWaveLds lds_all[4];
    __shared__ double fold[36];
    __shared__ int s_hist[3 * SIFT_MAX_OCTAVES];          // oriented keypoints per (octave, detection scale): Counters::o_scale
    const int lane = threadIdx.x & 63;
    OriWaveLds &L = lds_all[threadIdx.x >> 6];
    // the group's refined keypoints: its list is complete (the refinement launches precede this one on the stream)
    const int n = min(cnt->g_kp[group], kp_capacity);
    const int first = 0;
    // The launch is sized for a dense group (4096 workgroups: finer strides balance better, 154 k-keypoint frame -3.6 %);
    // the count, known here only, cuts it down for smaller groups (the rest of the chip is busy with the later octaves'
    // pyramid at that point: 9 k keypoints on 512 workgroups 0.893 ms per call, on 1024 0.907).  A workgroup beyond the
    // cut has nothing parked and nothing to wait for.
    const int count = n - first;
    const int nblocks = min((int)gridDim.x, count < 16384 ? small_blocks : (count < 65536 ? max(1024, small_blocks) : 4096));
    if ((int)blockIdx.x >= nblocks) return;
#ifdef SIFT_PHASE_CLOCK
    PhaseClock ph;
    ph.start(L.ph, lane);
#endif
    siftmath::load_atan_fold(fold);
    if (lane < 36) L.mask[lane] = make_uint2(0u, 0u);
    if (threadIdx.x < 3 * SIFT_MAX_OCTAVES) s_hist[threadIdx.x] = 0;
    __syncthreads();
    PH_MARK(0);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (nblocks * blockDim.x) >> 6;
    const float4 *pool4 = reinterpret_cast<const float4 *>(L.pool);
    __shared__ int s_pending[4], s_base;
    int pending = 0;                     // entries parked in L.obuf (wave uniform)
    auto store_pending = [&](int slot, int count) {
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < count; e += 64)
            if (slot + e < out_capacity) { okp[slot + e] = L.obuf[e]; oaux[slot + e] = L.oaux[e]; }       // (a cut list shows in the counter: the host grows it)
        __builtin_amdgcn_wave_barrier();
    };
    auto flush_wave = [&]() {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&cnt->g_out[group], pending);
        store_pending(__shfl(slot, 0), pending);
        pending = 0;
    };
    // Sparse groups (fewer than team_below keypoints): the four waves of a workgroup take ONE keypoint, each evaluates
    // every fourth batch of 64 window samples into its own vote masks / pool, and after a workgroup barrier the bin
    // owners (wave 0, lanes 0-35) add the four pools in batch order -- the same additions in the same order as with a
    // wave per keypoint, at a quarter of the latency (the launch lasts as long as its slowest keypoint).
    const bool team = count < team_below;                 // workgroup uniform
    const int w4 = threadIdx.x >> 6;
    const int boff = team ? 64 * w4 : 0, bstep = team ? 256 : 64;
    for (int i = first + (team ? (int)blockIdx.x : wave); i < n; i += (team ? nblocks : nwaves)) {
        const float4 k = kp[i];          // (peak, row, col, sigma)
        const int aux = __builtin_amdgcn_readfirstlane(kp_aux[i]);       // detection scale | octave << 8 (the keypoint is the same in every lane)
        const int scale = aux & 0xff, oct = aux >> 8;
        const int W = tab.W[oct], H = tab.H[oct], octsize = 1 << oct;
        if (!(k.y >= 0.0f)) continue;
        const float *I = tab.base + tab.off[oct] + (size_t)scale * W * H;
        const float *Gm = MAPS ? tab.gmap + map_offset(tab, oct, scale) : nullptr;
        const float *Om = MAPS ? tab.omap + map_offset(tab, oct, scale) : nullptr;
        auto taps_at = [&](int x, int y) {
            if (!MAPS) return gradient_fetch(I, x, y, W, H);
            GradTaps t = {};
            const size_t pos = (size_t)y * W + x;
            t.xa = Gm[pos]; t.xb = Om[pos];          // (magnitude, orientation)
            return t;
        };
        const int row = (int)((double)k.y + 0.5), col = (int)((double)k.z + 0.5);
        const float sigma = ori_sigma * k.w;
        const int radius = (int)((double)sigma * 3.0);
        const int rmin = max(0, row - radius), cmin = max(0, col - radius);
        const int rmax = min(row + radius, H - 2), cmax = min(col + radius, W - 2);
        const float lim = (float)(radius * radius) + 0.5f;
        const float two_s2 = 2.0f * sigma * sigma;
        // the two divisions per sample (orientation_cpu.cl:88-90) go through siftmath::div_by_reciprocal (same quotient)
        const float r_two_s2 = 1.0f / two_s2;
        const bool fast_div = two_s2 >= 1e-3f && two_s2 <= 1e6f;
        const int wc = cmax - cmin + 1, hr = rmax - rmin + 1;
        const int total = (wc > 0 && hr > 0) ? wc * hr : 0;
        const float inv_wc = 1.0f / (float)max(wc, 1);
        float h = 0.0f;                  // lane b < 36 owns hist[b]
        // sample position of this lane in the batch starting at `base`
        auto locate = [&](int base, int &r, int &c) {
            const int idx = base + lane;
            if (idx >= total) return false;
            int rem;
            const int q = div_exact(idx, wc, inv_wc, rem);
            r = rmin + q; c = cmin + rem;
            return true;
        };
        int nr = 0, nc = 0;
        bool nvalid = locate(boff, nr, nc);
        GradTaps ntaps = {};
        if (nvalid) ntaps = taps_at(nc, nr);   // the loads of batch b+1 are issued before batch b is evaluated
        PH_COUNT(13);
        PH_MARK(1);
        for (int base = 0; base < total; base += bstep) {         // (workgroup uniform in team form)
            bool valid = nvalid;
            const int r = nr, c = nc;
            const GradTaps taps = ntaps;
            nvalid = locate(base + bstep + boff, nr, nc);
            if (nvalid) ntaps = taps_at(nc, nr);
#ifdef SIFT_PHASE_CLOCK
            PH_COUNT(14);
            if (base + bstep + boff < total) PH_WAIT_VM(4); else PH_WAIT_VM(0);
            PH_MARK(2);
#endif
            int bin = 0;
            float val = 0.0f;
            if (valid) {
                float gx = taps.xa - taps.xb, gy = taps.ya - taps.yb;
                if (taps.bx) gx = 2.0f * gx;
                if (taps.by) gy = 2.0f * gy;
                const float gval = MAPS ? taps.xa : sqrtf(gx * gx + gy * gy);
                float dif = (float)r - k.y;
                float distsq = dif * dif;
                dif = (float)c - k.z;
                distsq = distsq + dif * dif;
                valid = (gval > 0.0f) && (distsq < lim);
                if (valid) {
                    // the two Ziv candidates in one basic block (their binary64 chains interleave), one branch for both fall-backs
                    const float earg = fast_div ? siftmath::div_by_reciprocal(-distsq, two_s2, r_two_s2) : -distsq / two_s2;
                    bool ok_a = true, ok_e;
                    float a = MAPS ? taps.xb : siftmath::atan2f_fast_try(-gy, gx, fold, ok_a);
                    float ew = siftmath::expf_fast_try(earg, ok_e);
                    if (!(ok_a && ok_e)) {
                        if (!MAPS && !ok_a) a = siftmath::atan2f_(-gy, gx);
                        if (!ok_e) ew = siftmath::expf_(earg);
                    }
                    bin = (int)siftmath::div_by_reciprocal(36.0f * (a + SM_PI_F + 0.001f), 2.0f * SM_PI_F, 1.0f / (2.0f * SM_PI_F));
                    valid = (bin >= 0) && (bin <= 36);
                    bin = min(max(bin, 0), 35);
                    val = ew * gval;
                }
                if (valid) atomicOr(reinterpret_cast<unsigned *>(L.mask) + 2 * bin + (lane >> 5), 1u << (lane & 31));
            }
            __builtin_amdgcn_wave_barrier();
            PH_MARK(3);
            // owners: vote counts -> aligned pool segments
            const uint2 mine = (lane < 36) ? L.mask[lane] : make_uint2(0u, 0u);
            const int votes = __popc(mine.x) + __popc(mine.y);
            const int padded = (votes + 3) & ~3;
            const int seg = wave_prefix_incl(padded) - padded;
            if (lane < 36) L.mbase[lane] = (unsigned)seg;
            // the last group of four of a segment starts as +0: its padding then adds +0 (an exact no-op on these non-negative
            // sums) and the owners' loops need no per-element masks
            if (votes) reinterpret_cast<float4 *>(L.pool)[(seg + padded - 4) >> 2] = make_float4(0.f, 0.f, 0.f, 0.f);
            __builtin_amdgcn_wave_barrier();
            PH_MARK(4);
            // voters: value to segment base + rank among the voters of the same bin (mbcnt: set bits below this lane)
            {
                const uint2 mk = L.mask[bin];
                const unsigned mb = L.mbase[bin];
                const int pos = mb + __builtin_amdgcn_mbcnt_hi(mk.y, __builtin_amdgcn_mbcnt_lo(mk.x, 0u));
                if (valid) L.pool[pos] = val;
            }
            __builtin_amdgcn_wave_barrier();
            PH_MARK(5);
            if (!team) {
                // owners: ordered sum of the segment
                for (int k0 = 0; k0 < padded; k0 += 4) {
                    const float4 v = pool4[(seg + k0) >> 2];
                    h = h + v.x; h = h + v.y; h = h + v.z; h = h + v.w;
                }
                if (votes) L.mask[lane] = make_uint2(0u, 0u);
                __builtin_amdgcn_wave_barrier();
                PH_MARK(6);
            } else {
                __syncthreads();
                if (w4 == 0 && lane < 36) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {                     // the four waves' batches, in raster order
                        const uint2 mq = lds_all[q].mask[lane];
                        const int vq = __popc(mq.x) + __popc(mq.y);
                        const int sq = (int)lds_all[q].mbase[lane];
                        const float4 *pq = reinterpret_cast<const float4 *>(lds_all[q].pool);
                        for (int k0 = 0; k0 < vq; k0 += 4) {
                            const float4 v = pq[(sq + k0) >> 2];
                            h = h + v.x; h = h + v.y; h = h + v.z; h = h + v.w;
                        }
                        if (vq) lds_all[q].mask[lane] = make_uint2(0u, 0u);
                    }
                }
                __syncthreads();
            }
        }
        if (team && w4 != 0) continue;       // the histogram lives in wave 0
        // six passes of circular [1 1 1]/3 smoothing; hist[35] sees the already updated hist[0].  The reference divides
        // in double, (float)((double)s / 3.0) (orientation_cpu.cl:101-109): for a float s the double quotient lies at
        // least |s| / (24 ulp) away from every float rounding boundary (3 x midpoint is never a float), so the double
        // rounding is harmless and the value is the correctly rounded float quotient s / 3.0f -- computed here by
        // siftmath::div_by_reciprocal (tiny sums, where its residuals could underflow, take the double division).
        const int lp = (lane == 0) ? 35 : lane - 1, ln = (lane >= 35) ? 0 : lane + 1;
        auto third = [&](float s) {
            return (__builtin_fabsf(s) >= 1e-25f && __builtin_fabsf(s) <= 1e30f) ? siftmath::div_by_reciprocal(s, 3.0f, 1.0f / 3.0f)
                                                                            : (float)((double)s / 3.0);
        };
#pragma unroll 1
        for (int pass = 0; pass < 6; pass++) {
            const float prev = __shfl(h, lp), nxt = __shfl(h, ln);
            float nh = third((prev + h) + nxt);
            const float nh0 = __shfl(nh, 0);
            if (lane == 35) nh = third((prev + h) + nh0);
            h = (lane < 36) ? nh : 0.0f;
        }
        float mx = (lane < 36) ? h : 0.0f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float maxval = fmaxf(mx, 0.0f);
        const uint64_t eq = __ballot(lane < 36 && h == maxval);
        const int argmax = (maxval > 0.0f && eq) ? (__ffsll((unsigned long long)eq) - 1) : 0;
        const float hp = __shfl(h, argmax == 0 ? 35 : argmax - 1);
        const float hn = __shfl(h, argmax == 35 ? 0 : argmax + 1);
        const float interp = 0.5f * (hp - hn) / (hp - 2.0f * maxval + hn);
        const float angle = 2.0f * SM_PI_F * ((float)argmax + 0.5f + interp) / 36.0f - SM_PI_F;
        // further peaks >= 80 % of the maximum (orientation_cpu.cl:153-172)
        const float hpp = __shfl(h, lp), hnn = __shfl(h, ln);
        bool extra = (lane < 36) && h > hpp && h > hnn && h >= 0.8f * maxval && lane != argmax;
        float a2 = 0.0f;
        if (extra) {
            const float it = 0.5f * (hpp - hnn) / (hpp - 2.0f * h + hnn);
            a2 = (float)((double)(2.0f * SM_PI_F * ((float)lane + 0.5f + it)) / 36.0 - (double)SM_PI_F);
            extra = (a2 >= -SM_PI_F) && (a2 <= SM_PI_F);
        }
        const uint64_t emask = __ballot(extra);
        const float ox = k.z * (float)octsize, oy = k.y * (float)octsize, os = k.w * (float)octsize;
        const float sum4 = ((ox + oy) + os) + angle;
        const int nmain = (sum4 == sum4) ? 1 : 0;     // host NaN sieve of plan.py:545-550, done here
        const int nextra = __popcll((unsigned long long)emask);
        // (the reference's counter also counts the rows its host drops for a NaN: orientation_cpu.cl:150-172, plan.py:545-550)
        if (lane == 0) atomicAdd(&s_hist[3 * oct + min(max(scale - 1, 0), 2)], 1 + nextra);
        // park the results of this keypoint in the wave's LDS buffer (flushed when the next keypoint might not fit)
        if (pending + nmain + nextra > SIFT_ORI_OBUF) flush_wave();
        if (lane == 0 && nmain) { L.obuf[pending] = make_float4(ox, oy, os, angle); L.oaux[pending] = aux; }
        if (extra) {
            const int at = pending + nmain + __popcll((unsigned long long)(emask & ((1ull << lane) - 1ull)));
            L.obuf[at] = make_float4(ox, oy, os, a2);
            L.oaux[at] = aux;
        }
        pending += nmain + nextra;
        PH_MARK(7);
    }
    // ---- the workgroup's remaining entries leave with a single atomicAdd
    if (lane == 0) s_pending[threadIdx.x >> 6] = pending;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_pending[0] + s_pending[1] + s_pending[2] + s_pending[3];
        s_base = tot ? atomicAdd(&cnt->g_out[group], tot) : 0;
    }
    __syncthreads();
    {
        const int w = threadIdx.x >> 6;
        int slot = s_base;
        for (int q = 0; q < w; q++) slot += s_pending[q];
        store_pending(slot, pending);
    }
    for (int t = threadIdx.x; t < 3 * SIFT_MAX_OCTAVES; t += blockDim.x)
        if (s_hist[t]) atomicAdd(&cnt->o_scale[0][0] + t, s_hist[t]);
#ifdef SIFT_PHASE_CLOCK
    PH_MARK(8);
    ph.flush(0, lane);
#endif
}


// ------------------------------------------------------------------------------------------
// Descriptor, streaming form: ONE WAVEFRONT per oriented keypoint, no workgroup barriers (keypoints_cpu.cl:36-161).
// Handles windows of any size; since round 2 only used for keypoint lists handed to the stage entry point whose windows
// exceed 2 * SIFT_DESC_MAXRAD + 1 rows, or on request (option "desc_stream") -- k_descriptor.hpp holds the row-interval form.
//
//  1. The (2R+1)^2 raster scan is filtered to the samples that fall inside the rotated 5x5-cell
//     window by an ORDER-PRESERVING compaction (wave ballot + popcount prefix) into a small LDS
//     list that also keeps the sample's (rx, cx).
//  2. 64 listed samples at a time, every lane evaluates one sample (gradient, atan2, exp, the
//     trilinear weights) and publishes in LDS its <= 8 contribution values, its packed cell origin
//     (ri, ci, oi) and one bit per touched bin in that bin's 64-bit "who contributes" mask
//     (LDS atomic OR: order independent).
//  3. Lane l owns bins l and l+64: it walks the set bits of each mask in ascending order -- the raster
//     order of the samples -- and adds the matching values one by one, so every bin sees exactly
//     the reference's sequence of float additions.
// Waves of a block are independent (4 keypoints per 256-thread block); latency is hidden by
// occupancy instead of by barriers.
struct DescWaveLds {
    // per bin: 64-bit mask of contributing lanes (two words) and the pool base, as three separate arrays: with one
    // 16-byte record per bin every access had a 4-bank stride (48 % of the kernel's LDS cycles were bank conflicts)
    unsigned int mlo[128], mhi[128], mbase[128];
    float pool[8 * 64];           // contribution values, grouped by bin, in lane (= raster) order inside a bin
    int pool_cnt;
    int sij[128];                 // packed (ii + 32768) | (jj + 32768) << 16
    float srx[128], scx[128];
    float V[128];
};

// 5 waves per SIMD (96 VGPRs, 20 bytes of scratch): +12 % on keypoint-dense frames against the natural 116 VGPRs / 4 waves
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void descriptor_stream_kernel(OctaveTable tab, const float4 *__restrict__ okp,
                                                         const int *__restrict__ oaux, Counters *cnt, int group,
                                                         int range_start, int range_end,  // used when cnt == nullptr
                                                         int out_capacity, KpRecord *__restrict__ records, int rec_capacity,
                                                         KpRecord *host_records, int host_capacity) {
    __shared__ DescWaveLds lds_all[4];
    __shared__ RecordSink sink;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    DescWaveLds &L = lds_all[wave];
    int start = range_start, end = range_end;
    if (cnt) { start = 0; end = min(cnt->g_out[group], out_capacity); }
    descriptor_open(cnt, group, end, records, rec_capacity, host_records, host_capacity, &sink);
    L.mlo[lane] = 0u; L.mhi[lane] = 0u; L.mlo[lane + 64] = 0u; L.mhi[lane + 64] = 0u;
    if (lane == 0) L.pool_cnt = 0;
    const int gwave = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int i = start + gwave; i < end; i += nwaves) {
        const float4 kq = okp[i];        // (x, y, sigma*oct, angle)
        const int aux = oaux[i];         // detection scale | octave << 8
        const int scale = aux & 0xff, oct = aux >> 8;
        const int W = tab.W[oct], H = tab.H[oct], octsize = 1 << oct;
        if (!(kq.y >= 0.0f)) {
            store_record(&sink, i, kq, 0, 0, lane, reinterpret_cast<unsigned char *>(L.V));
            continue;
        }
        const float *I = tab.base + tab.off[oct] + (size_t)scale * W * H;
        const float foct = (float)octsize;
        const float row = kq.y / foct, col = kq.x / foct, angle = kq.w;
        const int irow = (int)(row + 0.5f), icol = (int)(col + 0.5f);
        float sine, cosine;
        siftmath::sincosf_(angle, &sine, &cosine);
        const float spacing = kq.z / foct * 3.0f;
        const int iradius = (int)((1.414f * spacing * 2.5f) + 0.5f);
        const float drow = row - (float)irow, dcol = col - (float)icol;
        const int S = 2 * iradius + 1;
        const int total = S * S;
        const float inv_S = 1.0f / (float)S;
        float acc0 = 0.0f, acc1 = 0.0f;  // bins lane and lane + 64
        int list_n = 0, pos = 0;
        while (pos < total || list_n > 0) {
            // ---- 1. refill the ordered list of in-window samples
            while (list_n < 64 && pos < total) {
                const int idx = pos + lane;
                bool inside = false;
                float rx = 0.f, cx = 0.f;
                int ii = 0, jj = 0;
                if (idx < total) {
                    int rem;
                    const int q = div_exact(idx, S, inv_S, rem);
                    ii = q - iradius; jj = rem - iradius;
                    rx = ((cosine * (float)ii - sine * (float)jj) - drow) / spacing + 1.5f;
                    cx = ((sine * (float)ii + cosine * (float)jj) - dcol) / spacing + 1.5f;
                    const int yy = irow + ii, xx = icol + jj;
                    inside = rx > -1.0f && rx < 4.0f && cx > -1.0f && cx < 4.0f && yy >= 0 && yy < H && xx >= 0 && xx < W;
                }
                const unsigned long long bal = __ballot(inside);
                if (inside) {
                    const int at = list_n + __popcll(bal & lt_mask);
                    L.sij[at] = (ii + 32768) | ((jj + 32768) << 16);
                    L.srx[at] = rx;
                    L.scx[at] = cx;
                }
                list_n += __popcll(bal);
                pos += 64;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- 2. evaluate up to 64 listed samples
            const int m = min(list_n, 64);
            int cbin[8];
            float cval[8];
#pragma unroll
            for (int n8 = 0; n8 < 8; n8++) { cbin[n8] = -1; cval[n8] = 0.0f; }
            if (lane < m && !ABL(3)) {
                const int pk = L.sij[lane];
                const int ii = (pk & 0xffff) - 32768, jj = ((pk >> 16) & 0xffff) - 32768;
                const float rx = L.srx[lane], cx = L.scx[lane];
                float g, o;
                if (ABL(2)) { g = I[(size_t)(irow + ii) * W + icol + jj]; o = g * 0.01f; }
                else gradient_at(I, icol + jj, irow + ii, W, H, g, o);
                const float er = rx - 1.5f, ec = cx - 1.5f;
                const float mag = g * (ABL(2) ? (er * ec) : siftmath::expf_(-0.125f * (er * er + ec * ec)));
                o = o - angle;
                while (o > 2.0f * SM_PI_F) o -= 2.0f * SM_PI_F;
                while (o < 0.0f) o += 2.0f * SM_PI_F;
                const float oval = 4.0f * o * SM_1_PI_F;
                const int ri = (int)((rx >= 0.0f) ? rx : rx - 1.0f);
                const int ci = (int)((cx >= 0.0f) ? cx : cx - 1.0f);
                const int oi = (int)((oval >= 0.0f) ? oval : oval - 1.0f);
                const float rf = rx - (float)ri, cf = cx - (float)ci, of = oval - (float)oi;
                const bool contributes = ri >= -1 && ri < 4 && oi >= 0 && oi <= 8 && rf >= 0.0f && rf <= 1.0f;
                if (contributes) {
                    const unsigned int bit = 1u << (lane & 31);
                    const int word = lane >> 5;
#pragma unroll
                    for (int a = 0; a < 2; a++) {
                        const int rb = ri + a;
                        const float rw = mag * (a == 0 ? 1.0f - rf : rf);
#pragma unroll
                        for (int bb = 0; bb < 2; bb++) {
                            const int cb = ci + bb;
                            const float cw = rw * (bb == 0 ? 1.0f - cf : cf);
                            const bool ok = rb >= 0 && rb < 4 && cb >= 0 && cb < 4;
#pragma unroll
                            for (int e = 0; e < 2; e++) {
                                int ob = oi + e;
                                // oi == 8 only for oval == 8.0f exactly (ori == 2*pi_f), where of == 0:
                                // e=0 adds cw*1 to bin 0, e=1 adds cw*0 == +0 (no effect) -> skipped.
                                const bool dup = (e == 1 && oi == 8);
                                if (ob >= 8) ob = 0;
                                const int n8 = a * 4 + bb * 2 + e;
                                if (ok && !dup) {
                                    cbin[n8] = (rb * 4 + cb) * 8 + ob;
                                    cval[n8] = cw * (e == 0 ? 1.0f - of : of);
                                    atomicOr((word ? L.mhi : L.mlo) + cbin[n8], bit);
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- 3. ordered accumulation.  (a) every bin owner (lane owns bins lane, lane+64) reserves a
            //         pool segment for its contributors; (b) every sample lane writes each of its values at
            //         segment base + (number of lower lanes contributing to the same bin); (c) the owner adds
            //         its segment front to back: ascending lane == raster order of the samples.
            if (!(ABL(1) || ABL(3))) {
                const uint2 ia = make_uint2(L.mlo[lane], L.mhi[lane]), ib = make_uint2(L.mlo[lane + 64], L.mhi[lane + 64]);
                const int cnta = __popc(ia.x) + __popc(ia.y), cntb = __popc(ib.x) + __popc(ib.y);
                int base_a = 0;
                if (cnta + cntb) base_a = atomicAdd(&L.pool_cnt, cnta + cntb);
                const int base_b = base_a + cnta;
                if (cnta) L.mbase[lane] = (unsigned)base_a;
                if (cntb) L.mbase[lane + 64] = (unsigned)base_b;
                __builtin_amdgcn_wave_barrier();
                const unsigned lo_mask = (lane < 32) ? ((1u << lane) - 1u) : 0xffffffffu;
                const unsigned hi_mask = (lane < 32) ? 0u : ((1u << (lane - 32)) - 1u);
#pragma unroll
                for (int n8 = 0; n8 < 8; n8++)
                    if (cbin[n8] >= 0) {
                        const int b = cbin[n8];
                        L.pool[L.mbase[b] + __popc(L.mlo[b] & lo_mask) + __popc(L.mhi[b] & hi_mask)] = cval[n8];
                    }
                __builtin_amdgcn_wave_barrier();
                const int nmax = max(cnta, cntb);
                for (int r0 = 0; r0 < nmax; r0 += 4) {
                    float va[4], vb[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float ra = L.pool[(base_a + r0 + u) & 511], rb_ = L.pool[(base_b + r0 + u) & 511];
                        va[u] = (r0 + u < cnta) ? ra : 0.0f;
                        vb[u] = (r0 + u < cntb) ? rb_ : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) { acc0 = acc0 + va[u]; acc1 = acc1 + vb[u]; }
                }
                __builtin_amdgcn_wave_barrier();
                if (cnta) { L.mlo[lane] = 0u; L.mhi[lane] = 0u; }
                if (cntb) { L.mlo[lane + 64] = 0u; L.mhi[lane + 64] = 0u; }
                if (lane == 0) L.pool_cnt = 0;
            } else {
                L.mlo[lane] = 0u; L.mhi[lane] = 0u; L.mlo[lane + 64] = 0u; L.mhi[lane + 64] = 0u;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- drop the consumed entries, keep order
            const int rest = list_n - m;
            int mij = 0; float mrx = 0.f, mcx = 0.f;
            if (lane < rest) { mij = L.sij[m + lane]; mrx = L.srx[m + lane]; mcx = L.scx[m + lane]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < rest) { L.sij[lane] = mij; L.srx[lane] = mrx; L.scx[lane] = mcx; }
            list_n = rest;
            __builtin_amdgcn_wave_barrier();
        }
        // ---- normalise, clamp at 0.2, renormalise, quantise (keypoints_cpu.cl:125-160).
        // The reference sums the 128 squares sequentially in index order: reproduce that order.
        L.V[lane] = acc0; L.V[lane + 64] = acc1;
        __builtin_amdgcn_wave_barrier();
        float norm = 0.0f;
#pragma unroll 8
        for (int k2 = 0; k2 < 128; k2++) { const float t = L.V[k2]; norm = norm + t * t; }
        norm = 1.0f / sqrtf(norm);       // rsqrt
        acc0 = acc0 * norm; acc1 = acc1 * norm;
        const bool ch = (acc0 > 0.2f) || (acc1 > 0.2f);
        if (acc0 > 0.2f) acc0 = 0.2f;
        if (acc1 > 0.2f) acc1 = 0.2f;
        __builtin_amdgcn_wave_barrier();
        L.V[lane] = acc0; L.V[lane + 64] = acc1;
        __builtin_amdgcn_wave_barrier();
        if (__ballot(ch)) {
            float n2 = 0.0f;
#pragma unroll 8
            for (int k2 = 0; k2 < 128; k2++) { const float t = L.V[k2]; n2 = n2 + t * t; }
            n2 = 1.0f / sqrtf(n2);
            acc0 = acc0 * n2; acc1 = acc1 * n2;
        }
        __builtin_amdgcn_wave_barrier();
        // (int)(512.0*v) in double, MIN(255, .), NaN -> 0 (see oracle note)
        const int i0 = (acc0 == acc0) ? (int)(512.0 * (double)acc0) : 0;
        const int i1 = (acc1 == acc1) ? (int)(512.0 * (double)acc1) : 0;
        store_record(&sink, i, kq, min(255, i0), min(255, i1), lane, reinterpret_cast<unsigned char *>(L.V));
    }
}

// full-map gradient (stage replay of image.cl:47-80)
__global__ void gradient_kernel(const float *__restrict__ img, float *__restrict__ grad, float *__restrict__ ori, int W, int H) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    float g, o;
    gradient_at(img, x, y, W, H, g, o);
    grad[(size_t)y * W + x] = g;
    ori[(size_t)y * W + x] = o;
}

// Gradient maps of the three detection scales of the octaves [oct_lo, oct_hi): compute_gradient_orientation (image.cl:47-80)
// as the reference runs it on blur[1..3] of every octave (plan.py:658-662).  A work item is a block of 256 columns x
// SIFT_MAP_ROWS rows of one plane; workgroups walk the items of all the octaves with a grid stride (`total` of them, counted
// by gradient_map_items on the host with the same formula).  The arc tangent takes the Ziv fast path of the per-keypoint
// kernels, the defining function where it gives up: the same bits either way.
#define SIFT_MAP_ROWS 32
__host__ __device__ inline long long gradient_map_items_of(int W, int H) {
    return 3LL * ((H + SIFT_MAP_ROWS - 1) / SIFT_MAP_ROWS) * ((W + 255) / 256);
}
__global__ __launch_bounds__(256) void gradient_maps_kernel(OctaveTable tab, int oct_lo, int oct_hi, long long total,
                                                            float *__restrict__ gmap, float *__restrict__ omap) {
    __shared__ double fold[36];
    siftmath::load_atan_fold(fold);
    __syncthreads();
    for (long long item = blockIdx.x; item < total; item += gridDim.x) {      // workgroup uniform
        long long rem = item;
        int oct = oct_lo;
        while (oct < oct_hi - 1 && rem >= gradient_map_items_of(tab.W[oct], tab.H[oct])) { rem -= gradient_map_items_of(tab.W[oct], tab.H[oct]); oct++; }
        const int W = tab.W[oct], H = tab.H[oct];
        const int nx = (W + 255) / 256, nyp = (H + SIFT_MAP_ROWS - 1) / SIFT_MAP_ROWS;
        const int plane = (int)(rem / ((long long)nyp * nx));
        const int r2 = (int)(rem - (long long)plane * nyp * nx);
        const int cy = r2 / nx, xb = r2 - cy * nx;
        if (plane > 2) continue;
        const int x = xb * 256 + (int)threadIdx.x;
        if (x >= W) continue;
        const float *I = tab.base + tab.off[oct] + (size_t)(plane + 1) * W * H;
        const size_t mo = map_offset(tab, oct, plane + 1);
        const int y1 = min((cy + 1) * SIFT_MAP_ROWS, H);
        // (independent rows: a march down the rows with the column's three pixels kept in registers -- three loads per pixel
        // instead of four -- measured slower, 0.214 against 0.176 ms on a 4096^2 octave: its loads wait for one another)
        for (int y = cy * SIFT_MAP_ROWS; y < y1; y++) {
            const GradTaps t = gradient_fetch(I, x, y, W, H);
            float gx = t.xa - t.xb, gy = t.ya - t.yb;
            if (t.bx) gx = 2.0f * gx;
            if (t.by) gy = 2.0f * gy;
            bool ok;
            float a = siftmath::atan2f_fast_try(-gy, gx, fold, ok);
            if (!ok) a = siftmath::atan2f_(-gy, gx);
            gmap[mo + (size_t)y * W + x] = sqrtf(gx * gx + gy * gy);
            omap[mo + (size_t)y * W + x] = a;
        }
    }
}

// elementwise siftmath (test hook); fn 5 / 6: the Ziv fast paths of exp / atan2
__global__ void math_kernel(int fn, const float *__restrict__ a, const float *__restrict__ bb, float *__restrict__ out, int64_t n) {
    __shared__ double fold[36];
    siftmath::load_atan_fold(fold);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s, c;
        switch (fn) {
            case 0: out[i] = siftmath::expf_(a[i]); break;
            case 1: out[i] = siftmath::exp2f_(a[i]); break;
            case 2: siftmath::sincosf_(a[i], &s, &c); out[i] = s; break;
            case 3: siftmath::sincosf_(a[i], &s, &c); out[i] = c; break;
            case 5: out[i] = siftmath::expf_fast(a[i]); break;
            case 6: out[i] = siftmath::atan2f_fast(a[i], bb[i], fold); break;
            case 7: out[i] = siftmath::div_by_reciprocal(a[i], bb[i], 1.0f / bb[i]); break;
            default: out[i] = siftmath::atan2f_(a[i], bb[i]); break;
        }
    }
}

}  // namespace siftk
