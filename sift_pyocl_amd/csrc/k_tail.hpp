// k_tail.hpp -- the small octaves of a pyramid in ONE launch.
//
// Below 64 x 64 an octave is eight dependent launches (shrink, five blurs, extrema, refinement) of a few microseconds
// each, and the host cannot even enqueue them faster than ~3 us apiece: on a 512 x 512 frame the four smallest octaves
// are half of the launches of the call and sit at the end of its critical path.  Here one launch takes all of them, a
// 512-thread workgroup per octave with the plane being filtered resident in LDS:
//   shrink (preprocess.cl:267-285)  ->  5 x separable blur (convolution.cl:16-101)  ->  DoG extrema (image.cl:119-213)
//   ->  refinement + compaction (image.cl:235-369, algebra.cl:57-84),
// writing the six blur planes of every octave to HBM for the orientation / descriptor kernels as it goes.
// Same arithmetic as the per-launch kernels, operation for operation (acc = acc + in * tap, taps descending, no FMA;
// the extrema and refinement code is literally shared), so the records are bit-identical.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "k_pyramid.hpp"
#include "k_extrema.hpp"

namespace siftk {

#define SIFT_TAIL_MAX_OCT 6          // octaves one launch can walk
#define SIFT_TAIL_SPIN (1 << 21)     // polls (~0.5 us each) before a workgroup stops waiting for the octave above
#define SIFT_TAIL_MAX_PIXELS 4096    // largest plane (W * H) taken; both sides also <= 128 and >= 14 (reflection stays in range)
#define SIFT_TAIL_THREADS 512      // measured on a 512^2 frame (64^2 + 32^2 + 16^2 octaves): 512 threads 55 us, 1024 (128 VGPRs, spills) 62 us, 256 64 us
#define SIFT_TAIL_EXT_BUF 32

struct TailOctave {
    float *plane[6];
    int W, H, oct;
    float edth;
};
struct TailArgs {
    const float *src;                  // plane 3 of the octave above the first one here, row pitch src_w
    int src_w, n;
    TailOctave o[SIFT_TAIL_MAX_OCT];
    const float *taps[5];              // device tap arrays of the five blurs
    int ntaps[5];
};

__host__ __device__ inline int tail_pitch(int W) { return (W + 3) & ~3; }
// dynamic LDS of the launch for a first (largest) octave W x H: plane + padded copy + row-filtered copy, or the
// extrema waves' parking buffers where those are larger
inline size_t tail_lds_bytes(int W, int H) {
    const size_t pt = (size_t)tail_pitch(W);
    const size_t blur = (size_t)H * (pt + 32) + (size_t)(H + 26) * pt;
    const size_t ext = (size_t)(SIFT_TAIL_THREADS / 64) * sizeof(ExtWaveLdsT<SIFT_TAIL_EXT_BUF>) / 4;
    return 4 * ((size_t)H * pt + (blur > ext ? blur : ext));
}

// One blur of the LDS-resident plane P (pitch PT = tail_pitch(W)): P -> P, and -> gout (pitch W).
// A: H rows x (PT + 32), the plane with its reflected left/right margin materialised; T: (H + N - 1) rows x PT, the
// row-filtered plane with its reflected top/bottom margin materialised (row y lives at y + C).
template <int N>
__device__ __forceinline__ void tail_blur(float *P, float *A, float *T, float *__restrict__ gout, int W, int H,
                                          const float *__restrict__ taps) {
    constexpr int C = (N & 1) ? N / 2 : N / 2 - 1;   // convolution.cl:27-37
    constexpr int NW = (N + 3 + 3) & ~3;
    const int PT = tail_pitch(W), PA = PT + 32;
    const int tid = threadIdx.x;
    // 1. margins in x
    for (int idx = tid; idx < H * (W + N - 1); idx += SIFT_TAIL_THREADS) {
        const int row = idx / (W + N - 1), col = idx - row * (W + N - 1);
        A[row * PA + col] = P[row * PT + reflect_index(col - C, W)];
    }
    __syncthreads();
    // 2. rows: a task = 4 consecutive outputs of a row
    const int ht = PT >> 2;
    for (int task = tid; task < H * ht; task += SIFT_TAIL_THREADS) {
        const int row = task / ht, t = task - row * ht;
        const float *rowp = A + row * PA + 4 * t;
        float w[NW];
#pragma unroll
        for (int k = 0; k < NW / 4; k++) {
            const float4 v = *reinterpret_cast<const float4 *>(rowp + 4 * k);
            w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float tp = taps[N - 1 - q];
            a0 = a0 + w[q] * tp;
            a1 = a1 + w[q + 1] * tp;
            a2 = a2 + w[q + 2] * tp;
            a3 = a3 + w[q + 3] * tp;
        }
        *reinterpret_cast<float4 *>(T + (row + C) * PT + 4 * t) = make_float4(a0, a1, a2, a3);
    }
    __syncthreads();
    // 3. margins in y (reflection commutes with the row filter)
    for (int idx = tid; idx < (N - 1) * ht; idx += SIFT_TAIL_THREADS) {
        const int m = idx / ht, t = idx - m * ht;
        const int row = m < C ? m - C : H + (m - C);            // plane row of this margin row
        *reinterpret_cast<float4 *>(T + (row + C) * PT + 4 * t) =
            *reinterpret_cast<const float4 *>(T + (reflect_index(row, H) + C) * PT + 4 * t);
    }
    __syncthreads();
    // 4. columns: a task = 2 adjacent columns x 2 rows
    const int cp = PT >> 1, rg = (H + 1) >> 1;
    for (int task = tid; task < cp * rg; task += SIFT_TAIL_THREADS) {
        const int g = task / cp, c2 = (task - g * cp) * 2;
        const int r0 = g * 2;
        float2 w[N + 1];
#pragma unroll
        for (int k = 0; k < N + 1; k++)
            w[k] = (r0 + k < H + N - 1) ? *reinterpret_cast<const float2 *>(T + (r0 + k) * PT + c2) : make_float2(0.f, 0.f);
        float2 acc[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float tp = taps[N - 1 - q];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                acc[i].x = acc[i].x + w[i + q].x * tp;
                acc[i].y = acc[i].y + w[i + q].y * tp;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int y = r0 + i;
            if (y < H) {
                *reinterpret_cast<float2 *>(P + y * PT + c2) = acc[i];
                if (c2 < W) gout[(size_t)y * W + c2] = acc[i].x;
                if (c2 + 1 < W) gout[(size_t)y * W + c2 + 1] = acc[i].y;
            }
        }
    }
    __syncthreads();
}

// One workgroup per octave (at most SIFT_TAIL_MAX_OCT of them, checked on the host).  HIP promises nothing about dispatch
// order or residency: on this device a grid is dealt round-robin over eight XCDs that dispatch independently, so
// workgroup k may be resident and polling while workgroup k-1 is still queued (e.g. behind the persistent descriptor
// workgroups of another batch lane).  That is not a deadlock -- at most five polling workgroups can never keep the sixth
// from being placed -- but the wait is bounded all the same (SIFT_TAIL_SPIN polls, ~1 s): a workgroup that gives up raises
// bit 1 of Counters::overflow's neighbour `tail_timeout` and leaves, and the host runs the image again with the
// per-octave launches (plan_wait / siftmi_plan_keypoints).  Octave k+1 needs only plane 3
// of octave k: workgroup k raises ready[k] as soon as that plane is in HBM -- before its last two blurs, its extrema and
// its refinement -- and workgroup k+1 starts from there.  The chain through the tail is then the first three blurs of
// each octave, not the octaves end to end.  Each workgroup has its own candidate list (cand + k * cand_capacity).
// (one 512-thread workgroup per CU: two waves per SIMD whatever the register count -- the whole 256-register budget is
// there to be used; under the default occupancy heuristic the kernel kept 167 and spilled 56 bytes per lane)
__global__ __launch_bounds__(SIFT_TAIL_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void octave_tail_kernel(TailArgs a, int border, double contrast, float peak_thresh,
                                                                       float init_sigma, float4 *__restrict__ cand_all, int cand_capacity,
                                                                       int *__restrict__ n_cand, int *__restrict__ ready,
                                                                       float4 *__restrict__ kp,
                                                                       int *__restrict__ kp_aux, int *__restrict__ n_kp, int kp_capacity,
                                                                       int *__restrict__ c_scale_all, int *__restrict__ timed_out) {
    extern __shared__ float4 tail_smem4[];
    float *smem = reinterpret_cast<float *>(tail_smem4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWAVES = SIFT_TAIL_THREADS / 64;
    const int k = blockIdx.x;
    const TailOctave &o = a.o[k];
    const float *src = k ? a.o[k - 1].plane[3] : a.src;
    const int src_w = k ? a.o[k - 1].W : a.src_w;
    float4 *cand = cand_all + (size_t)k * cand_capacity;
    const int W = o.W, H = o.H, PT = tail_pitch(W);
    float *P = smem, *A = P + H * PT, *T = A + H * (PT + 32);
    if (k) {   // plane 3 of the octave above: acquire at device scope (its stores were released by the raise below)
        int *gave_up = reinterpret_cast<int *>(smem);
        if (tid == 0) {
            int polls = 0, v;
            // relaxed polls (an acquire per poll costs 2-3x per hop), one acquire fence once the flag is up
            while ((v = __hip_atomic_load(ready + k - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++polls < SIFT_TAIL_SPIN)
                __builtin_amdgcn_s_sleep(8);
            *gave_up = (v <= 0);               // never raised, or the producer itself gave up (-1)
        }
        __syncthreads();
        if (*gave_up) {                        // workgroup uniform
            if (tid == 0) {
                __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (k + 1 < a.n) __hip_atomic_store(ready + k, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // pass it on at once
            }
            return;
        }
        // one agent-scope acquire per workgroup (it invalidates this CU's L1; the XCD's L2 was written back by the
        // producer's release), then the barrier: no thread loads plane 3 before the invalidate, nor overwrites the flag word
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    }
    // ---- octave hand-off: every second sample of the plane above
    for (int i = tid; i < W * H; i += SIFT_TAIL_THREADS) {
        const int y = i / W, x = i - y * W;
        const float v = src[(size_t)(2 * y) * src_w + 2 * x];
        P[y * PT + x] = v;
        o.plane[0][i] = v;
    }
    __syncthreads();
    // ---- five blurs
    for (int s = 0; s < 5; s++) {
        switch (a.ntaps[s]) {
            case 11: tail_blur<11>(P, A, T, o.plane[s + 1], W, H, a.taps[s]); break;
            case 15: tail_blur<15>(P, A, T, o.plane[s + 1], W, H, a.taps[s]); break;
            case 17: tail_blur<17>(P, A, T, o.plane[s + 1], W, H, a.taps[s]); break;
            case 21: tail_blur<21>(P, A, T, o.plane[s + 1], W, H, a.taps[s]); break;
            default: tail_blur<27>(P, A, T, o.plane[s + 1], W, H, a.taps[s]); break;
        }
        if (s == 2 && k + 1 < a.n) {   // plane 3 is stored: release it to the next octave's workgroup
            // ONE agent-scope release, by the thread that raises the flag, behind the workgroup barrier (which orders every
            // wave's stores before it): a device-scope release writes the XCD's L2 back, and one per wave is fifteen too many
            __syncthreads();
            if (tid == 0) __hip_atomic_store(ready + k, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // (tail_blur ends with a workgroup barrier: the planes just stored are visible to every wave of this workgroup)
    // ---- extrema of the three detection scales
    BlurPlanes b;
#pragma unroll
    for (int s = 0; s < 6; s++) b.p[s] = o.plane[s];
    int *counter = n_cand + o.oct;
    if (W > 2 * border && H > 2 * border) {
        ExtWaveLdsT<SIFT_TAIL_EXT_BUF> &L = reinterpret_cast<ExtWaveLdsT<SIFT_TAIL_EXT_BUF> *>(A)[wave];
        const int rows = 4;
        const int nx = (W - 2 * border + 61) / 62, ny = (H - 2 * border + rows - 1) / rows;
        int pending = 0;
        for (int wid = wave; wid < nx * ny; wid += NWAVES)
            extrema_strip(b, W, H, border, rows, true, wid % nx, wid / nx, contrast, o.edth, cand, counter, cand_capacity, L, pending);
        if (pending) {                                            // wave uniform
            int slot = 0;
            if (lane == 0) slot = atomicAdd(counter, pending);
            ext_store_pending(L, cand, cand_capacity, __shfl(slot, 0), pending, lane);
        }
    }
    __syncthreads();
    // ---- refinement of the candidates
    const int found = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (cand_capacity is three candidates per sample of the largest tail plane: the list cannot be cut)
    refine_candidates(b, W, H, cand, min(found, cand_capacity), peak_thresh, init_sigma, kp, kp_aux, n_kp, kp_capacity,
                      o.oct, tid, SIFT_TAIL_THREADS, c_scale_all ? c_scale_all + 3 * o.oct : nullptr);
}

}  // namespace siftk
