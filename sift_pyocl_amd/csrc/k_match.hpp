// k_match.hpp -- brute-force L1 matching with ratio test (matching_cpu.cl:57-109 == matching_gpu.cl:52-106).
//
// This is integer-VALU bound (v_sad_u8), not HBM or MFMA bound: 128 byte-SADs per descriptor pair.
// Work decomposition: a 2-D grid (query block) x (partition of the second list).  Each thread keeps QPT
// query descriptors in registers (32 dwords each); its block streams one partition of the second list
// through a double-buffered LDS tile that every lane reads at the same address (broadcast).  A block
// leaves a partial (best distance, its earliest index, second-best distance) per query; a small second
// kernel folds the partitions in ascending index order and applies the ratio test.
//
// Tie-breaking parity: the reference scans j ascending with strict '<', so the earliest index of the
// minimum wins and dist2 is the second smallest value of the multiset.  Both properties survive the
// partition merge when partitions are folded in ascending order with '<' for "later beats earlier".
//
// Inside a partition the two smallest are kept as packed keys (distance << 16 | index in the partition): the 32
// accumulations of a pair are v_sad_hi_u8 (adds the byte SAD to the UPPER half), started from the index, so the key
// costs nothing; then key2 = med3(key1, key2, key) (the second smallest of the three, key1 <= key2 being invariant)
// and key1 = min(key1, key): two instructions per pair instead of the compare / select chain (six).  An equal
// distance at a later index is a larger key: it never displaces the first and it does become the second -- exactly
// the reference's `if (d < d1) ... else if (d < d2)`.  (Round 3: the kernel issues one VALU instruction per ~3.6
// cycles per SIMD whatever it is, so 100k x 100k is 10^10 pairs x (32 + bookkeeping) / 64 lanes of pure issue.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace siftk {

#define SIFT_MATCH_QPT 2
#define SIFT_MATCH_TILE 64
#define SIFT_MATCH_NONE 0x7fffffff
#define SIFT_MATCH_MAX_PART 65472        // list elements per partition: the index shares a key with the distance (16 bits)

struct MatchPartial { int d1, best, d2, pad; };

// median of three unsigned values, written in the form the backend selects v_med3_u32 for
__device__ __forceinline__ uint32_t match_umed3(uint32_t x, uint32_t y, uint32_t z) { return max(min(x, y), min(max(x, y), z)); }

// FLAGS (ROI-masked / mutual variants, matching_cpu.cl:136-199): qflag[i] bit 0 = every distance of this query is
// forced to 0; lflag[j] = 1: distance to this list element is forced to 0 (the literal `matching_valid` behaviour for
// a masked-out list-2 keypoint), 2: the element does not take part.
template <bool FLAGS>
__global__ __launch_bounds__(256) void match_partial_kernel(const uint8_t *__restrict__ kp1, int n1,
                                                            const uint8_t *__restrict__ kp2, int n2, int part_len,
                                                            MatchPartial *__restrict__ partial,
                                                            const uint8_t *__restrict__ qflag,
                                                            const uint8_t *__restrict__ lflag) {
    __shared__ uint4 tile[2][SIFT_MATCH_TILE * 8];
    __shared__ uint8_t tflag[2][SIFT_MATCH_TILE];
    const int tid = threadIdx.x;
    const int j_begin = blockIdx.y * part_len, j_end = min(j_begin + part_len, n2);
    uint32_t q[SIFT_MATCH_QPT][32];
    int qi[SIFT_MATCH_QPT];
    uint32_t key1[SIFT_MATCH_QPT], key2[SIFT_MATCH_QPT];      // (distance << 16 | index - j_begin) of the smallest / second smallest
    bool qzero[SIFT_MATCH_QPT];
#pragma unroll
    for (int u = 0; u < SIFT_MATCH_QPT; u++) {
        qi[u] = (blockIdx.x * SIFT_MATCH_QPT + u) * 256 + tid;
        key1[u] = SIFT_MATCH_NONE; key2[u] = SIFT_MATCH_NONE;
        const int src = min(qi[u], n1 - 1);
        qzero[u] = FLAGS && (qflag[src] & 1);
        const uint4 *p = reinterpret_cast<const uint4 *>(kp1 + (size_t)src * 144 + 16);
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const uint4 v = p[w];
            q[u][4 * w] = v.x; q[u][4 * w + 1] = v.y; q[u][4 * w + 2] = v.z; q[u][4 * w + 3] = v.w;
        }
    }
    // each thread stages two 16-byte pieces of a 64-descriptor tile
    auto fetch = [&](int j0, uint4 &a, uint4 &b) {
        const int ja = min(j0 + (tid >> 3), n2 - 1), jb = min(j0 + 32 + (tid >> 3), n2 - 1);
        a = reinterpret_cast<const uint4 *>(kp2 + (size_t)ja * 144 + 16)[tid & 7];
        b = reinterpret_cast<const uint4 *>(kp2 + (size_t)jb * 144 + 16)[tid & 7];
    };
    uint4 fa, fb;
    uint8_t ff = 0;
    if (j_begin < j_end) { fetch(j_begin, fa, fb); if (FLAGS && tid < SIFT_MATCH_TILE) ff = lflag[min(j_begin + tid, n2 - 1)]; }
    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += SIFT_MATCH_TILE, buf ^= 1) {
        tile[buf][tid] = fa;
        tile[buf][256 + tid] = fb;
        if (FLAGS && tid < SIFT_MATCH_TILE) tflag[buf][tid] = ff;
        __syncthreads();                      // one barrier per tile: the other buffer is free by construction
        if (j0 + SIFT_MATCH_TILE < j_end) {
            fetch(j0 + SIFT_MATCH_TILE, fa, fb);
            if (FLAGS && tid < SIFT_MATCH_TILE) ff = lflag[min(j0 + SIFT_MATCH_TILE + tid, n2 - 1)];
        }
        const int jn = min(SIFT_MATCH_TILE, j_end - j0);
        const uint4 *tb = tile[buf];
        const uint32_t jl0 = (uint32_t)(j0 - j_begin);
#pragma unroll 4
        for (int j = 0; j < jn; j++) {
            int lf = 0;
            if (FLAGS) { lf = tflag[buf][j]; if (lf == 2) continue; }      // wave-uniform
            uint32_t key[SIFT_MATCH_QPT];
#pragma unroll
            for (int u = 0; u < SIFT_MATCH_QPT; u++) key[u] = jl0 + (uint32_t)j;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const uint4 v = tb[j * 8 + w];
#pragma unroll
                for (int u = 0; u < SIFT_MATCH_QPT; u++) {
                    key[u] = __builtin_amdgcn_sad_hi_u8(q[u][4 * w], v.x, key[u]);
                    key[u] = __builtin_amdgcn_sad_hi_u8(q[u][4 * w + 1], v.y, key[u]);
                    key[u] = __builtin_amdgcn_sad_hi_u8(q[u][4 * w + 2], v.z, key[u]);
                    key[u] = __builtin_amdgcn_sad_hi_u8(q[u][4 * w + 3], v.w, key[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < SIFT_MATCH_QPT; u++) {
                if (FLAGS && (lf == 1 || qzero[u])) key[u] = jl0 + (uint32_t)j;           // distance forced to 0
                // ascending index inside equal distances: the earliest index wins ties (matching_cpu.cl:92-100)
                key2[u] = match_umed3(key1[u], key2[u], key[u]);
                key1[u] = min(key1[u], key[u]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SIFT_MATCH_QPT; u++)
        if (qi[u] < n1) {
            MatchPartial r;
            r.d1 = (key1[u] >> 16) == 0x7fffu ? SIFT_MATCH_NONE : (int)(key1[u] >> 16);
            r.best = (key1[u] >> 16) == 0x7fffu ? 0 : j_begin + (int)(key1[u] & 0xffffu);
            r.d2 = (key2[u] >> 16) == 0x7fffu ? SIFT_MATCH_NONE : (int)(key2[u] >> 16);
            r.pad = 0;
            partial[(size_t)blockIdx.y * n1 + qi[u]] = r;
        }
}

// fold the partitions (ascending) and apply the ratio test (matching_cpu.cl:103-108).
// qflag bit 1 (may be null): this query is dropped (`matching_valid`: keypoint on a masked-out pixel).
// nearest != null: "nearest only" mode for the mutual check -- writes the index of the minimum (-1 if the query was
// dropped or had no candidate) instead of appending pairs.
__global__ __launch_bounds__(256) void match_merge_kernel(const MatchPartial *__restrict__ partial, int n1, int nparts,
                                                          float ratio_th, int2 *__restrict__ pairs,
                                                          int *__restrict__ counter, int capacity,
                                                          const uint8_t *__restrict__ qflag, int *__restrict__ nearest) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const bool dropped = qflag && (qflag[i] & 2);
    int d1 = SIFT_MATCH_NONE, d2 = SIFT_MATCH_NONE, best = 0;
    for (int p = 0; p < nparts && !dropped; p++) {
        const MatchPartial r = partial[(size_t)p * n1 + i];
        // feed the partition's two smallest distances through the reference's update rule, in order
        if (r.d1 < d1) { d2 = d1; d1 = r.d1; best = r.best; }
        else if (r.d1 < d2) d2 = r.d1;
        if (r.d2 < d2) d2 = r.d2;
    }
    if (nearest) { nearest[i] = (dropped || d1 == SIFT_MATCH_NONE) ? -1 : best; return; }
    if (dropped) return;
    // distances are stored as float in the reference, initialised to 1e12f
    const float f1 = (d1 == SIFT_MATCH_NONE) ? 1000000000000.0f : (float)d1;
    const float f2 = (d2 == SIFT_MATCH_NONE) ? 1000000000000.0f : (float)d2;
    if (f2 != 0.0f && f1 / f2 < ratio_th) {
        const int old = atomicAdd(counter, 1);
        if (old < capacity) pairs[old] = make_int2(i, best);
    }
}

// ROI flags of one keypoint list (matching_cpu.cl:155-158,171-174): (c, r) = (int)x, (int)y;
// inside = 0 <= r < roi_height && 0 <= c < roi_width;  on = inside && valid[r*roi_width + c] != 0.
//   as_query: bit 0 = 0, bit 1 = dropped       (mode 1: inside && !on;  mode 2: !on)
//   as_list : 0 / 1 (distance forced to 0, mode 1 && !on) / 2 (excluded, mode 2 && !on)
__global__ void match_roi_flags_kernel(const uint8_t *__restrict__ kp, int n, const int8_t *__restrict__ valid,
                                       int rw, int rh, int mode, uint8_t *__restrict__ as_query, uint8_t *__restrict__ as_list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *k = reinterpret_cast<const float *>(kp + (size_t)i * 144);
    const int c = (int)k[0], r = (int)k[1];
    const bool inside = r < rh && c < rw && r >= 0 && c >= 0;
    const bool on = inside && valid[(size_t)r * rw + c] != 0;
    as_query[i] = (mode == 1 ? (inside && !on) : !on) ? 2 : 0;
    as_list[i] = on ? 0 : (mode == 1 ? 1 : 2);
}

// reverse-direction query flags from the list flags of kp2: list flag 1 -> all distances 0 (bit 0), 2 -> dropped (bit 1)
__global__ void match_reverse_flags_kernel(const uint8_t *__restrict__ l2, int n2, uint8_t *__restrict__ q2) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n2) q2[j] = l2[j] == 1 ? 1 : (l2[j] == 2 ? 2 : 0);
}
// list flags of kp1 for the reverse direction: a dropped query does not take part
__global__ void match_reverse_list_flags_kernel(const uint8_t *__restrict__ q1, int n1, uint8_t *__restrict__ l1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n1) l1[i] = (q1[i] & 2) ? 2 : 0;
}

// mutual check: keep (i, j) iff nearest1_of_2[j] == i; compacts in place order-independently into `out`
__global__ void match_mutual_filter_kernel(const int2 *__restrict__ pairs, int n, const int *__restrict__ nearest,
                                           int2 *__restrict__ out, int *__restrict__ counter) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int2 pr = pairs[t];
    if (nearest[pr.y] == pr.x) out[atomicAdd(counter, 1)] = pr;
}

}  // namespace siftk
