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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace siftk {

#define SIFT_MATCH_QPT 2
#define SIFT_MATCH_TILE 64
#define SIFT_MATCH_NONE 0x7fffffff

struct MatchPartial { int d1, best, d2, pad; };

__global__ __launch_bounds__(256) void match_partial_kernel(const uint8_t *__restrict__ kp1, int n1,
                                                            const uint8_t *__restrict__ kp2, int n2, int part_len,
                                                            MatchPartial *__restrict__ partial) {
    __shared__ uint4 tile[2][SIFT_MATCH_TILE * 8];
    const int tid = threadIdx.x;
    const int j_begin = blockIdx.y * part_len, j_end = min(j_begin + part_len, n2);
    uint32_t q[SIFT_MATCH_QPT][32];
    int qi[SIFT_MATCH_QPT];
    int d1[SIFT_MATCH_QPT], d2[SIFT_MATCH_QPT], best[SIFT_MATCH_QPT];
#pragma unroll
    for (int u = 0; u < SIFT_MATCH_QPT; u++) {
        qi[u] = (blockIdx.x * SIFT_MATCH_QPT + u) * 256 + tid;
        d1[u] = SIFT_MATCH_NONE; d2[u] = SIFT_MATCH_NONE; best[u] = 0;
        const int src = min(qi[u], n1 - 1);
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
    if (j_begin < j_end) fetch(j_begin, fa, fb);
    int buf = 0;
    for (int j0 = j_begin; j0 < j_end; j0 += SIFT_MATCH_TILE, buf ^= 1) {
        tile[buf][tid] = fa;
        tile[buf][256 + tid] = fb;
        __syncthreads();                      // one barrier per tile: the other buffer is free by construction
        if (j0 + SIFT_MATCH_TILE < j_end) fetch(j0 + SIFT_MATCH_TILE, fa, fb);
        const int jn = min(SIFT_MATCH_TILE, j_end - j0);
        const uint4 *tb = tile[buf];
        for (int j = 0; j < jn; j++) {
            uint32_t dist[SIFT_MATCH_QPT];
#pragma unroll
            for (int u = 0; u < SIFT_MATCH_QPT; u++) dist[u] = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const uint4 v = tb[j * 8 + w];
#pragma unroll
                for (int u = 0; u < SIFT_MATCH_QPT; u++) {
                    dist[u] = __builtin_amdgcn_sad_u8(q[u][4 * w], v.x, dist[u]);
                    dist[u] = __builtin_amdgcn_sad_u8(q[u][4 * w + 1], v.y, dist[u]);
                    dist[u] = __builtin_amdgcn_sad_u8(q[u][4 * w + 2], v.z, dist[u]);
                    dist[u] = __builtin_amdgcn_sad_u8(q[u][4 * w + 3], v.w, dist[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < SIFT_MATCH_QPT; u++) {
                const int d = (int)dist[u];
                // strict '<' and ascending j: the earliest index wins ties (matching_cpu.cl:92-100)
                if (d < d1[u]) { d2[u] = d1[u]; d1[u] = d; best[u] = j0 + j; }
                else if (d < d2[u]) d2[u] = d;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SIFT_MATCH_QPT; u++)
        if (qi[u] < n1) {
            MatchPartial r; r.d1 = d1[u]; r.best = best[u]; r.d2 = d2[u]; r.pad = 0;
            partial[(size_t)blockIdx.y * n1 + qi[u]] = r;
        }
}

// fold the partitions (ascending) and apply the ratio test (matching_cpu.cl:103-108)
__global__ __launch_bounds__(256) void match_merge_kernel(const MatchPartial *__restrict__ partial, int n1, int nparts,
                                                          float ratio_th, int2 *__restrict__ pairs,
                                                          int *__restrict__ counter, int capacity) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    int d1 = SIFT_MATCH_NONE, d2 = SIFT_MATCH_NONE, best = 0;
    for (int p = 0; p < nparts; p++) {
        const MatchPartial r = partial[(size_t)p * n1 + i];
        // feed the partition's two smallest distances through the reference's update rule, in order
        if (r.d1 < d1) { d2 = d1; d1 = r.d1; best = r.best; }
        else if (r.d1 < d2) d2 = r.d1;
        if (r.d2 < d2) d2 = r.d2;
    }
    // distances are stored as float in the reference, initialised to 1e12f
    const float f1 = (d1 == SIFT_MATCH_NONE) ? 1000000000000.0f : (float)d1;
    const float f2 = (d2 == SIFT_MATCH_NONE) ? 1000000000000.0f : (float)d2;
    if (f2 != 0.0f && f1 / f2 < ratio_th) {
        const int old = atomicAdd(counter, 1);
        if (old < capacity) pairs[old] = make_int2(i, best);
    }
}

}  // namespace siftk
