// k_match.hpp -- brute-force L1 matching with ratio test (matching_cpu.cl:57-109 == matching_gpu.cl:52-106).
//
// This is integer-VALU bound (v_sad_u8), not HBM or MFMA bound: 128 byte-SADs per descriptor pair.
// Each thread keeps QPT query descriptors in registers (32 dwords each); the second list is streamed
// through LDS in tiles of 64 descriptors that every lane reads at the same address (broadcast).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace siftk {

#define SIFT_MATCH_QPT 2
#define SIFT_MATCH_TILE 64

__global__ __launch_bounds__(256) void match_kernel(const uint8_t *__restrict__ kp1, int n1,
                                                    const uint8_t *__restrict__ kp2, int n2, float ratio_th,
                                                    int2 *__restrict__ pairs, int *__restrict__ counter, int capacity) {
    __shared__ uint4 tile[SIFT_MATCH_TILE * 8];
    const int tid = threadIdx.x;
    uint32_t q[SIFT_MATCH_QPT][32];
    int qi[SIFT_MATCH_QPT];
    int d1[SIFT_MATCH_QPT], d2[SIFT_MATCH_QPT], best[SIFT_MATCH_QPT];
#pragma unroll
    for (int u = 0; u < SIFT_MATCH_QPT; u++) {
        qi[u] = (blockIdx.x * SIFT_MATCH_QPT + u) * 256 + tid;
        d1[u] = 0x7fffffff; d2[u] = 0x7fffffff; best[u] = 0;
        const int src = min(qi[u], n1 - 1);
        const uint4 *p = reinterpret_cast<const uint4 *>(kp1 + (size_t)src * 144 + 16);
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const uint4 v = p[w];
            q[u][4 * w] = v.x; q[u][4 * w + 1] = v.y; q[u][4 * w + 2] = v.z; q[u][4 * w + 3] = v.w;
        }
    }
    for (int j0 = 0; j0 < n2; j0 += SIFT_MATCH_TILE) {
        __syncthreads();
        for (int k = tid; k < SIFT_MATCH_TILE * 8; k += 256) {
            const int j = min(j0 + (k >> 3), n2 - 1);
            tile[k] = reinterpret_cast<const uint4 *>(kp2 + (size_t)j * 144 + 16)[k & 7];
        }
        __syncthreads();
        const int jn = min(SIFT_MATCH_TILE, n2 - j0);
        for (int j = 0; j < jn; j++) {
            uint32_t dist[SIFT_MATCH_QPT];
#pragma unroll
            for (int u = 0; u < SIFT_MATCH_QPT; u++) dist[u] = 0;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const uint4 v = tile[j * 8 + w];
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
    for (int u = 0; u < SIFT_MATCH_QPT; u++) {
        if (qi[u] < n1) {
            // distances are stored as float in the reference, initialised to 1e12f
            const float f1 = (d1[u] == 0x7fffffff) ? 1000000000000.0f : (float)d1[u];
            const float f2 = (d2[u] == 0x7fffffff) ? 1000000000000.0f : (float)d2[u];
            if (f2 != 0.0f && f1 / f2 < ratio_th) {
                const int old = atomicAdd(counter, 1);
                if (old < capacity) pairs[old] = make_int2(qi[u], best[u]);
            }
        }
    }
}

}  // namespace siftk
