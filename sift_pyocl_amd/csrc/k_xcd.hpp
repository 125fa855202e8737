// k_xcd.hpp -- XCD-aware workgroup order, shared by the kernels that tile a large plane (marching blur, extrema).
#pragma once
#include <hip/hip_runtime.h>

namespace siftk {

// Workgroup id -> position in an order in which every XCD owns ONE contiguous range.  The dispatcher deals the linear
// workgroup ids of a grid round-robin over the eight XCDs (id L runs on XCD L % 8, in ascending order of L / 8), and each
// XCD has its own L2: neighbours in id share nothing.  M(L) = (start of XCD L % 8's range) + L / 8 is a bijection of
// [0, n) for every n (the first n % 8 XCDs own one workgroup more), ascending in time on every XCD.
__host__ __device__ __forceinline__ int xcd_contiguous(int L, int n) {
    const int c = L & 7, q = n >> 3, r = n & 7;
    return c * q + (c < r ? c : r) + (L >> 3);
}

}  // namespace siftk
