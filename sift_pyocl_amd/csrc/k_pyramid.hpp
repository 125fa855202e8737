// k_pyramid.hpp -- scale-space pyramid kernels for gfx950: input conversion, min/max, fused
// separable Gaussian blur (LDS tile with halo), octave hand-off.
//
// Numerics contract (bit-identical to the reference's OpenCL-CPU path compiled without FMA
// contraction): every multiply and add is a separate IEEE binary32 operation in the reference's
// order -- this file must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace siftk {

// ------------------------------------------------------------------------------------------
// order-preserving float <-> uint encoding for atomicMin/Max
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// mm[0] = ord(min), mm[1] = ord(max); reset to {0xffffffff, 0}
__global__ void minmax_init(uint32_t *mm) { mm[0] = 0xffffffffu; mm[1] = 0u; }

// Global min / max of an f32 image (replaces max_min_global_stage1/2, reductions.cl:62-199).
// min/max are order independent, so any reduction tree gives the reference's bits.
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ img, int64_t n, uint32_t *mm) {
    float lo = __builtin_inff(), hi = -__builtin_inff();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = n >> 2;
    const float4 *img4 = reinterpret_cast<const float4 *>(img);
    for (int64_t k = i; k < n4; k += stride) {
        float4 v = img4[k];
        lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
        hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (int64_t k = (n4 << 2) + i; k < n; k += stride) {
        float v = img[k];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    __shared__ float slo[4], shi[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { slo[wave] = lo; shi[wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
        if (lo <= hi) {   // skip all-NaN partials
            atomicMin(&mm[0], f2ord(lo));
            atomicMax(&mm[1], f2ord(hi));
        }
    }
}

// ------------------------------------------------------------------------------------------
// integer / RGB inputs -> float plane  (preprocess.cl:53-223): (float)x per element,
// RGB: 0.299f*R + 0.587f*G + 0.114f*B evaluated left to right.
template <typename T>
__global__ void convert_kernel(const T *__restrict__ in, float *__restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
}
__global__ void convert_rgb_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float r = (float)in[3 * i], g = (float)in[3 * i + 1], b = (float)in[3 * i + 2];
        out[i] = (0.299f * r + 0.587f * g) + 0.114f * b;
    }
}

// ------------------------------------------------------------------------------------------
// Fused separable Gaussian blur.  Replaces horizontal_convolution + vertical_convolution
// (convolution.cl:16-101) and, when NORM, the preceding `normalizes` (preprocess.cl:239-252).
//
// One 256-thread workgroup produces a TX x TY = 128 x 64 output tile:
//   1. the (TY+N-1) x (TX+N-1) input window is staged in LDS (symmetric boundary handled here),
//   2. horizontal pass in place in LDS: each thread owns 4 consecutive outputs of a row and slides
//      over N+3 inputs fetched with ds_read_b128 (one row per half-wave: conflict free),
//   3. vertical pass from LDS: each thread owns 2 adjacent columns x 8 rows per step
//      (ds_read_b64), result stored with coalesced 8-byte stores.
// Taps live in the kernel argument segment (SGPRs).  Accumulation order and rounding are exactly
// the reference's: acc = 0; acc = acc + in[x-c+j] * taps[N-1-j], j ascending, no FMA.
template <int N> struct TapsArg { float t[N]; };

template <int N> struct BlurGeom {
    static constexpr int TX = 128, TY = 64;
    static constexpr int C = (N & 1) ? N / 2 : N / 2 - 1;   // convolution.cl:27-37
    static constexpr int ROWS = TY + N - 1;
    static constexpr int COLS = TX + N - 1;
    static constexpr int PITCH = (COLS + 3) & ~3;
    static constexpr int NW = (N + 3 + 3) & ~3;             // floats read per 4-output H task
    static constexpr int LDS_BYTES = ROWS * PITCH * 4;
};

__device__ __forceinline__ int reflect_index(int i, int n) {
    if (i < 0) i = -i - 1;
    else if (i > n - 1) i = 2 * n - 1 - i;
    return min(max(i, 0), n - 1);   // clamp only matters for lanes whose output is masked
}

template <int N, bool NORM>
__global__ __launch_bounds__(256) void blur_hv_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                      int W, int H, TapsArg<N> taps,
                                                      const uint32_t *__restrict__ mm) {
    using G = BlurGeom<N>;
    extern __shared__ float4 smem4[];
    float *s = reinterpret_cast<float *>(smem4);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * G::TX, y0 = blockIdx.y * G::TY;

    float mn = 0.f, range = 1.f;
    if (NORM) { mn = ord2f(mm[0]); range = ord2f(mm[1]) - mn; }

    // ---- 1. stage input window ---------------------------------------------------------------
    for (int idx = tid; idx < G::ROWS * G::COLS; idx += 256) {
        const int row = idx / G::COLS, col = idx - row * G::COLS;
        const int gy = reflect_index(y0 - G::C + row, H);
        const int gx = reflect_index(x0 - G::C + col, W);
        float v = in[(size_t)gy * W + gx];
        if (NORM) v = 255.0f * (v - mn) / range;     // preprocess.cl:250
        s[row * G::PITCH + col] = v;
    }
    __syncthreads();

    // ---- 2. horizontal pass, in place -------------------------------------------------------
    for (int task = tid; task < G::ROWS * 32; task += 256) {
        const int row = task >> 5, t = task & 31;
        float *rowp = s + row * G::PITCH + 4 * t;
        float w[G::NW];
#pragma unroll
        for (int k = 0; k < G::NW / 4; k++) {
            const float4 v = *reinterpret_cast<const float4 *>(rowp + 4 * k);
            w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float tp = taps.t[N - 1 - q];
            a0 = a0 + w[q] * tp;
            a1 = a1 + w[q + 1] * tp;
            a2 = a2 + w[q + 2] * tp;
            a3 = a3 + w[q + 3] * tp;
        }
        // All lanes of this wave have issued their reads (same row lives in one half-wave and LDS
        // executes a wave's operations in order); keep the compiler from moving the store up.
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<float4 *>(rowp) = make_float4(a0, a1, a2, a3);
    }
    __syncthreads();

    // ---- 3. vertical pass --------------------------------------------------------------------
    const int c2 = (tid & 63) * 2, q4 = tid >> 6;
    const int gx = x0 + c2;
#pragma unroll 1
    for (int chunk = 0; chunk < 2; chunk++) {
        const int r0 = q4 * 16 + chunk * 8;
        float2 w[8 + N - 1];
#pragma unroll
        for (int k = 0; k < 8 + N - 1; k++)
            w[k] = *reinterpret_cast<const float2 *>(s + (r0 + k) * G::PITCH + c2);
        float2 acc[8];
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float tp = taps.t[N - 1 - q];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[i].x = acc[i].x + w[i + q].x * tp;
                acc[i].y = acc[i].y + w[i + q].y * tp;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int gy = y0 + r0 + i;
            if (gy < H) {
                float *o = out + (size_t)gy * W + gx;
                if (gx + 1 < W && ((W & 1) == 0)) *reinterpret_cast<float2 *>(o) = acc[i];
                else {
                    if (gx < W) o[0] = acc[i].x;
                    if (gx + 1 < W) o[1] = acc[i].y;
                }
            }
        }
    }
}

// Generic (any tap count, incl. even sizes) two-pass blur: plain global loads, used only for
// non-default init_sigma schedules and stage replay.  Same arithmetic.
__global__ void blur_generic_pass(const float *__restrict__ in, float *__restrict__ out, int W, int H,
                                  const float *__restrict__ taps, int n, int vertical,
                                  const uint32_t *__restrict__ mm, int norm) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const int c = (n & 1) ? n / 2 : n / 2 - 1;
    float mn = 0.f, range = 1.f;
    if (norm) { mn = ord2f(mm[0]); range = ord2f(mm[1]) - mn; }
    float acc = 0.f;
    for (int j = 0; j < n; j++) {
        float v;
        if (vertical) v = in[(size_t)reflect_index(y - c + j, H) * W + x];
        else v = in[(size_t)y * W + reflect_index(x - c + j, W)];
        if (norm) v = 255.0f * (v - mn) / range;
        acc = acc + v * taps[n - 1 - j];
    }
    out[(size_t)y * W + x] = acc;
}

// ------------------------------------------------------------------------------------------
// octave hand-off (shrink, preprocess.cl:267-285): next[y][x] = cur[2y][2x]
__global__ void shrink_kernel(const float *__restrict__ in, float *__restrict__ out, int LW, int SW, int SH) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < SW && y < SH) out[(size_t)y * SW + x] = in[(size_t)(2 * y) * LW + 2 * x];
}

// normalise only (stage replay of preprocess.cl:239-252)
__global__ void normalize_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n,
                                 const uint32_t *__restrict__ mm) {
    const float mn = ord2f(mm[0]), range = ord2f(mm[1]) - mn;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = 255.0f * (in[i] - mn) / range;
}

}  // namespace siftk
