// k_pyramid.hpp -- scale-space pyramid kernels for gfx950: input conversion, min/max, fused
// separable Gaussian blur (LDS tile with halo), octave hand-off.
//
// Numerics contract (bit-identical to the reference's OpenCL-CPU path compiled without FMA
// contraction): every multiply and add is a separate IEEE binary32 operation in the reference's
// order -- this file must be compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include "k_xcd.hpp"

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
// `reset` (may be null): the counter block of the NEXT image, zeroed here by workgroup 0 except word `reset_ones`, which becomes
// 0xffffffff (the order-encoded +inf of its min slot).  The plan alternates between two counter blocks, so the image that is
// running never touches the block being reset, and the per-image begin_image launch of rounds 1-3 (3.5 us + an 8 us gap in
// front of this kernel, profiles/r04/timeline_white4096.txt) is gone.
__device__ __forceinline__ void minmax_reset_next(uint32_t *reset, int reset_words, int reset_ones) {
    if (reset && blockIdx.x == 0)
        for (int i = threadIdx.x; i < reset_words; i += blockDim.x) reset[i] = (i == reset_ones) ? 0xffffffffu : 0u;
}

// Workgroups of 256 or 1024 threads (blockDim): the pass is a latency-bound stream -- 256 workgroups of four waves are one wave
// per SIMD (3.5 TB/s), and more workgroups pay for their two same-address atomics (8-10 ns each, served one at a time);
// sixteen waves per workgroup give four waves per SIMD at the same 256 atomic pairs.
__global__ __launch_bounds__(1024) void minmax_kernel(const float *__restrict__ img, int64_t n, uint32_t *mm, uint32_t *reset = nullptr,
                                                      int reset_words = 0, int reset_ones = -1) {
    minmax_reset_next(reset, reset_words, reset_ones);
    float lo = __builtin_inff(), hi = -__builtin_inff();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = n >> 2;
    const float4 *img4 = reinterpret_cast<const float4 *>(img);
    int64_t k = i;
    for (; k + 3 * stride < n4; k += 4 * stride) {    // four independent 16-byte loads in flight per lane
        const float4 v0 = img4[k], v1 = img4[k + stride], v2 = img4[k + 2 * stride], v3 = img4[k + 3 * stride];
        lo = fminf(lo, fminf(fminf(fminf(v0.x, v0.y), fminf(v0.z, v0.w)), fminf(fminf(v1.x, v1.y), fminf(v1.z, v1.w))));
        lo = fminf(lo, fminf(fminf(fminf(v2.x, v2.y), fminf(v2.z, v2.w)), fminf(fminf(v3.x, v3.y), fminf(v3.z, v3.w))));
        hi = fmaxf(hi, fmaxf(fmaxf(fmaxf(v0.x, v0.y), fmaxf(v0.z, v0.w)), fmaxf(fmaxf(v1.x, v1.y), fmaxf(v1.z, v1.w))));
        hi = fmaxf(hi, fmaxf(fmaxf(fmaxf(v2.x, v2.y), fmaxf(v2.z, v2.w)), fmaxf(fmaxf(v3.x, v3.y), fmaxf(v3.z, v3.w))));
    }
    for (; k < n4; k += stride) {
        float4 v = img4[k];
        lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
        hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (int64_t k2 = (n4 << 2) + i; k2 < n; k2 += stride) {
        float v = img[k2];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
    __shared__ float slo[16], shi[16];
    const int wave = threadIdx.x >> 6, nwaves = (int)blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { slo[wave] = lo; shi[wave] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = slo[0]; hi = shi[0];
        for (int w = 1; w < nwaves; w++) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
        if (lo <= hi) {   // skip all-NaN partials
            atomicMin(&mm[0], f2ord(lo));
            atomicMax(&mm[1], f2ord(hi));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Typed pixel load: the integer / RGB converters of preprocess.cl:53-223 applied at the point of use, so that a
// u8 / u16 / RGB frame is never materialised as an f32 plane in HBM.  DT = SIFTMI_* dtype code
// (0 f32, 1 u8, 2 u16, 3 u32, 4 u64, 5 i32, 6 i64, 8 rgb8).  (float)x rounds to nearest even exactly like the
// OpenCL implicit conversion; RGB is 0.299f*R + 0.587f*G + 0.114f*B left to right, unfused (preprocess.cl:221).
template <int DT> __device__ __forceinline__ float load_px(const void *__restrict__ base, size_t i) {
    if constexpr (DT == 0) return static_cast<const float *>(base)[i];
    else if constexpr (DT == 1) return (float)static_cast<const uint8_t *>(base)[i];
    else if constexpr (DT == 2) return (float)static_cast<const uint16_t *>(base)[i];
    else if constexpr (DT == 3) return (float)static_cast<const uint32_t *>(base)[i];
    else if constexpr (DT == 4) return (float)static_cast<const uint64_t *>(base)[i];
    else if constexpr (DT == 5) return (float)static_cast<const int32_t *>(base)[i];
    else if constexpr (DT == 6) return (float)static_cast<const int64_t *>(base)[i];
    else {
        static_assert(DT == 8, "unsupported pixel type");
        const uint8_t *q = static_cast<const uint8_t *>(base) + 3 * i;
        const float r = (float)q[0], g = (float)q[1], b = (float)q[2];
        return (0.299f * r + 0.587f * g) + 0.114f * b;
    }
}

// min / max of the converted values of a typed frame.  A lane consumes chunks of R 16-byte words (u8: 16 pixels,
// u16: 8, 32-bit: 4, 64-bit: 4 in two words, RGB8: 16 pixels in three words), four chunks in flight per step;
// the frame base must be 16-byte aligned (checked on the host).
template <int DT> struct TypedChunk {
    static constexpr int R = (DT == 8) ? 3 : ((DT == 4 || DT == 6) ? 2 : 1);      // uint4 words per chunk
    static constexpr int PX = (DT == 1 || DT == 8) ? 16 : (DT == 2 ? 8 : 4);     // pixels per chunk
};

template <int DT>
__device__ __forceinline__ void typed_chunk_minmax(const uint4 (&w)[TypedChunk<DT>::R], float &lo, float &hi) {
    auto upd = [&](float v) { lo = fminf(lo, v); hi = fmaxf(hi, v); };
    if constexpr (DT == 1) {
        const uint32_t d[4] = {w[0].x, w[0].y, w[0].z, w[0].w};
#pragma unroll
        for (int q = 0; q < 4; q++) { upd((float)(d[q] & 0xff)); upd((float)((d[q] >> 8) & 0xff)); upd((float)((d[q] >> 16) & 0xff)); upd((float)(d[q] >> 24)); }
    } else if constexpr (DT == 2) {
        const uint32_t d[4] = {w[0].x, w[0].y, w[0].z, w[0].w};
#pragma unroll
        for (int q = 0; q < 4; q++) { upd((float)(d[q] & 0xffff)); upd((float)(d[q] >> 16)); }
    } else if constexpr (DT == 3) {
        upd((float)w[0].x); upd((float)w[0].y); upd((float)w[0].z); upd((float)w[0].w);
    } else if constexpr (DT == 5) {
        upd((float)(int32_t)w[0].x); upd((float)(int32_t)w[0].y); upd((float)(int32_t)w[0].z); upd((float)(int32_t)w[0].w);
    } else if constexpr (DT == 4 || DT == 6) {
        const uint64_t q[4] = {(uint64_t)w[0].x | ((uint64_t)w[0].y << 32), (uint64_t)w[0].z | ((uint64_t)w[0].w << 32),
                               (uint64_t)w[1].x | ((uint64_t)w[1].y << 32), (uint64_t)w[1].z | ((uint64_t)w[1].w << 32)};
#pragma unroll
        for (int k = 0; k < 4; k++) upd(DT == 4 ? (float)q[k] : (float)(int64_t)q[k]);
    } else {
        const uint32_t d[12] = {w[0].x, w[0].y, w[0].z, w[0].w, w[1].x, w[1].y, w[1].z, w[1].w, w[2].x, w[2].y, w[2].z, w[2].w};
        auto byte = [&](int b) { return (float)((d[b >> 2] >> (8 * (b & 3))) & 0xff); };
#pragma unroll
        for (int k = 0; k < 16; k++) upd((0.299f * byte(3 * k) + 0.587f * byte(3 * k + 1)) + 0.114f * byte(3 * k + 2));
    }
}

template <int DT>
__global__ __launch_bounds__(256) void minmax_typed_kernel(const void *__restrict__ img, int64_t n, uint32_t *mm, uint32_t *reset = nullptr,
                                                           int reset_words = 0, int reset_ones = -1) {
    minmax_reset_next(reset, reset_words, reset_ones);
    using C = TypedChunk<DT>;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nchunks = n / C::PX;
    const uint4 *img4 = static_cast<const uint4 *>(img);
    int64_t k = i;
    for (; k + 3 * stride < nchunks; k += 4 * stride) {
        uint4 w[4][C::R];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int r = 0; r < C::R; r++) w[u][r] = img4[(k + u * stride) * C::R + r];
#pragma unroll
        for (int u = 0; u < 4; u++) typed_chunk_minmax<DT>(w[u], lo, hi);
    }
    for (; k < nchunks; k += stride) {
        uint4 w[C::R];
#pragma unroll
        for (int r = 0; r < C::R; r++) w[r] = img4[k * C::R + r];
        typed_chunk_minmax<DT>(w, lo, hi);
    }
    for (int64_t k2 = nchunks * C::PX + i; k2 < n; k2 += stride) {
        const float v = load_px<DT>(img, (size_t)k2);
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
        if (lo <= hi) {
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

template <int N, int TX_ = 128, int TY_ = 64> struct BlurGeom {
    static constexpr int TX = TX_, TY = TY_;
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

// TX x TY = 128 x 64 is the throughput shape; planes too small to give every CU a 128 x 64 tile use 64 x 32 or
// 32 x 16 tiles (VR = rows per vertical task: 8, or 4 to spread the small tiles' vertical pass over more lanes).
// A pyramid of a small frame is a chain of short launches, so it is the latency of one workgroup that matters there.
template <int N, bool NORM, int DT = 0, int TX = 128, int TY = 64, int VR = 8>
__global__ __launch_bounds__(256) void blur_hv_kernel(const void *__restrict__ in, float *__restrict__ out,
                                                      int W, int H, TapsArg<N> taps,
                                                      const uint32_t *__restrict__ mm,
                                                      float *__restrict__ half) {   // not null: also out[2y][2x] -> half (the next octave's plane 0)
    using G = BlurGeom<N, TX, TY>;
    static_assert(TX % 4 == 0 && 64 % (TX / 4) == 0, "the H tasks of one row must sit in one wave");
    static_assert(TY % VR == 0, "vertical tasks tile the rows");
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
        float v = load_px<DT>(in, (size_t)gy * W + gx);
        if (NORM) v = 255.0f * (v - mn) / range;     // preprocess.cl:250
        s[row * G::PITCH + col] = v;
    }
    __syncthreads();

    // ---- 2. horizontal pass, in place -------------------------------------------------------
    constexpr int HT = TX / 4;   // 4-output tasks per row
    for (int task = tid; task < G::ROWS * HT; task += 256) {
        const int row = task / HT, t = task - row * HT;
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
        // All tasks of a row sit in one wave (64 % HT == 0) and LDS executes a wave's operations in order: every
        // lane has issued its reads before any lane stores.  Keep the compiler from moving the store up.
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<float4 *>(rowp) = make_float4(a0, a1, a2, a3);
    }
    __syncthreads();

    // ---- 3. vertical pass: a task = 2 adjacent columns x VR rows --------------------------------
    constexpr int CP = TX / 2;
#pragma unroll 1
    for (int task = tid; task < CP * (TY / VR); task += 256) {
        const int rg = task / CP, c2 = (task - rg * CP) * 2;
        const int r0 = rg * VR;
        const int gx = x0 + c2;
        float2 w[VR + N - 1];
#pragma unroll
        for (int k = 0; k < VR + N - 1; k++)
            w[k] = *reinterpret_cast<const float2 *>(s + (r0 + k) * G::PITCH + c2);
        float2 acc[VR];
#pragma unroll
        for (int i = 0; i < VR; i++) acc[i] = make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < N; q++) {
            const float tp = taps.t[N - 1 - q];
#pragma unroll
            for (int i = 0; i < VR; i++) {
                acc[i].x = acc[i].x + w[i + q].x * tp;
                acc[i].y = acc[i].y + w[i + q].y * tp;
            }
        }
#pragma unroll
        for (int i = 0; i < VR; i++) {
            const int gy = y0 + r0 + i;
            if (gy < H) {
                float *o = out + (size_t)gy * W + gx;
                if (gx + 1 < W && ((W & 1) == 0)) *reinterpret_cast<float2 *>(o) = acc[i];
                else {
                    if (gx < W) o[0] = acc[i].x;
                    if (gx + 1 < W) o[1] = acc[i].y;
                }
                // octave hand-off (preprocess.cl:267-285) fused into the launch that writes plane 3 (gx is even)
                if (half && !(gy & 1) && (gy >> 1) < (H >> 1) && (gx >> 1) < (W >> 1))
                    half[(size_t)(gy >> 1) * (W >> 1) + (gx >> 1)] = acc[i].x;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused separable blur, "marching" form, for large planes.
//
// (Round 1's one-block form -- a 128-thread workgroup doing both passes -- is in the history, commit 55027f7: blur_march_kernel;
// what follows describes the arithmetic both forms share, the team form below is the one the product launches.)
// A workgroup owns a strip 256 columns wide and marches down `nblocks` blocks of N rows:
//   * the next block's N x (256+N-1) inputs are prefetched into registers while the current block
//     is processed, then staged in LDS with the rows interleaved in pairs ([row pair][col][row&1]),
//   * horizontal pass in place in LDS: a lane owns 4 consecutive outputs of a ROW PAIR and slides
//     over N+3 float2 inputs (ds_read_b128); the two rows ride in the two halves of packed-f32
//     instructions (v_pk_mul_f32 / v_pk_add_f32, measured 1.5x the rate of scalar mul+add),
//   * vertical pass without a window: thread t owns columns 2t, 2t+1 (one packed pair) and keeps
//     the N in-flight output rows as N rotating packed accumulators.  Row k adds h*taps[N-1-j] to
//     the accumulator of output row k-j (j = 0..N-1); rows arrive in ascending order, so every
//     output receives its taps in exactly the reference's order (j ascending, starting from 0.0f).
//     The taps are bitwise symmetric (taps[j] == taps[N-1-j], checked on the host), so only
//     (N+1)/2 distinct products per row are formed; each is the same IEEE product the reference
//     computes.  With the march unrolled N times every accumulator index is a compile-time constant.
// No FMA anywhere: the pyramid is bit-identical to the reference's unfused arithmetic.
// dev-only ablation switches for tools/ubench/blur_abl.hip (0 in every product build)
#ifndef BLUR_ABL
#define BLUR_ABL 0
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


template <int N, int S> struct SubSplit {
    static constexpr int RB = (((N + S - 1) / S) + 1) & ~1;
    static constexpr int rows(int s) { return (N - s * RB) < RB ? (N - s * RB) : RB; }
    static constexpr int pairs(int s) { return (rows(s) + 1) / 2; }
    static constexpr int NPS = RB / 2;
    static_assert(N - (S - 1) * RB > 0, "empty last sub-block");
};

template <int N, int NT, int S> struct March2Geom {
    using SS = SubSplit<N, S>;
    static constexpr int TX = 2 * NT;
    static constexpr int C = (N & 1) ? N / 2 : N / 2 - 1;
    static constexpr int NPS = SS::NPS;
    static constexpr int COLS = TX + N - 1;
    static constexpr int PITCH = (COLS + 3) & ~3;
    static constexpr int NW = (N + 3 + 1) & ~1;
    static constexpr int LDS_BYTES = NPS * PITCH * 2 * 4;
    static constexpr int HALO = N - 1;
    static constexpr int NB = (NPS * HALO + NT - 1) / NT;
};


// ------------------------------------------------------------------------------------------
// Marching blur, team form (used for >= 15 taps on large planes; bit-identical to the one-block form of round 1).
// The N rows of one accumulator period are staged and filtered in S sub-blocks of RB rows (SubSplit), and the two
// passes run on different waves of the workgroup:
// One workgroup = 256 threads = an H team (waves 0-1) and a V team (waves 2-3) on one 256-column strip.
// Step g = (block, sub-block): the H team filters sub-block g horizontally in LDS buffer g % 3 while the V team
// marches sub-block g-1 vertically out of buffer (g-1) % 3 (accumulators, global stores) and then stages sub-block
// g+1 (prefetched registers -> buffer (g+1) % 3) and issues the loads of g+2.  One barrier per step.
template <int N, bool NORM, int S, int DT = 0, int HW = 2>
__global__ __launch_bounds__(64 * HW + 128) void blur_team_kernel(const void *__restrict__ in, float *__restrict__ out,
                                                          int W, int H, int nblocks, int last_subs, int rows_out,
                                                          TapsArg<N> taps, const uint32_t *__restrict__ mm,
                                                          float *__restrict__ next0,     // not null: also out[2y][2x] -> next0 (the next octave's plane 0)
                                                          int xcd_map, int prio) {
    constexpr int NT = 128;                       // threads per team
    using G = March2Geom<N, NT, S>;
    using SS = SubSplit<N, S>;
    static_assert(N & 1, "marching blur needs an odd tap count");
    constexpr int BUF = G::NPS * G::PITCH * 2;    // floats per LDS buffer
    extern __shared__ float4 smem4[];
    float *sbase = reinterpret_cast<float *>(smem4);
    // HW waves form the H team (threads 0 .. 64*HW-1), the last two waves the V team
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 64 * HW ? 1 : 0;     // wave-uniform (a scalar: the team split below is a scalar branch)
    const int tid = role ? (int)threadIdx.x - 64 * HW : (int)threadIdx.x;
    // Workgroup -> (strip, segment).  The dispatcher deals workgroup ids round-robin over the eight XCDs, each with its own L2:
    // with (strip, segment) = (blockIdx.x, blockIdx.y) horizontally adjacent strips -- which read the same halo columns at the
    // same time -- sit on different XCDs and every halo line comes out of HBM twice.  xcd_map: XCD c takes a contiguous
    // range of the strip-fastest order (xcd_contiguous), i.e. whole segment rows -- the neighbours' halo lines are L2 hits
    // (FETCH_SIZE of a 4096^2 launch: 11 taps 101 -> 82 MB, 27 taps 113 -> 91 MB for a 67 MB plane; what is left are the
    // warm-up rows, which the segment above reads a whole launch later).
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (xcd_map) {
        const int gx = (int)gridDim.x;
        const int M = xcd_contiguous(bx + gx * by, gx * (int)gridDim.y);
        by = M / gx; bx = M - by * gx;
    }
    const int x0 = bx * G::TX;
    // A segment outputs `rows_out` rows; it marches nblocks - 1 full accumulator periods of N rows and `last_subs` (1..S)
    // sub-blocks of the last one: just enough rows to complete its outputs, so that the host can pick the segment height
    // that balances the workgroups over the CUs instead of one quantised to multiples of N rows.
    const int ys = by * rows_out;
    const int yend = min(ys + rows_out, H);
    float mn = 0.f, range = 1.f;
    if (NORM) { mn = ord2f(mm[0]); range = ord2f(mm[1]) - mn; }

    // ---- V-team state: staging duty, accumulators, look-ahead registers
    const int gx_a = reflect_index(x0 - G::C + tid, W);
    const int gx_b = reflect_index(x0 - G::C + NT + tid, W);
    int hb_rp[G::NB], hb_col[G::NB], hb_gx[G::NB];
#pragma unroll
    for (int u = 0; u < G::NB; u++) {
        const int e = tid + NT * u;
        hb_rp[u] = (e < G::NPS * G::HALO) ? e / G::HALO : 1 << 20;
        hb_col[u] = G::TX + e % G::HALO;
        hb_gx[u] = reflect_index(x0 - G::C + hb_col[u], W);
    }
    auto ld = [&](unsigned byte_off) {
        if constexpr (DT == 0) return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(in) + byte_off);
        else return load_px<DT>(in, (size_t)(byte_off >> 2));
    };
    const unsigned W4 = (unsigned)W * 4u;
    auto norm2 = [&](f32x2 v) {
        if (NORM) { v.x = 255.0f * (v.x - mn) / range; v.y = 255.0f * (v.y - mn) / range; }
        return v;
    };
    f32x2 acc[N];
#pragma unroll
    for (int k = 0; k < N; k++) acc[k] = (f32x2){0.f, 0.f};
    const int gxo = x0 + 2 * tid;
    const bool vec_store = ((W & 1) == 0) && (gxo + 1 < W);
    f32x2 pa[G::NPS], pb[G::NPS], ph[G::NB];

    auto prefetch = [&](int blk, int sub, int np) {
        const int v0 = ys - G::C + blk * N + sub * SS::RB;
        if (BLUR_ABL & 4) {
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++) { pa[rp] = (f32x2){1.f, 2.f}; pb[rp] = (f32x2){3.f, 4.f}; }
#pragma unroll
            for (int u = 0; u < G::NB; u++) ph[u] = (f32x2){5.f, 6.f};
            return;
        }
        if (v0 >= 0 && v0 + 2 * np <= H) {
            unsigned oa = ((unsigned)v0 * (unsigned)W + (unsigned)gx_a) * 4u;
            unsigned ob = ((unsigned)v0 * (unsigned)W + (unsigned)gx_b) * 4u;
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp < np) {
                    pa[rp].x = ld(oa); pa[rp].y = ld(oa + W4);
                    pb[rp].x = ld(ob); pb[rp].y = ld(ob + W4);
                    oa += 2u * W4; ob += 2u * W4;
                }
        } else {
#pragma unroll
            for (int rp = 0; rp < G::NPS; rp++)
                if (rp < np) {
                    const unsigned r0 = (unsigned)reflect_index(v0 + 2 * rp, H) * W4, r1 = (unsigned)reflect_index(v0 + 2 * rp + 1, H) * W4;
                    pa[rp].x = ld(r0 + 4u * gx_a); pa[rp].y = ld(r1 + 4u * gx_a);
                    pb[rp].x = ld(r0 + 4u * gx_b); pb[rp].y = ld(r1 + 4u * gx_b);
                }
        }
#pragma unroll
        for (int u = 0; u < G::NB; u++) {
            ph[u] = (f32x2){0.f, 0.f};
            if (hb_rp[u] < np) {
                ph[u].x = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u], H) * W4 + 4u * hb_gx[u]);
                ph[u].y = ld((unsigned)reflect_index(v0 + 2 * hb_rp[u] + 1, H) * W4 + 4u * hb_gx[u]);
            }
        }
    };
    auto stage = [&](float *s, int np) {
#pragma unroll
        for (int rp = 0; rp < G::NPS; rp++)
            if (rp < np) {
                *reinterpret_cast<f32x2 *>(s + (rp * G::PITCH + tid) * 2) = norm2(pa[rp]);
                *reinterpret_cast<f32x2 *>(s + (rp * G::PITCH + NT + tid) * 2) = norm2(pb[rp]);
            }
#pragma unroll
        for (int u = 0; u < G::NB; u++)
            if (hb_rp[u] < np) *reinterpret_cast<f32x2 *>(s + (hb_rp[u] * G::PITCH + hb_col[u]) * 2) = norm2(ph[u]);
    };
    // horizontal pass of one sub-block (np row pairs) in place, by the HW waves of the H team (one wave per row pair)
    auto hpass = [&](float *s, int np) {
        for (int task = tid; task < ((BLUR_ABL & 1) ? 0 : np * (NT / 2)); task += 64 * HW) {
            const int rp = task / (NT / 2), t4 = task % (NT / 2);
            float *rowp = s + (rp * G::PITCH + 4 * t4) * 2;
            f32x2 w[G::NW];
            constexpr int PRE = 4;
#pragma unroll
            for (int k = 0; k < PRE && k < G::NW / 2; k++) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(rowp + 4 * k);
                w[2 * k] = v.xy; w[2 * k + 1] = v.zw;
            }
            f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
            for (int q = 0; q < N; q++) {
                if ((q & 1) == 0) {
                    const int k = q / 2 + PRE;
                    if (k < G::NW / 2) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(rowp + 4 * k);
                        w[2 * k] = v.xy; w[2 * k + 1] = v.zw;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float tp = taps.t[N - 1 - q];
                const f32x2 tp2 = {tp, tp};
                a0 = a0 + w[q] * tp2;
                a1 = a1 + w[q + 1] * tp2;
                a2 = a2 + w[q + 2] * tp2;
                a3 = a3 + w[q + 3] * tp2;
            }
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<f32x4 *>(rowp) = (f32x4){a0.x, a1.x, a0.y, a1.y};
            *reinterpret_cast<f32x4 *>(rowp + 4) = (f32x4){a2.x, a3.x, a2.y, a3.y};
        }
    };

    // vertical march over the rows of sub-block `sub` (compile-time after unrolling) of block blk
#define VPASS(sbuf, blk_, sub_)                                                                              \
    {                                                                                                        \
        const int np_ = SS::pairs(sub_), nrows_ = SS::rows(sub_);                                            \
        const int ybase_ = ys + (blk_) * N - (N - 1);                                                        \
        float *optr = out + ((ptrdiff_t)(ybase_ + (sub_) * SS::RB) * W + gxo);                              \
        f32x4 hv_next = *reinterpret_cast<const f32x4 *>((sbuf) + (2 * tid) * 2);                           \
        _Pragma("unroll") for (int rp = 0; rp < G::NPS; rp++) {                                              \
            if (rp < np_) {                                                                                  \
                const f32x4 hv = hv_next;                                                                    \
                if (rp + 1 < np_) hv_next = *reinterpret_cast<const f32x4 *>((sbuf) + ((rp + 1) * G::PITCH + 2 * tid) * 2); \
                __builtin_amdgcn_sched_barrier(0);                                                           \
                _Pragma("unroll") for (int half = 0; half < 2; half++) {                                     \
                    if (2 * rp + half < nrows_) {                                                            \
                        const int kk = (sub_) * SS::RB + 2 * rp + half;                                      \
                        const f32x2 h = half ? hv.zw : hv.xy;                                                \
                        _Pragma("unroll") for (int k = 0; k < ((BLUR_ABL & 2) ? 1 : (N + 1) / 2); k++) {     \
                            const f32x2 t2 = {taps.t[k], taps.t[k]};                                         \
                            const f32x2 prod = h * t2;                                                       \
                            const int slot_a = (kk - k + N) % N, slot_b = (kk - (N - 1 - k) + N) % N;        \
                            if (k == 0) acc[slot_a] = (f32x2){0.f, 0.f} + prod;                              \
                            else acc[slot_a] = acc[slot_a] + prod;                                           \
                            asm volatile("" : "+v"(acc[slot_a]));                                            \
                            if (k != N - 1 - k) { acc[slot_b] = acc[slot_b] + prod; asm volatile("" : "+v"(acc[slot_b])); } \
                        }                                                                                    \
                        const int done = (kk + 1) % N;                                                       \
                        const int y = ybase_ + kk;                                                           \
                        if (y >= ys && y < yend && !((BLUR_ABL & 8) && acc[done].x != 12345.678f)) {        \
                            if (vec_store) *reinterpret_cast<f32x2 *>(optr) = acc[done];                     \
                            else { if (gxo < W) optr[0] = acc[done].x; if (gxo + 1 < W) optr[1] = acc[done].y; } \
                            if (next0 && !(y & 1) && (y >> 1) < (H >> 1) && (gxo >> 1) < (W >> 1))           \
                                next0[(size_t)(y >> 1) * (W >> 1) + (gxo >> 1)] = acc[done].x;              \
                        }                                                                                    \
                        optr += W;                                                                           \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }

    // step (blk, sub) exists?  (only the last period is partial)
    auto exists = [&](int blk, int sub) { return blk < nblocks - 1 || (blk == nblocks - 1 && sub < last_subs); };
    // ---- prologue: the V team stages step 0 and looks ahead to step 1
    if (role == 1) {
        prefetch(0, 0, SS::pairs(0));
        stage(sbase, SS::pairs(0));
        if (S > 1) { if (exists(0, 1 % S)) prefetch(0, 1 % S, SS::pairs(1 % S)); }
        else if (exists(1, 0)) prefetch(1, 0, SS::pairs(0));
    }
    __syncthreads();
    int g = 0;                                     // step counter: buffer of step g is g % 3
    // Wave priority as negative feedback on progress (prio != 0).  The whole grid is resident at once and the SIMD arbiter serves
    // the OLDEST wave first: of the three workgroups of a CU the first one marches as if it were alone (its own step latency,
    // the SIMDs two thirds idle), the second gets what it leaves, and the third does most of its march alone at the end -- a
    // traced 27-tap launch: they end at 54 / 76 / 100 % of its time (tools/ubench/blur_var.hip, profiles/r06/blur_timeline.txt).
    // Every wave lowers its priority by one level per quarter of its march: a workgroup that is behind outranks one that is
    // ahead, the three advance together and end within 16 % of one another: 27 / 21 / 17 / 15 taps -8 / -6 / -5 / -4.5 % per
    // launch alone (11 taps: +1-2 % alone, -1 % inside a frame).  Thirds, eighths, levels dithered from step to step, a short
    // last phase: all worse than quarters.  s_setprio only orders waves that compete for the same SIMD; which launches ask for
    // it: siftmi.hip, launch_team.
    const int total_steps = (nblocks - 1) * S + last_subs;
    int prio_level = -1;
    auto set_prio = [&](int step) {
        if (!prio) return;
        const int q = (step * 4) / total_steps;    // quarter of the march this workgroup is in
        if (q == prio_level) return;
        prio_level = q;
        if (q <= 0) __builtin_amdgcn_s_setprio(3);
        else if (q == 1) __builtin_amdgcn_s_setprio(2);
        else if (q == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    };
    set_prio(0);
    // Each team runs its own copy of the step loop (same barrier count).  With one loop and a role test inside, the V team's
    // accumulators and look-ahead registers are live across the H team's code; apart, the H team's window registers and the V
    // team's state share one allocation (27 taps: 34 -> 6 spilled SGPRs, 17 taps: 96 -> 88 VGPRs; whole call -0.8 %).
    if (role == 0) {
        for (int blk = 0; blk < nblocks; blk++) {
#pragma unroll
            for (int sub = 0; sub < S; sub++) {
                if (blk == nblocks - 1 && sub >= last_subs) break;      // workgroup uniform
                hpass(sbase + (g % 3) * BUF, SS::pairs(sub));
                __syncthreads();
                g++;
                set_prio(g);
            }
        }
        return;
    }
    for (int blk = 0; blk < nblocks; blk++) {
#pragma unroll
        for (int sub = 0; sub < S; sub++) {
            if (blk == nblocks - 1 && sub >= last_subs) break;      // workgroup uniform
            // (1) vertical march of the previous step
            if (g > 0) {
                float *prev = sbase + ((g + 2) % 3) * BUF;
                if (sub == 0) { VPASS(prev, blk - 1, S - 1) } else { VPASS(prev, blk, (sub + S - 1) % S) }
            }
            // (2) stage step g+1 (already in registers) and look ahead to g+2
            const int sub1 = (sub + 1) % S, blk1 = blk + (sub + 1) / S;
            if (exists(blk1, sub1)) {
                float *nxt = sbase + ((g + 1) % 3) * BUF;
                stage(nxt, SS::pairs(sub1));
                const int sub2 = (sub + 2) % S;
                const int blk2 = blk + (sub + 2) / S;
                if (exists(blk2, sub2)) prefetch(blk2, sub2, SS::pairs(sub2));
            }
            __syncthreads();
            g++;
            set_prio(g);
        }
    }
    // ---- epilogue: vertical march of the last step, (nblocks - 1, last_subs - 1)
    {
        float *prev = sbase + ((g + 2) % 3) * BUF;
        if (last_subs >= S) { VPASS(prev, nblocks - 1, S - 1) }
        if constexpr (S > 1) { if (last_subs == 1) { VPASS(prev, nblocks - 1, 0) } }
        if constexpr (S > 2) { if (last_subs == 2) { VPASS(prev, nblocks - 1, 1) } }
        if constexpr (S > 3) { if (last_subs == 3) { VPASS(prev, nblocks - 1, 2) } }
    }
#undef VPASS
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
// A thread moves four consecutive outputs (two 16-byte loads, one 16-byte store where the pitches allow it) of two rows;
// one output per thread made the 4096 -> 2048 hand-off a 16 k-workgroup launch that took 47-57 us for 84 MB of traffic.
__global__ __launch_bounds__(256) void shrink_kernel(const float *__restrict__ in, float *__restrict__ out, int LW, int SW, int SH) {
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 2;
    if (x >= SW) return;
    const bool vec_in = (LW & 1) == 0 && x + 3 < SW;     // row starts 2y * LW + 2x are then multiples of 4 floats
    const bool vec_out = (SW & 3) == 0;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = y0 + r;
        if (y >= SH) break;
        const float *src = in + (size_t)(2 * y) * LW + 2 * x;
        float *dst = out + (size_t)y * SW + x;
        if (vec_in) {
            const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
            if (vec_out) *reinterpret_cast<float4 *>(dst) = make_float4(a.x, a.z, b.x, b.z);
            else { dst[0] = a.x; dst[1] = a.z; dst[2] = b.x; dst[3] = b.z; }
        } else {
            for (int k = 0; k < 4 && x + k < SW; k++) dst[k] = src[2 * k];
        }
    }
}
inline dim3 shrink_grid(int SW, int SH) { return dim3((unsigned)((SW + 255) / 256), (unsigned)((SH + 7) / 8)); }

// normalise only (stage replay of preprocess.cl:239-252)
template <int DT = 0>
__global__ void normalize_kernel(const void *__restrict__ in, float *__restrict__ out, int64_t n,
                                 const uint32_t *__restrict__ mm) {
    const float mn = ord2f(mm[0]), range = ord2f(mm[1]) - mn;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = 255.0f * (load_px<DT>(in, (size_t)i) - mn) / range;
}

}  // namespace siftk
