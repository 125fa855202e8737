// k_align.hpp -- affine warp with bilinear interpolation (transform.cl:22-110 `transform`, 116-204 `transform_RGB`).
//
// HBM bound: 1 gathered read + 1 write of the plane (8 B / pixel for f32, 6 B / pixel for RGB8).  One thread
// per output pixel in a 64 x 4 tile; for the near-identity matrices LinearAlign produces the four taps of a
// wave fall in two image rows, so the gather is served by the vector L1 / L2 and HBM sees each line once.
//
// Parity notes (all reproduced bit for bit):
//   * matrix rows act on (y, x): ty = m0*y + m1*x + off0, tx = m2*y + m3*x + off1 (dot() = mul, mul, add; then +off)
//   * taps right of / below the image are replaced by `fill`, and the result is `fill` whenever
//     tx >= W - 0.5 or ty >= H - 0.5 (transform.cl:101-106) or the point is outside [0,W) x [0,H)
//   * mode 1 = bilinear, anything else = nearest-lower tap
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace siftk {

struct AffineArgs { float m0, m1, m2, m3, off0, off1, fill; int mode; };

__device__ __forceinline__ float bilinear_mix(float tx, float ty, int tx_prev, int ty_prev, float p, float px, float py, float pn) {
    const float fx1 = (float)(tx_prev + 1) - tx, fx0 = tx - (float)tx_prev;
    const float i1 = fx1 * p + fx0 * px;
    const float i2 = fx1 * py + fx0 * pn;
    return ((float)(ty_prev + 1) - ty) * i1 + (ty - (float)ty_prev) * i2;
}

__global__ __launch_bounds__(256) void transform_kernel(const float *__restrict__ image, float *__restrict__ out, AffineArgs a,
                                                        int W, int H, int OW, int OH) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= OW || y >= OH) return;
    float tx = a.m2 * (float)y + a.m3 * (float)x;
    float ty = a.m0 * (float)y + a.m1 * (float)x;
    tx += a.off1; ty += a.off0;
    float interp = a.fill;
    if (0.0f <= tx && tx < (float)W && 0.0f <= ty && ty < (float)H) {
        const int tx_prev = (int)tx, ty_prev = (int)ty;
        const float *r0 = image + (size_t)ty_prev * W + tx_prev;
        const float p = r0[0];
        if (a.mode == 1) {
            const bool xin = tx_prev + 1 < W, yin = ty_prev + 1 < H;
            const float px = xin ? r0[1] : a.fill;
            const float py = yin ? r0[W] : a.fill;
            const float pn = (xin && yin) ? r0[W + 1] : a.fill;
            interp = bilinear_mix(tx, ty, tx_prev, ty_prev, p, px, py, pn);
        } else interp = p;
    }
    if (tx >= (float)W + -0.5f) interp = a.fill;
    if (ty >= (float)H + -0.5f) interp = a.fill;
    out[(size_t)y * OW + x] = interp;
}

// RGB8: one thread produces 4 consecutive output pixels = 12 bytes, stored as three dwords when the row segment is
// 4-byte aligned (byte stores with stride 3 were store-issue bound: 0.13 ms at 4096^2 against 0.04 ms for f32).
__device__ __forceinline__ void transform_rgb_pixel(const uint8_t *__restrict__ image, const AffineArgs &a, int W, int H,
                                                    int x, int y, uint8_t rgb[3]) {
    float tx = a.m2 * (float)y + a.m3 * (float)x;
    float ty = a.m0 * (float)y + a.m1 * (float)x;
    tx += a.off1; ty += a.off0;
    const bool inside = (0.0f <= tx && tx < (float)W && 0.0f <= ty && ty < (float)H);
    const bool cut = (tx >= (float)W + -0.5f) || (ty >= (float)H + -0.5f);
    const int tx_prev = inside ? (int)tx : 0, ty_prev = inside ? (int)ty : 0;
    const bool xin = tx_prev + 1 < W, yin = ty_prev + 1 < H;
    // the 2 x 2 taps are 6 + 6 contiguous bytes: two unaligned 8-byte loads (gfx950 global loads need no alignment)
    // instead of 12 byte loads; the last pixels of the buffer, where 8 bytes would overrun it, take the byte path
    const size_t base = 3 * ((size_t)ty_prev * W + tx_prev), total = 3 * (size_t)W * H;
    const uint8_t *r0 = image + base;
    uint64_t q0 = 0, q1 = 0;
    const bool wide0 = base + 8 <= total, wide1 = yin && base + 3 * (size_t)W + 8 <= total;
    if (inside) {
        if (wide0) __builtin_memcpy(&q0, r0, 8);
        else for (int k = 0; k < 6 && base + k < total; k++) q0 |= (uint64_t)r0[k] << (8 * k);
        if (wide1) __builtin_memcpy(&q1, r0 + 3 * (size_t)W, 8);
        else if (yin) for (int k = 0; k < 6 && base + 3 * (size_t)W + k < total; k++) q1 |= (uint64_t)r0[3 * (size_t)W + k] << (8 * k);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float interp = a.fill;
        if (inside) {
            const float p = (float)(uint32_t)((q0 >> (8 * c)) & 0xff);
            if (a.mode == 1) {
                const float px = xin ? (float)(uint32_t)((q0 >> (8 * c + 24)) & 0xff) : a.fill;
                const float py = yin ? (float)(uint32_t)((q1 >> (8 * c)) & 0xff) : a.fill;
                const float pn = (xin && yin) ? (float)(uint32_t)((q1 >> (8 * c + 24)) & 0xff) : a.fill;
                interp = bilinear_mix(tx, ty, tx_prev, ty_prev, p, px, py, pn);
            } else interp = p;
        }
        if (cut) interp = a.fill;
        rgb[c] = (uint8_t)interp;
    }
}

__global__ __launch_bounds__(256) void transform_rgb_kernel(const uint8_t *__restrict__ image, uint8_t *__restrict__ out, AffineArgs a,
                                                            int W, int H, int OW, int OH) {
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= OW || y >= OH) return;
    uint8_t *o = out + 3 * ((size_t)y * OW + x0);
    if (x0 + 4 <= OW && (reinterpret_cast<uintptr_t>(o) & 3) == 0) {
        uint8_t b[12];
#pragma unroll
        for (int k = 0; k < 4; k++) transform_rgb_pixel(image, a, W, H, x0 + k, y, b + 3 * k);
        uint32_t w[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
            w[k] = (uint32_t)b[4 * k] | ((uint32_t)b[4 * k + 1] << 8) | ((uint32_t)b[4 * k + 2] << 16) | ((uint32_t)b[4 * k + 3] << 24);
        uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
        o32[0] = w[0]; o32[1] = w[1]; o32[2] = w[2];
    } else {
        for (int k = 0; k < 4 && x0 + k < OW; k++) {
            uint8_t b[3];
            transform_rgb_pixel(image, a, W, H, x0 + k, y, b);
            o[3 * k] = b[0]; o[3 * k + 1] = b[1]; o[3 * k + 2] = b[2];
        }
    }
}

}  // namespace siftk
