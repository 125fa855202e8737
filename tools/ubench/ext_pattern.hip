// dev: the READ PATTERN of the extrema kernel alone (no arithmetic but a sum): one wave marches a strip of `SW` columns down `rows`
// rows of six 4096^2 planes, one dword per lane and plane and row, LA rows of loads in flight ahead of the row being consumed.
// Against it: the same bytes as a grid-stride stream (tools/ubench/read_bw.hip: 65 us = 6.2 TB/s).
//   hipcc --offload-arch=gfx950 -O3 -o ext_pattern_bench ext_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Planes { const float *p[6]; };
// ORDER 0: strips of a band side by side (wid % nx), bands one after the other; 1: column-major (wid % ny rows first)
template <int LA, int SW, int ORDER, int WAVES> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))
void march(Planes b, int W, int H, int rows, float *out) {
    const int lane = threadIdx.x & 63;
    const int nx = (W + SW - 1) / SW, ny = (H + rows - 1) / rows;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nx * ny) return;
    const int sx = ORDER ? wid / ny : wid % nx, sy = ORDER ? wid % ny : wid / nx;
    const int x = min(sx * SW + lane, W - 1);
    const int ya = sy * rows, yb = min(ya + rows, H);
    float ring[LA][6];
    unsigned off = ((unsigned)ya * (unsigned)W + (unsigned)x) * 4u;
    const unsigned pitch = (unsigned)W * 4u;
    auto ld = [&](int k, unsigned o) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(b.p[k]) + o); };
#pragma unroll
    for (int a = 0; a < LA; a++) {
#pragma unroll
        for (int k = 0; k < 6; k++) ring[a][k] = ld(k, off);
        off += pitch;
    }
    float acc = 0.f;
    for (int y = ya; y < yb; y += LA) {
#pragma unroll
        for (int a = 0; a < LA; a++) {
            float v[6];
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = ring[a][k];
            if (y + a + LA < yb) {
#pragma unroll
                for (int k = 0; k < 6; k++) ring[a][k] = ld(k, off);
                off += pitch;
            }
#pragma unroll
            for (int k = 0; k < 6; k++) acc += v[k];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int LA, int SW, int ORDER, int WAVES> int run(Planes b, float *out, int rows) {
    const int W = 4096, H = 4096;
    const int nx = (W + SW - 1) / SW, ny = (H + rows - 1) / rows;
    const int blocks = (nx * ny + 3) / 4;
    hipEvent_t a, c; CK(hipEventCreate(&a)); CK(hipEventCreate(&c));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((march<LA, SW, ORDER, WAVES>), dim3(blocks), dim3(256), 0, 0, b, W, H, rows, out);
    CK(hipEventRecord(a, 0));
    for (int w = 0; w < 10; w++) hipLaunchKernelGGL((march<LA, SW, ORDER, WAVES>), dim3(blocks), dim3(256), 0, 0, b, W, H, rows, out);
    CK(hipEventRecord(c, 0)); CK(hipEventSynchronize(c));
    float ms; CK(hipEventElapsedTime(&ms, a, c));
    printf("look-ahead %d rows, strip %2d columns, %s order, %d waves/SIMD wanted, %3d-row strips (%5d waves): %6.1f us  %5.2f TB/s\n", LA, SW,
           ORDER ? "column" : "row   ", WAVES, rows, nx * ny, 100.0 * ms, 6.0 * W * H * 4 / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
int main() {
    const size_t plane = (size_t)4096 * 4096 * 4;
    char *buf; float *out; CK(hipMalloc(&buf, 6 * plane)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 0, 6 * plane));
    Planes b; for (int k = 0; k < 6; k++) b.p[k] = (const float *)(buf + k * plane);
    for (int rows : {64, 32, 128}) {
        run<1, 62, 0, 5>(b, out, rows); run<2, 62, 0, 5>(b, out, rows); run<4, 62, 0, 5>(b, out, rows);
        run<1, 64, 0, 5>(b, out, rows); run<2, 64, 0, 5>(b, out, rows); run<4, 64, 0, 4>(b, out, rows);
        run<1, 62, 1, 5>(b, out, rows); run<2, 62, 1, 5>(b, out, rows);
        run<2, 64, 0, 8>(b, out, rows); run<1, 64, 0, 8>(b, out, rows);
    }
    return 0;
}
