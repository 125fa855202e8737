// ubench: in which order does ONE ds_add_f32 instruction apply the lanes that hit the same LDS address?
// If it is ascending lane order (and IEEE round-to-nearest, denormals kept), the ordered per-bin sums of the descriptor
// need no routing at all.  Each wave: 64 lanes add a value to one of `nb` addresses; the host replays ascending-lane order
// (and descending, for contrast) in binary32 and compares bit for bit.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/lds_fadd_order tools/ubench/lds_fadd_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __attribute__((address_space(3))) float lds_f32;
__global__ __launch_bounds__(256) void fadd_kernel(const int *bin, const float *val, float *out, int nb, int rounds) {
    __shared__ float acc[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave;
    acc[wave][lane] = 0.0f;
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < rounds; r++) {
        const int b = bin[(size_t)(gw * rounds + r) * 64 + lane];
        const float v = val[(size_t)(gw * rounds + r) * 64 + lane];
        (void)__hip_atomic_fetch_add(&acc[wave][b], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    out[(size_t)gw * 64 + lane] = acc[wave][lane];
    (void)nb;
}

int main(int argc, char **argv) {
    const int waves = 4096 * 4, rounds = 8;
    int bad_asc_total = 0, bad_desc_total = 0, cases = 0;
    for (int mode = 0; mode < 4; mode++) {
        const int nb = mode == 0 ? 1 : (mode == 1 ? 4 : (mode == 2 ? 16 : 64));
        std::vector<int> bin((size_t)waves * rounds * 64); std::vector<float> val(bin.size());
        srand(1234 + mode);
        for (size_t i = 0; i < bin.size(); i++) {
            bin[i] = rand() % nb;
            // wide range of magnitudes, non-negative like the descriptor's values, some denormals and zeros
            const int e = rand() % 40 - 30;
            float v = ldexpf((float)(rand() % 16777216) / 16777216.0f + 1.0f, e);
            const int k = rand() % 64;
            if (k == 0) v = 0.0f;
            if (k == 1) v = ldexpf(1.0f, -140 + rand() % 10);      // denormal
            if (argc > 1 && (rand() & 1)) v = -v;                    // with an argument: mixed signs
            val[i] = v;
        }
        int *dbin; float *dval, *dout;
        CHK(hipMalloc(&dbin, bin.size() * 4)); CHK(hipMalloc(&dval, val.size() * 4)); CHK(hipMalloc(&dout, (size_t)waves * 64 * 4));
        CHK(hipMemcpy(dbin, bin.data(), bin.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dval, val.data(), val.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(fadd_kernel, dim3(waves / 4), dim3(256), 0, 0, dbin, dval, dout, nb, rounds);
        CHK(hipDeviceSynchronize());
        std::vector<float> out((size_t)waves * 64);
        CHK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        int bad_asc = 0, bad_desc = 0, denorm_in = 0, differ = 0;
        for (int w = 0; w < waves; w++) {
            float a[64], d[64];
            for (int b = 0; b < 64; b++) a[b] = d[b] = 0.0f;
            for (int r = 0; r < rounds; r++) {
                const size_t o = (size_t)(w * rounds + r) * 64;
                for (int l = 0; l < 64; l++) { volatile float t = a[bin[o + l]] + val[o + l]; a[bin[o + l]] = t; }
                for (int l = 63; l >= 0; l--) { volatile float t = d[bin[o + l]] + val[o + l]; d[bin[o + l]] = t; }
            }
            for (int b = 0; b < nb; b++) {
                const float g = out[(size_t)w * 64 + b];
                if (memcmp(&g, &a[b], 4)) bad_asc++;
                if (memcmp(&g, &d[b], 4)) bad_desc++;
                if (memcmp(&a[b], &d[b], 4)) differ++;
            }
        }
        for (float v : val) if (v != 0.0f && fabsf(v) < 1.17549435e-38f) denorm_in++;
        printf("%2d addresses: %d sums; ascending-lane replay differs in %d, descending in %d (the two replays differ from each other in %d); %d denormal inputs\n",
               nb, waves * nb, bad_asc, bad_desc, differ, denorm_in);
        bad_asc_total += bad_asc; bad_desc_total += bad_desc; cases += waves * nb;
        hipFree(dbin); hipFree(dval); hipFree(dout);
    }
    printf("TOTAL: %d sums, ascending mismatches %d, descending mismatches %d\n", cases, bad_asc_total, bad_desc_total);
    return 0;
}
