// siftmath.hpp -- "siftmath v1": bit-reproducible float transcendentals for the SIFT hot path.
//
// The reference's OpenCL kernels call exp / sin / cos / atan2 / pow (orientation_cpu.cl:88-91,
// keypoints_cpu.cl:60-74, image.cl:77,354, gaussian.cl:69); OpenCL leaves their last bits to the
// implementation.  For results that are identical on every device we evaluate each function in
// IEEE binary64 with a fixed sequence of add/mul/div (no FMA contraction: build with
// -ffp-contract=off), then round once to binary32.  FP64 is cheap on CDNA4 (78 TFLOP/s), and
// these calls are per keypoint-window sample, not per pixel.  The result is the correctly rounded
// f32 value except within ~2^-48 of a rounding boundary.
//
// Specification (identical constants and operation order are restated independently by the test
// oracle): exp: x = k ln2 + r, Taylor degree 14 in r; exp2: r = (y-k) ln2; sin/cos: Cody-Waite
// pi/2 reduction (33-bit hi part), Taylor to r^19 / r^20; atan2: octant table atan(k/8), k from an
// exactly rounded f32 quotient, t = (num - c den)/(den + c num), Taylor to t^17.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SM_HD __host__ __device__ __forceinline__

#define SM_PI_F 3.14159274101257f    // OpenCL M_PI_F
#define SM_1_PI_F 0.31830987334251f  // OpenCL M_1_PI_F

namespace siftmath {

SM_HD double from_bits(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; __builtin_memcpy(&d, &u, 8); return d;
#endif
}

// round to nearest-even integer for |z| < 2^51, additions only
SM_HD double rint_magic(double z) {
    // v_rndne_f64 on the device / roundsd on the host: round-half-even, == (z + 1.5*2^52) - 1.5*2^52
    return __builtin_rint(z);
}

SM_HD double exp_core(double r) {  // exp(r), |r| <= 0.35
    double p = 0x1.93974a8c07c9dp-37;
    p = p * r + 0x1.6124613a86d09p-33;
    p = p * r + 0x1.1eed8eff8d898p-29;
    p = p * r + 0x1.ae64567f544e4p-26;
    p = p * r + 0x1.27e4fb7789f5cp-22;
    p = p * r + 0x1.71de3a556c734p-19;
    p = p * r + 0x1.a01a01a01a01ap-16;
    p = p * r + 0x1.a01a01a01a01ap-13;
    p = p * r + 0x1.6c16c16c16c17p-10;
    p = p * r + 0x1.1111111111111p-7;
    p = p * r + 0x1.5555555555555p-5;
    p = p * r + 0x1.5555555555555p-3;
    p = p * r + 0x1.0p-1;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p;
}

SM_HD double pow2i(int k) { return from_bits((uint64_t)(k + 1023) << 52); }

SM_HD float expf_(float xf) {
    if (xf != xf) return xf;
    if (xf > 89.0f) return __builtin_inff();
    if (xf < -104.0f) return 0.0f;
    double x = (double)xf;
    double kd = rint_magic(x * 0x1.71547652b82fep+0);
    double r = (x - kd * 0x1.62e42fee00000p-1) - kd * 0x1.a39ef35793c76p-33;
    double v = exp_core(r) * pow2i((int)kd);
    return (float)v;
}

SM_HD float exp2f_(float yf) {
    if (yf != yf) return yf;
    if (yf > 128.0f) return __builtin_inff();
    if (yf < -150.0f) return 0.0f;
    double y = (double)yf;
    double kd = rint_magic(y);
    double r = (y - kd) * 0x1.62e42fefa39efp-1;
    double v = exp_core(r) * pow2i((int)kd);
    return (float)v;
}

SM_HD void sincosf_(float xf, float *sn, float *cs) {
    if (xf != xf || xf - xf != 0.0f) { *sn = xf - xf; *cs = xf - xf; return; }
    double x = (double)xf;
    double kd = rint_magic(x * 0x1.45f306dc9c883p-1);
    double r = (x - kd * 0x1.921fb54400000p+0) - kd * 0x1.0b4611a626331p-34;
    double r2 = r * r;
    double ps = 0x1.2f49b46814157p-57;
    ps = 0x1.952c77030ad4ap-49 - ps * r2;
    ps = 0x1.ae7f3e733b81fp-41 - ps * r2;
    ps = 0x1.6124613a86d09p-33 - ps * r2;
    ps = 0x1.ae64567f544e4p-26 - ps * r2;
    ps = 0x1.71de3a556c734p-19 - ps * r2;
    ps = 0x1.a01a01a01a01ap-13 - ps * r2;
    ps = 0x1.1111111111111p-7 - ps * r2;
    ps = 0x1.5555555555555p-3 - ps * r2;
    double s = r - (ps * r2) * r;
    double pc = 0x1.e542ba4020225p-62;
    pc = 0x1.6827863b97d97p-53 - pc * r2;
    pc = 0x1.ae7f3e733b81fp-45 - pc * r2;
    pc = 0x1.93974a8c07c9dp-37 - pc * r2;
    pc = 0x1.1eed8eff8d898p-29 - pc * r2;
    pc = 0x1.27e4fb7789f5cp-22 - pc * r2;
    pc = 0x1.a01a01a01a01ap-16 - pc * r2;
    pc = 0x1.6c16c16c16c17p-10 - pc * r2;
    pc = 0x1.5555555555555p-5 - pc * r2;
    pc = 0x1.0p-1 - pc * r2;
    double c = 1.0 - pc * r2;
    int q = (int)kd & 3;
    double so = (q & 1) ? c : s;
    double co = (q & 1) ? s : c;
    if (q == 1 || q == 2) co = -co;
    if (q >= 2) so = -so;
    *sn = (float)so;
    *cs = (float)co;
}

SM_HD float atan2f_(float yf, float xf) {
    if (yf != yf || xf != xf || yf - yf != 0.0f || xf - xf != 0.0f) return (yf - yf) + (xf - xf);
    double ay = (double)__builtin_fabsf(yf), ax = (double)__builtin_fabsf(xf);
    double res;
    if (ay == 0.0 && ax == 0.0) {
        res = 0.0;
    } else {
        bool swap = ay > ax;
        double num = swap ? ax : ay, den = swap ? ay : ax;
        float af = (float)num / (float)den;      // exactly rounded f32 division
        int k = (int)(af * 8.0f + 0.5f);
        double c = (double)k * 0.125;
        double t = (num - c * den) / (den + c * num);
        double t2 = t * t;
        double p = 0x1.e1e1e1e1e1e1ep-5;
        p = 0x1.1111111111111p-4 - p * t2;
        p = 0x1.3b13b13b13b14p-4 - p * t2;
        p = 0x1.745d1745d1746p-4 - p * t2;
        p = 0x1.c71c71c71c71cp-4 - p * t2;
        p = 0x1.2492492492492p-3 - p * t2;
        p = 0x1.999999999999ap-3 - p * t2;
        p = 0x1.5555555555555p-2 - p * t2;
        double at = t - (p * t2) * t;
        double tab = 0.0;
        tab = (k == 1) ? 0x1.fd5ba9aac2f6ep-4 : tab;
        tab = (k == 2) ? 0x1.f5b75f92c80ddp-3 : tab;
        tab = (k == 3) ? 0x1.6f61941e4def1p-2 : tab;
        tab = (k == 4) ? 0x1.dac670561bb4fp-2 : tab;
        tab = (k == 5) ? 0x1.1e00babdefeb4p-1 : tab;
        tab = (k == 6) ? 0x1.4978fa3269ee1p-1 : tab;
        tab = (k == 7) ? 0x1.700a7c5784634p-1 : tab;
        tab = (k >= 8) ? 0x1.921fb54442d18p-1 : tab;
        res = tab + at;
        if (swap) res = 0x1.921fb54442d18p+0 - res;
    }
    if (__builtin_signbitf(xf)) res = 0x1.921fb54442d18p+1 - res;
    if (__builtin_signbitf(yf)) res = -res;
    return (float)res;
}


// ---------------------------------------------------------------------------------------------------------------
// Device fast paths (Ziv's strategy).  expf_ / atan2f_ above DEFINE the values: a fixed binary64 evaluation, rounded
// once to binary32, which is the correctly rounded result unless the binary64 value lies within ~2^-48 (relative) of
// a binary32 rounding boundary.  The functions below compute the same quantity with fewer, fused operations
// (error <= 2^-45 relative, see the notes at each) and return its rounding only when the binary64 value is farther
// than 2^-39 from every rounding boundary -- then both evaluations sit on the same side of the boundary and round to
// the same binary32 number.  Otherwise (probability 2^-15 per call), and outside the guarded argument range, the
// defining function is evaluated.  Results are therefore bit-identical to expf_ / atan2f_ for every argument
// (checked on 10^7 arguments per function against the CPU oracle, tests/test_gpu_parity.py).
#if defined(__HIPCC__)

// T[swap + 2 * (x < 0)][k]: the octant fold applied to atan(k / 8), correctly rounded:
//   {atan(k/8), pi/2 - atan(k/8), pi - atan(k/8), pi/2 + atan(k/8)};  atan2 = T -/+ atan(t) (minus for folds 1 and 2)
__device__ const double c_atan_fold[36] = {
    0x0.0p+0, 0x1.fd5ba9aac2f6ep-4, 0x1.f5b75f92c80ddp-3, 0x1.6f61941e4def1p-2, 0x1.dac670561bb4fp-2, 0x1.1e00babdefeb4p-1, 0x1.4978fa3269ee1p-1, 0x1.700a7c5784634p-1, 0x1.921fb54442d18p-1,
    0x1.921fb54442d18p+0, 0x1.7249faa996a21p+0, 0x1.5368c951e9cfdp+0, 0x1.3647503caf55cp+0, 0x1.1b6e192ebbe44p+0, 0x1.031f57e54adbep+0, 0x1.dac670561bb4fp-1, 0x1.b434ee31013fdp-1, 0x1.921fb54442d18p-1,
    0x1.921fb54442d18p+1, 0x1.8234d7f6ecb9dp+1, 0x1.72c43f4b1650ap+1, 0x1.643382c07913ap+1, 0x1.56c6e7397f5aep+1, 0x1.4a9f8694c6d6bp+1, 0x1.3fc176b7a8560p+1, 0x1.361d162e61b8bp+1, 0x1.2d97c7f3321d2p+1,
    0x1.921fb54442d18p+0, 0x1.b1f56fdeef00fp+0, 0x1.d0d6a1369bd34p+0, 0x1.edf81a4bd64d4p+0, 0x1.0468a8ace4df6p+1, 0x1.109009519d639p+1, 0x1.1b6e192ebbe44p+1, 0x1.251279b802819p+1, 0x1.2d97c7f3321d2p+1};

// every thread of the block copies the fold table into its LDS copy (36 doubles); the caller synchronises
__device__ __forceinline__ void load_atan_fold(double *lds_tab) {
    for (int i = threadIdx.x; i < 36; i += blockDim.x) lds_tab[i] = c_atan_fold[i];
}

// binary64 value (normal binary32 range) farther than 2^13 units of 2^-52 from the round-to-nearest boundary of binary32?
__device__ __forceinline__ bool f32_rounding_is_safe(double v) {
    // the 29 discarded bits against the half-way pattern 2^28: |low - 2^28| > 8192 as one unsigned range test
    // (and / add / compare: three instructions; round 3 took the absolute value through min / max / sub: five)
    const unsigned low = (unsigned)__double2loint(v) & 0x1fffffffu;
    return (low - (0x10000000u - 8192u)) > 16384u;
}

// p * r + c as ONE v_fma_f64 with all three operands in registers.  Written out because the compiler selects the
// two-address v_fmac_f64 for a Horner step and, the coefficient being live across the loop, copies it into the
// destination first (v_mov_b64 + v_fmac_f64: 13 extra moves per batch of window samples, round-4 disassembly).
__device__ __forceinline__ double fma3(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}

// exp(x): one-step reduction x = k ln2 + r with a fused two-part ln2 (|r| <= 0.3466, reduction error < 2^-60),
// Taylor degree 11 by fused Horner (truncation r^12/12! < 2^-47 relative), scaling by an exponent-field add.
// The *_try forms are branch free: they return the candidate and whether it may be used, so that a caller can run
// several of them in one basic block (their dependent binary64 chains then interleave) and branch once.
__device__ __forceinline__ float expf_fast_try(float xf, bool &ok) {
    const double x = (double)xf;
    const double kd = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = __builtin_fma(-kd, 0x1.62e42fee00000p-1, x);
    r = __builtin_fma(-kd, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.ae64567f544e4p-26;
    p = fma3(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma3(p, r, 0x1.71de3a556c734p-19);
    p = fma3(p, r, 0x1.a01a01a01a01ap-16);
    p = fma3(p, r, 0x1.a01a01a01a01ap-13);
    p = fma3(p, r, 0x1.6c16c16c16c17p-10);
    p = fma3(p, r, 0x1.1111111111111p-7);
    p = fma3(p, r, 0x1.5555555555555p-5);
    p = fma3(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0x1.0p-1);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    // p in [0.70, 1.42], |k| <= 116: adding k to the exponent field is an exact scaling
    const double v = __hiloint2double(__double2hiint(p) + (int)kd * (1 << 20), __double2loint(p));
    // usable: argument inside the guarded range (result in the normal binary32 range, not NaN) and a safe rounding
    ok = (__builtin_fabsf(xf) <= 80.0f) && f32_rounding_is_safe(v);
    return (float)v;
}
__device__ __forceinline__ float expf_fast(float xf) {
    bool ok;
    const float v = expf_fast_try(xf, ok);
    return ok ? v : expf_(xf);
}

// atan2(y, x) for finite non-zero arguments of comparable size (the gradient components of a window sample):
// the same octant / table decomposition as atan2f_ with k taken from an approximate quotient (any k with
// |num/den - k/8| <= 1/16 + 2^-20 serves), t = (num - c den) / (den + c num) by a Newton reciprocal (two steps from the
// hardware seed: error < 2^-50), odd Taylor polynomial through t^11 (|t| <= 0.0626: truncation < 2^-51 relative), and
// the three reflections of atan2f_ folded into one table value (c_atan_fold).  `fold` is the block's LDS copy.
// (in two halves: the table value is requested from LDS by the first and used by the second, so that a caller can put
// other work between them -- the descriptor kernel's pipelined batch loop does)
struct Atan2Try { double at, base; bool in_range, sub, neg; };
__device__ __forceinline__ void atan2f_fast_begin(float yf, float xf, const double *fold, Atan2Try &T) {
    const float ay = __builtin_fabsf(yf), ax = __builtin_fabsf(xf);
    const float mx = __builtin_fmaxf(ax, ay), mn = __builtin_fminf(ax, ay);
    // guarded range: no zero / huge / NaN operand (fmin / fmax drop a NaN), no sub-normal result
    T.in_range = ax <= 1e18f && ay <= 1e18f && mn >= 1e-18f;
    const bool swap = ay > ax;
    int k = (int)(mn * __builtin_amdgcn_rcpf(mx) * 8.0f + 0.5f);               // 0..8 inside the guarded range
    k = k < 0 ? 0 : (k > 8 ? 8 : k);                                           // (keeps the table index in bounds outside it)
    const int f = (swap ? 1 : 0) + (__builtin_signbitf(xf) ? 2 : 0);
    T.base = fold[f * 9 + k];
    T.sub = (f == 1 || f == 2);
    T.neg = __builtin_signbitf(yf);
    const double c = (double)((float)k * 0.125f);
    const double num = (double)mn, den = (double)mx;
    const double tn = __builtin_fma(-c, den, num), td = __builtin_fma(c, num, den);
    double r = __builtin_amdgcn_rcp(td);
    r = __builtin_fma(__builtin_fma(-td, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-td, r, 1.0), r, r);
    const double t = tn * r;
    const double t2 = t * t;
    double p = -0x1.745d1745d1746p-4;
    p = fma3(p, t2, 0x1.c71c71c71c71cp-4);
    p = fma3(p, t2, -0x1.2492492492492p-3);
    p = fma3(p, t2, 0x1.999999999999ap-3);
    p = fma3(p, t2, -0x1.5555555555555p-2);
    T.at = __builtin_fma(t * t2, p, t);               // atan(t), same sign as t
}
__device__ __forceinline__ float atan2f_fast_end(const Atan2Try &T, bool &ok) {
    const double res = T.sub ? T.base - T.at : T.base + T.at;
    ok = T.in_range && f32_rounding_is_safe(res);
    const float out = (float)res;
    return T.neg ? -out : out;
}
__device__ __forceinline__ float atan2f_fast_try(float yf, float xf, const double *fold, bool &ok) {
    Atan2Try T;
    atan2f_fast_begin(yf, xf, fold, T);
    return atan2f_fast_end(T, ok);
}
__device__ __forceinline__ float atan2f_fast(float yf, float xf, const double *fold) {
    bool ok;
    const float v = atan2f_fast_try(yf, xf, fold, ok);
    return ok ? v : atan2f_(yf, xf);
}

// a / b correctly rounded, given rb = 1.0f / b correctly rounded (Markstein): q0 = RN(a rb) is within 2 ulp of a / b;
// one residual step makes it faithful, the second one rounds correctly (Handbook of Floating-Point Arithmetic, 2nd ed.,
// Theorem 4.9: y = RN(1/b), q faithful, r = a - b q exact  =>  RN(q + r y) = RN(a / b)).  Five dependent operations instead
// of the 12-instruction IEEE sequence; valid without over / underflow, i.e. for the window coordinates below
// (|a| <= 2^8, spacing in [0.1, 2^7]).  Checked against the IEEE division on 10^9 operand pairs (tests/test_gpu_parity.py).
__device__ __forceinline__ float div_by_reciprocal(float a, float b, float rb) {
    float q = a * rb;
    float r = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(r, rb, q);
    r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, rb, q);
}

#endif  // __HIPCC__

}  // namespace siftmath
