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

}  // namespace siftmath
