/*
 * oracle_math.h -- TEST INFRASTRUCTURE (part of oracle/, never linked into the product).
 *
 * "siftmath v1": the transcendental functions the SIFT hot path needs, specified as
 * float -> float functions evaluated in IEEE binary64 with a FIXED sequence of
 * add/mul/div operations (no FMA contraction, no libm calls), then rounded once to
 * binary32.  The result is the correctly rounded f32 value except when the true value
 * lies within ~2^-48 (relative) of a rounding boundary, i.e. for all practical purposes
 * it is the "centre" of what OpenCL's <=4 ulp exp/sin/cos/atan2/pow may return.
 *
 * Why the oracle does not call glibc here: the reference's kernels call the OpenCL
 * builtins exp/sin/cos/atan2/pow (orientation_cpu.cl:88-91, keypoints_cpu.cl:60-74,
 * image.cl:77,354, gaussian.cl:69), whose last bits are implementation-defined.  A
 * fixed-sequence f64 evaluation is the only definition that a CPU and a GPU can both
 * reproduce bit-for-bit.  The HIP product carries its own device implementation of the
 * same specification (sift_pyocl_amd/csrc/siftmath.hpp); tests/ check (i) this file
 * against mpmath/libm and (ii) the device version against this file.
 *
 * All constants are hex-float literals (IEEE doubles nearest to 1/n!, atan(k/8), ...)
 * so no compiler performs decimal->binary conversion on them.
 * Compile with -ffp-contract=off.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <stdint.h>
#include <string.h>

static inline double om_from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t om_to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

/* round-to-nearest-even integer of |z| < 2^51 using only IEEE additions */
static inline double om_rint(double z) {
    volatile double t = z + 0x1.8p52; /* volatile: forbid algebraic simplification */
    return t - 0x1.8p52;
}

/* exp(r) for |r| <= 0.35 : Taylor degree 14, Horner, binary64 */
static inline double om_exp_core(double r) {
    double p = 0x1.93974a8c07c9dp-37;        /* 1/14! */
    p = p * r + 0x1.6124613a86d09p-33;       /* 1/13! */
    p = p * r + 0x1.1eed8eff8d898p-29;       /* 1/12! */
    p = p * r + 0x1.ae64567f544e4p-26;       /* 1/11! */
    p = p * r + 0x1.27e4fb7789f5cp-22;       /* 1/10! */
    p = p * r + 0x1.71de3a556c734p-19;       /* 1/9!  */
    p = p * r + 0x1.a01a01a01a01ap-16;       /* 1/8!  */
    p = p * r + 0x1.a01a01a01a01ap-13;       /* 1/7!  */
    p = p * r + 0x1.6c16c16c16c17p-10;       /* 1/6!  */
    p = p * r + 0x1.1111111111111p-7;        /* 1/5!  */
    p = p * r + 0x1.5555555555555p-5;        /* 1/4!  */
    p = p * r + 0x1.5555555555555p-3;        /* 1/3!  */
    p = p * r + 0x1.0p-1;                    /* 1/2!  */
    p = p * r + 1.0;
    p = p * r + 1.0;
    return p;
}

/* 2^k as a double for -1022 <= k <= 1023 */
static inline double om_pow2i(int k) { return om_from_bits((uint64_t)(k + 1023) << 52); }

static inline float om_expf(float xf) {
    if (xf != xf) return xf;
    if (xf > 89.0f) return __builtin_inff();
    if (xf < -104.0f) return 0.0f;
    double x = (double)xf;
    double kd = om_rint(x * 0x1.71547652b82fep+0);     /* x*log2(e) */
    double r = (x - kd * 0x1.62e42fee00000p-1)          /* ln2 hi (32 bits): product exact */
                  - kd * 0x1.a39ef35793c76p-33;         /* ln2 lo */
    double v = om_exp_core(r) * om_pow2i((int)kd);
    return (float)v;
}

/* 2^y, used for InitSigma * pow(2, (s+ds)/3)  (image.cl:354) */
static inline float om_exp2f(float yf) {
    if (yf != yf) return yf;
    if (yf > 128.0f) return __builtin_inff();
    if (yf < -150.0f) return 0.0f;
    double y = (double)yf;
    double kd = om_rint(y);
    double r = (y - kd) * 0x1.62e42fefa39efp-1;         /* (y-k)*ln2 */
    double v = om_exp_core(r) * om_pow2i((int)kd);
    return (float)v;
}

/* sin and cos of a float, binary64 evaluation.  Accurate for |x| < ~1e5 (angles here
 * are in [-pi, pi]); beyond that the quadrant reduction loses accuracy but stays
 * deterministic.  NaN/Inf in -> NaN out. */
static inline void om_sincosf(float xf, float *sn, float *cs) {
    if (xf != xf || xf - xf != 0.0f) { *sn = xf - xf; *cs = xf - xf; return; }
    double x = (double)xf;
    double kd = om_rint(x * 0x1.45f306dc9c883p-1);      /* x * 2/pi */
    double r = (x - kd * 0x1.921fb54400000p+0)          /* pi/2 hi (33 bits) */
                  - kd * 0x1.0b4611a626331p-34;         /* pi/2 lo */
    double r2 = r * r;
    double ps = 0x1.2f49b46814157p-57;                  /* 1/19! */
    ps = 0x1.952c77030ad4ap-49 - ps * r2;               /* 1/17! */
    ps = 0x1.ae7f3e733b81fp-41 - ps * r2;               /* 1/15! */
    ps = 0x1.6124613a86d09p-33 - ps * r2;               /* 1/13! */
    ps = 0x1.ae64567f544e4p-26 - ps * r2;               /* 1/11! */
    ps = 0x1.71de3a556c734p-19 - ps * r2;               /* 1/9!  */
    ps = 0x1.a01a01a01a01ap-13 - ps * r2;               /* 1/7!  */
    ps = 0x1.1111111111111p-7 - ps * r2;                /* 1/5!  */
    ps = 0x1.5555555555555p-3 - ps * r2;                /* 1/3!  */
    double s = r - (ps * r2) * r;
    double pc = 0x1.e542ba4020225p-62;                  /* 1/20! */
    pc = 0x1.6827863b97d97p-53 - pc * r2;               /* 1/18! */
    pc = 0x1.ae7f3e733b81fp-45 - pc * r2;               /* 1/16! */
    pc = 0x1.93974a8c07c9dp-37 - pc * r2;               /* 1/14! */
    pc = 0x1.1eed8eff8d898p-29 - pc * r2;               /* 1/12! */
    pc = 0x1.27e4fb7789f5cp-22 - pc * r2;               /* 1/10! */
    pc = 0x1.a01a01a01a01ap-16 - pc * r2;               /* 1/8!  */
    pc = 0x1.6c16c16c16c17p-10 - pc * r2;               /* 1/6!  */
    pc = 0x1.5555555555555p-5 - pc * r2;                /* 1/4!  */
    pc = 0x1.0p-1 - pc * r2;                            /* 1/2!  */
    double c = 1.0 - pc * r2;
    int q = (int)kd & 3;
    double so = (q & 1) ? c : s;
    double co = (q & 1) ? s : c;
    if (q == 1 || q == 2) co = -co;
    if (q >= 2) so = -so;
    *sn = (float)so;
    *cs = (float)co;
}

/* atan2(y, x) with C99 special-case semantics for zeros; NaN in -> NaN out.
 * Infinities are not expected on this path and are treated as NaN. */
static inline float om_atan2f(float yf, float xf) {
    if (yf != yf || xf != xf || yf - yf != 0.0f || xf - xf != 0.0f) return (yf - yf) + (xf - xf);
    double ay = (double)__builtin_fabsf(yf), ax = (double)__builtin_fabsf(xf);
    double res;
    if (ay == 0.0 && ax == 0.0) {
        res = 0.0;
    } else {
        int swap = ay > ax;
        double num = swap ? ax : ay, den = swap ? ay : ax;
        /* octant table index from an exactly rounded f32 quotient */
        float af = (float)num / (float)den;
        int k = (int)(af * 8.0f + 0.5f);                /* 0..8 */
        double c = (double)k * 0.125;
        double t = (num - c * den) / (den + c * num);   /* |t| <= ~1/16 */
        double t2 = t * t;
        double p = 0x1.e1e1e1e1e1e1ep-5;                /* 1/17 */
        p = 0x1.1111111111111p-4 - p * t2;              /* 1/15 */
        p = 0x1.3b13b13b13b14p-4 - p * t2;              /* 1/13 */
        p = 0x1.745d1745d1746p-4 - p * t2;              /* 1/11 */
        p = 0x1.c71c71c71c71cp-4 - p * t2;              /* 1/9  */
        p = 0x1.2492492492492p-3 - p * t2;              /* 1/7  */
        p = 0x1.999999999999ap-3 - p * t2;              /* 1/5  */
        p = 0x1.5555555555555p-2 - p * t2;              /* 1/3  */
        double at = t - (p * t2) * t;
        double tab;
        switch (k) {
            case 0: tab = 0.0; break;
            case 1: tab = 0x1.fd5ba9aac2f6ep-4; break;
            case 2: tab = 0x1.f5b75f92c80ddp-3; break;
            case 3: tab = 0x1.6f61941e4def1p-2; break;
            case 4: tab = 0x1.dac670561bb4fp-2; break;
            case 5: tab = 0x1.1e00babdefeb4p-1; break;
            case 6: tab = 0x1.4978fa3269ee1p-1; break;
            case 7: tab = 0x1.700a7c5784634p-1; break;
            default: tab = 0x1.921fb54442d18p-1; break;
        }
        res = tab + at;
        if (swap) res = 0x1.921fb54442d18p+0 - res;     /* pi/2 - res */
    }
    if (__builtin_signbitf(xf)) res = 0x1.921fb54442d18p+1 - res;  /* pi - res */
    if (__builtin_signbitf(yf)) res = -res;
    return (float)res;
}

#endif
