/*
 * eu_math.h -- deterministic double-precision elementary functions for the trace kernel
 * (host + device).  The reference calls Rust's f64::{acos,asin,sin,cos,tan,atan2}, i.e. the
 * platform libm (/root/reference/src/util.rs:712-722, universe/entity/surface.rs:214-288,
 * universe/d3/entity/surface.rs:60-68, universe/d3/entity/camera.rs:164-185).  ROCm's ocml and a
 * host libm do not agree to the last bit, and colours are truncated to u8 inside the recursion
 * (surface.rs:72-76,104-112), so the kernel carries its own routines: the classic fdlibm 5.3
 * algorithms (e_acos, e_asin, k_sin, k_cos, e_rem_pio2 medium path, s_atan, e_atan2, k_tan),
 * written with IEEE + - * / sqrt and integer tests on the high word only -- no FMA contraction
 * (build with -ffp-contract=off), no fast-math.  <= 1 ulp against glibc (tests/test_math.py).
 * Coefficients are fdlibm's:
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *   Developed at SunSoft, a Sun Microsystems, Inc. business.  Permission to use, copy,
 *   modify, and distribute this software is freely granted, provided that this notice
 *   is preserved.
 * Only exact libm/ocml operations are used from the platform: sqrt (correctly rounded),
 * fabs, floor, fmod.
 */
#ifndef EU_MATH_H
#define EU_MATH_H

#if defined(__cplusplus)
#include "eu_platform.h"
#else
#include <stdint.h>
#include <math.h>
#endif

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define EU_HD __host__ __device__ __attribute__((always_inline))
#else
#define EU_HD
#endif

EU_HD static inline uint32_t eu_hi(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
EU_HD static inline uint32_t eu_lo(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return (uint32_t)u; }
EU_HD static inline double eu_clear_lo(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); u &= 0xffffffff00000000ull; __builtin_memcpy(&x, &u, 8); return x; }
EU_HD static inline double eu_from_words(uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double x; __builtin_memcpy(&x, &u, 8); return x; }
EU_HD static inline int eu_isnan(double x) { return x != x; }

#define EU_PI      3.14159265358979311600e+00
#define EU_PIO2_HI 1.57079632679489655800e+00
#define EU_PIO2_LO 6.12323399573676603587e-17
#define EU_PIO4_HI 7.85398163397448278999e-01

/* rational approximation shared by acos and asin: R(z) = p(z)/q(z) */
EU_HD static inline double eu_asin_p(double z) {
    return z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 +
           z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
}
EU_HD static inline double eu_asin_q(double z) {
    return 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 +
           z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
}

EU_HD static inline double eu_acos(double x) {
    uint32_t hx = eu_hi(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {                     /* |x| >= 1 */
        if (((ix - 0x3ff00000u) | eu_lo(x)) == 0) {
            if ((int32_t)hx > 0) return 0.0;
            return EU_PI + 2.0 * EU_PIO2_LO;
        }
        return (x - x) / (x - x);                /* NaN */
    }
    if (ix < 0x3fe00000u) {                      /* |x| < 0.5 */
        if (ix <= 0x3c600000u) return EU_PIO2_HI + EU_PIO2_LO;
        double z = x * x;
        double r = eu_asin_p(z) / eu_asin_q(z);
        return EU_PIO2_HI - (x - (EU_PIO2_LO - x * r));
    } else if ((int32_t)hx < 0) {                /* x < -0.5 */
        double z = (1.0 + x) * 0.5;
        double p = eu_asin_p(z), q = eu_asin_q(z);
        double s = sqrt(z);
        double r = p / q;
        double w = r * s - EU_PIO2_LO;
        return EU_PI - 2.0 * (s + w);
    } else {                                     /* x > 0.5 */
        double z = (1.0 - x) * 0.5;
        double s = sqrt(z);
        double df = eu_clear_lo(s);
        double c = (z - df * df) / (s + df);
        double p = eu_asin_p(z), q = eu_asin_q(z);
        double r = p / q;
        double w = r * s + c;
        return 2.0 * (df + w);
    }
}

EU_HD static inline double eu_asin(double x) {
    uint32_t hx = eu_hi(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {
        if (((ix - 0x3ff00000u) | eu_lo(x)) == 0) return x * EU_PIO2_HI + x * EU_PIO2_LO;
        return (x - x) / (x - x);
    } else if (ix < 0x3fe00000u) {
        if (ix < 0x3e400000u) return x;          /* |x| < 2^-27 */
        double t = x * x;
        double w = eu_asin_p(t) / eu_asin_q(t);
        return x + x * w;
    }
    double w = 1.0 - fabs(x);
    double t = w * 0.5;
    double p = eu_asin_p(t), q = eu_asin_q(t);
    double s = sqrt(t);
    if (ix >= 0x3FEF3333u) {                     /* |x| > 0.975 */
        w = p / q;
        t = EU_PIO2_HI - (2.0 * (s + s * w) - EU_PIO2_LO);
    } else {
        w = eu_clear_lo(s);
        double c = (t - w * w) / (s + w);
        double r = p / q;
        p = 2.0 * s * r - (EU_PIO2_LO - 2.0 * c);
        q = EU_PIO4_HI - 2.0 * w;
        t = EU_PIO4_HI - (p - q);
    }
    return ((int32_t)hx > 0) ? t : -t;
}

EU_HD static inline double eu_ksin(double x, double y, int iy) {
    uint32_t ix = eu_hi(x) & 0x7fffffffu;
    if (ix < 0x3e400000u) { if ((int)x == 0) return x; }
    double z = x * x;
    double v = z * x;
    double r = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
               z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    if (iy == 0) return x + v * (-1.66666666666666324348e-01 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * -1.66666666666666324348e-01);
}

EU_HD static inline double eu_kcos(double x, double y) {
    uint32_t ix = eu_hi(x) & 0x7fffffffu;
    if (ix < 0x3e400000u) { if ((int)x == 0) return 1.0; }
    double z = x * x;
    double r = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
               z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000u) qx = 0.28125; else qx = eu_from_words(ix - 0x00200000u, 0);
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

/* argument reduction: x = n*(pi/2) + y0 + y1, |y0+y1| <= pi/4.  Cody-Waite 3-stage path
 * (valid for |x| up to ~2^19*pi/2); larger finite arguments are first folded with fmod
 * (documented accuracy loss; the trace loop never produces them). */
EU_HD static inline int eu_rem_pio2(double x, double *y0, double *y1) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
    uint32_t hx = eu_hi(x), ix = hx & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) { *y0 = x; *y1 = 0.0; return 0; }
    if (ix > 0x413921fbu) {
        x = fmod(x, 6.28318530717958623200e+00 * 65536.0);
        hx = eu_hi(x); ix = hx & 0x7fffffffu;
        if (ix <= 0x3fe921fbu) { *y0 = x; *y1 = 0.0; return 0; }
    }
    double t = fabs(x);
    int n = (int)(t * invpio2 + 0.5);
    double fn = (double)n;
    double r = t - fn * pio2_1;
    double w = fn * pio2_1t;
    int j = (int)(ix >> 20);
    double a = r - w;
    int i = j - (int)((eu_hi(a) >> 20) & 0x7ff);
    if (i > 16) {
        t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        a = r - w;
        i = j - (int)((eu_hi(a) >> 20) & 0x7ff);
        if (i > 49) {
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            a = r - w;
        }
    }
    double b = (r - a) - w;
    if ((int32_t)hx < 0) { *y0 = -a; *y1 = -b; return -n; }
    *y0 = a; *y1 = b; return n;
}

EU_HD static inline double eu_sin(double x) {
    uint32_t ix = eu_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eu_ksin(x, 0.0, 0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eu_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return eu_ksin(y0, y1, 1);
        case 1: return eu_kcos(y0, y1);
        case 2: return -eu_ksin(y0, y1, 1);
        default: return -eu_kcos(y0, y1);
    }
}

EU_HD static inline double eu_cos(double x) {
    uint32_t ix = eu_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eu_kcos(x, 0.0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eu_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return eu_kcos(y0, y1);
        case 1: return -eu_ksin(y0, y1, 1);
        case 2: return -eu_kcos(y0, y1);
        default: return eu_ksin(y0, y1, 1);
    }
}

EU_HD static inline double eu_ktan(double x, double y, int iy) {
    constexpr double T[13] = {
        3.33333333333334091986e-01, 1.33333333333201242699e-01, 5.39682539762260521377e-02,
        2.18694882948595424599e-02, 8.86323982359930005737e-03, 3.59207910759131235356e-03,
        1.45620945432529025516e-03, 5.88041240820264096874e-04, 2.46463134818469906812e-04,
        7.81794442939557092300e-05, 7.14072491382608190305e-05, -1.85586374855275456654e-05,
        2.59073051863633712884e-05 };
    const double pio4 = 7.85398163397448278999e-01, pio4lo = 3.06161699786838301793e-17;
    uint32_t hx = eu_hi(x), ix = hx & 0x7fffffffu;
    if (ix < 0x3e300000u) {
        if ((int)x == 0) {
            if (((ix | eu_lo(x)) | (uint32_t)(iy + 1)) == 0) return 1.0 / fabs(x);
            if (iy == 1) return x;
            return -1.0 / x;
        }
    }
    if (ix >= 0x3FE59428u) {
        if ((int32_t)hx < 0) { x = -x; y = -y; }
        double z = pio4 - x;
        double w = pio4lo - y;
        x = z + w; y = 0.0;
    }
    double z = x * x;
    double w = z * z;
    double r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
    double v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
    double s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T[0] * s;
    w = x + r;
    if (ix >= 0x3FE59428u) {
        v = (double)iy;
        return (double)(1 - (int)((hx >> 30) & 2)) * (v - 2.0 * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    {
        double a, t;
        z = eu_clear_lo(w);
        v = r - (z - x);
        t = a = -1.0 / w;
        t = eu_clear_lo(t);
        s = 1.0 + t * z;
        return t + a * (s + t * v);
    }
}

EU_HD static inline double eu_tan(double x) {
    uint32_t ix = eu_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eu_ktan(x, 0.0, 1);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eu_rem_pio2(x, &y0, &y1);
    return eu_ktan(y0, y1, 1 - ((n & 1) << 1));
}

EU_HD static inline double eu_atan(double x) {
    constexpr double atanhi[4] = { 4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                      9.82793723247329054082e-01, 1.57079632679489655800e+00 };
    constexpr double atanlo[4] = { 2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                      1.39033110312309984516e-17, 6.12323399573676603587e-17 };
    constexpr double aT[11] = {
        3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
        -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
        6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
        -3.65315727442169155270e-02, 1.62858201153657823623e-02 };
    uint32_t hx = eu_hi(x), ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x44100000u) {                     /* |x| >= 2^66 */
        if (eu_isnan(x)) return x + x;
        if ((int32_t)hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3fdc0000u) {                      /* |x| < 0.4375 */
        if (ix < 0x3e200000u) return x;
        id = -1;
    } else {
        x = fabs(x);
        if (ix < 0x3ff30000u) {
            if (ix < 0x3fe60000u) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
            else { id = 1; x = (x - 1.0) / (x + 1.0); }
        } else {
            if (ix < 0x40038000u) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
            else { id = 3; x = -1.0 / x; }
        }
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return ((int32_t)hx < 0) ? -z : z;
}

EU_HD static inline double eu_atan2(double y, double x) {
    const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00;
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    if (eu_isnan(x) || eu_isnan(y)) return x + y;
    uint32_t hx = eu_hi(x), hy = eu_hi(y);
    uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    uint32_t lx = eu_lo(x), ly = eu_lo(y);
    if (((hx - 0x3ff00000u) | lx) == 0) return eu_atan(y);      /* x == 1.0 */
    int m = (int)(((hy >> 31) & 1) | ((hx >> 30) & 2));
    if ((iy | ly) == 0) {
        switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if ((ix | lx) == 0) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7ff00000u) {
        if (iy == 0x7ff00000u) {
            switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny;
                         case 2: return 3.0 * pi_o_4 + tiny; default: return -3.0 * pi_o_4 - tiny; }
        } else {
            switch (m) { case 0: return 0.0; case 1: return -0.0; case 2: return pi + tiny; default: return -pi - tiny; }
        }
    }
    if (iy == 0x7ff00000u) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int k = ((int)iy - (int)ix) >> 20;
    double z;
    if (k > 60) z = pi_o_2 + 0.5 * pi_lo;
    else if ((int32_t)hx < 0 && k < -60) z = 0.0;
    else z = eu_atan(fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

/* f64 entry points under names eu_real.h does not touch (it makes eu_acos(x) etc. round to the path's F at once): the
 * LinearSpace expressions are evaluated in f64 whatever F is */
EU_HD inline double eu_sin_f64(double x) { return eu_sin(x); }
EU_HD inline double eu_cos_f64(double x) { return eu_cos(x); }
EU_HD inline double eu_tan_f64(double x) { return eu_tan(x); }
EU_HD inline double eu_asin_f64(double x) { return eu_asin(x); }
EU_HD inline double eu_acos_f64(double x) { return eu_acos(x); }
EU_HD inline double eu_atan_f64(double x) { return eu_atan(x); }
EU_HD inline double eu_atan2_f64(double y, double x) { return eu_atan2(y, x); }

#endif /* EU_MATH_H */
