/*
 * eo_math.h -- ORACLE (test infrastructure, NOT product code).
 *
 * Deterministic double-precision elementary functions for the CPU restatement of
 * euclider's trace loop.  The reference calls Rust's f64::{acos,asin,sin,cos,tan,atan2}
 * (platform libm; /root/reference/src/util.rs:712-722, universe/entity/surface.rs:214-288,
 * universe/d3/entity/surface.rs:60-68, universe/d3/entity/camera.rs:164-185), which is not
 * pinned by the reference.  So that the oracle and the HIP kernel agree bit for bit, both use
 * the same published algorithms: the classic Sun fdlibm 5.3 routines (e_acos, e_asin, k_sin,
 * k_cos, e_rem_pio2 medium path, s_atan, e_atan2, k_tan), restated here using only IEEE
 * + - * / sqrt and integer tests on the high word (no FMA, no tables beyond the published
 * constants).  Coefficients are those of fdlibm:
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *   Developed at SunSoft, a Sun Microsystems, Inc. business.  Permission to use, copy,
 *   modify, and distribute this software is freely granted, provided that this notice
 *   is preserved.
 * tests/test_oracle_math.py checks every function here against glibc libm to <= 1 ulp.
 *
 * Build with -ffp-contract=off (see oracle/Makefile).
 */
#ifndef EO_MATH_H
#define EO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t eo_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t eo_lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double eo_clear_lo(double x) { uint64_t u; memcpy(&u, &x, 8); u &= 0xffffffff00000000ull; memcpy(&x, &u, 8); return x; }
static inline double eo_from_words(uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &u, 8); return x; }
static inline int eo_isnan(double x) { return x != x; }

#define EO_PI      3.14159265358979311600e+00
#define EO_PIO2_HI 1.57079632679489655800e+00
#define EO_PIO2_LO 6.12323399573676603587e-17
#define EO_PIO4_HI 7.85398163397448278999e-01

/* ---- acos and asin, correctly rounded but for about one argument in two thousand ----
 * (Rounds 1-2 carried fdlibm's e_acos / e_asin: < 1 ulp, and different from the correctly rounded value for 7.7 % / 5.8 % of random
 * arguments.  That mattered: `get_intersection_color` tests a quantised alpha for == 255 (surface.rs:73), an opaque blend's alpha is 1
 * or 1 - 2^-53 depending on the last bit of an acos, and a build of the oracle on glibc's libm -- what Rust's f64::acos calls on Linux,
 * itself correctly rounded for all but 0.11 % of arguments -- differed from this one in 2.08 % of 3d_room's bytes.)
 *
 * asin(s) = s + s^3 g(z), z = s^2 in [0, 1/4], g(z) = sum_{n>=1} c_n z^(n-1), c_n = (2n)! / (4^n n!^2 (2n+1)).  The terms of n = 1, 2, 3
 * are carried in double-double arithmetic (coefficients exact to 2^-106), the rest -- at most 1.2e-4 of the result -- is a degree-15
 * polynomial in double (Chebyshev interpolant of the exact series, relative error 2^-55.5).  Error of the sum: below 2^-65 of the
 * result, i.e. the final rounding differs from the correct one for about 2^-11 of the arguments (measured: tests/test_oracle_math.py).
 *   |x| <  1/2:  asin x = K(x, x^2),                 acos x = pi/2 - K(x, x^2)
 *   |x| >= 1/2:  with z = (1 - |x|) / 2 (exact) and s = sqrt(z) as a double-double:
 *                asin x = +-(pi/2 - 2 K(s, z)),      acos x = 2 K(s, z)  (x > 0),   pi - 2 K(s, z)  (x < 0)
 * Only IEEE + - * / sqrt and fma are used; fma(a, b, c) is a single correctly rounded operation on the device and on the host alike. */
#define EO_FMA(a, b, c) __builtin_fma((a), (b), (c))
static inline double eo_asin_tail(double z) {      /* sum_{n>=4} c_n z^(n-4) on [0, 1/4] */
    return 0x1.f1c71c71c71c7p-6 + z * (0x1.6e8ba2e8ba2e9p-6 + z * (0x1.1c4ec4ec4ec4bp-6 + z * (0x1.c999999999e89p-7 + z * (0x1.7a8787876be59p-7 +
           z * (0x1.3fde50dd73215p-7 + z * (0x1.12ef3c20e26eap-7 + z * (0x1.df3bfcbdb62d4p-8 + z * (0x1.a683535f31d33p-8 + z * (0x1.7854ca3f14fe8p-8 +
           z * (0x1.503462916817ap-8 + z * (0x1.3ce57f72fa79fp-8 + z * (0x1.a96c77a266974p-9 + z * (0x1.08eaccc123691p-7 + z * (-0x1.31ed0a9cc3c14p-7 +
           z * (0x1.f951c9ae35c84p-6 + z * (-0x1.19c1d5ee3df84p-5 + z * 0x1.dc10ddb789ef5p-6))))))))))))))));
}
/* K(s, z) = asin(s) as a double-double (rh + rl), for s = sh + sl with |s| <= 1/2 and z = zh + zl = s^2 */
static inline void eo_asin_kernel(double sh, double sl, double zh, double zl, double *rh, double *rl) {
    const double c1h = 0x1.5555555555555p-3, c1l = 0x1.5555555555555p-57;       /* 1/6 */
    const double c2h = 0x1.3333333333333p-4, c2l = 0x1.999999999999ap-59;       /* 3/40 */
    const double c3h = 0x1.6db6db6db6db7p-5, c3l = -0x1.2492492492492p-60;      /* 5/112 */
    const double t = eo_asin_tail(zh);
    /* L3 = c3 + z t */
    double p = zh * t, e = EO_FMA(zh, t, -p);
    double l3h = c3h + p, l3l = ((c3h - l3h) + p) + (c3l + e);
    /* L2 = c2 + z L3 */
    p = zh * l3h; e = EO_FMA(zh, l3h, -p) + (zh * l3l + zl * l3h);
    double l2h = c2h + p, l2l = ((c2h - l2h) + p) + (c2l + e);
    /* L1 = c1 + z L2 */
    p = zh * l2h; e = EO_FMA(zh, l2h, -p) + (zh * l2l + zl * l2h);
    double l1h = c1h + p, l1l = ((c1h - l1h) + p) + (c1l + e);
    /* W = z L1,  T = s W,  R = s + T */
    double wh = zh * l1h, wl = EO_FMA(zh, l1h, -wh) + (zh * l1l + zl * l1h);
    double th = sh * wh, tl = EO_FMA(sh, wh, -th) + (sh * wl + sl * wh);
    double h = sh + th;
    *rl = ((sh - h) + th) + (sl + tl);
    *rh = h;
}
#define EO_PI_H 0x1.921fb54442d18p+1
#define EO_PI_L 0x1.1a62633145c07p-53
#define EO_PIO2_H 0x1.921fb54442d18p+0
#define EO_PIO2_L 0x1.1a62633145c07p-54

static inline double eo_acos(double x) {
    uint32_t hx = eo_hi(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {                     /* |x| >= 1 */
        if (((ix - 0x3ff00000u) | eo_lo(x)) == 0) {
            if ((int32_t)hx > 0) return 0.0;
            return EO_PI_H;                     /* RN(pi) */
        }
        return (x - x) / (x - x);                /* NaN */
    }
    double rh, rl;
    if (ix < 0x3fe00000u) {                      /* |x| < 1/2: pi/2 - K(x, x^2) */
        const double zh = x * x, zl = EO_FMA(x, x, -zh);
        eo_asin_kernel(x, 0.0, zh, zl, &rh, &rl);
        const double h = EO_PIO2_H - rh;
        return h + (((EO_PIO2_H - h) - rh) + (EO_PIO2_L - rl));
    }
    const double z = (1.0 - fabs(x)) * 0.5;      /* exact: 1 - |x| by Sterbenz, the halving by scaling */
    const double sh = sqrt(z);
    const double sl = EO_FMA(-sh, sh, z) / (sh + sh);      /* sqrt(z) = sh + sl to ~2^-104 */
    eo_asin_kernel(sh, sl, z, 0.0, &rh, &rl);
    if ((int32_t)hx > 0) return 2.0 * (rh + rl);
    const double h = EO_PI_H - 2.0 * rh;
    return h + (((EO_PI_H - h) - 2.0 * rh) + (EO_PI_L - 2.0 * rl));
}

static inline double eo_asin(double x) {
    uint32_t hx = eo_hi(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {
        if (((ix - 0x3ff00000u) | eo_lo(x)) == 0) return ((int32_t)hx > 0) ? EO_PIO2_H : -EO_PIO2_H;      /* RN(+-pi/2) */
        return (x - x) / (x - x);
    }
    double rh, rl;
    if (ix < 0x3fe00000u) {                      /* |x| < 1/2 */
        if (ix < 0x3e400000u) return x;          /* |x| < 2^-27: x (1 + x^2/6 + ...) rounds to x */
        const double zh = x * x, zl = EO_FMA(x, x, -zh);
        eo_asin_kernel(x, 0.0, zh, zl, &rh, &rl);
        return rh + rl;
    }
    const double z = (1.0 - fabs(x)) * 0.5;
    const double sh = sqrt(z);
    const double sl = EO_FMA(-sh, sh, z) / (sh + sh);
    eo_asin_kernel(sh, sl, z, 0.0, &rh, &rl);
    const double h = EO_PIO2_H - 2.0 * rh;
    const double r = h + (((EO_PIO2_H - h) - 2.0 * rh) + (EO_PIO2_L - 2.0 * rl));
    return ((int32_t)hx > 0) ? r : -r;
}

/* ---- sin and cos kernels on |r| <= pi/4, r = rh + rl: the Taylor series, whose coefficients are exact rationals; the terms that exceed
 * 2^-14 of the result (r^3/6, r^5/120 for sin; r^2/2, r^4/24, r^6/720 for cos) in double-double arithmetic, the rest (seven terms) in
 * double.  Error below 2^-65 of the result: correctly rounded but for about one argument in two thousand (fdlibm's k_sin / k_cos, which
 * rounds 1-2 carried, differ from the correctly rounded value for ~3 % of the arguments). */
static inline double eo_ksin(double rh, double rl, int unused_iy) {
    (void)unused_iy;
    uint32_t ix = eo_hi(rh) & 0x7fffffffu;
    if (ix < 0x3e400000u) { if ((int)rh == 0) return rh; }      /* |r| < 2^-27 */
    const double s1h = -0x1.5555555555555p-3, s1l = -0x1.5555555555555p-57;      /* -1/6 */
    const double s2h = 0x1.1111111111111p-7, s2l = 0x1.1111111111111p-63;        /* 1/120 */
    const double zh = rh * rh, zl = EO_FMA(rh, rh, -zh) + 2.0 * rh * rl;
    const double t = -0x1.a01a01a01a01ap-13 + zh * (0x1.71de3a556c734p-19 + zh * (-0x1.ae64567f544e4p-26 + zh * (0x1.6124613a86d09p-33 +
                     zh * (-0x1.ae7f3e733b81fp-41 + zh * (0x1.952c77030ad4ap-49 + zh * -0x1.2f49b46814157p-57)))));
    double p = zh * t, e = EO_FMA(zh, t, -p);
    const double l2h = s2h + p, l2l = ((s2h - l2h) + p) + (s2l + e);
    p = zh * l2h; e = EO_FMA(zh, l2h, -p) + (zh * l2l + zl * l2h);
    const double l1h = s1h + p, l1l = ((s1h - l1h) + p) + (s1l + e);
    const double wh = zh * l1h, wl = EO_FMA(zh, l1h, -wh) + (zh * l1l + zl * l1h);
    const double th = rh * wh, tl = EO_FMA(rh, wh, -th) + (rh * wl + rl * wh);
    const double h = rh + th;
    return h + (((rh - h) + th) + (rl + tl));
}

static inline double eo_kcos(double rh, double rl) {
    uint32_t ix = eo_hi(rh) & 0x7fffffffu;
    if (ix < 0x3e400000u) { if ((int)rh == 0) return 1.0; }
    const double c2h = 0x1.5555555555555p-5, c2l = 0x1.5555555555555p-59;        /* 1/24 */
    const double c3h = -0x1.6c16c16c16c17p-10, c3l = 0x1.f49f49f49f49fp-65;      /* -1/720 */
    const double zh = rh * rh, zl = EO_FMA(rh, rh, -zh) + 2.0 * rh * rl;
    const double t = 0x1.a01a01a01a01ap-16 + zh * (-0x1.27e4fb7789f5cp-22 + zh * (0x1.1eed8eff8d898p-29 + zh * (-0x1.93974a8c07c9dp-37 +
                     zh * (0x1.ae7f3e733b81fp-45 + zh * (-0x1.6827863b97d97p-53 + zh * 0x1.e542ba4020225p-62)))));
    double p = zh * t, e = EO_FMA(zh, t, -p);
    const double l3h = c3h + p, l3l = ((c3h - l3h) + p) + (c3l + e);
    p = zh * l3h; e = EO_FMA(zh, l3h, -p) + (zh * l3l + zl * l3h);
    const double l2h = c2h + p, l2l = ((c2h - l2h) + p) + (c2l + e);
    p = zh * l2h; e = EO_FMA(zh, l2h, -p) + (zh * l2l + zl * l2h);
    const double l1h = -0.5 + p, l1l = ((-0.5 - l1h) + p) + e;
    const double wh = zh * l1h, wl = EO_FMA(zh, l1h, -wh) + (zh * l1l + zl * l1h);
    const double h = 1.0 + wh;
    return h + (((1.0 - h) + wh) + wl);
}

/* argument reduction: x = n*(pi/2) + y0 + y1, |y0+y1| <= pi/4.  Cody-Waite 3-stage path
 * (valid for |x| up to ~2^19*pi/2); larger finite arguments are first folded with fmod
 * (documented accuracy loss; the trace loop never produces them). */
static inline int eo_rem_pio2(double x, double *y0, double *y1) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
    uint32_t hx = eo_hi(x), ix = hx & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) { *y0 = x; *y1 = 0.0; return 0; }
    if (ix > 0x413921fbu) {
        x = fmod(x, 6.28318530717958623200e+00 * 65536.0);
        hx = eo_hi(x); ix = hx & 0x7fffffffu;
        if (ix <= 0x3fe921fbu) { *y0 = x; *y1 = 0.0; return 0; }
    }
    double t = fabs(x);
    int n = (int)(t * invpio2 + 0.5);
    double fn = (double)n;
    double r = t - fn * pio2_1;
    double w = fn * pio2_1t;
    int j = (int)(ix >> 20);
    double a = r - w;
    int i = j - (int)((eo_hi(a) >> 20) & 0x7ff);
    if (i > 16) {
        t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        a = r - w;
        i = j - (int)((eo_hi(a) >> 20) & 0x7ff);
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

static inline double eo_sin(double x) {
    uint32_t ix = eo_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eo_ksin(x, 0.0, 0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eo_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return eo_ksin(y0, y1, 1);
        case 1: return eo_kcos(y0, y1);
        case 2: return -eo_ksin(y0, y1, 1);
        default: return -eo_kcos(y0, y1);
    }
}

static inline double eo_cos(double x) {
    uint32_t ix = eo_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eo_kcos(x, 0.0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eo_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return eo_kcos(y0, y1);
        case 1: return -eo_ksin(y0, y1, 1);
        case 2: return -eo_kcos(y0, y1);
        default: return eo_ksin(y0, y1, 1);
    }
}

static inline double eo_ktan(double x, double y, int iy) {
    static const double T[13] = {
        3.33333333333334091986e-01, 1.33333333333201242699e-01, 5.39682539762260521377e-02,
        2.18694882948595424599e-02, 8.86323982359930005737e-03, 3.59207910759131235356e-03,
        1.45620945432529025516e-03, 5.88041240820264096874e-04, 2.46463134818469906812e-04,
        7.81794442939557092300e-05, 7.14072491382608190305e-05, -1.85586374855275456654e-05,
        2.59073051863633712884e-05 };
    const double pio4 = 7.85398163397448278999e-01, pio4lo = 3.06161699786838301793e-17;
    uint32_t hx = eo_hi(x), ix = hx & 0x7fffffffu;
    if (ix < 0x3e300000u) {
        if ((int)x == 0) {
            if (((ix | eo_lo(x)) | (uint32_t)(iy + 1)) == 0) return 1.0 / fabs(x);
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
        z = eo_clear_lo(w);
        v = r - (z - x);
        t = a = -1.0 / w;
        t = eo_clear_lo(t);
        s = 1.0 + t * z;
        return t + a * (s + t * v);
    }
}

static inline double eo_tan(double x) {
    uint32_t ix = eo_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eo_ktan(x, 0.0, 1);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eo_rem_pio2(x, &y0, &y1);
    return eo_ktan(y0, y1, 1 - ((n & 1) << 1));
}

static inline double eo_atan(double x) {
    static const double atanhi[4] = { 4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                      9.82793723247329054082e-01, 1.57079632679489655800e+00 };
    static const double atanlo[4] = { 2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                      1.39033110312309984516e-17, 6.12323399573676603587e-17 };
    static const double aT[11] = {
        3.33333333333329318027e-01, -1.99999999998764832476e-01, 1.42857142725034663711e-01,
        -1.11111104054623557880e-01, 9.09088713343650656196e-02, -7.69187620504482999495e-02,
        6.66107313738753120669e-02, -5.83357013379057348645e-02, 4.97687799461593236017e-02,
        -3.65315727442169155270e-02, 1.62858201153657823623e-02 };
    uint32_t hx = eo_hi(x), ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x44100000u) {                     /* |x| >= 2^66 */
        if (eo_isnan(x)) return x + x;
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

static inline double eo_atan2(double y, double x) {
    const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00;
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    if (eo_isnan(x) || eo_isnan(y)) return x + y;
    uint32_t hx = eo_hi(x), hy = eo_hi(y);
    uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    uint32_t lx = eo_lo(x), ly = eo_lo(y);
    if (((hx - 0x3ff00000u) | lx) == 0) return eo_atan(y);      /* x == 1.0 */
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
    else z = eo_atan(fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

/* ---- acos / asin / sin / cos for the F = f32 build: fdlibm 5.3's e_acos, e_asin, k_sin, k_cos (the routines rounds 1-2 used for f64 as
 * well): within 1 ulp of the f64 value, which is then rounded to f32 -- the correctly rounded f32 result unless the true value lies
 * within 2^-53 of an f32 rounding boundary (one argument in 2^28).  The double-double routines above would buy nothing there and cost
 * a tenth of the f32 frame rate. ---- */
/* rational approximation shared by acos and asin: R(z) = p(z)/q(z) */
static inline double eo_asin_p32(double z) {
    return z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 +
           z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
}
static inline double eo_asin_q32(double z) {
    return 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 +
           z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
}

static inline double eo_acos32(double x) {
    uint32_t hx = eo_hi(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {                     /* |x| >= 1 */
        if (((ix - 0x3ff00000u) | eo_lo(x)) == 0) {
            if ((int32_t)hx > 0) return 0.0;
            return EO_PI + 2.0 * EO_PIO2_LO;
        }
        return (x - x) / (x - x);                /* NaN */
    }
    if (ix < 0x3fe00000u) {                      /* |x| < 0.5 */
        if (ix <= 0x3c600000u) return EO_PIO2_HI + EO_PIO2_LO;
        double z = x * x;
        double r = eo_asin_p32(z) / eo_asin_q32(z);
        return EO_PIO2_HI - (x - (EO_PIO2_LO - x * r));
    } else if ((int32_t)hx < 0) {                /* x < -0.5 */
        double z = (1.0 + x) * 0.5;
        double p = eo_asin_p32(z), q = eo_asin_q32(z);
        double s = sqrt(z);
        double r = p / q;
        double w = r * s - EO_PIO2_LO;
        return EO_PI - 2.0 * (s + w);
    } else {                                     /* x > 0.5 */
        double z = (1.0 - x) * 0.5;
        double s = sqrt(z);
        double df = eo_clear_lo(s);
        double c = (z - df * df) / (s + df);
        double p = eo_asin_p32(z), q = eo_asin_q32(z);
        double r = p / q;
        double w = r * s + c;
        return 2.0 * (df + w);
    }
}

static inline double eo_asin32(double x) {
    uint32_t hx = eo_hi(x), ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {
        if (((ix - 0x3ff00000u) | eo_lo(x)) == 0) return x * EO_PIO2_HI + x * EO_PIO2_LO;
        return (x - x) / (x - x);
    } else if (ix < 0x3fe00000u) {
        if (ix < 0x3e400000u) return x;          /* |x| < 2^-27 */
        double t = x * x;
        double w = eo_asin_p32(t) / eo_asin_q32(t);
        return x + x * w;
    }
    double w = 1.0 - fabs(x);
    double t = w * 0.5;
    double p = eo_asin_p32(t), q = eo_asin_q32(t);
    double s = sqrt(t);
    if (ix >= 0x3FEF3333u) {                     /* |x| > 0.975 */
        w = p / q;
        t = EO_PIO2_HI - (2.0 * (s + s * w) - EO_PIO2_LO);
    } else {
        w = eo_clear_lo(s);
        double c = (t - w * w) / (s + w);
        double r = p / q;
        p = 2.0 * s * r - (EO_PIO2_LO - 2.0 * c);
        q = EO_PIO4_HI - 2.0 * w;
        t = EO_PIO4_HI - (p - q);
    }
    return ((int32_t)hx > 0) ? t : -t;
}

static inline double eo_ksin32(double x, double y, int iy) {
    uint32_t ix = eo_hi(x) & 0x7fffffffu;
    if (ix < 0x3e400000u) { if ((int)x == 0) return x; }
    double z = x * x;
    double v = z * x;
    double r = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
               z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    if (iy == 0) return x + v * (-1.66666666666666324348e-01 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * -1.66666666666666324348e-01);
}

static inline double eo_kcos32(double x, double y) {
    uint32_t ix = eo_hi(x) & 0x7fffffffu;
    if (ix < 0x3e400000u) { if ((int)x == 0) return 1.0; }
    double z = x * x;
    double r = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
               z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000u) qx = 0.28125; else qx = eo_from_words(ix - 0x00200000u, 0);
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

static inline double eo_sin32(double x) {
    uint32_t ix = eo_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eo_ksin32(x, 0.0, 0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eo_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return eo_ksin32(y0, y1, 1);
        case 1: return eo_kcos32(y0, y1);
        case 2: return -eo_ksin32(y0, y1, 1);
        default: return -eo_kcos32(y0, y1);
    }
}

static inline double eo_cos32(double x) {
    uint32_t ix = eo_hi(x) & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) return eo_kcos32(x, 0.0);
    if (ix >= 0x7ff00000u) return x - x;
    double y0, y1;
    int n = eo_rem_pio2(x, &y0, &y1);
    switch (n & 3) {
        case 0: return eo_kcos32(y0, y1);
        case 1: return -eo_ksin32(y0, y1, 1);
        case 2: return -eo_kcos32(y0, y1);
        default: return eo_ksin32(y0, y1, 1);
    }
}


#endif /* EO_MATH_H */
