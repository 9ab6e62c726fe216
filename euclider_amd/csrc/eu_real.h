/*
 * eu_real.h -- the arithmetic type of the trace path.
 *
 * The reference is generic over `F`: f64 by default, f32 with the cargo feature `low_precision`
 * (/root/reference/Cargo.toml:18-20, src/main.rs:46-49).  A cargo feature makes a different binary; so does this: the same
 * sources built with -DEU_LOW_PRECISION give libeuclider_amd_f32.so, in which `real` -- the type of rays, hits, colours, scene
 * parameters and the loader's constructor arithmetic -- is float.  Conventions:
 *   - R(x) around every floating literal of the path, so that expressions are evaluated in F like the reference's
 *     `<F as NumCast>::from(x)` constants, not in double with a final rounding;
 *   - eu_f64 (= double) wherever a value stays 64-bit whatever F is: the public ABI (camera pose, hit distances,
 *     trace_screen_point's colour), the elementary functions' internals (eu_math.h), and the LinearSpace expressions (meval
 *     evaluates in f64 and the result is cast to F: material.rs:99-111).
 * (Rounds 1-2 redefined the keyword `double` instead; round 3 replaced that by this typedef, frames unchanged.)
 */
#ifndef EU_REAL_H
#define EU_REAL_H

typedef double eu_f64;

#ifdef EU_LOW_PRECISION
#define EU_REAL_BITS 32
typedef float real;
#else
#define EU_REAL_BITS 64
typedef double real;
#endif
#define R(x) ((real)(x))

/* the elementary functions are evaluated in f64 (eu_math.h) and rounded to F at once, so that no expression continues in double
 * behind a call (in the default build the casts are no-ops); eu_*_f64 are the untouched entry points.  F = f32: acos / asin / sin / cos
 * take the 1-ulp f64 routines (eu_*32): after the rounding to f32 they give the correctly rounded f32 value but for one argument in 2^28,
 * and the double-double routines of the f64 build would cost a tenth of the frame rate for nothing. */
#if EU_REAL_BITS == 32
#define eu_acos(x) ((real)eu_acos32(x))
#define eu_asin(x) ((real)eu_asin32(x))
#define eu_sin(x) ((real)eu_sin32(x))
#define eu_cos(x) ((real)eu_cos32(x))
#else
#define eu_acos(x) ((real)eu_acos(x))
#define eu_asin(x) ((real)eu_asin(x))
#define eu_sin(x) ((real)eu_sin(x))
#define eu_cos(x) ((real)eu_cos(x))
#endif
#define eu_tan(x) ((real)eu_tan(x))
#define eu_atan(x) ((real)eu_atan(x))
#define eu_atan2(y, x) ((real)eu_atan2(y, x))

#endif
