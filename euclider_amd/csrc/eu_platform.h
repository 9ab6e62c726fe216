/*
 * eu_platform.h -- what the device headers need from the platform, in both compilation modes:
 *   - ahead of time (hipcc, csrc/Makefile): the system headers;
 *   - at run time (hiprtc, jit.cpp: scene-specialised kernels): hiprtc has no system headers, only its built-in HIP runtime
 *     declarations (device math included), so the fixed-width integer names are declared here.
 */
#ifndef EU_PLATFORM_H
#define EU_PLATFORM_H

#if defined(__HIPCC_RTC__)
typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long long uint64_t;
typedef signed char int8_t; typedef short int16_t; typedef int int32_t; typedef long long int64_t;
typedef unsigned long uintptr_t;
#else
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#endif

template <bool B, class T, class F> struct eu_conditional { typedef T type; };
template <class T, class F> struct eu_conditional<false, T, F> { typedef F type; };

#endif
