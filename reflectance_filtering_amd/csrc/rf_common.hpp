// rf_common.hpp -- shared host/device helpers of librf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "reflectance_filtering.h"

namespace rf {

// Thread-local message behind rf_last_error().
char *last_error_buf();
int fail(int code, const char *fmt, ...);

#define RF_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::rf::fail(RF_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                              __FILE__, __LINE__);                                           \
    } while (0)

// cv::borderInterpolate (opencv/modules/core/src/copy.cpp); -1 = outside for CONSTANT.
__host__ __device__ inline int border_interpolate(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (border == RF_BORDER_REPLICATE)
        return p < 0 ? 0 : len - 1;
    if (border == RF_BORDER_REFLECT || border == RF_BORDER_REFLECT_101) {
        const int delta = border == RF_BORDER_REFLECT_101;
        if (len == 1)
            return 0;
        do {
            p = p < 0 ? -p - 1 + delta : len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (border == RF_BORDER_WRAP) {
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    }
    return -1;
}

// saturate_cast<uchar>(float): cvRound (round-half-even) then clamp to 0..255.  cvRound is
// cvtss2si on x86: NaN and anything outside the int range become INT_MIN, i.e. 0 after the clamp
// (v_cvt_i32_f32 would saturate +inf to INT_MAX instead).
__device__ inline uint8_t saturate_u8(float v)
{
    if (!(fabsf(v) < 2147483648.0f))
        return 0;
    int iv = __float2int_rn(v);
    iv = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
    return (uint8_t)iv;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Do the byte ranges [a, a + na) and [b, b + nb) intersect?
inline bool ranges_overlap(const void *a, size_t na, const void *b, size_t nb)
{
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + nb && b0 < a0 + na;
}

// Test / benchmark switches behind rf_debug_option() (include/reflectance_filtering_debug.h).
// All zero unless a test or tool sets them; none of them changes the bytes of a result except
// kDbgJbfStageOnly, which leaves dst unwritten, and kDbgGfExpSkip, which leaves kernels out.
enum DebugOption {
    kDbgGfTwoKernel = 0,   // guided filter: row-sum / column-sum kernel pair instead of the fused stage 2
    kDbgJbfStageOnly,      // joint bilateral: stage the tile and return (timing only)
    kDbgJbfCompilerLoop,   // joint bilateral: compiler-scheduled tap loop instead of the asm one
    kDbgJbfTile64Only,     // joint bilateral: 64x64 tiles only (no strip tiles)
    kDbgJbfTune,           // joint bilateral: kernel-variant override (tools/jbf_tune.py), 0 = auto
    kDbgJbfF32Untiled,     // float joint bilateral: one-thread-per-pixel kernel
    kDbgCnnLdsColumns,     // CNN: activations handed between layers through LDS columns (round-1 form)
    kDbgGfSegRows,         // guided filter: rows per stage-1 segment (0 = chosen by the library)
    kDbgGfOneStream,       // guided filter: the whole chunk on the caller's stream (no side stream)
    kDbgGfForceTwoStreams, // guided filter: fork the side stream for any chunk of two or more images
    kDbgGfGuideCache,      // guided filter: keep the guide statistics of the first pass of an iterated call (experiment)
    kDbgGfChained,         // guided filter: chained column walk (no row-walk kernel; measured slower, profiles/r04_gf_chained.md)
    kDbgGfNoCompact,       // guided filter: iterated calls hand grey images on as three channels in dst (not one byte per pixel)
    kDbgGfExpSkip,         // guided filter, TIMING ONLY (wrong results): bit 0 no stage 1, bit 1 no row states, bit 2 no column walk
    kDbgGfStagger,         // guided filter: staggered two-stream schedule (stage 1 of one part beside the walks of the other)
    kDbgGfParts,           // guided filter, staggered schedule: parts per chunk (even, default 2)
    kDbgGfS1Cap,           // guided filter: stage-1 workgroups per CU (dynamic-LDS pad), 0 = whatever fits
    kDbgGfS1MinWgs,        // guided filter: workgroups a stage-1 launch should at least have (0 = chosen by the library)
    kDbgJbfLookahead1,     // joint bilateral: grey asm loop with the gathers one column step ahead (round-4 form)
    kDbgGfS1LegacyStrips,  // guided filter: stage-1 strips with a halo of exactly r columns on either side (rounds 1-5)
    kDbgGfExactAllFlagged, // guided filter, exact-row form: treat every row as failing the test (exercises the list path)
    kDbgGfCwChanRun,       // guided filter, colour src, planar passes: n + 1 = runs of n pairs per channel in the column walk's item order (0: the library's 64; 1: channel fastest)
    kDbgGfExact,           // guided filter: exact-row stage 2 (off by default: measured slower, profiles/r06_gf_exact.md)
    kDbgCount
};
int debug_get(int id);

}  // namespace rf
