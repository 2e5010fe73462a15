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

}  // namespace rf
