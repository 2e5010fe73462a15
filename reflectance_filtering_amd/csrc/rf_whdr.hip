// rf_whdr.hip -- WHDR of a batch of reflectance predictions against IIW judgements (gfx950).
//
// Replaces the per-comparison Python loop of /root/reference/training/layers/whdr_layer.py:253-287
// (with _lightness, :180-196) for device-resident predictions: one wave per image; lanes decide
// 64 comparisons at a time (two point gathers, float32 lightness = max(eps, mean over channels),
// float32 ratios against float32(1 + delta)), lane 0 then adds the weights in comparison order in
// float64, which is the order and precision of the reference's scalar accumulation.
#include <cfloat>

#include "rf_common.hpp"

namespace rf {
namespace {

__device__ inline float lightness(const float *__restrict__ img, int c, size_t plane, size_t pix)
{
    float m;
    if (c == 3) {
        // np.mean of a float32 3-vector: ((r0 + r1) + r2) / 3 in float32
        m = __fdiv_rn(__fadd_rn(__fadd_rn(img[pix], img[plane + pix]), img[2 * plane + pix]), 3.0f);
    } else {
        m = img[pix];
    }
    // Python's max(eps, m): m only if m > eps
    return m > FLT_EPSILON ? m : FLT_EPSILON;
}

__global__ __launch_bounds__(64) void whdr_kernel(const float *__restrict__ refl, int c, int h,
                                                  int w, const int *__restrict__ pts,
                                                  const double *__restrict__ wts,
                                                  const int *__restrict__ offsets, float thresh,
                                                  double *__restrict__ out)
{
    __shared__ double err_w[64];
    const int img = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t plane = (size_t)h * w;
    const float *R = refl + (size_t)img * c * plane;
    const int k0 = offsets[img], k1 = offsets[img + 1];
    double error_sum = 0.0, weight_sum = 0.0;
    for (int base = k0; base < k1; base += 64) {
        const int k = base + lane;
        double e = 0.0;
        if (k < k1) {
            const int *p = pts + (size_t)k * 5;
            const float l1 = lightness(R, c, plane, (size_t)p[1] * w + p[0]);
            const float l2 = lightness(R, c, plane, (size_t)p[3] * w + p[2]);
            int alg = 0;
            if (__fdiv_rn(l2, l1) > thresh)
                alg = 1;
            else if (__fdiv_rn(l1, l2) > thresh)
                alg = 2;
            if (p[4] != alg)
                e = wts[k];
        }
        err_w[lane] = e;
        __syncthreads();
        if (lane == 0) {
            const int cnt = min(64, k1 - base);
            for (int j = 0; j < cnt; j++) {
                // `error_sum += weight` only happens on a mismatch; adding +0.0 otherwise leaves
                // the (non-negative) running sum unchanged
                error_sum += err_w[j];
                weight_sum += wts[base + j];
            }
        }
        __syncthreads();
    }
    if (lane == 0)
        out[img] = weight_sum != 0.0 ? error_sum / weight_sum : 0.0;
}

}  // namespace
}  // namespace rf

extern "C" int rf_whdr_f32(const float *refl, int n, int c, int h, int w, const int *points,
                           const double *weights, const int *offsets, double delta, double *out,
                           void *stream_)
{
    using namespace rf;
    if (n == 0)
        return RF_OK;
    if (!refl || !points || !weights || !offsets || !out)
        return fail(RF_E_BADARG, "rf_whdr_f32: NULL pointer");
    if (n < 0 || h <= 0 || w <= 0)
        return fail(RF_E_BADARG, "rf_whdr_f32: bad size n=%d h=%d w=%d", n, h, w);
    if (c != 1 && c != 3)
        return fail(RF_E_UNSUPPORTED, "rf_whdr_f32: 1 or 3 channels (got %d)", c);
    if (!(delta >= 0))
        return fail(RF_E_BADARG, "rf_whdr_f32: delta must be >= 0");
    const float thresh = (float)(1.0 + delta);
    hipLaunchKernelGGL(whdr_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream_, refl, c, h, w,
                       points, weights, offsets, thresh, out);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
