// rf_cnn.hip -- the shipped 1x1 reflectance CNN, one lane per pixel, for gfx950 (MI355X).
//
// Replaces caffe.Net.forward() on /root/reference/network_definition.prototxt:9-165 with
// /root/reference/learned_weights.caffemodel, fed by imgCV2_to_caffeBlob
// (/root/reference/decompose_with_trained_CNN.py:57-69,82-95):
//   x = srgb_lut[R,G,B]                       (uint8 BGR in; the LUT is the reference's float64
//                                              sRGB->linear curve rounded to float32)
//   h0 = relu(W0 x + b0); h_l = relu(W_l h_{l-1} + b_l), l=1..4      (32 channels each)
//   z  = wf . [h0|h1|h2|h3|h4] + bf ;  r = 1/(1+exp(-z))
// Each dot product is a float32 FMA chain over k ascending starting from 0, then + bias (the
// bias is a second rank-1 gemm in Caffe).  No pass is a large dense contraction (K = 3/32/160
// per pixel) and exact-f32 MFMA issues at the VALU rate on gfx950, so this is plain VALU code:
// activations live in VGPRs, the 4,513 weights are wave-uniform scalar operands.
#include "rf_common.hpp"

namespace rf {
namespace {

__global__ __launch_bounds__(256) void cnn_reflectance_kernel(
    const uint8_t *__restrict__ bgr, float *__restrict__ r_out, uint8_t *__restrict__ r_u8_out,
    size_t npix, const float *__restrict__ wts, const float *__restrict__ srgb_lut)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = srgb_lut[threadIdx.x];
    __syncthreads();
    const float *W0 = wts, *b0 = wts + 96;
    const float *wf = wts + 128 + 4 * 1056, *bf = wf + 160;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix;
         i += (size_t)gridDim.x * 256) {
        const uint8_t *px = bgr + i * 3;
        const float x0 = lut[px[2]], x1 = lut[px[1]], x2 = lut[px[0]];  // blob order is RGB
        float cur[32], nxt[32];
        float z = 0.f;
#pragma unroll
        for (int o = 0; o < 32; o++) {
            float acc = __fmaf_rn(W0[o * 3 + 0], x0, 0.f);
            acc = __fmaf_rn(W0[o * 3 + 1], x1, acc);
            acc = __fmaf_rn(W0[o * 3 + 2], x2, acc);
            acc = __fadd_rn(acc, b0[o]);
            cur[o] = fmaxf(acc, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 32; k++)
            z = __fmaf_rn(wf[k], cur[k], z);
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const float *W = wts + 128 + l * 1056;
            const float *b = W + 1024;
#pragma unroll
            for (int o = 0; o < 32; o++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 32; k++)
                    acc = __fmaf_rn(W[o * 32 + k], cur[k], acc);
                acc = __fadd_rn(acc, b[o]);
                nxt[o] = fmaxf(acc, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 32; k++) {
                cur[k] = nxt[k];
                z = __fmaf_rn(wf[32 * (l + 1) + k], cur[k], z);
            }
        }
        z = __fadd_rn(z, bf[0]);
        // caffe: 1. / (1. + exp(-x)) with a float exp; expf modelled as round(exp in double)
        const float e = (float)exp((double)(-z));
        const float r = (float)(1.0 / (1.0 + (double)e));
        if (r_out)
            r_out[i] = r;
        if (r_u8_out)
            r_u8_out[i] = (uint8_t)__fmul_rn(r, 255.0f);  // astype(uint8): truncation
    }
}

}  // namespace
}  // namespace rf

extern "C" int rf_cnn_reflectance_u8(const uint8_t *bgr, float *r_out, uint8_t *r_u8_out, int n,
                                     int h, int w, const float *weights, const float *srgb_lut,
                                     void *stream_)
{
    using namespace rf;
    if (!bgr || !weights || !srgb_lut || (!r_out && !r_u8_out))
        return fail(RF_E_BADARG, "rf_cnn_reflectance_u8: NULL pointer");
    if (n < 0 || h <= 0 || w <= 0)
        return fail(RF_E_BADARG, "rf_cnn_reflectance_u8: bad size n=%d h=%d w=%d", n, h, w);
    if (n == 0)
        return RF_OK;
    const size_t npix = (size_t)n * h * w;
    size_t blocks = (npix + 255) / 256;
    if (blocks > 256 * 32)
        blocks = 256 * 32;
    hipLaunchKernelGGL(cnn_reflectance_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream_, bgr, r_out, r_u8_out, npix, weights, srgb_lut);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
