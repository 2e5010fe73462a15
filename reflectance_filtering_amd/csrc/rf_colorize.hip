// rf_colorize.hip -- colourised reflectance / shading outputs of decompose_image, for gfx950.
//
// Replaces, for a batch that is already on the device, the host numpy chain of
// /root/reference/decompose_with_trained_CNN.py:121-128 and /root/reference/image_utils.py:60-92:
//
//   shading     = mean_c(image) / r                      (float64; image is the uint8 BGR input,
//   reflectance = image / max(shading, 1e-3)[..., None]   r the CNN's float32 intensity)
//   for out in (reflectance, shading):                   imwrite(..., sRGB=True)
//       if max(out) > 1: out = clip(out / percentile(out, 99.9, 'lower'), 0, 1)
//       out = rgb_to_srgb(out)                           (<= 0.0031308: *12.92,
//                                                          else (1.055*x)^(1/2.4) - 0.055)
//       bytes = (out * 255).astype(uint8)                (truncation)
//
// Everything is IEEE float64 and is reproduced operation by operation (one correctly rounded
// division / multiplication per numpy ufunc), with two exceptions that are made exact another way:
//  * the percentile is an order statistic (k-th smallest of the 3HW resp. HW values, k computed by
//    the host with numpy's own index rule): an 8-pass radix select on the bit patterns of the
//    non-negative doubles finds it without sorting and without storing the float64 images (they
//    are recomputed from the 7 input bytes per pixel in every pass);
//  * np.power is libm's pow, which the GPU cannot reproduce bit for bit.  But x -> byte is a
//    step function with at most 255 steps on the power branch, so the host computes the 255 step
//    positions once with numpy itself (image_utils.srgb_write_steps) and the kernel counts the
//    steps at or below x: exact for whatever libm the host has.
#include "rf_common.hpp"

namespace rf {
namespace {

struct SelState {
    unsigned long long prefix;  // high bits of the k-th smallest key found so far
    unsigned long long k;       // rank still to be located inside the current prefix bucket
    unsigned long long maxkey;  // largest key (for the `max > 1` test)
    unsigned int hist[256];
};

constexpr int kTargets = 2;  // 0: reflectance (3 values per pixel), 1: shading (1 per pixel)

__device__ inline unsigned long long key_of(double v) { return (unsigned long long)__double_as_longlong(v); }

// The float64 values numpy forms for one pixel.
__device__ inline void pixel_values(const uint8_t *px, float r, double (&refl)[3], double &shading)
{
    const double mean = __ddiv_rn((double)((int)px[0] + (int)px[1] + (int)px[2]), 3.0);
    shading = __ddiv_rn(mean, (double)r);
    const double den = shading > 1e-3 ? shading : (shading != shading ? shading : 1e-3);
#pragma unroll
    for (int c = 0; c < 3; c++)
        refl[c] = __ddiv_rn((double)px[c], den);
}

__global__ void colorize_init_kernel(SelState *st, int n, unsigned long long k_refl,
                                     unsigned long long k_shading)
{
    const int i = blockIdx.x;  // (image, target)
    SelState &s = st[i];
    if (threadIdx.x == 0) {
        s.prefix = 0;
        s.k = (i % kTargets) == 0 ? k_refl : k_shading;
        s.maxkey = 0;
    }
    s.hist[threadIdx.x] = 0;
}

// pass p = 0..7 looks at key byte 7-p of the values whose higher bytes equal the prefix
__global__ __launch_bounds__(256) void colorize_hist_kernel(const uint8_t *__restrict__ bgr,
                                                            const float *__restrict__ r,
                                                            SelState *__restrict__ st, size_t npx,
                                                            int pass)
{
    __shared__ unsigned int hist[kTargets][256];
    __shared__ unsigned long long lmax[kTargets];
    const int img = blockIdx.y;
    hist[0][threadIdx.x] = 0;
    hist[1][threadIdx.x] = 0;
    if (threadIdx.x < kTargets)
        lmax[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *b = bgr + (size_t)img * npx * 3;
    const float *rr = r + (size_t)img * npx;
    const int shift = 56 - 8 * pass;
    const unsigned long long pre0 = st[img * kTargets + 0].prefix;
    const unsigned long long pre1 = st[img * kTargets + 1].prefix;
    unsigned long long m0 = 0, m1 = 0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx;
         p += (size_t)gridDim.x * blockDim.x) {
        double refl[3], sh;
        pixel_values(b + p * 3, rr[p], refl, sh);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const unsigned long long k = key_of(refl[c]);
            if (pass == 0) {
                m0 = k > m0 ? k : m0;
                atomicAdd(&hist[0][k >> 56], 1u);
            } else if ((k >> (shift + 8)) == (pre0 >> (shift + 8))) {
                atomicAdd(&hist[0][(k >> shift) & 255u], 1u);
            }
        }
        const unsigned long long k = key_of(sh);
        if (pass == 0) {
            m1 = k > m1 ? k : m1;
            atomicAdd(&hist[1][k >> 56], 1u);
        } else if ((k >> (shift + 8)) == (pre1 >> (shift + 8))) {
            atomicAdd(&hist[1][(k >> shift) & 255u], 1u);
        }
    }
    if (pass == 0) {
        atomicMax(&lmax[0], m0);
        atomicMax(&lmax[1], m1);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kTargets; t++) {
        const unsigned int c = hist[t][threadIdx.x];
        if (c)
            atomicAdd(&st[img * kTargets + t].hist[threadIdx.x], c);
    }
    if (pass == 0 && threadIdx.x < kTargets)
        atomicMax(&st[img * kTargets + threadIdx.x].maxkey, lmax[threadIdx.x]);
}

// one workgroup per (image, target): bucket that holds rank k, then clear the histogram
__global__ __launch_bounds__(256) void colorize_pick_kernel(SelState *st, int pass)
{
    SelState &s = st[blockIdx.x];
    __shared__ unsigned int h[256];
    h[threadIdx.x] = s.hist[threadIdx.x];
    __syncthreads();
    s.hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        unsigned long long k = s.k;
        int bin = 0;
        for (; bin < 255; bin++) {
            if (k < h[bin])
                break;
            k -= h[bin];
        }
        s.k = k;
        s.prefix |= (unsigned long long)bin << (56 - 8 * pass);
    }
}

// rgb_to_srgb + (x*255).astype(uint8) of a value in [0,1] (NaN -> 0 like the zero-initialised
// result array of the reference)
__device__ inline uint8_t srgb_byte(double v, const double *__restrict__ steps)
{
    if (v <= 0.0031308)
        return (uint8_t)(int)__dmul_rn(__dmul_rn(v, 12.92), 255.0);
    if (!(v > 0.0031308))
        return 0;
    // number of steps k (1..255) with steps[k-1] <= v; the steps are non-decreasing
    int lo = 0, hi = 255;  // invariant: steps[lo-1] <= v < steps[hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (steps[mid] <= v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (uint8_t)lo;
}

__device__ inline double normalise(double v, bool on, double pct)
{
    if (!on)
        return v;
    v = __ddiv_rn(v, pct);
    // np.clip(v, 0, 1) = minimum(maximum(v, 0), 1), NaN propagating
    if (v != v)
        return v;
    v = v < 0.0 ? 0.0 : v;
    return v > 1.0 ? 1.0 : v;
}

__global__ __launch_bounds__(256) void colorize_write_kernel(
    const uint8_t *__restrict__ bgr, const float *__restrict__ r, const SelState *__restrict__ st,
    uint8_t *__restrict__ refl_out, uint8_t *__restrict__ shading_out, size_t npx,
    const double *__restrict__ steps_g)
{
    __shared__ double steps[256];
    steps[threadIdx.x] = threadIdx.x < 255 ? steps_g[threadIdx.x] : __longlong_as_double(0x7ff0000000000000LL);
    __syncthreads();
    const int img = blockIdx.y;
    const SelState &s0 = st[img * kTargets + 0];
    const SelState &s1 = st[img * kTargets + 1];
    const unsigned long long one = key_of(1.0);
    const bool n0 = s0.maxkey > one, n1 = s1.maxkey > one;  // np.max(img) > 1
    const double p0 = __longlong_as_double((long long)s0.prefix);
    const double p1 = __longlong_as_double((long long)s1.prefix);
    const uint8_t *b = bgr + (size_t)img * npx * 3;
    const float *rr = r + (size_t)img * npx;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx;
         p += (size_t)gridDim.x * blockDim.x) {
        double refl[3], sh;
        pixel_values(b + p * 3, rr[p], refl, sh);
        if (refl_out) {
            uint8_t *o = refl_out + ((size_t)img * npx + p) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++)
                o[c] = srgb_byte(normalise(refl[c], n0, p0), steps);
        }
        if (shading_out)
            shading_out[(size_t)img * npx + p] = srgb_byte(normalise(sh, n1, p1), steps);
    }
}

}  // namespace
}  // namespace rf

extern "C" size_t rf_colorize_workspace_bytes(int n)
{
    if (n <= 0)
        return 0;
    return (size_t)n * rf::kTargets * sizeof(rf::SelState);
}

extern "C" int rf_colorize_srgb_u8(const uint8_t *bgr, const float *r, uint8_t *refl_out,
                                   uint8_t *shading_out, int n, int h, int w,
                                   unsigned long long k_refl, unsigned long long k_shading,
                                   const double *srgb_steps, void *workspace,
                                   size_t workspace_bytes, void *stream_)
{
    using namespace rf;
    if (n == 0)
        return RF_OK;
    if (!bgr || !r || !srgb_steps || !workspace || (!refl_out && !shading_out))
        return fail(RF_E_BADARG, "rf_colorize_srgb_u8: NULL pointer");
    if (n < 0 || h <= 0 || w <= 0)
        return fail(RF_E_BADARG, "rf_colorize_srgb_u8: bad size n=%d h=%d w=%d", n, h, w);
    const size_t npx = (size_t)h * w;
    if (k_refl >= 3 * npx || k_shading >= npx)
        return fail(RF_E_BADARG, "rf_colorize_srgb_u8: percentile rank outside the image");
    if (workspace_bytes < rf_colorize_workspace_bytes(n))
        return fail(RF_E_WORKSPACE, "rf_colorize_srgb_u8: workspace %zu B < %zu B", workspace_bytes,
                    rf_colorize_workspace_bytes(n));
    if (n > 65535)
        return fail(RF_E_UNSUPPORTED, "rf_colorize_srgb_u8: n <= 65535 per call");
    hipStream_t stream = (hipStream_t)stream_;
    SelState *st = reinterpret_cast<SelState *>(workspace);
    hipLaunchKernelGGL(colorize_init_kernel, dim3(n * kTargets), dim3(256), 0, stream, st, n, k_refl,
                       k_shading);
    // enough workgroups to fill the chip, few enough that the per-block histogram flush is cheap
    int bx = (int)std::min<size_t>((npx + 256 * 8 - 1) / (256 * 8), 2048);
    if (bx * n < 1024)
        bx = (int)std::min<size_t>((npx + 255) / 256, (size_t)((1024 + n - 1) / n));
    if (bx < 1)
        bx = 1;
    for (int pass = 0; pass < 8; pass++) {
        hipLaunchKernelGGL(colorize_hist_kernel, dim3(bx, n), dim3(256), 0, stream, bgr, r, st, npx,
                           pass);
        hipLaunchKernelGGL(colorize_pick_kernel, dim3(n * kTargets), dim3(256), 0, stream, st, pass);
    }
    hipLaunchKernelGGL(colorize_write_kernel, dim3(bx, n), dim3(256), 0, stream, bgr, r, st, refl_out,
                       shading_out, npx, srgb_steps);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
