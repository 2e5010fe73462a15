// rf_jbf.hip -- joint bilateral filter, uint8, for gfx950 (MI355X).
//
// Replaces cv2.ximgproc.jointBilateralFilter as called at
// /root/reference/filter_reflectance.py:60-64.  Arithmetic contract (DESIGN.md "JBF"):
// per output pixel the taps of the radius-r disk are visited row-major, the weight is
// spaceW[k] * colorLUT[L1(joint0, jointTap)] in float32, and sum[c] += weight * src[c] is a
// separately rounded multiply then add -- the order of the 8u path of
// opencv_contrib/modules/ximgproc/src/joint_bilateral_filter.cpp on a non-FMA build.
//
// Kernels:
//   jbf_tiled_kernel   one 512-thread workgroup = 64x32 output tile.  The joint/src tile with
//                      its halo is staged once into LDS as packed {BGRx joint, BGRx src}
//                      8-byte texels (border handling happens at staging time, so the tap
//                      loop is branch-free), the colour LUT sits in LDS replicated 32x so that
//                      every lane gathers from its own bank, each lane owns 4 horizontally
//                      adjacent outputs and slides over the tap row so every LDS texel and
//                      its 3 byte->float conversions feed 4 outputs.
//   jbf_generic_kernel untiled, any radius, global-memory gathers (fallback + cross-check).
#include <cmath>
#include <mutex>
#include <vector>

#include "rf_common.hpp"

namespace rf {
namespace {

constexpr int kTileW = 64;
constexpr int kTileH = 32;
constexpr int kPix = 4;        // outputs per lane (horizontal)
constexpr int kThreads = 512;  // 16 lanes across x 32 rows
constexpr int kLutRep = 32;    // LUT replicas = LDS banks of ds_read_b32
constexpr int kMaxLds = 160 * 1024;

struct JbfTables {
    int device = -1;
    int radius = 0;
    int joint_cn = 0;
    double sigma_color = 0, sigma_space = 0;
    int maxk = 0;
    int lut_len = 0;   // entries kept: indices >= lut_len-1 are clamped (LUT value exactly 0)
    int sw_stride = 0; // floats per padded spatial-weight row
    float *d_lut = nullptr;     // [256*joint_cn]
    int *d_di = nullptr;        // [maxk]
    int *d_dj = nullptr;        // [maxk]
    float *d_sw = nullptr;      // [maxk]
    int *d_hw = nullptr;        // [2r+1] half-width of the disk on tap row i
    float *d_swpad = nullptr;   // [2r+1][sw_stride]: zeros | weights j=-hw..hw | zeros
};

std::mutex g_mu;
std::vector<JbfTables> g_tables;

void free_tables(JbfTables &t)
{
    (void)hipFree(t.d_lut);
    (void)hipFree(t.d_di);
    (void)hipFree(t.d_dj);
    (void)hipFree(t.d_sw);
    (void)hipFree(t.d_hw);
    (void)hipFree(t.d_swpad);
}

// Host-side parameter tables, computed in double exactly like jointBilateralFilter_8u does.
int get_tables(int radius, int joint_cn, double sigma_color, double sigma_space, JbfTables *out)
{
    int dev = 0;
    RF_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mu);
    for (const JbfTables &t : g_tables)
        if (t.device == dev && t.radius == radius && t.joint_cn == joint_cn &&
            t.sigma_color == sigma_color && t.sigma_space == sigma_space) {
            *out = t;
            return RF_OK;
        }
    JbfTables t;
    t.device = dev;
    t.radius = radius;
    t.joint_cn = joint_cn;
    t.sigma_color = sigma_color;
    t.sigma_space = sigma_space;
    const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
    const double gauss_space_coeff = -0.5 / (sigma_space * sigma_space);
    const int nlut = 256 * joint_cn;
    std::vector<float> lut(nlut);
    for (int i = 0; i < nlut; i++)
        lut[i] = (float)std::exp(i * i * gauss_color_coeff);
    // keep entries up to and including the first exact zero (the LUT is non-increasing)
    t.lut_len = nlut;
    for (int i = 0; i < nlut; i++)
        if (lut[i] == 0.0f) {
            t.lut_len = i + 1;
            break;
        }
    const int d = 2 * radius + 1;
    std::vector<int> di, dj, hw(d, -1);
    std::vector<float> sw;
    t.sw_stride = d + 2 * (kPix - 1);
    std::vector<float> swpad((size_t)d * t.sw_stride, 0.0f);
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = std::sqrt((double)i * i + (double)j * j);
            if (r > radius)
                continue;
            float wgt = (float)std::exp(r * r * gauss_space_coeff);
            di.push_back(i);
            dj.push_back(j);
            sw.push_back(wgt);
            if (j >= 0 && j > hw[i + radius])
                hw[i + radius] = j;
            swpad[(size_t)(i + radius) * t.sw_stride + (j + radius + kPix - 1)] = wgt;
        }
    t.maxk = (int)di.size();
    RF_HIP_CHECK(hipMalloc(&t.d_lut, sizeof(float) * nlut));
    RF_HIP_CHECK(hipMalloc(&t.d_di, sizeof(int) * t.maxk));
    RF_HIP_CHECK(hipMalloc(&t.d_dj, sizeof(int) * t.maxk));
    RF_HIP_CHECK(hipMalloc(&t.d_sw, sizeof(float) * t.maxk));
    RF_HIP_CHECK(hipMalloc(&t.d_hw, sizeof(int) * d));
    RF_HIP_CHECK(hipMalloc(&t.d_swpad, sizeof(float) * swpad.size()));
    RF_HIP_CHECK(hipMemcpy(t.d_lut, lut.data(), sizeof(float) * nlut, hipMemcpyHostToDevice));
    RF_HIP_CHECK(hipMemcpy(t.d_di, di.data(), sizeof(int) * t.maxk, hipMemcpyHostToDevice));
    RF_HIP_CHECK(hipMemcpy(t.d_dj, dj.data(), sizeof(int) * t.maxk, hipMemcpyHostToDevice));
    RF_HIP_CHECK(hipMemcpy(t.d_sw, sw.data(), sizeof(float) * t.maxk, hipMemcpyHostToDevice));
    RF_HIP_CHECK(hipMemcpy(t.d_hw, hw.data(), sizeof(int) * d, hipMemcpyHostToDevice));
    RF_HIP_CHECK(hipMemcpy(t.d_swpad, swpad.data(), sizeof(float) * swpad.size(),
                           hipMemcpyHostToDevice));
    g_tables.push_back(t);
    *out = t;
    return RF_OK;
}

// Packs up to 3 interleaved bytes into the low bytes of a dword (byte 3 = 0), so that
// v_sad_u8 on two such dwords is the L1 colour distance.
__device__ inline uint32_t load_packed(const uint8_t *img, size_t pix, int cn)
{
    const uint8_t *p = img + pix * cn;
    uint32_t v = p[0];
    if (cn == 3)
        v |= ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    return v;
}

__device__ inline void finish_pixel(uint8_t *o, const float *sum, float wsum, int scn, int flags)
{
    if (flags & RF_JBF_TRUE_DIVISION) {
        for (int c = 0; c < scn; c++)
            o[c] = saturate_u8(__fdiv_rn(sum[c], wsum));
    } else {
        const float inv = __fdiv_rn(1.0f, wsum);
        for (int c = 0; c < scn; c++)
            o[c] = saturate_u8(__fmul_rn(sum[c], inv));
    }
}

// ------------------------------------------------------------------------------------------
// Untiled fallback: one thread per output pixel, taps gathered from global memory.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void jbf_generic_kernel(
    const uint8_t *__restrict__ joint, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
    int h, int w, int jcn, int scn, int border, const float *__restrict__ lut,
    const int *__restrict__ di, const int *__restrict__ dj, const float *__restrict__ sw, int maxk,
    int flags)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    const size_t img = (size_t)blockIdx.z * h * w;
    const uint32_t j0 = load_packed(joint, img + (size_t)y * w + x, jcn);
    float sum[3] = {0.f, 0.f, 0.f};
    float wsum = 0.f;
    for (int k = 0; k < maxk; k++) {
        const int yy = border_interpolate(y + di[k], h, border);
        const int xx = border_interpolate(x + dj[k], w, border);
        uint32_t jt = 0, st = 0;
        if (yy >= 0 && xx >= 0) {
            const size_t q = img + (size_t)yy * w + xx;
            jt = load_packed(joint, q, jcn);
            st = load_packed(src, q, scn);
        }
        const uint32_t alpha = __builtin_amdgcn_sad_u8(j0, jt, 0u);
        const float wgt = __fmul_rn(sw[k], lut[alpha]);
        sum[0] = __fadd_rn(sum[0], __fmul_rn(wgt, (float)(st & 0xff)));
        if (scn == 3) {
            sum[1] = __fadd_rn(sum[1], __fmul_rn(wgt, (float)((st >> 8) & 0xff)));
            sum[2] = __fadd_rn(sum[2], __fmul_rn(wgt, (float)((st >> 16) & 0xff)));
        }
        wsum = __fadd_rn(wsum, wgt);
    }
    finish_pixel(dst + (img + (size_t)y * w + x) * scn, sum, wsum, scn, flags);
}

// ------------------------------------------------------------------------------------------
// Tiled kernel.
// LDS: [ lutrep: lut_len*32 floats ][ tile: tlh rows x tlw texels of uint2 ]
// Tile column X (0 = tile_x0 - radius) is stored at  (X & 3) * (tlw/4) + (X >> 2)  within its
// row: the lane that owns outputs 4*tx..4*tx+3 reads X = 4*tx + const, i.e. consecutive lanes
// read consecutive 8-byte texels (conflict-free ds_read_b64) although each lane's own outputs
// are adjacent.  tlw % 32 == 16 keeps the two 16-lane rows of a 32-lane group on disjoint banks.
// ------------------------------------------------------------------------------------------
template <int SCN>
__global__ __launch_bounds__(kThreads) void jbf_tiled_kernel(
    const uint8_t *__restrict__ joint, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
    int h, int w, int jcn, int radius, int border, const float *__restrict__ lut, int lut_len,
    const int *__restrict__ hwtab, const float *__restrict__ swpad, int sw_stride, int tlw,
    int tlh, int tiles_x, int tiles_per_img, int flags)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *lutrep = reinterpret_cast<float *>(smem);
    uint2 *tile = reinterpret_cast<uint2 *>(smem + (size_t)lut_len * kLutRep * sizeof(float));

    const int tid = threadIdx.x;
    const int img_idx = blockIdx.x / tiles_per_img;
    const int t_in_img = blockIdx.x - img_idx * tiles_per_img;
    const int tile_y0 = (t_in_img / tiles_x) * kTileH;
    const int tile_x0 = (t_in_img % tiles_x) * kTileW;
    const size_t img = (size_t)img_idx * h * w;
    const int q4 = tlw >> 2;

    // ---- stage the colour LUT (replicated across banks) and the texel tile ----
    for (int i = tid; i < lut_len * kLutRep; i += kThreads)
        lutrep[i] = lut[i / kLutRep];
    const int tlw_used = kTileW + 2 * radius + (kPix - 1);
    for (int ry = tid >> 6; ry < tlh; ry += kThreads >> 6) {
        const int gy = border_interpolate(tile_y0 - radius + ry, h, border);
        for (int X = tid & 63; X < tlw_used; X += 64) {
            const int gx = border_interpolate(tile_x0 - radius + X, w, border);
            uint2 t = make_uint2(0u, 0u);
            if (gy >= 0 && gx >= 0) {
                const size_t q = img + (size_t)gy * w + gx;
                t.x = load_packed(joint, q, jcn);
                t.y = load_packed(src, q, SCN);
            }
            tile[ry * tlw + (X & 3) * q4 + (X >> 2)] = t;
        }
    }
    __syncthreads();

    const int tx = tid & 15;
    const int ty = tid >> 4;
    const int lane_lut = (tid & (kLutRep - 1));
    const int amax = lut_len - 1;

    // centre joint texels of this lane's 4 outputs: X = 4*tx + p + radius
    uint32_t jc[kPix];
#pragma unroll
    for (int p = 0; p < kPix; p++) {
        const int X = 4 * tx + p + radius;
        jc[p] = tile[(ty + radius) * tlw + (X & 3) * q4 + (X >> 2)].x;
    }
    float sum[kPix][SCN];
    float wsum[kPix];
#pragma unroll
    for (int p = 0; p < kPix; p++) {
        wsum[p] = 0.f;
#pragma unroll
        for (int c = 0; c < SCN; c++)
            sum[p][c] = 0.f;
    }

    for (int i = -radius; i <= radius; i++) {
        const int hw = hwtab[i + radius];
        const uint2 *trow = tile + (ty + i + radius) * tlw + tx;
        // swr[j] = spatial weight of tap (i, j); zero for hw < |j| <= hw + 3
        const float *swr = swpad + (size_t)(i + radius) * sw_stride + (radius + kPix - 1);
        for (int c = -hw; c <= hw + kPix - 1; c++) {
            const int cc = c + radius;  // uniform, >= 0
            const uint2 t = trow[(cc & 3) * q4 + (cc >> 2)];
            float s[SCN];
            s[0] = (float)(t.y & 0xff);
            if (SCN == 3) {
                s[1] = (float)((t.y >> 8) & 0xff);
                s[2] = (float)((t.y >> 16) & 0xff);
            }
#pragma unroll
            for (int p = 0; p < kPix; p++) {
                uint32_t alpha = __builtin_amdgcn_sad_u8(t.x, jc[p], 0u);
                alpha = min(alpha, (uint32_t)amax);
                const float wgt = __fmul_rn(swr[c - p], lutrep[alpha * kLutRep + lane_lut]);
#pragma unroll
                for (int ch = 0; ch < SCN; ch++)
                    sum[p][ch] = __fadd_rn(sum[p][ch], __fmul_rn(wgt, s[ch]));
                wsum[p] = __fadd_rn(wsum[p], wgt);
            }
        }
    }

    const int oy = tile_y0 + ty;
    if (oy < h) {
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            const int ox = tile_x0 + 4 * tx + p;
            if (ox < w)
                finish_pixel(dst + (img + (size_t)oy * w + ox) * SCN, sum[p], wsum[p], SCN, flags);
        }
    }
}

int tiled_geometry(int radius, int lut_len, int *tlw, int *tlh, size_t *lds_bytes)
{
    int wv = kTileW + 2 * radius + (kPix - 1);
    int tw = (wv + 31) / 32 * 32;  // multiple of 32 ...
    if (tw - wv >= 16)
        tw -= 16;  // ... or of 16 with tlw % 32 == 16
    else
        tw += 16;
    *tlw = tw;
    *tlh = kTileH + 2 * radius;
    *lds_bytes = (size_t)lut_len * kLutRep * sizeof(float) + (size_t)tw * (*tlh) * sizeof(uint2);
    return *lds_bytes <= (size_t)kMaxLds;
}

}  // namespace

void jbf_shutdown()
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (JbfTables &t : g_tables)
        free_tables(t);
    g_tables.clear();
}

}  // namespace rf

extern "C" int rf_jbf_u8(const uint8_t *joint, const uint8_t *src, uint8_t *dst, int n, int h,
                         int w, int joint_cn, int src_cn, int d, double sigma_color,
                         double sigma_space, int border, int flags, void *stream_)
{
    using namespace rf;
    if (!joint || !src || !dst)
        return fail(RF_E_BADARG, "rf_jbf_u8: NULL image pointer");
    if (n < 0 || h <= 0 || w <= 0)
        return fail(RF_E_BADARG, "rf_jbf_u8: bad size n=%d h=%d w=%d", n, h, w);
    if ((joint_cn != 1 && joint_cn != 3) || (src_cn != 1 && src_cn != 3))
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: channels must be 1 or 3 (joint %d, src %d)",
                    joint_cn, src_cn);
    if (border < 0 || border > 4)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: border type %d", border);
    if (dst == joint || dst == src)
        return fail(RF_E_BADARG, "rf_jbf_u8: dst must not alias an input");
    if (n == 0)
        return RF_OK;
    // OpenCV: non-positive sigmas become 1; radius from d or from sigma_space
    if (sigma_color <= 0)
        sigma_color = 1;
    if (sigma_space <= 0)
        sigma_space = 1;
    int radius = d <= 0 ? (int)std::lrint(sigma_space * 1.5) : d / 2;
    if (radius < 1)
        radius = 1;
    if (radius > 4096)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: radius %d too large", radius);
    hipStream_t stream = (hipStream_t)stream_;
    JbfTables t;
    int rc = get_tables(radius, joint_cn, sigma_color, sigma_space, &t);
    if (rc != RF_OK)
        return rc;

    int tlw = 0, tlh = 0;
    size_t lds = 0;
    const bool tiled_ok = tiled_geometry(radius, t.lut_len, &tlw, &tlh, &lds);
    if (tiled_ok && !(flags & RF_JBF_FORCE_GENERIC)) {
        const int tiles_x = ceil_div(w, kTileW), tiles_y = ceil_div(h, kTileH);
        const long long blocks = (long long)tiles_x * tiles_y * n;
        if (blocks > 0x7fffffffLL)
            return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: batch too large for one launch");
        auto kern = src_cn == 3 ? jbf_tiled_kernel<3> : jbf_tiled_kernel<1>;
        RF_HIP_CHECK(hipFuncSetAttribute((const void *)kern,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kThreads), lds, stream, joint, src,
                           dst, h, w, joint_cn, radius, border, t.d_lut, t.lut_len, t.d_hw,
                           t.d_swpad, t.sw_stride, tlw, tlh, tiles_x, tiles_x * tiles_y, flags);
    } else {
        dim3 grid(ceil_div(w, 64), ceil_div(h, 4), n);
        if (n > 65535)
            return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: generic path supports n <= 65535");
        hipLaunchKernelGGL(jbf_generic_kernel, grid, dim3(256), 0, stream, joint, src, dst, h, w,
                           joint_cn, src_cn, border, t.d_lut, t.d_di, t.d_dj, t.d_sw, t.maxk,
                           flags);
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
