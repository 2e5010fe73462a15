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
constexpr int kTlw2 = 144;     // v2 tile row pitch in texels (covers radius <= 36)

typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

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
    // v2: rows |i| = 0..r, each sw_len = 2*(r4+8) floats, centre at index r4+8, zeros outside
    // the disk (the weights are symmetric in i and in j)
    int r4 = 0, sw_len = 0;
    float *d_swsym = nullptr;
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
    (void)hipFree(t.d_swsym);
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
    t.r4 = (radius + 3) & ~3;
    t.sw_len = 2 * (t.r4 + 8);
    std::vector<float> swsym((size_t)(radius + 1) * t.sw_len, 0.0f);
    for (size_t k = 0; k < di.size(); k++)
        if (di[k] >= 0)
            swsym[(size_t)di[k] * t.sw_len + (t.r4 + 8) + dj[k]] = sw[k];
    RF_HIP_CHECK(hipMalloc(&t.d_swsym, sizeof(float) * swsym.size()));
    RF_HIP_CHECK(hipMemcpy(t.d_swsym, swsym.data(), sizeof(float) * swsym.size(),
                           hipMemcpyHostToDevice));
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


// ------------------------------------------------------------------------------------------
// Tiled kernel, software-pipelined (v2).
//
// Same tile idea as above, plus:
//   * the tap row is walked in groups of 4 columns starting at a multiple of 4, so the texel
//     address of column (group g, u) is  lane_base + u*(TLW/4) + g : one VALU add per group,
//     immediates for the rest, and the spatial weights of the 4 columns x 4 outputs are a
//     7-float window of the (symmetric) weight row, fetched from LDS as two aligned float4
//     broadcasts per group (no scalar-memory loads inside the loop, so LDS waits stay counted);
//   * a 3-stage pipeline over columns: the texel of column c+2 and the four LUT gathers of
//     column c+1 are in flight while column c is accumulated;
//   * TH rows per tile (16*TH threads): 32 -> 2 waves/SIMD, 48 -> 3 waves/SIMD;
//   * LUTREP replicas of the colour LUT (32 = conflict-free, 16/8 trade conflicts for LDS).
// Columns outside the disk carry zero weight: w = 0 adds +0.0 to non-negative sums, which is
// bit-identical to skipping the tap.
// LDS: [lutrep lut_len*LUTREP f32][sw (r+1)*sw_len f32][tile (TH+2r) x TLW uint2]
// Tile column X <-> image x = tile_x0 - r4 + X, stored at (X&3)*(TLW/4) + (X>>2).
// ------------------------------------------------------------------------------------------
// LDS byte address of a pointer into the workgroup's LDS (low 32 bits of the generic pointer).
__device__ inline uint32_t lds_addr(const void *p)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));
}

// The LDS reads of the tap loop are issued through asm so that their order and their waits are
// exactly the pipeline described above (left to itself the compiler sinks each read next to its
// use and waits for lgkmcnt(0) after every gather).  The wait statement names everything it
// releases -- and the accumulators -- as in/out operands: that keeps consumers below the wait
// and the accumulation of the current column above it, i.e. underneath the reads in flight.
#define RF_LDS_READ_B64(dst, addr, off) \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RF_LDS_READ_B128(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RF_LDS_READ_B32(dst, addr) asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr))

// Accumulates all taps of one lane's 4 outputs.  NCH = channels accumulated (3, or 1 when the
// src is single-channel or every src texel of the tile is grey: identical bits, a third of the
// multiply-adds).  sum/wsum must be zero on entry.
template <int NCH, int LUTREP, bool CLAMP, int TLW>
__device__ __forceinline__ void jbf_tap_loop(uint32_t lut_lane_addr, uint32_t sw_addr0,
                                             uint32_t tile_lane_addr, const uint32_t (&jc)[kPix],
                                             uint32_t amax, int ty, int radius, int r4, int sw_len,
                                             const int *__restrict__ hwtab, float (&sum)[kPix][NCH],
                                             float (&wsum)[kPix])
{
    constexpr int Q4 = TLW / 4;
    auto issue_gathers = [&](uint32_t jtex, float *g) {
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            uint32_t alpha = __builtin_amdgcn_sad_u8(jtex, jc[p], 0u);
            if (CLAMP)
                alpha = min(alpha, amax);
            const uint32_t a = alpha * (LUTREP * 4u) + lut_lane_addr;
            RF_LDS_READ_B32(g[p], a);
        }
    };

    for (int i = -radius; i <= radius; i++) {
        const int hw = hwtab[i + radius];
        const int hw4 = (hw + 3) & ~3;
        const int ai = i < 0 ? -i : i;
        // column c = 4*gq + u - hw4 (gq = 0 .. hw4/2): tile column X = c + r4 + 4*tx, i.e. texel
        // address = ta + u*Q4*8 + gq*8 with ta the per-lane address of (row, group 0, u = 0)
        uint32_t ta = tile_lane_addr +
                      (uint32_t)(((ty + i + radius) * TLW + ((r4 - hw4) >> 2)) * 8);
        // weight of tap (i, j) = swc[j] = swc[-j]; group gq needs swc[hw4 - 4*gq - 4 .. +3]
        uint32_t wa_addr = sw_addr0 + (uint32_t)((ai * sw_len + (r4 + 8) + hw4 - 4) * 4);
        const int ngroups = (hw4 >> 1) + 1;

        // Register rings with compile-time indices only: column 4*gq+u lives in tq[u], its
        // gathers in gg[u & 1].  Every read issued in a step is released by the wait at the END
        // of that step, so nothing is in flight across the loop back-edge (a value in flight
        // there would be copied by the compiler's phi moves before it has landed).
        uint2v tq[4];
        float4v wna, wnb;
        float gg[2][kPix];
        RF_LDS_READ_B64(tq[0], ta, 0);
        RF_LDS_READ_B64(tq[1], ta, Q4 * 8);
        RF_LDS_READ_B128(wna, wa_addr, 0);
        RF_LDS_READ_B128(wnb, wa_addr, 16);
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(tq[0]));
        issue_gathers(tq[0].x, gg[0]);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(tq[1]), "+v"(wna), "+v"(wnb), "+v"(gg[0][0]), "+v"(gg[0][1]),
                       "+v"(gg[0][2]), "+v"(gg[0][3]));

#define RF_ACCUM(U)                                                                  \
    {                                                                                \
        const uint32_t sv = tq[(U)].y;                                               \
        float s[NCH];                                                                \
        s[0] = (float)(sv & 0xff);                                                   \
        if (NCH == 3) {                                                              \
            s[1] = (float)((sv >> 8) & 0xff);                                        \
            s[2] = (float)((sv >> 16) & 0xff);                                       \
        }                                                                            \
        _Pragma("unroll") for (int p = 0; p < kPix; p++)                             \
        {                                                                            \
            const float wgt = __fmul_rn(wv[4 + p - (U)], gg[(U) & 1][p]);            \
            _Pragma("unroll") for (int ch = 0; ch < NCH; ch++) sum[p][ch] =          \
                __fadd_rn(sum[p][ch], __fmul_rn(wgt, s[ch]));                        \
            wsum[p] = __fadd_rn(wsum[p], wgt);                                       \
        }                                                                            \
    }
#define RF_TEXEL_OFF(U) ((((U) + 2) & 3) * Q4 * 8 + (((U) + 2) >> 2) * 8)
        // Pins the accumulators at this point of the instruction stream (no instruction).
#define RF_PIN_ACC()                                                                            \
    if constexpr (NCH == 3) {                                                                   \
        asm volatile(""                                                                         \
                     : "+v"(sum[0][0]), "+v"(sum[1][0]), "+v"(sum[2][0]), "+v"(sum[3][0]),      \
                       "+v"(sum[0][1]), "+v"(sum[1][1]), "+v"(sum[2][1]), "+v"(sum[3][1]),      \
                       "+v"(sum[0][NCH - 1]), "+v"(sum[1][NCH - 1]), "+v"(sum[2][NCH - 1]),     \
                       "+v"(sum[3][NCH - 1]), "+v"(wsum[0]), "+v"(wsum[1]), "+v"(wsum[2]),      \
                       "+v"(wsum[3]));                                                          \
    } else {                                                                                    \
        asm volatile(""                                                                         \
                     : "+v"(sum[0][0]), "+v"(sum[1][0]), "+v"(sum[2][0]), "+v"(sum[3][0]),      \
                       "+v"(wsum[0]), "+v"(wsum[1]), "+v"(wsum[2]), "+v"(wsum[3]));             \
    }
        // one column: issue texel(+2) and gathers(+1), accumulate column +0 underneath them,
        // then release what was issued
#define RF_STEP(U)                                                                            \
    RF_LDS_READ_B64(tq[((U) + 2) & 3], ta, RF_TEXEL_OFF(U));                                  \
    issue_gathers(tq[((U) + 1) & 3].x, gg[((U) + 1) & 1]);                                    \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    RF_ACCUM(U)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    RF_PIN_ACC()                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                       \
                 : "+v"(tq[((U) + 2) & 3]), "+v"(gg[((U) + 1) & 1][0]),                       \
                   "+v"(gg[((U) + 1) & 1][1]), "+v"(gg[((U) + 1) & 1][2]),                    \
                   "+v"(gg[((U) + 1) & 1][3]));                                               \
    __builtin_amdgcn_sched_barrier(0);

        for (int gq = 0; gq < ngroups; gq++) {
            float wv[8];
            wv[0] = wna.x; wv[1] = wna.y; wv[2] = wna.z; wv[3] = wna.w;
            wv[4] = wnb.x; wv[5] = wnb.y; wv[6] = wnb.z; wv[7] = wnb.w;
            RF_STEP(0)
            RF_STEP(1)
            RF_STEP(2)
            // u = 3 also fetches the next group's weight window
            RF_LDS_READ_B64(tq[1], ta, RF_TEXEL_OFF(3));
            issue_gathers(tq[0].x, gg[0]);
            wa_addr -= 16;
            RF_LDS_READ_B128(wna, wa_addr, 0);
            RF_LDS_READ_B128(wnb, wa_addr, 16);
            __builtin_amdgcn_sched_barrier(0);
            RF_ACCUM(3)
            __builtin_amdgcn_sched_barrier(0);
            RF_PIN_ACC()
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(tq[1]), "+v"(wna), "+v"(wnb), "+v"(gg[0][0]), "+v"(gg[0][1]),
                           "+v"(gg[0][2]), "+v"(gg[0][3]));
            __builtin_amdgcn_sched_barrier(0);
            ta += 8;
        }
#undef RF_STEP
#undef RF_PIN_ACC
#undef RF_ACCUM
#undef RF_TEXEL_OFF
    }
}

// Same contract as jbf_tap_loop, deeper pipeline: one step = TWO columns.  While columns
// (2s, 2s+1) are accumulated, the 8 LUT gathers of columns (2s+2, 2s+3) and the texels of
// columns (2s+4, 2s+5) are in flight; all of them are released at the end of the step.  One loop
// trip = 8 columns = two 4-column weight groups.  (With one column per step the accumulation
// is shorter than an LDS round trip at 2 waves/SIMD and the loop is latency-bound.)
template <int NCH, int LUTREP, bool CLAMP, int TLW>
__device__ __forceinline__ void jbf_tap_loop_pairs(uint32_t lut_lane_addr, uint32_t sw_addr0,
                                                   uint32_t tile_lane_addr,
                                                   const uint32_t (&jc)[kPix], uint32_t amax, int ty,
                                                   int radius, int r4, int sw_len,
                                                   const int *__restrict__ hwtab,
                                                   float (&sum)[kPix][NCH], float (&wsum)[kPix])
{
    constexpr int Q4 = TLW / 4;
    auto issue_gathers = [&](uint32_t jtex, float *g) {
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            uint32_t alpha = __builtin_amdgcn_sad_u8(jtex, jc[p], 0u);
            if (CLAMP)
                alpha = min(alpha, amax);
            const uint32_t a = alpha * (LUTREP * 4u) + lut_lane_addr;
            RF_LDS_READ_B32(g[p], a);
        }
    };
    // byte offset of body column cb (0..11) from the body's base texel address
#define RF_COL_OFF(cb) ((((cb) & 3) * Q4 + ((cb) >> 2)) * 8)

    for (int i = -radius; i <= radius; i++) {
        const int hw = hwtab[i + radius];
        const int hw4 = (hw + 3) & ~3;
        const int ai = i < 0 ? -i : i;
        uint32_t ta = tile_lane_addr +
                      (uint32_t)(((ty + i + radius) * TLW + ((r4 - hw4) >> 2)) * 8);
        // group gq (4 columns) needs swc[hw4 - 4*gq - 4 .. +3]; a body holds groups 2b, 2b+1
        uint32_t wa_addr = sw_addr0 + (uint32_t)((ai * sw_len + (r4 + 8) + hw4 - 4) * 4);
        const int nbodies = ((hw4 >> 1) + 2) >> 1;

        uint2v tq[8];       // texel of body column cb lives in tq[cb & 7]
        float gg[2][2][kPix];  // gathers of pair s live in gg[s & 1][column in pair]
        // weight windows of the body's two groups overlap: group 2b uses (wf1, wf2), group
        // 2b+1 uses (wf0, wf1); wf0 starts 4 floats below the first group's window
        float4v wf0, wf1, wf2;
        RF_LDS_READ_B64(tq[0], ta, RF_COL_OFF(0));
        RF_LDS_READ_B64(tq[1], ta, RF_COL_OFF(1));
        RF_LDS_READ_B64(tq[2], ta, RF_COL_OFF(2));
        RF_LDS_READ_B64(tq[3], ta, RF_COL_OFF(3));
        wa_addr -= 16;
        RF_LDS_READ_B128(wf0, wa_addr, 0);
        RF_LDS_READ_B128(wf1, wa_addr, 16);
        RF_LDS_READ_B128(wf2, wa_addr, 32);
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(tq[0]), "+v"(tq[1]));
        issue_gathers(tq[0].x, gg[0][0]);
        issue_gathers(tq[1].x, gg[0][1]);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(tq[2]), "+v"(tq[3]), "+v"(wf0), "+v"(wf1), "+v"(wf2),
                       "+v"(gg[0][0][0]), "+v"(gg[0][0][1]), "+v"(gg[0][0][2]), "+v"(gg[0][0][3]),
                       "+v"(gg[0][1][0]), "+v"(gg[0][1][1]), "+v"(gg[0][1][2]), "+v"(gg[0][1][3]));

#define RF_ACCUM_COL(CB, G)                                                          \
    {                                                                                \
        const uint32_t sv = tq[(CB) & 7].y;                                          \
        float s[NCH];                                                                \
        s[0] = (float)(sv & 0xff);                                                   \
        if (NCH == 3) {                                                              \
            s[1] = (float)((sv >> 8) & 0xff);                                        \
            s[2] = (float)((sv >> 16) & 0xff);                                       \
        }                                                                            \
        _Pragma("unroll") for (int p = 0; p < kPix; p++)                             \
        {                                                                            \
            const float wgt = __fmul_rn(wv[(CB) >> 2][4 + p - ((CB) & 3)], (G)[p]);  \
            _Pragma("unroll") for (int ch = 0; ch < NCH; ch++) sum[p][ch] =          \
                __fadd_rn(sum[p][ch], __fmul_rn(wgt, s[ch]));                        \
            wsum[p] = __fadd_rn(wsum[p], wgt);                                       \
        }                                                                            \
    }
        // Pins the accumulators at this point of the instruction stream (no instruction).
#define RF_PIN_ACC()                                                                            \
    if constexpr (NCH == 3) {                                                                   \
        asm volatile(""                                                                         \
                     : "+v"(sum[0][0]), "+v"(sum[1][0]), "+v"(sum[2][0]), "+v"(sum[3][0]),      \
                       "+v"(sum[0][1]), "+v"(sum[1][1]), "+v"(sum[2][1]), "+v"(sum[3][1]),      \
                       "+v"(sum[0][NCH - 1]), "+v"(sum[1][NCH - 1]), "+v"(sum[2][NCH - 1]),     \
                       "+v"(sum[3][NCH - 1]), "+v"(wsum[0]), "+v"(wsum[1]), "+v"(wsum[2]),      \
                       "+v"(wsum[3]));                                                          \
    } else {                                                                                    \
        asm volatile(""                                                                         \
                     : "+v"(sum[0][0]), "+v"(sum[1][0]), "+v"(sum[2][0]), "+v"(sum[3][0]),      \
                       "+v"(wsum[0]), "+v"(wsum[1]), "+v"(wsum[2]), "+v"(wsum[3]));             \
    }
        // pair-step S (0..3): accumulate body columns 2S, 2S+1; gathers of 2S+2, 2S+3; texels
        // of 2S+4, 2S+5
#define RF_PAIR_ISSUE(S)                                                                      \
    RF_LDS_READ_B64(tq[(2 * (S) + 4) & 7], ta, RF_COL_OFF(2 * (S) + 4));                      \
    RF_LDS_READ_B64(tq[(2 * (S) + 5) & 7], ta, RF_COL_OFF(2 * (S) + 5));                      \
    issue_gathers(tq[(2 * (S) + 2) & 7].x, gg[((S) + 1) & 1][0]);                             \
    issue_gathers(tq[(2 * (S) + 3) & 7].x, gg[((S) + 1) & 1][1]);
#define RF_PAIR_ACCUM(S)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    RF_ACCUM_COL(2 * (S), gg[(S) & 1][0])                                                     \
    RF_ACCUM_COL(2 * (S) + 1, gg[(S) & 1][1])                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    RF_PIN_ACC()                                                                              \
    __builtin_amdgcn_sched_barrier(0);
#define RF_PAIR_WAIT(S)                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                       \
                 : "+v"(tq[(2 * (S) + 4) & 7]), "+v"(tq[(2 * (S) + 5) & 7]),                  \
                   "+v"(gg[((S) + 1) & 1][0][0]), "+v"(gg[((S) + 1) & 1][0][1]),              \
                   "+v"(gg[((S) + 1) & 1][0][2]), "+v"(gg[((S) + 1) & 1][0][3]),              \
                   "+v"(gg[((S) + 1) & 1][1][0]), "+v"(gg[((S) + 1) & 1][1][1]),              \
                   "+v"(gg[((S) + 1) & 1][1][2]), "+v"(gg[((S) + 1) & 1][1][3]));             \
    __builtin_amdgcn_sched_barrier(0);

        for (int b = 0; b < nbodies; b++) {
            float wv[2][8];
            wv[0][0] = wf1.x; wv[0][1] = wf1.y; wv[0][2] = wf1.z; wv[0][3] = wf1.w;
            wv[0][4] = wf2.x; wv[0][5] = wf2.y; wv[0][6] = wf2.z; wv[0][7] = wf2.w;
            wv[1][0] = wf0.x; wv[1][1] = wf0.y; wv[1][2] = wf0.z; wv[1][3] = wf0.w;
            wv[1][4] = wf1.x; wv[1][5] = wf1.y; wv[1][6] = wf1.z; wv[1][7] = wf1.w;
            RF_PAIR_ISSUE(0) RF_PAIR_ACCUM(0) RF_PAIR_WAIT(0)
            RF_PAIR_ISSUE(1) RF_PAIR_ACCUM(1) RF_PAIR_WAIT(1)
            RF_PAIR_ISSUE(2) RF_PAIR_ACCUM(2) RF_PAIR_WAIT(2)
            // last pair of the body: also fetch the next body's two weight windows
            RF_PAIR_ISSUE(3)
            wa_addr -= 32;
            RF_LDS_READ_B128(wf0, wa_addr, 0);
            RF_LDS_READ_B128(wf1, wa_addr, 16);
            RF_LDS_READ_B128(wf2, wa_addr, 32);
            RF_PAIR_ACCUM(3)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf0), "+v"(wf1), "+v"(wf2));
            RF_PAIR_WAIT(3)
            ta += 16;
        }
#undef RF_PAIR_ISSUE
#undef RF_PAIR_ACCUM
#undef RF_PAIR_WAIT
#undef RF_PIN_ACC
#undef RF_ACCUM_COL
    }
#undef RF_COL_OFF
}
#undef RF_LDS_READ_B64
#undef RF_LDS_READ_B128
#undef RF_LDS_READ_B32

template <int SCN, int TH, int LUTREP, bool CLAMP, bool PAIRS>
__global__ __launch_bounds__(16 * TH) void jbf_tiled2_kernel(
    const uint8_t *__restrict__ joint, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
    int h, int w, int jcn, int radius, int border, const float *__restrict__ lut, int lut_len,
    const int *__restrict__ hwtab, const float *__restrict__ swsym, int sw_len, int tiles_x,
    int tiles_per_img, int flags)
{
    constexpr int NT = 16 * TH;
    constexpr int TLW = kTlw2;
    constexpr int Q4 = TLW / 4;
    extern __shared__ __align__(16) unsigned char smem[];
    float *lutrep = reinterpret_cast<float *>(smem);
    const int lut_bytes = (lut_len * LUTREP * 4 + 15) & ~15;
    float *swl = reinterpret_cast<float *>(smem + lut_bytes);
    const int sw_bytes = ((radius + 1) * sw_len * 4 + 15) & ~15;
    uint2 *tile = reinterpret_cast<uint2 *>(smem + lut_bytes + sw_bytes);

    const int tid = threadIdx.x;
    const int img_idx = blockIdx.x / tiles_per_img;
    const int t_in_img = blockIdx.x - img_idx * tiles_per_img;
    const int tile_y0 = (t_in_img / tiles_x) * TH;
    const int tile_x0 = (t_in_img % tiles_x) * kTileW;
    const size_t img = (size_t)img_idx * h * w;
    const int r4 = (radius + 3) & ~3;
    const int tlh = TH + 2 * radius;

    for (int i = tid; i < lut_len * LUTREP; i += NT)
        lutrep[i] = lut[i / LUTREP];
    for (int i = tid; i < (radius + 1) * sw_len; i += NT)
        swl[i] = swsym[i];
    const int tlw_used = TLW;  // includes the pipeline's look-ahead / zero-weight columns
    int grey = 1;  // every src texel staged by this thread has B == G == R
    for (int ry = tid >> 6; ry < tlh; ry += NT >> 6) {
        const int gy = border_interpolate(tile_y0 - radius + ry, h, border);
        for (int X = tid & 63; X < tlw_used; X += 64) {
            const int gx = border_interpolate(tile_x0 - r4 + X, w, border);
            uint2 t = make_uint2(0u, 0u);
            if (gy >= 0 && gx >= 0) {
                const size_t q = img + (size_t)gy * w + gx;
                t.x = load_packed(joint, q, jcn);
                t.y = load_packed(src, q, SCN);
            }
            if (SCN == 3)
                grey &= (int)(((t.y ^ (t.y >> 8)) & 0xffffu) == 0u);
            tile[ry * TLW + (X & 3) * Q4 + (X >> 2)] = t;
        }
    }
    const int all_grey = __syncthreads_and(grey);  // also the barrier that publishes the tile

    const int tx = tid & 15;
    const int ty = tid >> 4;
    const uint32_t lane_lut = (uint32_t)(tid & (LUTREP - 1));
    const uint32_t amax = (uint32_t)(lut_len - 1);
    uint32_t jc[kPix];
#pragma unroll
    for (int p = 0; p < kPix; p++) {
        const int X = 4 * tx + p + r4;
        jc[p] = tile[(ty + radius) * TLW + (X & 3) * Q4 + (X >> 2)].x;
    }
    const uint32_t lut_lane_addr = lds_addr(lutrep) + lane_lut * 4u;
    const uint32_t sw_addr0 = lds_addr(swl);
    const uint32_t tile_lane_addr = lds_addr(tile) + (uint32_t)tx * 8u;

    float sum[kPix][SCN];
    float wsum[kPix];
#pragma unroll
    for (int p = 0; p < kPix; p++) {
        wsum[p] = 0.f;
#pragma unroll
        for (int c = 0; c < SCN; c++)
            sum[p][c] = 0.f;
    }
    if (SCN == 3 && !all_grey) {
        if (PAIRS)
            jbf_tap_loop_pairs<SCN, LUTREP, CLAMP, TLW>(lut_lane_addr, sw_addr0, tile_lane_addr,
                                                        jc, amax, ty, radius, r4, sw_len, hwtab,
                                                        sum, wsum);
        else
            jbf_tap_loop<SCN, LUTREP, CLAMP, TLW>(lut_lane_addr, sw_addr0, tile_lane_addr, jc,
                                                  amax, ty, radius, r4, sw_len, hwtab, sum, wsum);
    } else {
        // single-channel accumulation; for a grey 3-channel src the three sums are the same
        // sequence of float operations, so replicating one of them is bit-identical
        float sum1[kPix][1];
#pragma unroll
        for (int p = 0; p < kPix; p++)
            sum1[p][0] = 0.f;
        if (PAIRS)
            jbf_tap_loop_pairs<1, LUTREP, CLAMP, TLW>(lut_lane_addr, sw_addr0, tile_lane_addr, jc,
                                                      amax, ty, radius, r4, sw_len, hwtab, sum1,
                                                      wsum);
        else
            jbf_tap_loop<1, LUTREP, CLAMP, TLW>(lut_lane_addr, sw_addr0, tile_lane_addr, jc, amax,
                                                ty, radius, r4, sw_len, hwtab, sum1, wsum);
#pragma unroll
        for (int p = 0; p < kPix; p++)
#pragma unroll
            for (int c = 0; c < SCN; c++)
                sum[p][c] = sum1[p][0];
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

struct Tiled2Config {
    int th, lutrep;
    bool full_lut;  // stage all 256*jcn LUT entries and skip the per-tap clamp
    bool pairs;     // two columns per pipeline step
};

size_t tiled2_lds_bytes(const JbfTables &t, int th, int lutrep, bool full_lut)
{
    const int lut_len = full_lut ? 256 * t.joint_cn : t.lut_len;
    const size_t lut_bytes = ((size_t)lut_len * lutrep * 4 + 15) & ~(size_t)15;
    const size_t sw_bytes = ((size_t)(t.radius + 1) * t.sw_len * 4 + 15) & ~(size_t)15;
    return lut_bytes + sw_bytes + (size_t)kTlw2 * (th + 2 * t.radius) * sizeof(uint2);
}

template <int SCN, int TH, int LUTREP, bool PAIRS>
int launch_tiled2(const JbfTables &t, bool full_lut, const uint8_t *joint, const uint8_t *src,
                  uint8_t *dst, int n, int h, int w, int jcn, int border, int flags,
                  hipStream_t stream)
{
    const int lut_len = full_lut ? 256 * jcn : t.lut_len;
    const bool clamp = lut_len < 256 * jcn;
    const size_t lds = tiled2_lds_bytes(t, TH, LUTREP, full_lut);
    const int tiles_x = ceil_div(w, kTileW), tiles_y = ceil_div(h, TH);
    const long long blocks = (long long)tiles_x * tiles_y * n;
    if (blocks > 0x7fffffffLL)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: batch too large for one launch");
    auto kern = clamp ? jbf_tiled2_kernel<SCN, TH, LUTREP, true, PAIRS>
                      : jbf_tiled2_kernel<SCN, TH, LUTREP, false, PAIRS>;
    RF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(16 * TH), lds, stream, joint, src, dst, h,
                       w, jcn, t.radius, border, t.d_lut, lut_len, t.d_hw, t.d_swsym, t.sw_len,
                       tiles_x, tiles_x * tiles_y, flags);
    return RF_OK;
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

    // ---- kernel selection -------------------------------------------------------------
    // tune = 0: automatic.  1: first-generation tiled kernel.  2..12: v2 with a fixed
    // (tile height, LUT replicas, full LUT, pair steps) configuration from kCfg.
    const int tune = (flags >> RF_JBF_TUNE_SHIFT) & 0xf;
    static const Tiled2Config kCfg[] = {
        {32, 32, false, false}, {32, 16, false, false}, {48, 16, false, false},
        {48, 8, false, false},  {32, 8, false, false},  {32, 8, true, false},
        {48, 4, true, false},   {32, 8, true, true},    {48, 4, true, true},
        {32, 32, false, true},  {48, 8, false, true}};
    // measured on MI355X at 1080p (tools/jbf_tune.py): 3 waves/SIMD with a 16x replicated,
    // clamped LUT wins for grey src; the clamp-free 8x table is next
    static const int kAutoOrder[] = {2, 5, 0, 1, 4};
    int cfg = -1;
    const bool v2_radius_ok = t.r4 <= 36;
    if (!(flags & RF_JBF_FORCE_GENERIC) && v2_radius_ok) {
        if (tune >= 2 && tune <= 12) {
            const Tiled2Config &c = kCfg[tune - 2];
            if (tiled2_lds_bytes(t, c.th, c.lutrep, c.full_lut) <= (size_t)kMaxLds)
                cfg = tune - 2;
        } else if (tune == 0) {
            for (int ci : kAutoOrder) {
                const Tiled2Config &c = kCfg[ci];
                if (tiled2_lds_bytes(t, c.th, c.lutrep, c.full_lut) <= (size_t)kMaxLds) {
                    cfg = ci;
                    break;
                }
            }
        }
    }
    int tlw = 0, tlh = 0;
    size_t lds = 0;
    const bool tiled_ok = tiled_geometry(radius, t.lut_len, &tlw, &tlh, &lds);
    if (cfg >= 0) {
#define RF_T2_(TH_, REP_, P_)                                                                   \
    rc = src_cn == 3                                                                            \
             ? launch_tiled2<3, TH_, REP_, P_>(t, kCfg[cfg].full_lut, joint, src, dst, n, h, w,     \
                                               joint_cn, border, flags, stream)                 \
             : launch_tiled2<1, TH_, REP_, P_>(t, kCfg[cfg].full_lut, joint, src, dst, n, h, w,     \
                                               joint_cn, border, flags, stream)
#define RF_T2(TH_, REP_, P_) RF_T2_(TH_, REP_, P_)
        switch (cfg) {
        case 0: RF_T2(32, 32, false); break;
        case 1: RF_T2(32, 16, false); break;
        case 2: RF_T2(48, 16, false); break;
        case 3: RF_T2(48, 8, false); break;
        case 4: RF_T2(32, 8, false); break;
        case 5: RF_T2(32, 8, false); break;
        case 6: RF_T2(48, 4, false); break;
        case 7: RF_T2(32, 8, true); break;
        case 8: RF_T2(48, 4, true); break;
        case 9: RF_T2(32, 32, true); break;
        default: RF_T2(48, 8, true); break;
        }
#undef RF_T2_
#undef RF_T2
        if (rc != RF_OK)
            return rc;
    } else if (tiled_ok && !(flags & RF_JBF_FORCE_GENERIC) && (tune == 0 || tune == 1)) {
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
