// rf_gf.hip -- colour guided filter, uint8, for gfx950 (MI355X).
//
// Replaces cv2.ximgproc.guidedFilter(guide, src, radius, eps) as called at
// /root/reference/filter_reflectance.py:67-70.  Arithmetic contract (DESIGN.md "GF"): the
// operation order of opencv_contrib/modules/ximgproc/src/guided_filter.cpp (3-channel guide)
// over cv::boxFilter(CV_32F, normalize, BORDER_REFLECT) whose sums are double:
//
//   stage 1  box means of I_g, I_g*I_g', p_s, p_s*I_g.  Inputs are integers <= 65025, so the
//            double running sums OpenCV forms are exact integers; we form the same integers
//            with uint32 arithmetic (any order is exact) and round once:
//            mean = (float)((double)S * (1.0/k^2)).                         -> gf_stage1_kernel
//            The per-pixel algebra up to alpha/beta is fused behind it.
//   stage 2  box means of the float planes alpha_{s,g}, beta_s.  Not exact in double, so the
//            summation ORDER matters: RowSum<float,double> is a running sum along the
//            border-extended row starting at its left end, ColumnSum<double,float> a running
//            sum down the image starting 2r rows above the first output row.  Both are
//            reproduced as sequential chains (one lane per row / per column), exposed to the
//            GPU as parallelism over rows x planes x images, in two forms with identical bytes:
//            radius 1..128: gf_rowstate_kernel<R> + gf_colwalk_kernel<R> (rf_gf_fused.hpp,
//            instantiated per radius in rf_gf_fused_inst.hip) - the double row sums never reach
//            HBM; radius 0 (and the switch gf_two_kernel): gf_rowsum_kernel + gf_colsum_apply_kernel, which write
//            every row sum (8 B per pixel and plane) and read it twice; radius 129..4096: the
//            float kernels of rf_gf_f32 on float copies of the images, each pass rounded to uint8
//            (on 8-bit data their double window sums are the same exact integers).
//
// Grey sources: the reference filters the CNN's grey `-r.png`, which imread turns into three
// identical channels.  The src channels never mix, so identical channels give identical
// outputs; gf_grey_probe_kernel marks such images (a device-side flag, no host round trip) and
// they run the one-channel instantiation with the result byte written three times (1/3 of the
// per-channel planes).  Every stage is launched in both instantiations; workgroups of the one
// that does not apply to their image exit at once.  The probe also leaves channel 0 of every image
// as one byte per pixel in the workspace: that copy is what stage 1 reads for a grey image and
// what the passes of an iterated call hand on, until the last pass writes dst.
#include "rf_gf_fused.hpp"

#include <algorithm>
#include <mutex>
#include <vector>

namespace rf {
namespace {

// ------------------------------------------------------------------------------------------
// stage 1 + per-pixel algebra
// ------------------------------------------------------------------------------------------
// Threads per workgroup and columns per thread (strip width incl. halo = threads x columns) by
// the number of src channels computed.  One channel (13 quantities): 256 x 3 - 40 KB of prefix
// sums, 4 workgroups = 16 waves per CU, 768-column strips (2r of them halo).  Three channels
// (21 quantities): 256 x 2 - 43 KB, 3 workgroups = 12 waves per CU.  Measured alternatives for
// three channels (8 x 4K pass, best rows per segment each): 320 x 2 (15 waves per CU, seven
// strips cover 3840 columns exactly, but five waves per workgroup load the four SIMDs unevenly
// between barriers) 5.04 ms against 4.95; 512 x 1 at 80 registers (24 waves per CU, a fifth
// more scan work per pixel) 5.56 ms.
constexpr int stage1_threads(int) { return 256; }
constexpr int stage1_cols(int scn) { return scn == 1 ? 3 : 2; }

// Stage-1 quantities.  GUIDE = true: all of them, q: 0..2 I_g | 3..8 I_aI_b (00 01 02 11 12 22) |
// 9.. p_s | then p_s*I_g (s major).  GUIDE = false (later passes of an iterated call, whose guide
// statistics come from the record the first pass left): only p_s | p_s*I_g.
template <int SCN, bool GUIDE = true>
struct Quant {
    static constexpr int G0 = GUIDE ? 9 : 0;  // first src quantity
    static constexpr int NQ = G0 + 4 * SCN;
    // acc[q] += (NEG ? -1 : +1) * quantity q of the pixel with guide bytes g0..g2 and src bytes p[]
    // (uint32 wrap-around is exact).  A row that leaves the window is added with one factor of
    // every product negated: one v_mad_i32_i24 per product either way, instead of a multiply and
    // a subtract.
    template <bool NEG>
    __device__ static inline void accumulate(int g0, int g1, int g2, const int *p, uint32_t *acc)
    {
        const int s0 = NEG ? -g0 : g0, s1 = NEG ? -g1 : g1, s2 = NEG ? -g2 : g2;
        if (GUIDE) {
            acc[0] += (uint32_t)s0;
            acc[1] += (uint32_t)s1;
            acc[2] += (uint32_t)s2;
            acc[3] += (uint32_t)(s0 * g0);
            acc[4] += (uint32_t)(s0 * g1);
            acc[5] += (uint32_t)(s0 * g2);
            acc[6] += (uint32_t)(s1 * g1);
            acc[7] += (uint32_t)(s1 * g2);
            acc[8] += (uint32_t)(s2 * g2);
        }
#pragma unroll
        for (int s = 0; s < SCN; s++) {
            const int ps = NEG ? -p[s] : p[s];
            acc[G0 + s] += (uint32_t)ps;
            acc[G0 + SCN + 3 * s + 0] += (uint32_t)(ps * g0);
            acc[G0 + SCN + 3 * s + 1] += (uint32_t)(ps * g1);
            acc[G0 + SCN + 3 * s + 2] += (uint32_t)(ps * g2);
        }
    }
};

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (4 shifts inside each row of
// 16 lanes, then the row totals are broadcast forward): 6 VALU instructions instead of the 6
// ds_bpermute round trips of a __shfl_up scan.
__device__ inline uint32_t wave_inclusive_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31
    return v;
}

// mean = (float)((double)S * scale), OpenCV's `(float)(sum * scale)` on the exact window sum S.
// (double)S is formed without a conversion instruction: 2^52 + S has the bit pattern
// {0x43300000, S}, and fma(2^52 + S, scale, -(2^52 * scale)) rounds the exact product S * scale
// once (2^52 * scale is a power-of-two multiple of scale, hence exact) - the same double as
// cvt + mul, one 4-cycle instruction less per quantity and pixel.  nbias = -(2^52 * scale).
__device__ inline float mean_of(uint32_t s, double scale, double nbias)
{
    return (float)__fma_rn(__hiloint2double(0x43300000, (int)s), scale, nbias);
}

// index into the 6-entry symmetric store: (0,0)=0 (0,1)=1 (0,2)=2 (1,1)=3 (1,2)=4 (2,2)=5
__device__ constexpr int sym(int i, int j)
{
    return i <= j ? (i * 3 - i * (i - 1) / 2 + (j - i)) : (j * 3 - j * (j - 1) / 2 + (i - j));
}

// Per-pixel algebra of guided_filter.cpp, every operation the separately rounded float op of the
// corresponding OpenCV helper, in two halves:
//   gf_guide_algebra  from the 9 guide means m (order of Quant): covariance of the guide (+eps on
//                     the diagonal) and its inverse by cofactors -> gs[0..2] = mean I_g,
//                     gs[3..8] = inverse (symmetric store).  Depends on the guide only: the passes
//                     of an iterated call share it (kGsFloats floats per pixel).
//   gf_src_algebra    from gs and the 4*SCN src means ms (p_s | p_s*I_g): for every src channel s
//                     the coefficients alpha_{s,g} (out[4s + g]) and beta_s (out[4s + 3]).
constexpr int kGsFloats = kGsFloatsPublic;

// out[i] = num[i] / den for six numerators, every quotient the correctly rounded IEEE result
// (__fdiv_rn), i.e. what OpenCV's per-element division gives.  hipcc expands an IEEE division into
// v_div_scale x 2, v_rcp, two FMAs that refine the reciprocal, a multiply, three FMAs, v_div_fmas
// and v_div_fixup (46 issue cycles); the reciprocal and its refinement depend on the denominator
// alone as long as v_div_scale leaves both operands unscaled, which it does whenever the
// exponents are far from the float range's ends.  That case is recognised up front (|den| in
// [2^-30, 2^50], every numerator zero or in [2^-40, 2^36]: no scaling, no denormal quotient) and
// runs the SAME instruction sequence with the denominator's part done once; v_div_fixup stays,
// it is what gives a zero numerator its sign.  Anything else takes the plain divisions.
__device__ inline void div6_by(const float *num, float den, float *out)
{
    // |den| in [2^-30, 2^50]; every numerator zero or in [2^-40, 2^36].  The upper bound is tested on
    // the SUM of the magnitudes (source modifiers, no masking instructions; a NaN or an infinity
    // anywhere propagates and fails the comparison; a sum above 2^36 whose terms are all below it
    // merely takes the plain divisions); the smallest non-zero numerator is an unsigned minimum
    // over (bits << 1) - 2, which drops the sign and sends a zero to the top (one v_lshl_add_u32
    // per numerator).
    const float ad = fabsf(den);
    const float hi = ((fabsf(num[0]) + fabsf(num[1])) + (fabsf(num[2]) + fabsf(num[3]))) +
                     (fabsf(num[4]) + fabsf(num[5]));
    uint32_t lo = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 6; i++)
        lo = min(lo, (__float_as_uint(num[i]) << 1) - 2u);
    // (comparisons written so that a NaN anywhere fails them)
    const bool fast = ad >= 0x1p-30f && ad <= 0x1p50f && hi <= 0x1p36f &&
                      lo >= (0x2b800000u << 1) - 2u;  // 2^-40
    if (fast) {
        const float r0 = __builtin_amdgcn_rcpf(den);
        const float e = __fmaf_rn(-den, r0, 1.0f);
        const float r1 = __fmaf_rn(e, r0, r0);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const float q0 = __fmul_rn(num[i], r1);
            const float t = __fmaf_rn(-den, q0, num[i]);
            const float q1 = __fmaf_rn(t, r1, q0);
            const float t2 = __fmaf_rn(-den, q1, num[i]);
            out[i] = __builtin_amdgcn_div_fixupf(__fmaf_rn(t2, r1, q1), den, num[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++)
            out[i] = __fdiv_rn(num[i], den);
    }
}

__device__ inline void gf_guide_algebra(const float *m, float eps_f, int eps_small, float *gs)
{
    const float *mI = m;
    float cov[6];
    // cov(c1,c2) = mean(I1*I2) - mean1*mean2 ; diagonal: - (mean*mean + (-eps))
    cov[sym(0, 1)] = __fsub_rn(m[4], __fmul_rn(mI[0], mI[1]));
    cov[sym(0, 2)] = __fsub_rn(m[5], __fmul_rn(mI[0], mI[2]));
    cov[sym(1, 2)] = __fsub_rn(m[7], __fmul_rn(mI[1], mI[2]));
    cov[sym(0, 0)] = __fsub_rn(m[3], __fadd_rn(__fmul_rn(mI[0], mI[0]), -eps_f));
    cov[sym(1, 1)] = __fsub_rn(m[6], __fadd_rn(__fmul_rn(mI[1], mI[1]), -eps_f));
    cov[sym(2, 2)] = __fsub_rn(m[8], __fadd_rn(__fmul_rn(mI[2], mI[2]), -eps_f));
    float inv[6];
#pragma unroll
    for (int kk = 0; kk < 3; kk++)
#pragma unroll
        for (int l = 0; l <= kk; l++) {
            const float a00 = cov[sym((kk + 1) % 3, (l + 1) % 3)];
            const float a01 = cov[sym((kk + 1) % 3, (l + 2) % 3)];
            const float a10 = cov[sym((kk + 2) % 3, (l + 1) % 3)];
            const float a11 = cov[sym((kk + 2) % 3, (l + 2) % 3)];
            inv[sym(kk, l)] = __fsub_rn(__fmul_rn(a00, a11), __fmul_rn(a01, a10));
        }
    float det = __fmul_rn(cov[sym(0, 0)], inv[sym(0, 0)]);
    det = __fadd_rn(det, __fmul_rn(cov[sym(1, 0)], inv[sym(1, 0)]));
    det = __fadd_rn(det, __fmul_rn(cov[sym(2, 0)], inv[sym(2, 0)]));
    if (eps_small && fabsf(det) < 1e-6f)
        det = 1.f;
    gs[0] = mI[0];
    gs[1] = mI[1];
    gs[2] = mI[2];
    div6_by(inv, det, gs + 3);
}

template <int SCN>
__device__ inline void gf_src_algebra(const float *gs, const float *ms, float *out)
{
    const float *mI = gs, *inv = gs + 3;
#pragma unroll
    for (int s = 0; s < SCN; s++) {
        const float mp = ms[s];
        float cp[3];
#pragma unroll
        for (int g = 0; g < 3; g++)
            cp[g] = __fsub_rn(ms[SCN + 3 * s + g], __fmul_rn(mp, mI[g]));
        float beta = mp;
#pragma unroll
        for (int g = 0; g < 3; g++) {
            float a = __fmul_rn(inv[sym(g, 0)], cp[0]);
            a = __fadd_rn(a, __fmul_rn(inv[sym(g, 1)], cp[1]));
            a = __fadd_rn(a, __fmul_rn(inv[sym(g, 2)], cp[2]));
            out[4 * s + g] = a;
            beta = __fsub_rn(beta, __fmul_rn(a, mI[g]));
        }
        out[4 * s + 3] = beta;
    }
}

// both halves: from the 9 + 4*SCN window means m (order of Quant<SCN, true>)
template <int SCN>
__device__ inline void gf_pixel_algebra(const float *m, float eps_f, int eps_small, float *out)
{
    float gs[kGsFloats];
    gf_guide_algebra(m, eps_f, eps_small, gs);
    gf_src_algebra<SCN>(gs, m + 9, out);
}

// colour[img] != 0  <=>  some pixel of the 3-channel image has unequal channels.
// grid: (blocks per image, images); colour[] zeroed beforehand.  compact (optional): [img][npx]
// bytes that receive channel 0 of every pixel - for an image that turns out grey this IS the
// image, one byte per pixel, and what the first pass's stage 1 then reads (see rf_gf_u8).
__global__ __launch_bounds__(256) void gf_grey_probe_kernel(const uint8_t *__restrict__ src,
                                                            int *__restrict__ colour, size_t npx,
                                                            uint8_t *__restrict__ compact)
{
    const uint8_t *simg = src + (size_t)blockIdx.y * npx * 3;
    uint8_t *cimg = compact ? compact + (size_t)blockIdx.y * npx : nullptr;
    const size_t nquads = npx / 4;  // 4 pixels = 12 bytes = 3 dwords (image base is 4-aligned
                                    // only when npx*3*img is; use byte-safe loads)
    bool diff = false;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
         q += (size_t)gridDim.x * blockDim.x) {
        uint32_t d0, d1, d2;
        __builtin_memcpy(&d0, simg + q * 12, 4);
        __builtin_memcpy(&d1, simg + q * 12 + 4, 4);
        __builtin_memcpy(&d2, simg + q * 12 + 8, 4);
        // bytes: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3 ; grey <=> every pixel's 3 bytes equal
        const uint32_t p0 = d0 & 0xffffffu, p1 = (d0 >> 24) | ((d1 & 0xffffu) << 8);
        const uint32_t p2 = (d1 >> 16) | ((d2 & 0xffu) << 16), p3 = d2 >> 8;
        diff |= p0 != (p0 & 0xffu) * 0x010101u || p1 != (p1 & 0xffu) * 0x010101u ||
                p2 != (p2 & 0xffu) * 0x010101u || p3 != (p3 & 0xffu) * 0x010101u;
        if (cimg != nullptr) {
            const uint32_t c4 = (p0 & 0xffu) | ((p1 & 0xffu) << 8) | ((p2 & 0xffu) << 16) | (p3 << 24 & 0xff000000u);
            __builtin_memcpy(cimg + q * 4, &c4, 4);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(npx - nquads * 4)) {
        const uint8_t *p = simg + (nquads * 4 + threadIdx.x) * 3;
        diff |= p[0] != p[1] || p[1] != p[2];
        if (cimg != nullptr)
            cimg[nquads * 4 + threadIdx.x] = p[0];
    }
    if (diff)
        colour[blockIdx.y] = 1;
}

// Last pass of a colour image that travelled as planes: planes [img][3][npx] -> dst [img][npx][3].
// grid: (blocks per image, images); images whose flag colour[img] is 0 (grey: their last pass wrote dst
// itself) are skipped.  Four pixels per thread: three 4-byte loads, one 12-byte store.
__global__ __launch_bounds__(256) void gf_interleave3_kernel(const uint8_t *__restrict__ planes,
                                                             uint8_t *__restrict__ dst, size_t npx,
                                                             const int *__restrict__ colour)
{
    if (colour[blockIdx.y] == 0)
        return;
    const uint8_t *p = planes + (size_t)blockIdx.y * npx * 3;
    uint8_t *d = dst + (size_t)blockIdx.y * npx * 3;
    const size_t nquads = npx / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads;
         q += (size_t)gridDim.x * blockDim.x) {
        uint32_t c0, c1, c2;
        __builtin_memcpy(&c0, p + q * 4, 4);
        __builtin_memcpy(&c1, p + npx + q * 4, 4);
        __builtin_memcpy(&c2, p + 2 * npx + q * 4, 4);
        // bytes out: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3   (channel planes c0, c1, c2)
        const uint32_t o[3] = {
            (c0 & 0xffu) | ((c1 & 0xffu) << 8) | ((c2 & 0xffu) << 16) | ((c0 & 0xff00u) << 16),
            ((c1 >> 8) & 0xffu) | (((c2 >> 8) & 0xffu) << 8) | (((c0 >> 16) & 0xffu) << 16) |
                (((c1 >> 16) & 0xffu) << 24),
            ((c2 >> 16) & 0xffu) | ((c0 >> 24) << 8) | ((c1 >> 24) << 16) | ((c2 >> 24) << 24)};
        __builtin_memcpy(d + q * 12, o, 12);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(npx - nquads * 4)) {
        const size_t i = nquads * 4 + threadIdx.x;
        d[i * 3 + 0] = p[i];
        d[i * 3 + 1] = p[npx + i];
        d[i * 3 + 2] = p[2 * npx + i];
    }
}

// grid: (strips, row segments, images).  ab: [img][SPX][h][w][4] float (g<3 alpha, g=3 beta).
// SCN = src channels computed, SPX = src bytes per pixel (SCN, or 3 with SCN = 1 for a grey
// 3-channel image whose first channel stands for all three).
// MODE (iterated calls: the guide is the same in every pass, /root/reference/README.md:66; only
// kS1Full is used unless the experiment switch "gf_guide_cache" is set, see rf_gf_u8):
//   kS1Full    all 9 + 4*SCN quantities, nothing kept
//   kS1Keep    the same, and the guide half of the algebra (mean I_g, inverse covariance: kGsFloats
//              floats per pixel) is stored to gs                         - first pass
//   kS1Reuse   only the 4*SCN src quantities are box-summed; the guide half is read from gs
//              (requested at the top of a row, used at its end)          - later passes
enum { kS1Full = 0, kS1Keep = 1, kS1Reuse = 2 };

// "Exact rows" (rf_gf_fused.hpp): what stage 1 leaves for a stage 2 without a row walk.
//   xf    [img * groups + group][row][plane 0..3][nb]  double: the sum of the plane's 16 values of
//         image columns 16 b .. 16 b + 15 of that row (nb = w / 16), formed as a butterfly over the 16
//         lanes that hold them (ds_swizzle: the LDS crossbar, no LDS memory).  In a row that passes
//         the exactness test every partial sum is exact, so the order is immaterial; in a row that
//         does not, nobody reads them.
//   xstat [img * groups + group][row][strip * 8 + wave * 2 + half]  two packed words per half-wave:
//         {alpha_0..2 jointly, beta}, each (top 16 bits of the largest magnitude's bit pattern << 1)
//         << 16 | 0xffff - (top 16 bits of the smallest non-zero magnitude's (bits << 1) - 2).
//         gf_exact_rows_kernel turns a row's words into its exactness flag.
struct GfExactOut {
    double *xf;
    uint2 *xstat;
    int nb, slots;  // 16-column blocks per row; statistic slots per row (strips x 8)
};
template <int PATTERN>
__device__ __forceinline__ double swizzle_add(double d)
{
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(d), PATTERN);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(d), PATTERN);
    return d + __hiloint2double(hi, lo);
}
template <int PATTERN>
__device__ __forceinline__ uint32_t swizzle_pkmax(uint32_t v)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const uint32_t o = (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, PATTERN);
    u16x2 a, b;
    __builtin_memcpy(&a, &v, 4);
    __builtin_memcpy(&b, &o, 4);
    const u16x2 m = __builtin_elementwise_max(a, b);
    uint32_t r;
    __builtin_memcpy(&r, &m, 4);
    return r;
}
// ds_swizzle bit-mask mode: lane' = ((lane & and) | or) ^ xor within 32 lanes
constexpr int swz_xor(int x) { return (x << 10) | 0x1f; }
// Diagnostic build only (-DRF_GF_S1_STAMP, tools/gf_s1_stamp.py): shader cycles per phase of the row
// loop, summed over waves, in a buffer of their own; no output depends on them.
#ifdef RF_GF_S1_STAMP
__device__ unsigned long long g_s1_stamps[16];
#define RF_S1_STAMP(i)                                                  \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        st_acc[i] += now_ - st_t;                                       \
        st_t = now_;                                                    \
    } while (0)
#else
#define RF_S1_STAMP(i) do { } while (0)
#endif
// (waves per SIMD: the one-channel kernel is built for four - 128 registers -, which the exact-row
//  additions would otherwise miss by one register; the three-channel kernel for three)
template <int SCN, int SPX, int MODE, bool EXACT = false>
__global__ __launch_bounds__(stage1_threads(SCN))
    __attribute__((amdgpu_waves_per_eu(SCN == 1 ? 4 : 3))) void gf_stage1_kernel(
    const uint8_t *__restrict__ guide, const uint8_t *__restrict__ src, float *__restrict__ ab,
    int h, int w, int radius, float eps_f, int eps_small, int seg_rows,
    const int *__restrict__ colour, float *__restrict__ gs, int ab_groups, int hl, int out_w,
    const GfExactOut xo, const uint8_t *__restrict__ src_planar)
{
    // src_planar (three-channel kernel, later passes of an iterated call): the src channels as three
    // planes [img][3][h][w] - what the previous pass's column walk left - instead of interleaved
    static_assert(!EXACT || MODE == kS1Full, "exact rows: plain stage 1 only");
    // hl, out_w: the strip's geometry - its kACW columns are image columns xs - hl .. xs - hl + kACW - 1
    // (xs = strip index x out_w), of which columns hl .. hl + out_w - 1 are its outputs; hl >= radius
    // and kACW - hl - out_w >= radius (rf_gf_u8 picks them, see gf_strip_geometry)
    // ab_groups: plane groups (src channels) an image has in ab - SPX, or 3 when a grey 3-channel
    // image is read from its one-byte-per-pixel intermediate (SPX = 1, see rf_gf_u8)
    if (wrong_variant<SCN>(colour, blockIdx.z))
        return;
    using Q = Quant<SCN, MODE != kS1Reuse>;
    constexpr int NQ = Q::NQ;
    constexpr int kAThreads = stage1_threads(SCN), kAWaves = kAThreads / 64;
    constexpr int kACols = stage1_cols(SCN);
    constexpr int kACW = kAThreads * kACols;
    __shared__ uint32_t pfx[NQ][kACW + 1];
    __shared__ uint32_t wave_acc[NQ][kAWaves];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xs = blockIdx.x * out_w;
    const int ys = blockIdx.y * seg_rows;
    const int ye = min(ys + seg_rows, h);
    const size_t npx = (size_t)h * w;
    const uint8_t *gimg = guide + (size_t)blockIdx.z * npx * 3;
    const uint8_t *simg = src + (size_t)blockIdx.z * npx * SPX;
    float *abimg = ab + (size_t)blockIdx.z * npx * (ab_groups * 4);
    float *gsimg = MODE == kS1Full ? nullptr : gs + (size_t)blockIdx.z * npx * kGsFloats;
    const int ks = 2 * radius + 1;
    const double scale = 1.0 / (double)(ks * ks);
    const double nbias = -(4503599627370496.0 * scale);

    // byte offsets of the thread's columns within an image row (guide: 3 bytes per pixel, src: SPX);
    // a row's bytes are then (wave-uniform row pointer) + (32-bit lane offset): no 64-bit vector
    // arithmetic per load
    const bool planar = SCN == 3 && src_planar != nullptr;
    const uint8_t *pimg = planar ? src_planar + (size_t)blockIdx.z * npx * 3 : nullptr;
    uint32_t gx3[kACols], gxs[kACols];
#pragma unroll
    for (int k = 0; k < kACols; k++) {
        const int gx = border_interpolate(xs - hl + tid * kACols + k, w, RF_BORDER_REFLECT);
        gx3[k] = (uint32_t)gx * 3u;
        gxs[k] = planar ? (uint32_t)gx : (uint32_t)gx * (uint32_t)SPX;
    }

    uint32_t V[kACols][NQ];
#pragma unroll
    for (int k = 0; k < kACols; k++)
#pragma unroll
        for (int q = 0; q < NQ; q++)
            V[k][q] = 0;

    // (Byte loads, 3 x (3 + SCN) per row step and thread, are not what limits this kernel: fetching
    // a thread's three pixels as one 12-byte load and picking the bytes apart was measured 10 %
    // SLOWER - the extra VALU work costs more than the vector-memory instructions it saves.)
    // One image row of the strip enters (or leaves) the vertical running sums.  The row's bytes are
    // read through a buffer descriptor over that row (wave-uniform base: scalar arithmetic only)
    // with the lane's constant 32-bit byte offset: no vector address arithmetic per load.
    auto add_row = [&](int yy, auto neg_c) __attribute__((always_inline)) {
        constexpr bool NEG = decltype(neg_c)::value;
        const int gy = border_interpolate(yy, h, RF_BORDER_REFLECT);
        const auto rg = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t *>(gimg + (size_t)gy * w * 3), 0, w * 3, 0x00020000);
        // (planar: one descriptor over the three planes' common row offset .. the last plane's row,
        //  channel sc at byte sc * npx + column; interleaved: the row's w * SPX bytes)
        const auto rp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t *>(planar ? pimg + (size_t)gy * w : simg + (size_t)gy * w * SPX), 0,
            planar ? (int)(2 * npx) + w : w * SPX, 0x00020000);
        const int pstep = planar ? (int)npx : 1;
#pragma unroll
        for (int k = 0; k < kACols; k++) {
            const uint32_t g01 = __builtin_amdgcn_raw_buffer_load_b16(rg, (int)gx3[k], 0, 0);
            const int g2 = __builtin_amdgcn_raw_buffer_load_b8(rg, (int)gx3[k] + 2, 0, 0);
            int p[SCN];
#pragma unroll
            for (int sc = 0; sc < SCN; sc++)
                p[sc] = __builtin_amdgcn_raw_buffer_load_b8(rp, (int)gxs[k], sc * pstep, 0);
            Q::template accumulate<NEG>((int)(g01 & 0xffu), (int)(g01 >> 8), g2, p, V[k]);
        }
    };
    using RowIn = std::integral_constant<bool, false>;
    using RowOut = std::integral_constant<bool, true>;

    for (int yy = ys - radius; yy < ys + radius; yy++)
        add_row(yy, RowIn{});
    if (tid == 0)
#pragma unroll
        for (int q = 0; q < NQ; q++)
            pfx[q][0] = 0;

    // wave_acc[q][j] accumulates, over all rows so far, the totals of the waves to the left of
    // wave j (uint32 wrap-around is exact): a wave adds its total to the entries of the waves to
    // its right with one LDS atomic per quantity, and every thread gets its row's base as the
    // difference between the entry now and one row ago - one broadcast read and one subtraction
    // per quantity instead of kAWaves - 1 reads, selects and adds
    uint32_t acc_prev[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++)
        acc_prev[q] = 0;
    if (tid < kAWaves)
#pragma unroll
        for (int q = 0; q < NQ; q++)
            wave_acc[q][tid] = 0;
    __syncthreads();

    // Per-pixel phase (window means -> algebra -> alpha/beta): the strip's columns are dealt to the
    // threads one by one, thread t taking columns pc[k] = t' + kAThreads * k (the prefix sums are in LDS:
    // any thread can serve any column).  A wave then stores 64 consecutive pixels (1 KB) per
    // instruction, its 16-lane rows are 16 consecutive columns, and with a halo of a whole wave on
    // either side (hl = 64, out_w = kACW - 128: 3840 = 6 x 640, 1920 = 3 x 640) two of the strip's
    // twelve wave-iterations have no output column at all and are skipped - which waves those are
    // alternates with the workgroup's parity (t' = t rotated by two waves), so that the four SIMDs
    // of a CU see the same load.
    const int tidp = (tid + ((blockIdx.x + blockIdx.y + blockIdx.z) & 1) * (kAThreads / 2)) & (kAThreads - 1);
    bool col_ok[kACols];
#pragma unroll
    for (int k = 0; k < kACols; k++) {
        const int c = tidp + kAThreads * k;
        col_ok[k] = c >= hl && c < hl + out_w && xs - hl + c < w;
    }
#ifdef RF_GF_S1_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
    unsigned long long st_n = 0;
#endif
    for (int y = ys; y < ye; y++) {
        // later passes: the guide records of the row's pixels, used after the barriers
        // (gs: [img][row][kGsFloats][w] - the floats of a record are w apart, so the lanes of a wave
        //  read and write runs of consecutive floats; interleaved 36-byte records made every load
        //  and store instruction of a wave touch 54 cache lines and cost more than they saved)
        float gsr[kACols][kGsFloats];
        if (MODE == kS1Reuse) {
            const float *rec = gsimg + (size_t)y * kGsFloats * w + (xs - hl + tidp);
#pragma unroll
            for (int k = 0; k < kACols; k++) {
                if (!col_ok[k])
                    continue;
#pragma unroll
                for (int e = 0; e < kGsFloats; e++)
                    gsr[k][e] = rec[(size_t)e * w + kAThreads * k];
            }
        }
        add_row(y + radius, RowIn{});
        RF_S1_STAMP(0);
        // inclusive prefix over the strip's columns, per quantity.  The six DPP steps of the wave
        // scan run stage by stage across the quantities: back to back on one quantity every step
        // waits out the VALU-write -> DPP-read hazard (an s_nop per step and quantity)
        uint32_t incl[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            uint32_t tsum = V[0][q];
#pragma unroll
            for (int k = 1; k < kACols; k++)
                tsum += V[k][q];
            incl[q] = tsum;
        }
#define RF_SCAN_STAGE(CTRL, ROWMASK, BOUND)                                                  \
    _Pragma("unroll") for (int q = 0; q < NQ; q++) incl[q] +=                                \
        (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl[q], CTRL, ROWMASK, 0xf, BOUND);
        RF_SCAN_STAGE(0x111, 0xf, true)   // row_shr:1
        RF_SCAN_STAGE(0x112, 0xf, true)   // row_shr:2
        RF_SCAN_STAGE(0x114, 0xf, true)   // row_shr:4
        RF_SCAN_STAGE(0x118, 0xf, true)   // row_shr:8
        RF_SCAN_STAGE(0x142, 0xa, false)  // row_bcast:15
#undef RF_SCAN_STAGE
        // last step (row_bcast:31 into rows 2 and 3) spelled out: hipcc does not fold this one into
        // the add and emits v_mov 0 / v_mov_dpp / v_add per quantity.  Each statement reads a
        // register written NQ instructions earlier, so the DPP read hazard is covered.
        asm volatile("s_nop 1");  // ... whatever hipcc put last before the first statement
#pragma unroll
        for (int q = 0; q < NQ; q++)
            asm volatile("v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                         : "+v"(incl[q]));
        RF_S1_STAMP(7);
        {
            // lanes 0 .. kAWaves-2-wave add the wave total (lane 63's prefix) to the entries of the
            // waves to the right
            // (all the totals first, then ONE masked region with the atomics: per quantity hipcc
            //  opens and closes the mask around each)
            const int dstw = wave + 1 + lane;
            uint32_t tot[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++)
                tot[q] = (uint32_t)__builtin_amdgcn_readlane((int)incl[q], 63);
            if (dstw < kAWaves) {
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    atomicAdd(&wave_acc[q][dstw], tot[q]);
            }
        }
        RF_S1_STAMP(1);
        __syncthreads();
        RF_S1_STAMP(2);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint32_t acc = wave_acc[q][wave];
            const uint32_t base = acc - acc_prev[q];
            acc_prev[q] = acc;
            uint32_t pk = incl[q] + base;  // inclusive prefix at the thread's last column
#pragma unroll
            for (int k = kACols - 1; k >= 0; k--) {
                pfx[q][tid * kACols + k + 1] = pk;
                pk -= V[k][q];
            }
        }
        RF_S1_STAMP(3);
        __syncthreads();
        RF_S1_STAMP(4);
        // exact rows: largest magnitude (its bit pattern: an infinity or a NaN then counts as the largest
        // exponent there is and fails the row) and smallest non-zero magnitude (as (bits << 1) - 2, which
        // sends a zero to the top) of the row's alpha planes jointly and of its beta plane, per src
        // channel, over this lane's columns
        uint32_t xmx_a[SCN], xmx_b[SCN];
        uint32_t xmn_a[SCN], xmn_b[SCN];
        if (EXACT) {
#pragma unroll
            for (int sc = 0; sc < SCN; sc++) {
                xmx_a[sc] = xmx_b[sc] = 0u;
                xmn_a[sc] = xmn_b[sc] = 0xffffffffu;
            }
        }
#pragma unroll
        for (int k = 0; k < kACols; k++) {
            const int c = tidp + kAThreads * k;
            const int x = xs - hl + c;
            if (!col_ok[k])
                continue;
            float m[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++)
                m[q] = mean_of(pfx[q][c + radius + 1] - pfx[q][c - radius], scale, nbias);
            float ab_px[4 * SCN];
            if (MODE == kS1Reuse) {
                gf_src_algebra<SCN>(gsr[k], m, ab_px);
            } else {
                float gsn[kGsFloats];
                gf_guide_algebra(m, eps_f, eps_small, gsn);
                gf_src_algebra<SCN>(gsn, m + 9, ab_px);
                if (MODE == kS1Keep) {
                    float *rec = gsimg + (size_t)y * kGsFloats * w + x;
#pragma unroll
                    for (int e = 0; e < kGsFloats; e++)
                        rec[(size_t)e * w] = gsn[e];
                }
            }
            // the four planes of a src channel (alpha_0..2, beta) interleaved per pixel: one
            // 16-byte store, and stage 2 reads 256-byte runs per image row instead of 64-byte ones
            // (through a descriptor over the image row: wave-uniform base, the lane's constant offset)
#pragma unroll
            for (int sc = 0; sc < SCN; sc++) {
                const auto rab = __builtin_amdgcn_make_buffer_rsrc(
                    abimg + ((size_t)sc * npx + (size_t)y * w) * 4, 0, w * 16, 0x00020000);
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t o = {__float_as_uint(ab_px[4 * sc]), __float_as_uint(ab_px[4 * sc + 1]),
                                   __float_as_uint(ab_px[4 * sc + 2]), __float_as_uint(ab_px[4 * sc + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(o, rab, x * 16, 0, 0);
            }
            if (EXACT) {
                // (col_ok is uniform over a 16-lane row here: hl, out_w and w are multiples of 16)
#pragma unroll
                for (int sc = 0; sc < SCN; sc++) {
                    const float a0 = ab_px[4 * sc], a1 = ab_px[4 * sc + 1], a2 = ab_px[4 * sc + 2],
                                bt = ab_px[4 * sc + 3];
                    xmx_a[sc] = max(max(xmx_a[sc], __float_as_uint(a0) & 0x7fffffffu),
                                    max(__float_as_uint(a1) & 0x7fffffffu, __float_as_uint(a2) & 0x7fffffffu));
                    xmx_b[sc] = max(xmx_b[sc], __float_as_uint(bt) & 0x7fffffffu);
                    xmn_a[sc] = min(min(xmn_a[sc], (__float_as_uint(a0) << 1) - 2u),
                                    min((__float_as_uint(a1) << 1) - 2u, (__float_as_uint(a2) << 1) - 2u));
                    xmn_b[sc] = min(xmn_b[sc], (__float_as_uint(bt) << 1) - 2u);
                    double d[4];
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        d[p] = (double)ab_px[4 * sc + p];
                        d[p] = swizzle_add<swz_xor(1)>(d[p]);
                        d[p] = swizzle_add<swz_xor(2)>(d[p]);
                        d[p] = swizzle_add<swz_xor(4)>(d[p]);
                        d[p] = swizzle_add<swz_xor(8)>(d[p]);
                    }
                    // lane j < 4 of the 16-lane row stores plane j's sum
                    const double d01 = (lane & 1) ? d[1] : d[0], d23 = (lane & 1) ? d[3] : d[2];
                    const double dsel = (lane & 2) ? d23 : d01;
                    if ((lane & 15) < 4) {
                        double *xr = xo.xf + (((size_t)blockIdx.z * ab_groups + sc) * h + y) * 4 * xo.nb;
                        xr[(lane & 3) * xo.nb + (x >> 4)] = dsel;
                    }
                }
            }
        }
        if (EXACT) {
            // the half-wave's statistics: five butterfly steps over packed 16-bit fields (maximum of
            // the largest, maximum of the inverted smallest), lanes 0 and 32 store
#pragma unroll
            for (int sc = 0; sc < SCN; sc++) {
                uint32_t wa = ((xmx_a[sc] >> 15) << 16) | (0xffffu - (xmn_a[sc] >> 16));
                uint32_t wb = ((xmx_b[sc] >> 15) << 16) | (0xffffu - (xmn_b[sc] >> 16));
                wa = swizzle_pkmax<swz_xor(1)>(wa);
                wb = swizzle_pkmax<swz_xor(1)>(wb);
                wa = swizzle_pkmax<swz_xor(2)>(wa);
                wb = swizzle_pkmax<swz_xor(2)>(wb);
                wa = swizzle_pkmax<swz_xor(4)>(wa);
                wb = swizzle_pkmax<swz_xor(4)>(wb);
                wa = swizzle_pkmax<swz_xor(8)>(wa);
                wb = swizzle_pkmax<swz_xor(8)>(wb);
                wa = swizzle_pkmax<swz_xor(16)>(wa);
                wb = swizzle_pkmax<swz_xor(16)>(wb);
                if ((lane & 31) == 0)
                    xo.xstat[(((size_t)blockIdx.z * ab_groups + sc) * h + y) * xo.slots + blockIdx.x * 8 +
                             wave * 2 + (lane >> 5)] = make_uint2(wa, wb);
            }
        }
        RF_S1_STAMP(5);
        add_row(y - radius, RowOut{});
        RF_S1_STAMP(6);
#ifdef RF_GF_S1_STAMP
        st_n++;
#endif
    }
#ifdef RF_GF_S1_STAMP
    if (lane == 0) {
        for (int i = 0; i < 8; i++)
            atomicAdd(&g_s1_stamps[i], st_acc[i]);
        atomicAdd(&g_s1_stamps[15], st_n);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// stage 2a: RowSum<float,double>.  One wave = 64 rows of one plane, one lane per row, walking
// the border-extended row left to right; tiles are transposed through LDS so that global
// loads/stores stay row-contiguous.
// ------------------------------------------------------------------------------------------
constexpr int kBChunk = 32;

// planes: [img][src_np][h][w] (il = 1) or [img][src_np / 4][h][w][4] (il = 4: groups of four
// planes interleaved per pixel, the layout stage 1 writes), of which the first np per image are
// summed; rowsums: [img][np][h][w]
template <int il>
__global__ __launch_bounds__(64) void gf_rowsum_kernel(const float *__restrict__ planes,
                                                       double *__restrict__ rowsums, int h, int w,
                                                       int radius, int row_blocks, int np,
                                                       const int *__restrict__ colour, int src_np)
{
    {
        // grey 3-channel images only carry the 4 planes of their first channel
        const int pl = blockIdx.x / row_blocks;
        if (colour != nullptr && pl % np >= 4 && colour[pl / np] == 0)
            return;
    }
    __shared__ float t_in[kBRows][kBChunk + 1];
    __shared__ float t_out_lo[kBRows][kBChunk + 1];  // leaving values
    __shared__ double t_d[kBRows][kBChunk + 1];

    const int lane = threadIdx.x;
    const int plane = blockIdx.x / row_blocks;  // plane index across the whole chunk of images
    const int row0 = (blockIdx.x - plane * row_blocks) * kBRows;
    const int q = plane % np;  // plane of the image
    const float *S = planes + ((size_t)(plane / np) * src_np + q / il * il) * h * w + q % il;
    double *D = rowsums + (size_t)plane * h * w;
    const int ks = 2 * radius + 1;
    const int sub = lane >> 5, col = lane & 31;  // loader role: 2 rows x 32 columns per instruction

    double s = 0.0;
    // prologue: s = sum_{i<ks} ext[i], ext[i] = S[bi(i - r)]
    for (int i0 = 0; i0 < ks; i0 += kBChunk) {
        const int xi = i0 + col;
        const int sx = border_interpolate(min(xi, ks - 1) - radius, w, RF_BORDER_REFLECT);
        for (int rr = 0; rr < kBRows; rr += 2) {
            const int row = min(row0 + rr + sub, h - 1);
            t_in[rr + sub][col] = S[((size_t)row * w + sx) * il];
        }
        __syncthreads();
        const int cnt = min(kBChunk, ks - i0);
        for (int c = 0; c < cnt; c++)
            s += (double)t_in[lane][c];
        __syncthreads();
    }
    // D[0] = s; then D[o] for o = 1..w-1:  s += (double)ext[o-1+ks] - (double)ext[o-1]
    // chunk over o in [1, w): entering S[bi(o + r)], leaving S[bi(o - 1 - r)]
    for (int o0 = 0; o0 < w; o0 += kBChunk) {
        const int o = o0 + col;
        const int se = border_interpolate(min(o, w - 1) + radius, w, RF_BORDER_REFLECT);
        const int sl = border_interpolate(min(o, w - 1) - 1 - radius, w, RF_BORDER_REFLECT);
        for (int rr = 0; rr < kBRows; rr += 2) {
            const int row = min(row0 + rr + sub, h - 1);
            t_in[rr + sub][col] = S[((size_t)row * w + se) * il];
            t_out_lo[rr + sub][col] = S[((size_t)row * w + sl) * il];
        }
        __syncthreads();
        const int cnt = min(kBChunk, w - o0);
        for (int c = 0; c < cnt; c++) {
            if (o0 + c > 0)
                s += (double)t_in[lane][c] - (double)t_out_lo[lane][c];
            t_d[lane][c] = s;
        }
        __syncthreads();
        if (o < w)
            for (int rr = 0; rr < kBRows; rr += 2) {
                const int row = row0 + rr + sub;
                if (row < h)
                    D[(size_t)row * w + o] = t_d[rr + sub][col];
            }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// stage 2b: ColumnSum<double,float> + ApplyTransform + convertTo(uint8).
// block = 64 columns x (4*SCN) planes; every thread walks its column of one plane from the
// top of the image; per output row the 4 means of a src channel meet in LDS and the beta
// thread of that channel forms q = beta + a0*I0 + a1*I1 + a2*I2 and stores the byte.
// ------------------------------------------------------------------------------------------
// T = uint8_t (result rounded and saturated) or float (the CV_32F variant: result stored as is)
template <int SCN, int SPX, typename T = uint8_t>
__global__ __launch_bounds__(64 * 4 * SCN) void gf_colsum_apply_kernel(
    const double *__restrict__ rowsums, const T *__restrict__ guide, T *__restrict__ dst, int h,
    int w, int radius, const int *__restrict__ colour)
{
    if (wrong_variant<SCN>(colour, blockIdx.z))
        return;
    constexpr int NP = 4 * SCN;
    constexpr int kDepth = 4;
    __shared__ float means[2][NP][64];

    const int lane = threadIdx.x;
    const int plane = threadIdx.y;
    const int x = blockIdx.x * 64 + lane;
    const int xc = min(x, w - 1);
    const size_t npx = (size_t)h * w;
    const double *R = rowsums + ((size_t)blockIdx.z * (4 * SPX) + plane) * npx + xc;
    const T *gimg = guide + (size_t)blockIdx.z * npx * 3;
    T *dimg = dst + (size_t)blockIdx.z * npx * SPX;
    const int ks = 2 * radius + 1;
    const double scale = 1.0 / (double)(ks * ks);

    double SUM = 0.0;
    for (int yy = -radius; yy < radius; yy++)
        SUM += R[(size_t)border_interpolate(yy, h, RF_BORDER_REFLECT) * w];

    for (int y0 = 0; y0 < h; y0 += kDepth) {
        double sp[kDepth], sm[kDepth];
#pragma unroll
        for (int k = 0; k < kDepth; k++) {
            const int y = min(y0 + k, h - 1);
            sp[k] = R[(size_t)border_interpolate(y + radius, h, RF_BORDER_REFLECT) * w];
            sm[k] = R[(size_t)border_interpolate(y - radius, h, RF_BORDER_REFLECT) * w];
        }
#pragma unroll
        for (int k = 0; k < kDepth; k++) {
            const int y = y0 + k;
            if (y >= h)
                break;
            const double s0 = SUM + sp[k];
            means[y & 1][plane][lane] = (float)(s0 * scale);
            SUM = s0 - sm[k];
            __syncthreads();
            if ((plane & 3) == 3 && x < w) {
                const int s = plane >> 2;
                const size_t pix = (size_t)y * w + x;
                float q = means[y & 1][plane][lane];
#pragma unroll
                for (int g = 0; g < 3; g++)
                    q = __fadd_rn(q, __fmul_rn(means[y & 1][s * 4 + g][lane],
                                               (float)gimg[pix * 3 + g]));
                T o;
                if constexpr (sizeof(T) == 1)
                    o = saturate_u8(q);
                else
                    o = q;
                if (SCN == SPX) {
                    dimg[pix * SCN + s] = o;
                } else {  // grey image: the one computed channel stands for all three
                    dimg[pix * 3 + 0] = o;
                    dimg[pix * 3 + 1] = o;
                    dimg[pix * 3 + 2] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// CV_32F variant.  Same operations as the uint8 path, but the stage-1 window
// sums are no longer exact integers, so every box filter is the order-faithful pair
// RowSum<float,double> / ColumnSum<double,float>: products -> row sums -> column sums -> means
// -> per-pixel algebra -> row sums -> column sums + apply.
// ------------------------------------------------------------------------------------------
// P: [img][NQ][h][w], quantity order of Quant<SCN>
template <int SCN>
__global__ __launch_bounds__(256) void gff_products_kernel(const float *__restrict__ guide,
                                                           const float *__restrict__ src,
                                                           float *__restrict__ P, size_t npx)
{
    constexpr int NQ = Quant<SCN>::NQ;
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npx)
        return;
    const float *g = guide + ((size_t)blockIdx.y * npx + pix) * 3;
    const float *p = src + ((size_t)blockIdx.y * npx + pix) * SCN;
    float *out = P + (size_t)blockIdx.y * NQ * npx + pix;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    out[0 * npx] = g0;
    out[1 * npx] = g1;
    out[2 * npx] = g2;
    out[3 * npx] = __fmul_rn(g0, g0);
    out[4 * npx] = __fmul_rn(g0, g1);
    out[5 * npx] = __fmul_rn(g0, g2);
    out[6 * npx] = __fmul_rn(g1, g1);
    out[7 * npx] = __fmul_rn(g1, g2);
    out[8 * npx] = __fmul_rn(g2, g2);
#pragma unroll
    for (int s = 0; s < SCN; s++) {
        const float ps = p[s];
        out[(size_t)(9 + s) * npx] = ps;
        out[(size_t)(9 + SCN + 3 * s + 0) * npx] = __fmul_rn(ps, g0);
        out[(size_t)(9 + SCN + 3 * s + 1) * npx] = __fmul_rn(ps, g1);
        out[(size_t)(9 + SCN + 3 * s + 2) * npx] = __fmul_rn(ps, g2);
    }
}

// ColumnSum<double,float>: rowsums [img][np][h][w] -> means [img][np][h][w]; one lane per column
// of one plane.  grid (ceil(w/64), np, images)
__global__ __launch_bounds__(64) void gff_colsum_mean_kernel(const double *__restrict__ rowsums,
                                                             float *__restrict__ means, int h,
                                                             int w, int radius, int np)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w)
        return;
    const size_t npx = (size_t)h * w;
    const size_t plane = (size_t)blockIdx.z * np + blockIdx.y;
    const double *R = rowsums + plane * npx + x;
    float *M = means + plane * npx + x;
    const int ks = 2 * radius + 1;
    const double scale = 1.0 / (double)(ks * ks);
    double SUM = 0.0;
    for (int yy = -radius; yy < radius; yy++)
        SUM += R[(size_t)border_interpolate(yy, h, RF_BORDER_REFLECT) * w];
    for (int y = 0; y < h; y++) {
        const double s0 = SUM + R[(size_t)border_interpolate(y + radius, h, RF_BORDER_REFLECT) * w];
        M[(size_t)y * w] = (float)(s0 * scale);
        SUM = s0 - R[(size_t)border_interpolate(y - radius, h, RF_BORDER_REFLECT) * w];
    }
}

// means [img][NQ][h][w] -> alpha/beta in place in the first 4*SCN planes of each image
template <int SCN>
__global__ __launch_bounds__(256) void gff_algebra_kernel(float *__restrict__ P, size_t npx,
                                                          float eps_f, int eps_small)
{
    constexpr int NQ = Quant<SCN>::NQ;
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npx)
        return;
    float *base = P + (size_t)blockIdx.y * NQ * npx + pix;
    float m[NQ], ab[4 * SCN];
#pragma unroll
    for (int q = 0; q < NQ; q++)
        m[q] = base[(size_t)q * npx];
    gf_pixel_algebra<SCN>(m, eps_f, eps_small, ab);
#pragma unroll
    for (int e = 0; e < 4 * SCN; e++)
        base[(size_t)e * npx] = ab[e];
}

template <int SCN>
int gf_f32_chunk(const float *guide, const float *src, float *dst, int m, int h, int w, int radius,
                 float eps_f, int eps_small, int iterations, float *P, double *rows,
                 hipStream_t stream)
{
    constexpr int NQ = Quant<SCN>::NQ;
    const size_t npx = (size_t)h * w;
    const int row_blocks = ceil_div(h, kBRows);
    const unsigned pb = (unsigned)((npx + 255) / 256);
    for (int it = 0; it < iterations; it++) {
        const float *s0 = it == 0 ? src : dst;
        hipLaunchKernelGGL(gff_products_kernel<SCN>, dim3(pb, m), dim3(256), 0, stream, guide, s0, P,
                           npx);
        hipLaunchKernelGGL(gf_rowsum_kernel<1>, dim3((unsigned)(m * NQ * row_blocks)), dim3(64), 0,
                           stream, P, rows, h, w, radius, row_blocks, NQ, (const int *)nullptr, NQ);
        hipLaunchKernelGGL(gff_colsum_mean_kernel, dim3(ceil_div(w, 64), NQ, m), dim3(64), 0, stream,
                           rows, P, h, w, radius, NQ);
        hipLaunchKernelGGL(gff_algebra_kernel<SCN>, dim3(pb, m), dim3(256), 0, stream, P, npx, eps_f,
                           eps_small);
        hipLaunchKernelGGL(gf_rowsum_kernel<1>, dim3((unsigned)(m * 4 * SCN * row_blocks)), dim3(64), 0,
                           stream, P, rows, h, w, radius, row_blocks, 4 * SCN,
                           (const int *)nullptr, NQ);
        hipLaunchKernelGGL((gf_colsum_apply_kernel<SCN, SCN, float>), dim3(ceil_div(w, 64), 1, m),
                           dim3(64, 4 * SCN), 0, stream, rows, guide, dst, h, w, radius,
                           (const int *)nullptr);
    }
    return RF_OK;
}

// uint8 <-> float images for radii beyond the 8-bit kernels' range (rf_gf_u8 -> float kernels)
__global__ __launch_bounds__(256) void gf_u8_to_f32_kernel(const uint8_t *__restrict__ in,
                                                           float *__restrict__ out, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}
__global__ __launch_bounds__(256) void gf_f32_to_u8_kernel(const float *__restrict__ in,
                                                           uint8_t *__restrict__ out, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        out[i] = saturate_u8(in[i]);  // convertTo(CV_8U): round half to even, clamp, NaN -> 0
}

}  // namespace

// Side streams for the second half of a batch, one per CALLER stream (so two callers never meet
// on one side stream, and a caller that captures its stream into a graph pulls only its own side
// stream into that capture).  The table is bounded: when it is full the least recently used entry
// whose side stream is idle is destroyed and replaced; if none is idle, or the caller's stream
// belongs to another device than the current one, the call runs on the caller's stream alone
// (nullptr).  An entry is held (busy) from gf_side_stream until the call that got it has enqueued
// everything (gf_side_release): an idle-looking stream is not taken from under a call that is
// about to use it.  rf_shutdown() destroys them all.
namespace {
struct SideStream {
    hipStream_t caller, side;
    int device;
    unsigned long long used;
    int busy;  // calls that hold this entry and may not have enqueued their work yet: never recycled
};
constexpr size_t kMaxSideStreams = 16;
std::mutex g_side_mu;
SideStream g_side[kMaxSideStreams];
size_t g_side_n = 0;
unsigned long long g_side_tick = 0;
}  // namespace

hipStream_t gf_side_stream(hipStream_t caller)
{
    int dev = 0, sdev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return nullptr;
    if (caller != nullptr) {
        if (hipStreamGetDevice(caller, &sdev) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (sdev != dev)
            return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (size_t i = 0; i < g_side_n; i++)
        if (g_side[i].caller == caller && g_side[i].device == dev) {
            g_side[i].used = ++g_side_tick;
            g_side[i].busy++;
            return g_side[i].side;
        }
    size_t slot = g_side_n;
    if (g_side_n == kMaxSideStreams) {
        slot = kMaxSideStreams;
        for (size_t i = 0; i < g_side_n; i++) {
            if (g_side[i].busy > 0 || (slot != kMaxSideStreams && g_side[i].used > g_side[slot].used))
                continue;
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(g_side[i].side, &cap) != hipSuccess ||
                cap != hipStreamCaptureStatusNone || hipStreamQuery(g_side[i].side) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
            slot = i;
        }
        if (slot == kMaxSideStreams)
            return nullptr;
        (void)hipStreamDestroy(g_side[slot].side);
    }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        if (slot != g_side_n) {  // the recycled entry is gone: close the gap
            g_side[slot] = g_side[g_side_n - 1];
            g_side_n--;
        }
        return nullptr;
    }
    g_side[slot] = {caller, st, dev, ++g_side_tick, 1};
    if (slot == g_side_n)
        g_side_n++;
    return st;
}

void gf_side_release(hipStream_t side)
{
    if (side == nullptr)
        return;
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (size_t i = 0; i < g_side_n; i++)
        if (g_side[i].side == side && g_side[i].busy > 0) {
            g_side[i].busy--;
            return;
        }
}

void gf_shutdown()
{
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (size_t i = 0; i < g_side_n; i++)
        (void)hipStreamDestroy(g_side[i].side);
    g_side_n = 0;
}

// The per-radius launchers live in the rf_gf_fused_<part>.o units (Makefile: GF_PARTS = 8).
GfFusedLaunch gf_fused_part_0(int), gf_fused_part_1(int), gf_fused_part_2(int), gf_fused_part_3(int),
    gf_fused_part_4(int), gf_fused_part_5(int), gf_fused_part_6(int), gf_fused_part_7(int);
// ... and the radii above kGfFusedSmallMax in rf_gf_fused_L<part>.o (GF_LARGE_PARTS = 4)
GfFusedLaunch gf_fused_large_0(int), gf_fused_large_1(int), gf_fused_large_2(int), gf_fused_large_3(int);

GfFusedLaunch gf_fused_launcher(int radius)
{
    typedef GfFusedLaunch (*Part)(int);
    static const Part parts[8] = {gf_fused_part_0, gf_fused_part_1, gf_fused_part_2, gf_fused_part_3,
                                  gf_fused_part_4, gf_fused_part_5, gf_fused_part_6, gf_fused_part_7};
    if (radius < 1 || radius > kGfFusedMaxRadius)
        return nullptr;
    if (radius > kGfFusedSmallMax) {
        static const Part large[4] = {gf_fused_large_0, gf_fused_large_1, gf_fused_large_2, gf_fused_large_3};
        return large[(radius - kGfFusedSmallMax - 1) % 4](radius);
    }
    return parts[(radius - 1) % 8](radius);
}

size_t gf_workspace_cap()
{
    static size_t cap = 0;  // the answer cannot change within a process; a benign race at worst
    if (cap == 0) {
        size_t c = (size_t)6 << 30;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            c = std::min(std::max(c, (size_t)prop.totalGlobalMem / 8), (size_t)32 << 30);
        else
            (void)hipGetLastError();  // no device: not an error of this call
        cap = c;
    }
    return cap;
}

// per-image "has colour" flags at the head of the workspace
size_t gf_header_bytes(int n) { return (((size_t)n * sizeof(int)) + 255) & ~(size_t)255; }

}  // namespace rf

namespace rf {
// Scratch per image in flight, by stage-2 form (np = 4 x src channels float planes alpha/beta):
//   two-kernel   alpha/beta + a double row sum per pixel and plane
//   row walk     alpha/beta + a double state per 16 pixels and plane
//   chained      alpha/beta + the hand-off words (16 B per lane, 64 lanes per sub-tile and block)
//                + the head sums (a double per row and plane) + the part's sync words
size_t gf_per_img_two_kernel(size_t npx, int np) { return npx * np * (sizeof(float) + sizeof(double)); }
size_t gf_per_img_row_walk(size_t npx, int np, int nb, int h)
{
    return (size_t)np * (npx * sizeof(float) + (size_t)nb * h * sizeof(double));
}
// exact rows: per image and plane group the statistics of stage 1 (8 B per half-wave and row; a strip
// has at least 256 output columns for every radius the 8-bit kernels take), a flag byte and a list
// entry per row, the list's length
size_t gf_exact_extra(int np, int h, int w)
{
    const size_t slots = (size_t)ceil_div(w, 256) * 8;
    return (((size_t)(np / 4) * ((size_t)h * (slots * 8 + 8) + 16)) + 255) & ~(size_t)255;
}
constexpr size_t kGfSyncBytes = 256;
// radii above this run the float kernels (uint32 window sums: (2r+1)^2 * 255^2 < 2^32)
constexpr int kGfMaxRadiusU8 = 128;  // (2 * 128 + 1)^2 * 255^2 = 4,294,836,225 < 2^32: the last radius that fits
// ... on float copies of guide, src and result + the float kernels' own planes and row sums
constexpr size_t kGfF32Slack = 16;  // once per workspace: alignment of the float kernels' scratch
size_t gf_per_img_via_f32(size_t npx, int src_cn)
{
    return npx * ((3 + 2 * (size_t)src_cn) * sizeof(float) +
                  (9 + 4 * (size_t)src_cn) * (sizeof(float) + sizeof(double)));
}
size_t gf_chain_xst_bytes(int src_cn, int nb, int h, int radius)
{
    return (size_t)src_cn * nb * gf_chain_nsub(h, radius) * 64 * 16;
}
size_t gf_per_img_chained(size_t npx, int np, int nb, int h, int radius)
{
    return kGfSyncBytes + gf_chain_xst_bytes(np / 4, nb, h, radius) + (size_t)np * h * sizeof(double) +
           npx * np * sizeof(float);
}
}  // namespace rf

#ifdef RF_GF_S1_STAMP
extern "C" int rf_debug_gf_s1_stamps(unsigned long long *out16)
{
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rf::g_s1_stamps), sizeof(zero)) != hipSuccess)
        return -4;
    if (hipMemcpyToSymbol(HIP_SYMBOL(rf::g_s1_stamps), zero, sizeof(zero)) != hipSuccess)
        return -4;
    return 0;
}
#endif

extern "C" size_t rf_gf_workspace_bytes(int n, int h, int w, int guide_cn, int src_cn, int radius)
{
    (void)guide_cn;
    if (n <= 0 || h <= 0 || w <= 0 || (src_cn != 1 && src_cn != 3))
        return 0;
    // the larger of the forms a call may take (small images: the chained form's fixed part)
    size_t per_img = rf::gf_per_img_two_kernel((size_t)h * w, 4 * src_cn);
    if (radius > rf::kGfMaxRadiusU8)  // float kernels on float copies of the images (rf_gf_u8)
        per_img = rf::gf_per_img_via_f32((size_t)h * w, src_cn);
    if (radius >= 1 && radius <= rf::kGfFusedMaxRadius) {
        per_img = std::max(per_img, rf::gf_per_img_chained((size_t)h * w, 4 * src_cn,
                                                           rf::ceil_div(w, rf::kSB), h, radius));
        per_img = std::max(per_img, rf::gf_per_img_row_walk((size_t)h * w, 4 * src_cn,
                                                            rf::ceil_div(w, rf::kSB), h) +
                                        rf::gf_exact_extra(4 * src_cn, h, w) +
                                        (((size_t)h * w + 15) & ~(size_t)15) +
                                        (src_cn == 3 ? ((3 * (size_t)h * w + 15) & ~(size_t)15) : 0));
    }
    // enough images in flight to fill the chip and to make the tails of the launches small: capped
    // at 1/8 of the device's memory, at most 32 GiB (6 GiB when no device can be asked).  C5 shard
    // (128 x 4K, 3 passes): round 2 91.5 ms with 6 GiB (13 images per chunk), 86.6 ms with 16 GiB (37);
    // round 3 70.7 / 68.6 / 70.1 ms with 16 / 32 / 64 GiB (chunks of 37 / 74 / all 128 images).
    size_t imgs = (size_t)n;
    const size_t cap = rf::gf_workspace_cap();
    if (imgs * per_img > cap)
        imgs = cap / per_img;
    if (imgs < 1)
        imgs = 1;
    return rf::gf_header_bytes(n) + imgs * per_img + (radius > rf::kGfMaxRadiusU8 ? rf::kGfF32Slack : 0);
}

extern "C" int rf_gf_f32(const float *guide, const float *src, float *dst, int n, int h, int w,
                         int guide_cn, int src_cn, int radius, double eps, int iterations,
                         void *workspace, size_t workspace_bytes, void *stream_);

extern "C" int rf_gf_u8(const uint8_t *guide, const uint8_t *src, uint8_t *dst, int n, int h,
                        int w, int guide_cn, int src_cn, int radius, double eps, int iterations,
                        void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace rf;
    if (n == 0)  // an empty batch is valid whatever the (possibly NULL) pointers are
        return RF_OK;
    if (!guide || !src || !dst || !workspace)
        return fail(RF_E_BADARG, "rf_gf_u8: NULL pointer");
    if (n < 0 || h <= 0 || w <= 0 || iterations < 1)
        return fail(RF_E_BADARG, "rf_gf_u8: bad size n=%d h=%d w=%d iterations=%d", n, h, w,
                    iterations);
    if (guide_cn != 3)
        return fail(RF_E_UNSUPPORTED, "rf_gf_u8: guide must have 3 channels (got %d)", guide_cn);
    if (src_cn != 1 && src_cn != 3)
        return fail(RF_E_UNSUPPORTED, "rf_gf_u8: src channels must be 1 or 3 (got %d)", src_cn);
    if (radius < 0 || radius > 4096)
        return fail(RF_E_UNSUPPORTED, "rf_gf_u8: radius %d outside 0..4096", radius);
    if (w >= (1 << 27))  // the kernels address a row's alpha/beta (16 B per pixel) with 32-bit offsets
        return fail(RF_E_UNSUPPORTED, "rf_gf_u8: width %d beyond 2^27 - 1", w);
    {
        const size_t px = (size_t)n * h * w;
        if (ranges_overlap(dst, px * src_cn, guide, px * 3))
            return fail(RF_E_BADARG, "rf_gf_u8: dst must not overlap guide");
        if (dst != src && ranges_overlap(dst, px * src_cn, src, px * src_cn))
            return fail(RF_E_BADARG, "rf_gf_u8: dst may equal src but not partially overlap it");
    }
    if (n == 0)
        return RF_OK;
    const int np = 4 * src_cn;
    const size_t npx = (size_t)h * w;
    const int nb = ceil_div(w, kSB);
    const size_t header = gf_header_bytes(n);
    hipStream_t stream = (hipStream_t)stream_;
    if (radius > kGfMaxRadiusU8) {
        // Beyond the 8-bit kernels' range (int(sigma_spatial) is a free parameter of the reference's
        // tool, /root/reference/filter_reflectance.py:67-70,118): the float kernels of rf_gf_f32 on
        // float copies of the images, every pass rounded to uint8 like convertTo(CV_8U).  On 8-bit
        // data the float path's double window sums are the same exact integers the 8-bit stage 1
        // forms, so the bytes are what the 8-bit kernels would give (tested on either side of 128).
        const size_t per_img_f = gf_per_img_via_f32(npx, src_cn);
        if (workspace_bytes < header + per_img_f + kGfF32Slack)
            return fail(RF_E_WORKSPACE, "rf_gf_u8: workspace %zu B < %zu B needed for one image at "
                        "radius %d", workspace_bytes, header + per_img_f + kGfF32Slack, radius);
        const size_t f32_ws = npx * (9 + 4 * (size_t)src_cn) * (sizeof(float) + sizeof(double));
        // (kGfF32Slack: the float kernels' scratch starts with double row sums, so the float copies
        //  in front of it are rounded up to 16 bytes - an odd pixel count would leave it 4-aligned)
        int chunk = (int)std::min<size_t>((size_t)n, (workspace_bytes - header - kGfF32Slack) / per_img_f);
        char *ws0 = static_cast<char *>(workspace) + header;
        for (int i0 = 0; i0 < n; i0 += chunk) {
            const int m = std::min(chunk, n - i0);
            float *gF = reinterpret_cast<float *>(ws0);
            float *sF = gF + (size_t)m * npx * 3;
            float *dF = sF + (size_t)m * npx * src_cn;
            const size_t copies = ((size_t)m * npx * (3 + 2 * (size_t)src_cn) * sizeof(float) + 15) & ~(size_t)15;
            void *fw = ws0 + copies;
            const size_t cg = (size_t)m * npx * 3, cs = (size_t)m * npx * src_cn;
            const unsigned bg = (unsigned)std::min<size_t>((cg + 255) / 256, 65535);
            const unsigned bs = (unsigned)std::min<size_t>((cs + 255) / 256, 65535);
            hipLaunchKernelGGL(gf_u8_to_f32_kernel, dim3(bg), dim3(256), 0, stream,
                               guide + (size_t)i0 * npx * 3, gF, cg);
            uint8_t *d0 = dst + (size_t)i0 * npx * src_cn;
            for (int it = 0; it < iterations; it++) {
                const uint8_t *s0 = it == 0 ? src + (size_t)i0 * npx * src_cn : d0;
                hipLaunchKernelGGL(gf_u8_to_f32_kernel, dim3(bs), dim3(256), 0, stream, s0, sF, cs);
                const int rc = rf_gf_f32(gF, sF, dF, m, h, w, 3, src_cn, radius, eps, 1, fw,
                                         (size_t)m * f32_ws, stream_);
                if (rc != RF_OK)
                    return rc;
                hipLaunchKernelGGL(gf_f32_to_u8_kernel, dim3(bs), dim3(256), 0, stream, dF, d0, cs);
            }
        }
        RF_HIP_CHECK(hipGetLastError());
        return RF_OK;
    }
    // Stage 2 (box means of alpha/beta), three forms with identical bytes:
    //   row walk + column walk (default for the instantiated radii 1..96)
    //   chained column walk (debug option "gf_chained", radius 45 and 52 only: no row-walk kernel, every block takes its row
    //       sums from its left neighbour; identical bytes, measured SLOWER - the stagger between
    //       neighbouring blocks costs the L2 sharing of their operand lines, profiles/r04_gf_chained.md)
    //   row-sum / column-sum kernel pair (radius 0; debug option "gf_two_kernel")
    // (the fused kernels index planes with 32-bit element offsets: images below 2^28 pixels)
    const GfFusedLaunch fused_launch = gf_fused_launcher(radius);
    const bool can_fuse = !debug_get(kDbgGfTwoKernel) && fused_launch != nullptr &&
                          npx < ((size_t)1 << 28);
    const size_t per_img_chained = can_fuse ? gf_per_img_chained(npx, np, nb, h, radius) : 0;
    const size_t per_img_rw = gf_per_img_row_walk(npx, np, nb, h);
    const size_t per_img = gf_per_img_two_kernel(npx, np);
    const bool chained = can_fuse && debug_get(kDbgGfChained) && gf_chained_radius(radius) &&
                         workspace_bytes >= header + per_img_chained;
    const bool fused = chained || (can_fuse && workspace_bytes >= header + per_img_rw);
    const size_t per_img_fused = chained ? per_img_chained : per_img_rw;
    if (!fused && workspace_bytes < header + per_img)
        return fail(RF_E_WORKSPACE, "rf_gf_u8: workspace %zu B < %zu B needed for one image",
                    workspace_bytes, header + per_img);
    // EXPERIMENT, off by default (debug option "gf_guide_cache"): iterated calls keep the guide half
    // of the per-pixel algebra (kGsFloats floats per pixel) from the first pass for the later ones,
    // which then box-sum only the 4 src quantities - 3x fewer VALU instructions in stage 1, but 36
    // more bytes per pixel and pass through a memory system the other kernels of the pass already
    // load.  Measured twice (interleaved 36-byte records; row-planar records with coalesced
    // accesses), 8 x 4K grey, kernels alone: stage 1 of a later pass 1.085 ms against 1.044 ms
    // without the record, the pass that writes it 1.69 ms; C5 shard 11.4 against 14.0 GP/s
    // (profiles/r03_gf_guide_cache.md).  Kept as a switch so that the measurement can be repeated.
    const size_t gs_bytes = (npx * kGsFloats * sizeof(float) + 15) & ~(size_t)15;  // parts stay 16-byte aligned
    const bool keep_gs = iterations > 1 && debug_get(kDbgGfGuideCache) &&
                         workspace_bytes - header >= (fused ? per_img_fused : per_img) + gs_bytes;
    // 3-channel sources: an image whose channels are equal (the reference filters the CNN's grey map,
    // /root/reference/README.md:66) is handled as ONE byte per pixel in the workspace: the grey
    // probe, which reads every src byte anyway, leaves channel 0 there for the first pass's stage 1
    // (its one-byte-per-pixel instantiation fetches 4 instead of 6 bytes per pixel and row), and the
    // passes of an iterated call hand their result on the same way - the column walk writes a third
    // of the bytes - until the last pass writes dst.  Bytes identical (debug option "gf_no_compact"
    // keeps the three-channel reads and hand-offs).
    const size_t cmp_want = (npx + 15) & ~(size_t)15;
    const size_t cmp_bytes =
        (fused && src_cn == 3 && !debug_get(kDbgGfNoCompact) &&
         workspace_bytes - header >= per_img_fused + (keep_gs ? gs_bytes : 0) + cmp_want)
            ? cmp_want
            : 0;
    // Colour images leave every pass as three PLANES in the workspace (round 6): a channel's column
    // walk then stores 4 bytes per lane instead of four single bytes into the interleaved dst (its
    // vector-memory instructions are what it is short of), stage 1 of the next pass reads bytes either
    // way, and a small kernel interleaves the last pass's planes into dst (6 B/px at copy speed).
    const size_t cmp3_want = (3 * npx + 15) & ~(size_t)15;
    const size_t cmp3_bytes =
        (cmp_bytes && !keep_gs &&
         workspace_bytes - header >= per_img_fused + cmp_bytes + cmp3_want)
            ? cmp3_want
            : 0;
    // Exact rows (rf_gf_fused.hpp): rows whose alpha/beta pass the exactness test need no row walk -
    // stage 1 leaves block sums and per-row statistics, the rows that fail are listed and walked, the
    // column walk starts its chains from the block sums.  Needs a width that is a multiple of 16 and
    // room for the statistics.  OFF unless the debug option "gf_exact" is set: bit-identical, measured
    // slower than the row walk (profiles/r06_gf_exact.md); compiled for the radii of gf_exact_radius.
    const size_t exact_bytes = gf_exact_extra(np, h, w);
    const bool exact =
        fused && !chained && !keep_gs && gf_exact_radius(radius) && w % 16 == 0 && h <= kGfExactMaxH &&
        debug_get(kDbgGfExact) &&
        !debug_get(kDbgGfS1LegacyStrips) &&
        workspace_bytes - header >= per_img_fused + cmp_bytes + cmp3_bytes + exact_bytes;
    const size_t per_img_used = (fused ? per_img_fused : per_img) + (keep_gs ? gs_bytes : 0) +
                                cmp_bytes + cmp3_bytes + (exact ? exact_bytes : 0);
    int chunk = (int)std::min<size_t>((size_t)n, (workspace_bytes - header) / per_img_used);
    if (chunk > 16383)
        chunk = 16383;
    // chained column walk: a part's images are dealt to 8 ticket queues (one per XCD), so parts of
    // a multiple of 8 images - chunks of a multiple of 16 - load the queues evenly
    if (chained && chunk >= 16 && n > chunk)
        chunk &= ~15;
    const float eps_f = (float)eps;
    const int eps_small = eps < 1e-2;
    // stage-1 strips: stage1_threads x stage1_cols columns, at least r of them halo on either side.
    // The left halo and the output width are multiples of 16 (a 16-lane row of the per-pixel phase
    // is then one aligned 16-column block of the image); a halo of a whole wave on either side is
    // taken where it costs no strip (two of the strip's wave-iterations then have nothing to do, see
    // the kernel).  Debug option "gf_s1_legacy_strips": halo r on either side (rounds 1-5).
    struct StripGeom {
        int hl, out_w, strips;
    };
    auto strip_geometry = [&](int scn) {
        const int acw = stage1_threads(scn) * stage1_cols(scn);
        StripGeom g;
        if (debug_get(kDbgGfS1LegacyStrips)) {
            g.hl = radius;
            g.out_w = acw - 2 * radius;
        } else {
            g.hl = (radius + 15) & ~15;
            g.out_w = (acw - g.hl - radius) & ~15;
            if (radius <= 64 && ceil_div(w, acw - 128) <= ceil_div(w, g.out_w)) {
                g.hl = 64;
                g.out_w = acw - 128;
            }
        }
        g.strips = ceil_div(w, g.out_w);
        return g;
    };
    const StripGeom geo3 = strip_geometry(3), geo1 = strip_geometry(1);
    const int strips3 = geo3.strips, strips1 = geo1.strips;
    const int xslots = std::max(strips1, strips3) * 8;  // exact rows: statistic slots per row
    const int mask_words = ceil_div(h, 32);
    const GfStateLayout lay = exact ? GfStateLayout{nb, 1, 4 * nb} : GfStateLayout{nb * h, h, 1};

    // 3-channel sources: find the images whose channels are identical (see the file header)
    int *colour_all = nullptr;
    if (src_cn == 3) {
        colour_all = reinterpret_cast<int *>(workspace);
        RF_HIP_CHECK(hipMemsetAsync(colour_all, 0, sizeof(int) * (size_t)n, stream));
    }
    const int probe_blocks = (int)std::min<size_t>(1024, (npx / 4 + 255) / 256 + 1);

    // One part = m images starting at i0, their scratch at ws, every launch on st.
    // One part = m images starting at i0, their scratch at ws, every launch on st.  A part is
    // enqueued in steps - probe, then per pass stage 1 and stage 2 - so that the schedules below can
    // interleave the steps of several parts.
    struct Part {
        int i0, m;
        char *ws;
        hipStream_t st;
        // derived (part_setup)
        const int *colour;
        double *rows;
        float *ab, *gs;
        uint8_t *cmp, *cmp3;
        uint2 *xstat;            // exact rows: statistics, list and its lengths, flag bitmask
        unsigned *rowmask;
        int *xlist, *xcount;
        GfChain xc;
        const uint8_t *g0;
        uint8_t *d0;
        int seg_rows1, seg_rows3;
    };
    // stage-1 occupancy cap (debug option "gf_s1_cap" = workgroups per CU, 0 = whatever fits): a
    // dynamic-LDS pad makes one more workgroup than the cap exceed the CU's 160 KB, which leaves
    // registers and LDS on every CU for the walk kernels of the other part (see the schedule below)
    const int s1_cap = debug_get(kDbgGfS1Cap);
    auto s1_pad = [&](size_t static_lds) -> unsigned {
        if (s1_cap < 1 || s1_cap > 3)
            return 0;
        const size_t want = (size_t)163840 / (s1_cap + 1) + 1536;
        return want > static_lds ? (unsigned)(want - static_lds) : 0u;
    };
    const unsigned pad1 = s1_pad(sizeof(uint32_t) * (13 * (stage1_threads(1) * stage1_cols(1) + 1) + 13 * 4));
    const unsigned pad3 = s1_pad(sizeof(uint32_t) * (21 * (stage1_threads(3) * stage1_cols(3) + 1) + 21 * 4));
    auto part_setup = [&](Part &P, int m_fill) {
        const int i0 = P.i0, m = P.m;
        char *ws = P.ws;
        P.colour = colour_all ? colour_all + i0 : nullptr;
        // two-kernel form: [row sums (double)][alpha/beta]; row walk: [states (double)][alpha/beta];
        // chained: [sync words][hand-off words][head sums][alpha/beta]
        P.rows = reinterpret_cast<double *>(ws);
        P.ab = reinterpret_cast<float *>(P.rows + (fused ? (size_t)m * np * nb * h
                                                          : (size_t)m * np * npx));
        P.xc = GfChain{nullptr, nullptr, nullptr};
        if (chained) {
            P.xc.sync = reinterpret_cast<unsigned *>(ws);
            P.xc.xst = reinterpret_cast<unsigned long long *>(ws + kGfSyncBytes);
            double *head = reinterpret_cast<double *>(
                ws + kGfSyncBytes + (size_t)m * gf_chain_xst_bytes(src_cn, nb, h, radius));
            P.xc.head = head;
            P.ab = reinterpret_cast<float *>(head + (size_t)m * np * h);
            P.rows = nullptr;
        }
        P.gs = keep_gs ? P.ab + (size_t)m * np * npx : nullptr;  // [m][h][kGsFloats][w]
        // grey 3-channel images of an iterated call: the passes hand their result on as one byte per
        // pixel (see cmp_bytes above)
        P.cmp = cmp_bytes ? reinterpret_cast<uint8_t *>(P.ab + (size_t)m * np * npx) +
                                (keep_gs ? (size_t)m * gs_bytes : 0)
                          : nullptr;
        P.cmp3 = cmp3_bytes ? P.cmp + (size_t)m * cmp_bytes : nullptr;  // [m][3][npx]
        P.xstat = nullptr;
        P.rowmask = nullptr;
        P.xlist = P.xcount = nullptr;
        if (exact) {
            char *xb = reinterpret_cast<char *>(P.ab + (size_t)m * np * npx) +
                       (keep_gs ? (size_t)m * gs_bytes : 0) + (size_t)m * (cmp_bytes + cmp3_bytes);
            const size_t rows_all = (size_t)m * src_cn * h;
            P.xstat = reinterpret_cast<uint2 *>(xb);
            P.xlist = reinterpret_cast<int *>(P.xstat + rows_all * xslots);
            P.xcount = P.xlist + rows_all;  // [m x groups] lengths, then [m x groups][mask_words] flags
            P.rowmask = reinterpret_cast<unsigned *>(P.xcount + (size_t)m * src_cn);
        }
        P.g0 = guide + (size_t)i0 * npx * 3;
        P.d0 = dst + (size_t)i0 * npx * src_cn;
        // Rows per stage-1 segment (m_fill: the images whose stage 1 is in flight on the device
        // together - both halves of a chunk in the aligned schedule, the part alone otherwise).  A
        // workgroup walks its 2r warm-up rows (about a fifth of a main row each: sums only) and then its
        // segment; more, shorter segments spread small work over the chip, fewer and longer ones waste
        // less on warm-up rows and re-read fewer bytes.  A small cost model decides: k workgroups
        // resident on a CU take crowd[k] times one workgroup alone (one wave per SIMD issues every ~6
        // cycles, four saturate the pipe - stage 1 alone at 1 / 2 / 3 / 4 workgroups per CU,
        // profiles/r05_c5_overlap.md: k / 1, 1.38, 1.70, 1.87); a launch is whole rounds of resident
        // workgroups plus the rest; the segment count with the least modelled time wins.  Against the
        // round-4 rule (>= 960 workgroups, never below 3/4 of a window: debug option "gf_s1_min_wgs" =
        // 960), ms per call: 1 x 256x256 0.10 / 0.28, 1 x IIW 0.11 / 0.31, 1 x 1080p 0.26 / 0.44,
        // 1 x 4K 0.48 / 0.58, 2 x 4K 0.67 / 0.74, 16 x IIW 0.25 / 0.42, 4 x 4K colour 2.12 / 2.24,
        // 32 x 4K x 3 passes 16.6 / 17.5, 64 x 1080p x 3 8.7 / 9.2; equal at 4 x 4K, 256 x IIW and the
        // C5 shard, which keeps the three segments of 720 rows measured best in round 4
        // (profiles/r05_gf_seg_sweep.json; the measurements behind the earlier rules: HISTORY.md).
        // The 3-channel kernel (21 running sums, 3 workgroups per CU) is capped at 6 windows per segment.
        auto pick_seg = [&](int strips_k, long long min_wgs, int cap, int per_cu) {
            int seg;
            if (debug_get(kDbgGfS1MinWgs) > 0) {  // the round-4 rule, kept for A/B runs
                min_wgs = debug_get(kDbgGfS1MinWgs);
                const long long per_seg = std::max<long long>(1, (long long)strips_k * m_fill);
                long long k = (min_wgs + per_seg - 1) / per_seg;          // segments per image
                k = std::max<long long>(k, ceil_div(h, cap));
                k = std::min<long long>(std::max<long long>(k, 1), h);
                seg = ceil_div(h, (int)k);
                seg = std::max(seg, std::min(h, std::max(3 * (2 * radius + 1) / 4, 32)));
            } else {
                const double warm = 0.2 * 2.0 * radius;
                // time of k resident workgroups on one CU, in units of one workgroup alone
                static const double crowd[5] = {0.0, 1.0, 2.0 / 1.38, 3.0 / 1.70, 4.0 / 1.87};
                const long long slots = 256LL * per_cu;
                double best = 0.0;
                seg = h;
                // (k runs over the segment counts that change the segment length: O(sqrt(h)) of them)
                for (int k = std::max(1, ceil_div(h, cap)); k <= h;) {
                    const int sg = ceil_div(h, k);
                    const int k_next = sg > 1 ? (h - 1) / (sg - 1) + 1 : h + 1;
                    // full rounds of resident workgroups, then the rest on ceil(rest / 256) per CU: a
                    // launch a little over a whole round pays a whole workgroup's length for the rest
                    // (4 colour images at 4K: 800 workgroups on 768 places 2.47 ms, 640 on them 2.26)
                    // (two halves on two streams: the other half's kernels fill the tail of a launch,
                    //  so beyond one round the workgroups are priced as a fluid - at the C5 shard that
                    //  keeps round 4's three segments of 720 rows, 63.4 against 64.4 ms with 540)
                    const long long wgs = (long long)strips_k * m_fill * ceil_div(h, sg);
                    const long long full = wgs / slots, rest = wgs % slots;
                    const double t =
                        (m_fill > P.m && wgs > slots)
                            ? (warm + sg) * (double)wgs / (double)slots * crowd[per_cu]
                            : (warm + sg) * ((double)full * crowd[per_cu] +
                                             (rest ? crowd[(int)((rest + 255) / 256)] : 0.0));
                    if (best == 0.0 || t < best * 0.999) {
                        best = t;
                        seg = sg;
                    }
                    k = std::max(k + 1, k_next);
                }
            }
            seg = ceil_div(h, ceil_div(h, seg));  // equal segments: a launch ends with its longest one
            if (debug_get(kDbgGfSegRows) > 0)
                seg = std::min(h, debug_get(kDbgGfSegRows));
            return seg;
        };
        P.seg_rows1 = pick_seg(strips1, 960, h, 4);
        P.seg_rows3 = pick_seg(strips3, 1024, std::max(6 * (2 * radius + 1), 512), 3);
    };
    // the part's grey probe (flags zeroed by the caller's stream before the fork); with the
    // one-byte hand-off it also leaves every image's channel 0 in cmp for the first pass
    auto part_probe = [&](const Part &P) {
        if (colour_all != nullptr)
            hipLaunchKernelGGL(gf_grey_probe_kernel, dim3(probe_blocks, P.m), dim3(256), 0, P.st,
                               src + (size_t)P.i0 * npx * 3, colour_all + P.i0, npx, P.cmp);
    };
    auto part_stage1 = [&](const Part &P, int it) {
        hipStream_t st = P.st;
        const int m = P.m, seg_rows1 = P.seg_rows1, seg_rows3 = P.seg_rows3;
        const int *colour = P.colour;
        float *ab = P.ab, *gs = P.gs;
        const uint8_t *g0 = P.g0, *cmp = P.cmp;
        const uint8_t *s0 = (it == 0 ? src : (const uint8_t *)dst) + (size_t)P.i0 * npx * src_cn;
        const dim3 ga3(strips3, ceil_div(h, seg_rows3), m), ga1(strips1, ceil_div(h, seg_rows1), m);
        const GfExactOut xo = {P.rows, P.xstat, nb, xslots};
        // (the occupancy cap's dynamic-LDS pad can take a workgroup beyond the 64 KB a launch may use
        //  without asking: ask)
#define RF_GF_S1_ATTR(K, PAD)                                                                      \
    do {                                                                                           \
        if ((PAD) > 0)                                                                             \
            (void)hipFuncSetAttribute((const void *)(K), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)(PAD));                                                 \
    } while (0)
#define RF_GF_STAGE1(MODE, EX)                                                                     \
    do {                                                                                           \
        RF_GF_S1_ATTR((gf_stage1_kernel<3, 3, MODE, EX>), pad3);                                   \
        RF_GF_S1_ATTR((gf_stage1_kernel<1, 1, MODE, EX>), pad1);                                   \
        RF_GF_S1_ATTR((gf_stage1_kernel<1, 3, MODE, EX>), pad1);                                   \
        if (src_cn == 3) {                                                                         \
            hipLaunchKernelGGL((gf_stage1_kernel<3, 3, MODE, EX>), ga3, dim3(stage1_threads(3)), pad3, \
                               st, g0, s0, ab, h, w, radius, eps_f, eps_small, seg_rows3, colour,  \
                               gs, 3, geo3.hl, geo3.out_w, xo, it > 0 ? P.cmp3 : nullptr);         \
            if (cmp != nullptr)                                                                    \
                hipLaunchKernelGGL((gf_stage1_kernel<1, 1, MODE, EX>), ga1, dim3(stage1_threads(1)), \
                                   pad1, st, g0, cmp, ab, h, w, radius, eps_f, eps_small,          \
                                   seg_rows1, colour, gs, 3, geo1.hl, geo1.out_w, xo, nullptr);    \
            else                                                                                   \
                hipLaunchKernelGGL((gf_stage1_kernel<1, 3, MODE, EX>), ga1, dim3(stage1_threads(1)), \
                                   pad1, st, g0, s0, ab, h, w, radius, eps_f, eps_small,           \
                                   seg_rows1, colour, gs, 3, geo1.hl, geo1.out_w, xo, nullptr);    \
        } else {                                                                                   \
            hipLaunchKernelGGL((gf_stage1_kernel<1, 1, MODE, EX>), ga1, dim3(stage1_threads(1)), pad1, \
                               st, g0, s0, ab, h, w, radius, eps_f, eps_small, seg_rows1, colour,  \
                               gs, 1, geo1.hl, geo1.out_w, xo, nullptr);                           \
        }                                                                                          \
    } while (0)
        if (debug_get(kDbgGfExpSkip) & 1)
            ;  // timing experiment: no stage 1 (results wrong)
        else if (exact)
            RF_GF_STAGE1(kS1Full, true);
        else if (!keep_gs)
            RF_GF_STAGE1(kS1Full, false);
        else if (it == 0)
            RF_GF_STAGE1(kS1Keep, false);
        else
            RF_GF_STAGE1(kS1Reuse, false);
#undef RF_GF_STAGE1
#undef RF_GF_S1_ATTR
    };
    auto part_stage2 = [&](const Part &P, int it) {
        hipStream_t st = P.st;
        const int m = P.m;
        if (fused) {
            GfExact xr = {nullptr, nullptr, nullptr, 0, 0};
            if (exact) {
                // the rows that fail the exactness test: flagged for the column walk, listed for the row walk
                (void)hipMemsetAsync(P.xcount, 0, sizeof(int) * (size_t)m * src_cn * (1 + mask_words), st);
                hipLaunchKernelGGL(gf_exact_rows_kernel<0>, dim3((unsigned)ceil_div(h, 256), (unsigned)(m * src_cn)),
                                   dim3(256), 0, st, P.xstat, h, xslots, strips1 * 8, strips3 * 8, src_cn,
                                   debug_get(kDbgGfExactAllFlagged) ? -1 : gf_exact_limit(radius),
                                   P.colour, P.rowmask, mask_words, P.xlist, P.xcount);
                xr = GfExact{P.rowmask, P.xlist, P.xcount, mask_words, 1};
            }
            const GfFusedArgs fa = {P.ab, P.rows, P.g0, P.d0, m, h, w, nb, src_cn, P.colour, st, P.xc,
                                    debug_get(kDbgGfExpSkip),
                                    it + 1 < iterations ? P.cmp : nullptr, lay, xr, P.cmp3,
                                    // colour images (which leave the walk as planes): an XCD walks its (pair,
                                    // channel) items in runs of 64 pairs per channel - a channel's
                                    // neighbouring blocks then stay neighbours in time and find each other's
                                    // operands in the L2 (81.6 against 82.8 ms per colour chain).  Without
                                    // the planes (no room in the workspace, "gf_no_compact") the walk
                                    // stores single bytes into the interleaved dst, where the three
                                    // channels of a block want to run side by side (channel fastest: 0).
                                    // Debug option "gf_cw_chan_run": n + 1 forces runs of n (1: channel fastest)
                                    P.cmp3 != nullptr
                                        ? (debug_get(kDbgGfCwChanRun) ? debug_get(kDbgGfCwChanRun) - 1 : 64)
                                        : 0};
            fused_launch(fa);
            if (P.cmp3 != nullptr && it + 1 == iterations && !(debug_get(kDbgGfExpSkip) & 4))
                hipLaunchKernelGGL(gf_interleave3_kernel, dim3(probe_blocks, m), dim3(256), 0, st, P.cmp3,
                                   P.d0, npx, P.colour);
            return;
        }
        const int row_blocks = ceil_div(h, kBRows);
        hipLaunchKernelGGL(gf_rowsum_kernel<4>, dim3((unsigned)(m * np * row_blocks)), dim3(64), 0,
                           st, P.ab, P.rows, h, w, radius, row_blocks, np, P.colour, np);
        dim3 gc(ceil_div(w, 64), 1, m);
        if (src_cn == 3) {
            hipLaunchKernelGGL((gf_colsum_apply_kernel<3, 3>), gc, dim3(64, 12), 0, st, P.rows,
                               P.g0, P.d0, h, w, radius, P.colour);
            hipLaunchKernelGGL((gf_colsum_apply_kernel<1, 3>), gc, dim3(64, 4), 0, st, P.rows,
                               P.g0, P.d0, h, w, radius, P.colour);
        } else {
            hipLaunchKernelGGL((gf_colsum_apply_kernel<1, 1>), gc, dim3(64, 4), 0, st, P.rows,
                               P.g0, P.d0, h, w, radius, P.colour);
        }
    };
    auto run_part = [&](int i0, int m, int m_fill, char *ws, hipStream_t st) {
        Part P{};
        P.i0 = i0, P.m = m, P.ws = ws, P.st = st;
        part_setup(P, m_fill);
        part_probe(P);
        for (int it = 0; it < iterations; it++) {
            part_stage1(P, it);
            part_stage2(P, it);
        }
    };

    // A chunk of eight or more images runs as parts on two streams - the caller's and a side
    // stream of the library, forked and joined with events, so the call still looks stream-ordered
    // to the caller and can be captured into a graph.  The kernels of a pass are bound by different
    // things (stage 1: the issue rate of its 4-cycle VALU instructions; row states: the latency of
    // a chunk of loads; column walk: memory bandwidth).
    //   aligned schedule (debug option "gf_stagger" = 0): two halves, each running its passes on its
    //       own stream from the same instant: both are in stage 1 together, then both in the walks -
    //       the second stream fills launch tails, nothing else.
    //   staggered schedule ("gf_stagger" = 1): the stage-1 launches of the parts are chained by
    //       events in the order (pass, part), parts alternating between the streams, so that at any
    //       time one part is in its VALU-bound stage 1 and the other stream's part in its
    //       memory-bound walks.  Events only: stream-ordered for the caller, capturable.
    // The debug option "gf_one_stream" keeps everything on the caller's stream,
    // "gf_force_two_streams" forks from two images on (cross-checks; identical bytes).
    char *ws0 = static_cast<char *>(workspace) + header;
    hipStream_t side = nullptr;
    // (tools/gf_stream_sweep.py, 3 passes at 4K, two streams over one: grey 1.10 / 1.18 / 1.00 /
    //  0.93 / 0.92 / 0.87 / 0.91 and colour 0.99 / 0.97 / 0.91 / 0.93 / 0.91 / 0.89 / 0.89 at 2 / 4 /
    //  8 / 12 / 16 / 24 / 36 images: small grey launches leave XCDs idle when halved)
    const int fork_from = debug_get(kDbgGfForceTwoStreams) ? 2 : 8;
    if (fused && chunk >= fork_from && !debug_get(kDbgGfOneStream))
        side = gf_side_stream(stream);
    struct SideHold {  // the entry stays ours until every launch of this call is enqueued
        hipStream_t s;
        ~SideHold() { gf_side_release(s); }
    } side_hold{side};
    struct Events {  // destroyed on every path out (a pending event is released on completion)
        std::vector<hipEvent_t> ev;
        hipEvent_t make()
        {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
                return nullptr;
            ev.push_back(e);
            return e;
        }
        ~Events()
        {
            for (hipEvent_t e : ev)
                (void)hipEventDestroy(e);
        }
    };
    const int stagger = debug_get(kDbgGfStagger);
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        if (side == nullptr || m < fork_from) {
            run_part(i0, m, m, ws0, stream);
            continue;
        }
        Events evs;
        hipEvent_t fork = evs.make(), join = evs.make();
        if (!fork || !join)
            return fail(RF_E_HIP, "rf_gf_u8: hipEventCreate failed");
        RF_HIP_CHECK(hipEventRecord(fork, stream));
        RF_HIP_CHECK(hipStreamWaitEvent(side, fork, 0));
        if (!stagger) {
            const int ma = (m + 1) / 2;
            run_part(i0, ma, m, ws0, stream);
            run_part(i0 + ma, m - ma, m, ws0 + (size_t)ma * per_img_used, side);
        } else {
            // parts alternate between the two streams; stage 1 of (pass, part) waits for stage 1 of
            // its predecessor in that order, which runs on the other stream
            int nparts = debug_get(kDbgGfParts) > 0 ? debug_get(kDbgGfParts) : 2;
            nparts = std::max(2, std::min(std::min(nparts, 16), m)) & ~1;
            std::vector<Part> parts((size_t)nparts);
            int at = 0;
            for (int p = 0; p < nparts; p++) {
                Part &P = parts[(size_t)p];
                P = Part{};
                P.i0 = i0 + at;
                P.m = m / nparts + (p < m % nparts ? 1 : 0);
                P.ws = ws0 + (size_t)at * per_img_used;
                P.st = (p & 1) ? side : stream;
                at += P.m;
                part_setup(P, P.m);
                part_probe(P);
            }
            hipEvent_t prev = nullptr;
            for (int it = 0; it < iterations; it++)
                for (int p = 0; p < nparts; p++) {
                    const Part &P = parts[(size_t)p];
                    if (prev != nullptr)
                        RF_HIP_CHECK(hipStreamWaitEvent(P.st, prev, 0));
                    part_stage1(P, it);
                    prev = evs.make();
                    if (!prev)
                        return fail(RF_E_HIP, "rf_gf_u8: hipEventCreate failed");
                    RF_HIP_CHECK(hipEventRecord(prev, P.st));
                    part_stage2(P, it);
                }
        }
        RF_HIP_CHECK(hipEventRecord(join, side));
        RF_HIP_CHECK(hipStreamWaitEvent(stream, join, 0));
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

extern "C" size_t rf_gf_f32_workspace_bytes(int n, int h, int w, int guide_cn, int src_cn,
                                            int radius)
{
    (void)guide_cn;
    (void)radius;
    if (n <= 0 || h <= 0 || w <= 0 || (src_cn != 1 && src_cn != 3))
        return 0;
    const size_t per_img = (size_t)h * w * (9 + 4 * src_cn) * (sizeof(float) + sizeof(double));
    size_t imgs = (size_t)n;
    const size_t cap = (size_t)16 << 30;
    if (imgs * per_img > cap)
        imgs = cap / per_img;
    if (imgs < 1)
        imgs = 1;
    return imgs * per_img;
}

extern "C" int rf_gf_f32(const float *guide, const float *src, float *dst, int n, int h, int w,
                         int guide_cn, int src_cn, int radius, double eps, int iterations,
                         void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace rf;
    if (n == 0)
        return RF_OK;
    if (!guide || !src || !dst || !workspace)
        return fail(RF_E_BADARG, "rf_gf_f32: NULL pointer");
    if (n < 0 || h <= 0 || w <= 0 || iterations < 1)
        return fail(RF_E_BADARG, "rf_gf_f32: bad size n=%d h=%d w=%d iterations=%d", n, h, w,
                    iterations);
    if (guide_cn != 3)
        return fail(RF_E_UNSUPPORTED, "rf_gf_f32: guide must have 3 channels (got %d)", guide_cn);
    if (src_cn != 1 && src_cn != 3)
        return fail(RF_E_UNSUPPORTED, "rf_gf_f32: src channels must be 1 or 3 (got %d)", src_cn);
    if (radius < 0 || radius > 4096)
        return fail(RF_E_UNSUPPORTED, "rf_gf_f32: radius %d outside 0..4096", radius);
    {
        const size_t px = (size_t)n * h * w * sizeof(float);
        if (ranges_overlap(dst, px * src_cn, guide, px * 3))
            return fail(RF_E_BADARG, "rf_gf_f32: dst must not overlap guide");
        if (dst != src && ranges_overlap(dst, px * src_cn, src, px * src_cn))
            return fail(RF_E_BADARG, "rf_gf_f32: dst may equal src but not partially overlap it");
    }
    const int nq = 9 + 4 * src_cn;
    const size_t npx = (size_t)h * w;
    const size_t per_img = npx * nq * (sizeof(float) + sizeof(double));
    if (workspace_bytes < per_img)
        return fail(RF_E_WORKSPACE, "rf_gf_f32: workspace %zu B < %zu B needed for one image",
                    workspace_bytes, per_img);
    hipStream_t stream = (hipStream_t)stream_;
    int chunk = (int)std::min<size_t>((size_t)n, workspace_bytes / per_img);
    if (chunk > 65535)
        chunk = 65535;
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        double *rows = reinterpret_cast<double *>(workspace);
        float *P = reinterpret_cast<float *>(rows + (size_t)m * nq * npx);
        const float *g0 = guide + (size_t)i0 * npx * 3;
        const float *s0 = src + (size_t)i0 * npx * src_cn;
        float *d0 = dst + (size_t)i0 * npx * src_cn;
        const int rc = src_cn == 3 ? gf_f32_chunk<3>(g0, s0, d0, m, h, w, radius, (float)eps,
                                                     eps < 1e-2, iterations, P, rows, stream)
                                   : gf_f32_chunk<1>(g0, s0, d0, m, h, w, radius, (float)eps,
                                                     eps < 1e-2, iterations, P, rows, stream);
        if (rc != RF_OK)
            return rc;
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
