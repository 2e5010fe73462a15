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
//            GPU as parallelism over rows x planes x images.  -> gf_rowsum_kernel, gf_colsum_apply_kernel
//
// Grey sources: the reference filters the CNN's grey `-r.png`, which imread turns into three
// identical channels.  The src channels never mix, so identical channels give identical
// outputs; gf_grey_probe_kernel marks such images (a device-side flag, no host round trip) and
// they run the one-channel instantiation with the result byte written three times (1/3 of the
// per-channel planes).  Every stage is launched in both instantiations; workgroups of the one
// that does not apply to their image exit at once.
#include "rf_common.hpp"

namespace rf {
namespace {

// ------------------------------------------------------------------------------------------
// stage 1 + per-pixel algebra
// ------------------------------------------------------------------------------------------
constexpr int kAThreads = 256;
constexpr int kACols = 2;                      // columns per thread
constexpr int kACW = kAThreads * kACols;       // strip width incl. halo
constexpr int kAWaves = kAThreads / 64;

template <int SCN>
struct Quant {
    static constexpr int NQ = 9 + 4 * SCN;
    // q: 0..2 I_g | 3..8 I_aI_b (00 01 02 11 12 22) | 9.. p_s | then p_s*I_g (s major)
    __device__ static inline void eval(const uint8_t *g, const uint8_t *p, uint32_t *v)
    {
        const uint32_t g0 = g[0], g1 = g[1], g2 = g[2];
        v[0] = g0;
        v[1] = g1;
        v[2] = g2;
        v[3] = g0 * g0;
        v[4] = g0 * g1;
        v[5] = g0 * g2;
        v[6] = g1 * g1;
        v[7] = g1 * g2;
        v[8] = g2 * g2;
#pragma unroll
        for (int s = 0; s < SCN; s++) {
            const uint32_t ps = p[s];
            v[9 + s] = ps;
            v[9 + SCN + 3 * s + 0] = ps * g0;
            v[9 + SCN + 3 * s + 1] = ps * g1;
            v[9 + SCN + 3 * s + 2] = ps * g2;
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

__device__ inline float mean_of(uint32_t s, double scale) { return (float)((double)s * scale); }

// index into the 6-entry symmetric store: (0,0)=0 (0,1)=1 (0,2)=2 (1,1)=3 (1,2)=4 (2,2)=5
__device__ constexpr int sym(int i, int j)
{
    return i <= j ? (i * 3 - i * (i - 1) / 2 + (j - i)) : (j * 3 - j * (j - 1) / 2 + (i - j));
}

// Per-pixel algebra of guided_filter.cpp from the 9 + 4*SCN window means m (order of Quant):
// covariance of the guide (+eps on the diagonal), its inverse by cofactors, and for every src
// channel s the coefficients alpha_{s,g} (out[4s + g]) and beta_s (out[4s + 3]).  Every
// operation is the separately rounded float op of the corresponding OpenCV helper.
template <int SCN>
__device__ inline void gf_pixel_algebra(const float *m, float eps_f, int eps_small, float *out)
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
#pragma unroll
    for (int e = 0; e < 6; e++)
        inv[e] = __fdiv_rn(inv[e], det);
#pragma unroll
    for (int s = 0; s < SCN; s++) {
        const float mp = m[9 + s];
        float cp[3];
#pragma unroll
        for (int g = 0; g < 3; g++)
            cp[g] = __fsub_rn(m[9 + SCN + 3 * s + g], __fmul_rn(mp, mI[g]));
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

// colour[img] != 0  <=>  some pixel of the 3-channel image has unequal channels.
// grid: (blocks per image, images); colour[] zeroed beforehand.
__global__ __launch_bounds__(256) void gf_grey_probe_kernel(const uint8_t *__restrict__ src,
                                                            int *__restrict__ colour, size_t npx)
{
    const uint8_t *simg = src + (size_t)blockIdx.y * npx * 3;
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
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(npx - nquads * 4)) {
        const uint8_t *p = simg + (nquads * 4 + threadIdx.x) * 3;
        diff |= p[0] != p[1] || p[1] != p[2];
    }
    if (diff)
        colour[blockIdx.y] = 1;
}

// Does this workgroup's instantiation apply to image img?  (colour == nullptr: no choice to make)
template <int SCN>
__device__ inline bool wrong_variant(const int *__restrict__ colour, int img)
{
    return colour != nullptr && (colour[img] != 0) != (SCN == 3);
}

// grid: (strips, row segments, images).  ab: [img][SPX*4][h][w] float (g<3 alpha, g=3 beta).
// SCN = src channels computed, SPX = src bytes per pixel (SCN, or 3 with SCN = 1 for a grey
// 3-channel image whose first channel stands for all three).
template <int SCN, int SPX>
__global__ __launch_bounds__(kAThreads) void gf_stage1_kernel(
    const uint8_t *__restrict__ guide, const uint8_t *__restrict__ src, float *__restrict__ ab,
    int h, int w, int radius, float eps_f, int eps_small, int seg_rows,
    const int *__restrict__ colour)
{
    if (wrong_variant<SCN>(colour, blockIdx.z))
        return;
    constexpr int NQ = Quant<SCN>::NQ;
    __shared__ uint32_t pfx[NQ][kACW + 1];
    __shared__ uint32_t wave_tot[NQ][kAWaves];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int out_w = kACW - 2 * radius;
    const int xs = blockIdx.x * out_w;
    const int ys = blockIdx.y * seg_rows;
    const int ye = min(ys + seg_rows, h);
    const size_t npx = (size_t)h * w;
    const uint8_t *gimg = guide + (size_t)blockIdx.z * npx * 3;
    const uint8_t *simg = src + (size_t)blockIdx.z * npx * SPX;
    float *abimg = ab + (size_t)blockIdx.z * npx * (SPX * 4);
    const int ks = 2 * radius + 1;
    const double scale = 1.0 / (double)(ks * ks);

    int gx[kACols];
#pragma unroll
    for (int k = 0; k < kACols; k++)
        gx[k] = border_interpolate(xs - radius + tid * kACols + k, w, RF_BORDER_REFLECT);

    uint32_t V[kACols][NQ];
#pragma unroll
    for (int k = 0; k < kACols; k++)
#pragma unroll
        for (int q = 0; q < NQ; q++)
            V[k][q] = 0;

    auto add_row = [&](int yy, bool add) {
        const int gy = border_interpolate(yy, h, RF_BORDER_REFLECT);
#pragma unroll
        for (int k = 0; k < kACols; k++) {
            uint32_t v[NQ];
            const size_t pix = (size_t)gy * w + gx[k];
            Quant<SCN>::eval(gimg + pix * 3, simg + pix * SPX, v);
#pragma unroll
            for (int q = 0; q < NQ; q++)
                V[k][q] = add ? V[k][q] + v[q] : V[k][q] - v[q];
        }
    };

    for (int yy = ys - radius; yy < ys + radius; yy++)
        add_row(yy, true);
    if (tid == 0)
#pragma unroll
        for (int q = 0; q < NQ; q++)
            pfx[q][0] = 0;

    for (int y = ys; y < ye; y++) {
        add_row(y + radius, true);
        // inclusive prefix over the strip's columns, per quantity
        uint32_t incl[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const uint32_t s = wave_inclusive_scan(V[0][q] + V[1][q]);
            incl[q] = s;
            if (lane == 63)
                wave_tot[q][wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            uint32_t base = 0;
            for (int wv = 0; wv < wave; wv++)
                base += wave_tot[q][wv];
            const uint32_t p1 = incl[q] + base;
            pfx[q][tid * kACols + 2] = p1;
            pfx[q][tid * kACols + 1] = p1 - V[1][q];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kACols; k++) {
            const int c = tid * kACols + k;
            const int x = xs - radius + c;
            if (c < radius || c >= kACW - radius || x >= w)
                continue;
            float m[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++)
                m[q] = mean_of(pfx[q][c + radius + 1] - pfx[q][c - radius], scale);
            const size_t pix = (size_t)y * w + x;
            float ab_px[4 * SCN];
            gf_pixel_algebra<SCN>(m, eps_f, eps_small, ab_px);
#pragma unroll
            for (int e = 0; e < 4 * SCN; e++)
                abimg[(size_t)e * npx + pix] = ab_px[e];
        }
        add_row(y - radius, false);
    }
}

// ------------------------------------------------------------------------------------------
// stage 2a: RowSum<float,double>.  One wave = 64 rows of one plane, one lane per row, walking
// the border-extended row left to right; tiles are transposed through LDS so that global
// loads/stores stay row-contiguous.
// ------------------------------------------------------------------------------------------
constexpr int kBRows = 64;
constexpr int kBChunk = 32;

// planes: [img][src_np][h][w] of which the first np per image are summed; rowsums: [img][np][h][w]
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
    const float *S = planes + ((size_t)(plane / np) * src_np + plane % np) * h * w;
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
            t_in[rr + sub][col] = S[(size_t)row * w + sx];
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
            t_in[rr + sub][col] = S[(size_t)row * w + se];
            t_out_lo[rr + sub][col] = S[(size_t)row * w + sl];
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
        hipLaunchKernelGGL(gf_rowsum_kernel, dim3((unsigned)(m * NQ * row_blocks)), dim3(64), 0,
                           stream, P, rows, h, w, radius, row_blocks, NQ, (const int *)nullptr, NQ);
        hipLaunchKernelGGL(gff_colsum_mean_kernel, dim3(ceil_div(w, 64), NQ, m), dim3(64), 0, stream,
                           rows, P, h, w, radius, NQ);
        hipLaunchKernelGGL(gff_algebra_kernel<SCN>, dim3(pb, m), dim3(256), 0, stream, P, npx, eps_f,
                           eps_small);
        hipLaunchKernelGGL(gf_rowsum_kernel, dim3((unsigned)(m * 4 * SCN * row_blocks)), dim3(64), 0,
                           stream, P, rows, h, w, radius, row_blocks, 4 * SCN,
                           (const int *)nullptr, NQ);
        hipLaunchKernelGGL((gf_colsum_apply_kernel<SCN, SCN, float>), dim3(ceil_div(w, 64), 1, m),
                           dim3(64, 4 * SCN), 0, stream, rows, guide, dst, h, w, radius,
                           (const int *)nullptr);
    }
    return RF_OK;
}

}  // namespace

// per-image "has colour" flags at the head of the workspace
size_t gf_header_bytes(int n) { return (((size_t)n * sizeof(int)) + 255) & ~(size_t)255; }

}  // namespace rf

extern "C" size_t rf_gf_workspace_bytes(int n, int h, int w, int guide_cn, int src_cn, int radius)
{
    (void)guide_cn;
    (void)radius;
    if (n <= 0 || h <= 0 || w <= 0 || (src_cn != 1 && src_cn != 3))
        return 0;
    const size_t per_img = (size_t)h * w * (4 * src_cn) * (sizeof(float) + sizeof(double));
    // enough images in flight to fill the chip, capped at 16 GiB of scratch
    size_t imgs = (size_t)n;
    const size_t cap = (size_t)16 << 30;
    if (imgs * per_img > cap)
        imgs = cap / per_img;
    if (imgs < 1)
        imgs = 1;
    return rf::gf_header_bytes(n) + imgs * per_img;
}

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
    // uint32 window sums: (2r+1)^2 * 255^2 must stay below 2^32; strip width must hold the halo
    if (radius < 0 || radius > 120)
        return fail(RF_E_UNSUPPORTED, "rf_gf_u8: radius %d outside 0..120", radius);
    if (dst == guide)
        return fail(RF_E_BADARG, "rf_gf_u8: dst must not alias guide");
    if (n == 0)
        return RF_OK;
    const int np = 4 * src_cn;
    const size_t npx = (size_t)h * w;
    const size_t per_img = npx * np * (sizeof(float) + sizeof(double));
    const size_t header = gf_header_bytes(n);
    if (workspace_bytes < header + per_img)
        return fail(RF_E_WORKSPACE, "rf_gf_u8: workspace %zu B < %zu B needed for one image",
                    workspace_bytes, header + per_img);
    hipStream_t stream = (hipStream_t)stream_;
    int chunk = (int)std::min<size_t>((size_t)n, (workspace_bytes - header) / per_img);
    if (chunk > 65535)
        chunk = 65535;
    const float eps_f = (float)eps;
    const int eps_small = eps < 1e-2;
    const int out_w = kACW - 2 * radius;
    const int strips = ceil_div(w, out_w);

    // 3-channel sources: find the images whose channels are identical (see the file header)
    int *colour_all = nullptr;
    if (src_cn == 3) {
        colour_all = reinterpret_cast<int *>(workspace);
        RF_HIP_CHECK(hipMemsetAsync(colour_all, 0, sizeof(int) * (size_t)n, stream));
        const int pb = (int)std::min<size_t>(1024, (npx / 4 + 255) / 256 + 1);
        for (int i0 = 0; i0 < n; i0 += 65535) {
            const int m = std::min(65535, n - i0);
            hipLaunchKernelGGL(gf_grey_probe_kernel, dim3(pb, m), dim3(256), 0, stream,
                               src + (size_t)i0 * npx * 3, colour_all + i0, npx);
        }
    }

    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        const int *colour = colour_all ? colour_all + i0 : nullptr;
        double *rows = reinterpret_cast<double *>(static_cast<char *>(workspace) + header);
        float *ab = reinterpret_cast<float *>(rows + (size_t)m * np * npx);
        const uint8_t *g0 = guide + (size_t)i0 * npx * 3;
        uint8_t *d0 = dst + (size_t)i0 * npx * src_cn;
        // row segments: enough workgroups to fill 256 CUs, but segments no shorter than 2r+1
        int seg_rows = h;
        while ((long long)strips * ceil_div(h, seg_rows) * m < 1024 && seg_rows > 2 * (2 * radius + 1) &&
               seg_rows > 32)
            seg_rows = (seg_rows + 1) / 2;
        const int segs = ceil_div(h, seg_rows);
        for (int it = 0; it < iterations; it++) {
            const uint8_t *s0 = (it == 0 ? src : (const uint8_t *)dst) + (size_t)i0 * npx * src_cn;
            dim3 ga(strips, segs, m);
            if (src_cn == 3) {
                hipLaunchKernelGGL((gf_stage1_kernel<3, 3>), ga, dim3(kAThreads), 0, stream, g0, s0,
                                   ab, h, w, radius, eps_f, eps_small, seg_rows, colour);
                hipLaunchKernelGGL((gf_stage1_kernel<1, 3>), ga, dim3(kAThreads), 0, stream, g0, s0,
                                   ab, h, w, radius, eps_f, eps_small, seg_rows, colour);
            } else {
                hipLaunchKernelGGL((gf_stage1_kernel<1, 1>), ga, dim3(kAThreads), 0, stream, g0, s0,
                                   ab, h, w, radius, eps_f, eps_small, seg_rows, colour);
            }
            const int row_blocks = ceil_div(h, kBRows);
            hipLaunchKernelGGL(gf_rowsum_kernel, dim3((unsigned)(m * np * row_blocks)), dim3(64), 0,
                               stream, ab, rows, h, w, radius, row_blocks, np, colour, np);
            dim3 gc(ceil_div(w, 64), 1, m);
            if (src_cn == 3) {
                hipLaunchKernelGGL((gf_colsum_apply_kernel<3, 3>), gc, dim3(64, 12), 0, stream, rows,
                                   g0, d0, h, w, radius, colour);
                hipLaunchKernelGGL((gf_colsum_apply_kernel<1, 3>), gc, dim3(64, 4), 0, stream, rows,
                                   g0, d0, h, w, radius, colour);
            } else {
                hipLaunchKernelGGL((gf_colsum_apply_kernel<1, 1>), gc, dim3(64, 4), 0, stream, rows,
                                   g0, d0, h, w, radius, colour);
            }
        }
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
    if (dst == guide)
        return fail(RF_E_BADARG, "rf_gf_f32: dst must not alias guide");
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
