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
//            radius 45 / 52 (the reference's parameter sets): gf_rowstate_kernel +
//            gf_colwalk_kernel - the double row sums never reach HBM (see the comment block
//            above them); any other radius <= 120: gf_rowsum_kernel + gf_colsum_apply_kernel,
//            which write every row sum (8 B per pixel and plane) and read it twice.
//
// Grey sources: the reference filters the CNN's grey `-r.png`, which imread turns into three
// identical channels.  The src channels never mix, so identical channels give identical
// outputs; gf_grey_probe_kernel marks such images (a device-side flag, no host round trip) and
// they run the one-channel instantiation with the result byte written three times (1/3 of the
// per-channel planes).  Every stage is launched in both instantiations; workgroups of the one
// that does not apply to their image exit at once.
#include "rf_common.hpp"

#include <algorithm>
#include <mutex>

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

// grid: (strips, row segments, images).  ab: [img][SPX][h][w][4] float (g<3 alpha, g=3 beta).
// SCN = src channels computed, SPX = src bytes per pixel (SCN, or 3 with SCN = 1 for a grey
// 3-channel image whose first channel stands for all three).
template <int SCN, int SPX>
__global__ __launch_bounds__(stage1_threads(SCN)) void gf_stage1_kernel(
    const uint8_t *__restrict__ guide, const uint8_t *__restrict__ src, float *__restrict__ ab,
    int h, int w, int radius, float eps_f, int eps_small, int seg_rows,
    const int *__restrict__ colour)
{
    if (wrong_variant<SCN>(colour, blockIdx.z))
        return;
    constexpr int NQ = Quant<SCN>::NQ;
    constexpr int kAThreads = stage1_threads(SCN), kAWaves = kAThreads / 64;
    constexpr int kACols = stage1_cols(SCN);
    constexpr int kACW = kAThreads * kACols;
    __shared__ uint32_t pfx[NQ][kACW + 1];
    __shared__ uint32_t wave_acc[NQ][kAWaves];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    for (int y = ys; y < ye; y++) {
        add_row(y + radius, true);
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
        {
            // lanes 0 .. kAWaves-2-wave add the wave total (lane 63's prefix) to the entries of the
            // waves to the right
            const int dstw = wave + 1 + lane;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl[q], 63);
                if (dstw < kAWaves)
                    atomicAdd(&wave_acc[q][dstw], tot);
            }
        }
        __syncthreads();
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
            // the four planes of a src channel (alpha_0..2, beta) interleaved per pixel: one
            // 16-byte store, and stage 2 reads 256-byte runs per image row instead of 64-byte ones
#pragma unroll
            for (int sc = 0; sc < SCN; sc++)
                *reinterpret_cast<float4 *>(abimg + ((size_t)sc * npx + pix) * 4) =
                    make_float4(ab_px[4 * sc], ab_px[4 * sc + 1], ab_px[4 * sc + 2], ab_px[4 * sc + 3]);
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
// stage 2, fused form (radii 45 and 52, the reference's parameter sets): the double row sums
// never reach HBM.
//
//   gf_rowstate_kernel   walks every row of every alpha/beta plane once (one lane per row, the
//                        same RowSum<float,double> chain as gf_rowsum_kernel) but stores the
//                        running sum only at every 16th column: states[plane][x/16][row].
//   gf_colwalk_kernel    one wave owns 16 columns x the 4 planes (alpha_0..2, beta) of one src
//                        channel and walks down the image in sub-tiles of T rows (2R = NSUB T).
//                        Row phase (lane = plane x row): restart each row chain from its stored
//                        state and rebuild the sub-tile's T x 16 row sums in LDS.  Column phase
//                        (lane = plane x column): ColumnSum<double,float> down the sub-tile; the
//                        value leaving the window, R[y - r], is the value that entered 2r steps
//                        earlier in the same lane, so it is kept in a register FIFO of 2r doubles
//                        (statically indexed: the loop body is one period of NSUB sub-tiles,
//                        fully unrolled).  After each sub-tile the four means of a pixel meet in
//                        LDS and q = beta + sum alpha_g I_g is formed and stored.
//
// Every double add happens in the order of the two-kernel form above, so the bytes are the same;
// measured memory-side traffic of a whole pass at 8 x 4K, colour src: 548 -> 215 B/px
// (profiles/r02base_gf_cnn.md, r02_gf_cnn.md).
// ------------------------------------------------------------------------------------------
constexpr int kSB = 16;      // columns per state block and per column-walk wave

// planes: [img][src_np / 4][h][w][4] (the four planes of a src channel interleaved per pixel);
// states: [img * np + plane][nb][h], nb = ceil(w / 16); states[..][b][row] = RowSum at column 16 b.
// grid: (plane groups of the chunk) x (64-row blocks); one workgroup = 4 waves = the 4 planes of
// a group, lane = row, walking the border-extended row ext[i] = S[bi(i - r)] from its left end.
// The workgroup fetches a chunk of 16 columns x 64 rows as float4 pixels (256 contiguous bytes per
// image row and load) and hands each wave its plane through LDS (20 KB).  The value leaving the
// window, ext[i - ks], is the value that entered ks steps earlier in the same lane: it is kept in
// a register FIFO of F >= ks floats, F a multiple of 16 (slot = step mod F, static because the
// loop body is one period of F steps, fully unrolled) - no second read.  The stream is prefixed
// with PAD dummy steps so that every chunk of 16 steps is an aligned run of 16 source columns.
// A workgroup walks its rows alone from end to end (252 chunks at 4K), so the time of the kernel
// is the time of a chunk: two chunks of loads are in flight, and a full chunk reads its 16
// operands with four 16-byte LDS reads and forms all differences before the chain of dependent
// adds (one LDS round trip per chunk, not per step).  Measured at 8 x 4K (planar layout, one wave
// per plane, one chunk in flight: 0.36 ms grey, 0.99 ms colour): 0.33 / 0.83 ms; without the
// global loads (timing-only build) 0.20 / 0.40 ms - the rest is the issue rate of the one wave a
// SIMD holds when a grey batch gives every CU a single workgroup.  Capping the registers for a
// third / fourth workgroup per CU spills and is slower.
template <int R>
__global__ __launch_bounds__(256) void gf_rowstate_kernel(const float *__restrict__ planes,
                                                          double *__restrict__ states, int h, int w,
                                                          int row_blocks, int np,
                                                          const int *__restrict__ colour, int src_np,
                                                          int nb)
{
    constexpr int KS = 2 * R + 1;
    constexpr int F = (KS + 15) & ~15;
    constexpr int NCH = F / 16;
    constexpr int PAD = (16 - R % 16) % 16;  // step t <-> extended index i = t - PAD, column i - R
    static_assert(KS + PAD <= 2 * F, "the window fills within the two peeled periods");
    const int ng = np / 4;                     // plane groups (src channels) per image
    const int grp = blockIdx.x / row_blocks;   // plane group across the chunk of images
    const int img = grp / ng, gq = grp - img * ng;
    if (colour != nullptr && gq >= 1 && colour[img] == 0)
        return;  // grey 3-channel images only carry the 4 planes of their first channel
    __shared__ __align__(16) float tE[4][kBRows][20];  // pitch 20: 16-byte rows, conflict-free

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's plane of the group
    const int row0 = (blockIdx.x - grp * row_blocks) * kBRows;
    const float4 *S4 =
        reinterpret_cast<const float4 *>(planes + ((size_t)img * src_np + gq * 4) * h * w);
    double *ST = states + ((size_t)img * np + gq * 4 + wv) * nb * h + row0 + lane;
    const int total = w + 2 * R + PAD;  // steps
    // loader role: the workgroup fetches the chunk's 64 rows x 16 columns as 1024 float4 (all
    // four planes of a pixel), thread t the pixels t, t + 256, ...: 16 consecutive threads read
    // 256 contiguous bytes of one image row
    const int cc = tid & 15;
    const bool row_ok = row0 + lane < h;
    uint32_t srow[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        srow[k] = (uint32_t)min(row0 + ((tid + 256 * k) >> 4), h - 1) * (uint32_t)w;

    // two chunks in flight (A: even chunks, B: odd ones): a workgroup walks its rows alone, one
    // chunk after the other, so the time of a chunk is the memory latency it cannot hide
    float4 preA[4], preB[4];
    // (a chunk away from both ends of the row needs no border arithmetic; pixel offsets fit 32
    //  bits - the host admits images below 2^28 pixels here - so a load is SGPR base + byte offset)
#define RF_RS_FETCH(t0_, BUF)                                                                \
    do {                                                                                     \
        const int x0_ = (t0_) - PAD - R;                                                     \
        int sx_ = x0_ + cc;                                                                  \
        if (x0_ < 0 || x0_ + 15 >= w || (t0_) + 15 >= total)                                 \
            sx_ = border_interpolate(min((t0_) + cc, total - 1) - PAD - R, w,                \
                                     RF_BORDER_REFLECT);                                     \
        _Pragma("unroll") for (int k = 0; k < 4; k++) BUF[k] = *reinterpret_cast<const float4 *>( \
            reinterpret_cast<const char *>(S4) + ((srow[k] + (uint32_t)sx_) << 4));          \
    } while (0)
    double s = 0.0;
    float fifo[F];
    // one chunk of 16 steps; PER = period (0, 1: peeled, window still filling; 2: steady state),
    // KCH = chunk of the period: step t = t0 + c with (t mod F) = KCH*16 + c static
#define RF_RS_CHUNK(PER, KCH, t0_, BUF)                                                      \
    do {                                                                                     \
        const int t0c_ = (t0_);                                                              \
        if (t0c_ < total) {                                                                  \
            __syncthreads();                                                                 \
            _Pragma("unroll") for (int k = 0; k < 4; k++)                                    \
            {                                                                                \
                const int r_ = (tid + 256 * k) >> 4;                                         \
                tE[0][r_][cc] = BUF[k].x;                                                    \
                tE[1][r_][cc] = BUF[k].y;                                                    \
                tE[2][r_][cc] = BUF[k].z;                                                    \
                tE[3][r_][cc] = BUF[k].w;                                                    \
            }                                                                                \
            __syncthreads();                                                                 \
            if (t0c_ + 32 < total)                                                           \
                RF_RS_FETCH(t0c_ + 32, BUF);                                                 \
            if ((PER) == 2 && t0c_ + 16 <= total) {                                          \
                /* full chunk in the steady state: the 16 operands first (16-byte LDS reads), */ \
                /* the differences against the FIFO (the slot a step reads is overwritten    */ \
                /* F - KS steps later), then the dependent adds: one LDS round trip per      */ \
                /* chunk instead of one per step, no per-step branch                         */ \
                float4 e4_[4];                                                               \
                _Pragma("unroll") for (int c4 = 0; c4 < 4; c4++)                             \
                    e4_[c4] = *reinterpret_cast<const float4 *>(&tE[wv][lane][4 * c4]);      \
                const float e_[16] = {e4_[0].x, e4_[0].y, e4_[0].z, e4_[0].w, e4_[1].x, e4_[1].y, \
                                      e4_[1].z, e4_[1].w, e4_[2].x, e4_[2].y, e4_[2].z, e4_[2].w, \
                                      e4_[3].x, e4_[3].y, e4_[3].z, e4_[3].w};               \
                _Pragma("unroll") for (int hh = 0; hh < 2; hh++)                             \
                {                                                                            \
                    double d_[8];                                                            \
                    _Pragma("unroll") for (int c8 = 0; c8 < 8; c8++)                         \
                        d_[c8] = (double)e_[8 * hh + c8] -                                   \
                                 (double)fifo[((KCH) * 16 + 8 * hh + c8 + F - (KS % F)) % F]; \
                    _Pragma("unroll") for (int c8 = 0; c8 < 8; c8++)                         \
                    {                                                                        \
                        const int c = 8 * hh + c8;                                           \
                        s += d_[c8];                                                         \
                        if (((c - PAD - KS + 1) & (kSB - 1)) == 0 && row_ok)                 \
                            ST[(size_t)((t0c_ + c - PAD - KS + 1) >> 4) * h] = s;            \
                    }                                                                        \
                }                                                                            \
                _Pragma("unroll") for (int c = 0; c < 16; c++) fifo[(KCH) * 16 + c] = e_[c]; \
            } else                                                                           \
            _Pragma("unroll") for (int c = 0; c < 16; c++)                                   \
            {                                                                                \
                const int tp_ = (KCH) * 16 + c;           /* t mod F */                      \
                const int ip_ = (PER) * F + tp_ - PAD;     /* i (exact in the peeled periods) */ \
                if ((PER) < 2 && ip_ < 0) {                                                  \
                    /* dummy step in front of the row */                                     \
                } else if (t0c_ + c < total) {                                               \
                    const float e_ = tE[wv][lane][c];                                        \
                    if ((PER) < 2 && ip_ < KS)                                               \
                        s += (double)e_;                                                     \
                    else                                                                     \
                        s += (double)e_ - (double)fifo[(tp_ + F - (KS % F)) % F];            \
                    fifo[tp_] = e_;                                                          \
                    const int o_ = t0c_ + c - PAD - KS + 1; /* output column of this RowSum */ \
                    if (o_ >= 0 && (o_ & (kSB - 1)) == 0 && row_ok)                          \
                        ST[(size_t)(o_ >> 4) * h] = s;                                       \
                }                                                                            \
            }                                                                                \
        }                                                                                    \
    } while (0)
    // P0 = buffer of the period's first chunk (chunks alternate A, B)
#define RF_RS_PERIOD(PER, t0_, P0, P1)                                                       \
    do {                                                                                     \
        RF_RS_CHUNK(PER, 0, (t0_), P0);                                                      \
        RF_RS_CHUNK(PER, 1, (t0_) + 16, P1);                                                 \
        RF_RS_CHUNK(PER, 2, (t0_) + 32, P0);                                                 \
        RF_RS_CHUNK(PER, 3, (t0_) + 48, P1);                                                 \
        RF_RS_CHUNK(PER, 4, (t0_) + 64, P0);                                                 \
        RF_RS_CHUNK(PER, 5, (t0_) + 80, P1);                                                 \
        if constexpr (NCH > 6)                                                               \
            RF_RS_CHUNK(PER, 6, (t0_) + 96, P0);                                             \
    } while (0)
    static_assert(NCH == 6 || NCH == 7, "period of 96 or 112 steps");
    RF_RS_FETCH(0, preA);
    RF_RS_FETCH(16, preB);
    if constexpr (NCH == 6) {
        RF_RS_PERIOD(0, 0, preA, preB);
        RF_RS_PERIOD(1, F, preA, preB);
        for (int t0 = 2 * F; t0 < total; t0 += F)
            RF_RS_PERIOD(2, t0, preA, preB);
    } else {  // 7 chunks per period: the buffers swap roles from one period to the next
        RF_RS_PERIOD(0, 0, preA, preB);
        RF_RS_PERIOD(1, F, preB, preA);
        for (int t0 = 2 * F; t0 < total; t0 += 2 * F) {
            RF_RS_PERIOD(2, t0, preA, preB);
            RF_RS_PERIOD(2, t0 + F, preB, preA);
        }
    }
#undef RF_RS_PERIOD
#undef RF_RS_CHUNK
#undef RF_RS_FETCH
}

// grid: 8 * ceil(items / 8) single-wave workgroups; items = images x SCN x nb, walked so that each
// XCD (workgroup id mod 8) owns a contiguous run of column blocks: a block's "leaving" operands
// are the "entering" operands of the block ~1.4 places to its left, served by that XCD's L2.
//
// The walk advances in sub-tiles of T padded rows (2R = NSUB * T; T = 15 for R = 45, 13 for 52):
//   row phase     lane = (plane, row): 4 x T chains of 15 steps from the stored states -> Rt (LDS)
//   column phase  lane = (plane, column): T steps of ColumnSum<double,float>; FIFO slot of step jj
//                 of the k-th sub-tile of a period = k T + jj (the period loop is unrolled)
//   flush         the T x 16 finished pixels: q = beta + a0 I0 + a1 I1 + a2 I2 -> uint8
// One wave per workgroup: no s_barrier anywhere (__syncthreads() is the LDS fence of the wave).
// The wave hides memory latency itself: operands and states of sub-tile u+1 and the guide bytes of
// sub-tile u are requested before sub-tile u's chains run.
template <int R, int T, int SCN, int SPX>
__global__ __launch_bounds__(64) void gf_colwalk_kernel(
    const float *__restrict__ ab, const double *__restrict__ states,
    const uint8_t *__restrict__ guide, uint8_t *__restrict__ dst, int h, int w, int nb,
    int n_items, const int *__restrict__ colour)
{
    constexpr int KS = 2 * R + 1;
    constexpr int NSUB = 2 * R / T;
    static_assert(NSUB * T == 2 * R && 4 * T <= 64 && (T & 1), "sub-tile height");
    constexpr int NG = (T * 12 + 63) / 64;  // guide dwords per lane and sub-tile
    constexpr int NF = (T * kSB + 63) / 64; // flush pixels per lane and sub-tile
    const int per_xcd = (n_items + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || item >= n_items)
        return;
    const int b = item % nb;
    const int s_ch = (item / nb) % SCN;
    const int img = item / (nb * SCN);
    if (wrong_variant<SCN>(colour, img))
        return;

    __shared__ double Rt[64][T];                              // row sums [plane*16 + col][row]
    __shared__ float stE[4 * T][kSB + 1], stL[4 * T][kSB + 1]; // operands [plane*T + row][col]
    __shared__ float xch[T][4][kSB];                          // means of the sub-tile's rows
    __shared__ int rowtab[2][T + 1];                          // image row of each padded row
    __shared__ uint32_t rowoff[2][T + 1];                     // ... times 16 w (byte offset of the row in a plane group)
    __shared__ uint32_t gst[T][12];                           // guide bytes of the output rows

    const int lane = threadIdx.x;
    const int g4 = lane >> 4, cc = lane & 15;  // loader and column role: plane, column
    const int cp = lane / T, cl = lane - cp * T;  // chain role (lane < 4T): plane, row
    const bool chain = lane < 4 * T;
    const size_t npx = (size_t)h * w;
    constexpr int np = 4 * SPX;
    const float *abg = ab + ((size_t)img * np + 4 * s_ch) * npx;          // planes 4s .. 4s+3
    const double *stg = states + ((size_t)img * np + 4 * s_ch) * nb * h;  // their states
    const uint8_t *gimg = guide + (size_t)img * npx * 3;
    uint8_t *dimg = dst + (size_t)img * npx * SPX;
    // RowSum at output column o = 16 b + cc (cc >= 1):  + ext[o + 2r] - ext[o - 1], ext[i] = S[bi(i - r)]
    // per-lane BYTE offsets from the wave-uniform base abg of the channel's plane group
    // ([h][w][4] floats; 32 bits: the host admits images of less than 2^28 pixels here), so that
    // base + zero-extended (row offset + lane offset) is the whole address computation
    const uint32_t oe = 4u * (uint32_t)g4 +
                        16u * (uint32_t)border_interpolate(b * kSB + cc + R, w, RF_BORDER_REFLECT);
    const uint32_t ol = 4u * (uint32_t)g4 +
                        16u * (uint32_t)border_interpolate(b * kSB + cc - 1 - R, w, RF_BORDER_REFLECT);
    const char *abgb = reinterpret_cast<const char *>(abg);
    const double *Ps = stg + ((size_t)min(cp, 3) * nb + b) * h;
    const double scale = 1.0 / (double)(KS * KS);
    const int nsub = (h + 2 * R + T - 1) / T;  // > NSUB
    const int jmax = h + 2 * R - 1;
    const uint32_t gbytes = (uint32_t)(npx * 3);

    float pe[T], pl[T];
    double pst = 0.0;
    uint32_t gpre[NG];
    /* padded row -> image row (BORDER_REFLECT) */
#define RF_ROWTAB(u_)                                                                        \
    do {                                                                                     \
        if (lane < T) {                                                                      \
            const int row_ = border_interpolate(min((u_) * T + lane, jmax) - R, h,           \
                                                RF_BORDER_REFLECT);                          \
            rowtab[(u_) & 1][lane] = row_;                                                   \
            rowoff[(u_) & 1][lane] = 16u * (uint32_t)row_ * (uint32_t)w;                     \
        }                                                                                    \
    } while (0)
#define RF_FETCH(u_)                                                                         \
    do {                                                                                     \
        _Pragma("unroll") for (int L = 0; L < T; L++)                                        \
        {                                                                                    \
            const uint32_t ro_ = rowoff[(u_) & 1][L];                                        \
            pe[L] = *reinterpret_cast<const float *>(abgb + (oe + ro_));                     \
            pl[L] = *reinterpret_cast<const float *>(abgb + (ol + ro_));                     \
        }                                                                                    \
        if (chain)                                                                           \
            pst = Ps[rowtab[(u_) & 1][cl]];                                                  \
    } while (0)
    /* guide bytes of output rows y0 .. y0+T-1, 48 contiguous bytes per row */
#define RF_GUIDE_FETCH(y0_)                                                                  \
    do {                                                                                     \
        _Pragma("unroll") for (int k = 0; k < NG; k++)                                       \
        {                                                                                    \
            const int idx_ = lane + 64 * k;                                                  \
            const int gy_ = min((y0_) + idx_ / 12, h - 1);                                   \
            const uint32_t off_ = ((uint32_t)gy_ * w + b * kSB) * 3 + (idx_ % 12) * 4;       \
            uint32_t v_ = 0;                                                                 \
            if (off_ + 4 <= gbytes) {                                                        \
                __builtin_memcpy(&v_, gimg + off_, 4);                                       \
            } else {                                                                         \
                for (int q = 0; q < 4; q++)                                                  \
                    if (off_ + q < gbytes)                                                   \
                        v_ |= (uint32_t)gimg[off_ + q] << (8 * q);                           \
            }                                                                                \
            gpre[k] = v_;                                                                    \
        }                                                                                    \
    } while (0)

    double SUM = 0.0;
    double fifo[2 * R];  // R[y - r] is what entered 2r steps ago: slot = step mod 2r, all static
    // one sub-tile; KSLOT = its place in the FIFO period, FILL = prologue (padded rows -r .. r-1)
#define RF_SUB(KSLOT, FILL, u_)                                                              \
    do {                                                                                     \
        const int uu_ = (u_);                                                                \
        RF_ROWTAB(uu_ + 1);                                                                  \
        _Pragma("unroll") for (int L = 0; L < T; L++)                                        \
        {                                                                                    \
            stE[g4 * T + L][cc] = pe[L];                                                     \
            stL[g4 * T + L][cc] = pl[L];                                                     \
        }                                                                                    \
        double s_ = pst;                                                                     \
        __syncthreads();                                                                     \
        if (!(FILL))                                                                         \
            RF_GUIDE_FETCH((uu_ - NSUB) * T);                                                \
        if (uu_ + 1 < nsub)                                                                  \
            RF_FETCH(uu_ + 1);                                                               \
        if (chain) {                                                                         \
            /* all operand differences first (independent LDS reads and conversions), then   */ \
            /* the chain of dependent adds: the wave is alone on its SIMD, nothing else hides */ \
            /* an LDS round trip per step                                                     */ \
            double d_[kSB];                                                                  \
            _Pragma("unroll") for (int c = 1; c < kSB; c++)                                  \
                d_[c] = (double)stE[lane][c] - (double)stL[lane][c];                         \
            Rt[cp * kSB][cl] = s_;                                                           \
            _Pragma("unroll") for (int c = 1; c < kSB; c++)                                  \
            {                                                                                \
                s_ += d_[c];                                                                 \
                Rt[cp * kSB + c][cl] = s_;                                                   \
            }                                                                                \
        }                                                                                    \
        __syncthreads();                                                                     \
        {                                                                                    \
            /* the sub-tile's row sums first (independent LDS reads), then the dependent chain */ \
            double v_[T];                                                                    \
            _Pragma("unroll") for (int jj = 0; jj < T; jj++) v_[jj] = Rt[lane][jj];          \
            _Pragma("unroll") for (int jj = 0; jj < T; jj++)                                 \
            {                                                                                \
                if (FILL) {                                                                  \
                    SUM += v_[jj];                                                           \
                } else {                                                                     \
                    const double s0_ = SUM + v_[jj];                                         \
                    xch[jj][g4][cc] = (float)(s0_ * scale);                                  \
                    SUM = s0_ - fifo[(KSLOT) * T + jj];                                      \
                }                                                                            \
                fifo[(KSLOT) * T + jj] = v_[jj];                                             \
            }                                                                                \
        }                                                                                    \
        if (!(FILL)) {                                                                       \
            _Pragma("unroll") for (int k = 0; k < NG; k++)                                   \
            {                                                                                \
                const int idx_ = lane + 64 * k;                                              \
                if (idx_ < T * 12)                                                           \
                    gst[idx_ / 12][idx_ % 12] = gpre[k];                                     \
            }                                                                                \
            __syncthreads();                                                                 \
            const int y0_ = (uu_ - NSUB) * T;                                                \
            /* NF independent pixels per lane, computed branch-free so that their LDS reads and */ \
            /* dependent float chains overlap (one wave per SIMD: nobody else hides them);      */ \
            /* only the stores are predicated                                                   */ \
            uint8_t o_[NF];                                                                  \
            _Pragma("unroll") for (int k = 0; k < NF; k++)                                   \
            {                                                                                \
                const int fr_ = min((lane + 64 * k) >> 4, T - 1);                            \
                const uint8_t *gb_ = reinterpret_cast<const uint8_t *>(&gst[fr_][0]) + 3 * cc; \
                float q_ = xch[fr_][3][cc];                                                  \
                q_ = __fadd_rn(q_, __fmul_rn(xch[fr_][0][cc], (float)gb_[0]));               \
                q_ = __fadd_rn(q_, __fmul_rn(xch[fr_][1][cc], (float)gb_[1]));               \
                q_ = __fadd_rn(q_, __fmul_rn(xch[fr_][2][cc], (float)gb_[2]));               \
                o_[k] = saturate_u8(q_);                                                     \
            }                                                                                \
            _Pragma("unroll") for (int k = 0; k < NF; k++)                                   \
            {                                                                                \
                const int fr_ = (lane + 64 * k) >> 4;                                        \
                const int y_ = y0_ + fr_, x_ = b * kSB + cc;                                 \
                if (fr_ < T && y_ < h && x_ < w) {                                           \
                    const uint32_t pix_ = (uint32_t)y_ * w + x_;                             \
                    if (SCN == SPX) {                                                        \
                        dimg[(size_t)pix_ * SCN + s_ch] = o_[k];                             \
                    } else { /* grey image: the computed channel stands for all three */     \
                        dimg[(size_t)pix_ * 3 + 0] = o_[k];                                  \
                        dimg[(size_t)pix_ * 3 + 1] = o_[k];                                  \
                        dimg[(size_t)pix_ * 3 + 2] = o_[k];                                  \
                    }                                                                        \
                }                                                                            \
            }                                                                                \
            __syncthreads();                                                                 \
        }                                                                                    \
    } while (0)

    RF_ROWTAB(0);
    __syncthreads();
    RF_FETCH(0);
#pragma unroll
    for (int k = 0; k < NSUB; k++)
        RF_SUB(k, true, k);
    for (int u0 = NSUB; u0 < nsub; u0 += NSUB) {
#pragma unroll
        for (int k = 0; k < NSUB; k++)
            if (u0 + k < nsub)
                RF_SUB(k, false, u0 + k);
    }
#undef RF_SUB
#undef RF_GUIDE_FETCH
#undef RF_FETCH
#undef RF_ROWTAB
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

}  // namespace

// Side streams for the second half of a batch, one per CALLER stream (so two callers never meet
// on one side stream, and a caller that captures its stream into a graph pulls only its own side
// stream into that capture).  The table is bounded: when it is full the least recently used entry
// whose side stream is idle is destroyed and replaced; if none is idle, or the caller's stream
// belongs to another device than the current one, the call runs on the caller's stream alone
// (nullptr).  rf_shutdown() destroys them all.
namespace {
struct SideStream {
    hipStream_t caller, side;
    int device;
    unsigned long long used;
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
            return g_side[i].side;
        }
    size_t slot = g_side_n;
    if (g_side_n == kMaxSideStreams) {
        slot = kMaxSideStreams;
        for (size_t i = 0; i < g_side_n; i++) {
            if (slot != kMaxSideStreams && g_side[i].used > g_side[slot].used)
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
    g_side[slot] = {caller, st, dev, ++g_side_tick};
    if (slot == g_side_n)
        g_side_n++;
    return st;
}

void gf_shutdown()
{
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (size_t i = 0; i < g_side_n; i++)
        (void)hipStreamDestroy(g_side[i].side);
    g_side_n = 0;
}

size_t gf_workspace_cap()
{
    static size_t cap = 0;  // the answer cannot change within a process; a benign race at worst
    if (cap == 0) {
        size_t c = (size_t)6 << 30;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            c = std::min(std::max(c, (size_t)prop.totalGlobalMem / 16), (size_t)16 << 30);
        else
            (void)hipGetLastError();  // no device: not an error of this call
        cap = c;
    }
    return cap;
}

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
    // enough images in flight to fill the chip and to make the tails of the launches small: capped
    // at 1/16 of the device's memory, at most 16 GiB (6 GiB when no device can be asked).  C5 shard
    // (128 x 4K, 3 passes): 91.5 ms with 6 GiB (13 images per chunk), 86.6 ms with 16 GiB (36).
    size_t imgs = (size_t)n;
    const size_t cap = rf::gf_workspace_cap();
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
    const size_t per_img = npx * np * (sizeof(float) + sizeof(double));
    const size_t header = gf_header_bytes(n);
    if (workspace_bytes < header + per_img)
        return fail(RF_E_WORKSPACE, "rf_gf_u8: workspace %zu B < %zu B needed for one image",
                    workspace_bytes, header + per_img);
    hipStream_t stream = (hipStream_t)stream_;
    // fused stage 2 (row states + column walk) for the instantiated radii; the debug option
    // "gf_two_kernel" forces the row-sum / column-sum kernel pair (cross-check of tests and tools)
    // (the fused kernels index planes with 32-bit element offsets: images below 2^28 pixels)
    const bool fused = !debug_get(kDbgGfTwoKernel) && (radius == 45 || radius == 52) &&
                       npx < ((size_t)1 << 28);
    const int nb = ceil_div(w, kSB);
    const size_t per_img_fused = (size_t)np * (npx * sizeof(float) + (size_t)nb * h * sizeof(double));
    const size_t per_img_used = fused ? per_img_fused : per_img;
    int chunk = (int)std::min<size_t>((size_t)n, (workspace_bytes - header) / per_img_used);
    if (chunk > 16383)
        chunk = 16383;
    const float eps_f = (float)eps;
    const int eps_small = eps < 1e-2;
    // stage-1 strips: stage1_threads x stage1_cols columns, 2r of them halo
    const int strips3 = ceil_div(w, stage1_threads(3) * stage1_cols(3) - 2 * radius);
    const int strips1 = ceil_div(w, stage1_threads(1) * stage1_cols(1) - 2 * radius);
    const int strips = src_cn == 3 ? strips3 : strips1;

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

    // One part = m images starting at i0, their scratch at ws, every launch on st.
    auto run_part = [&](int i0, int m, int m_fill, char *ws, hipStream_t st) {
        const int *colour = colour_all ? colour_all + i0 : nullptr;
        // two-kernel form: [row sums (double)][alpha/beta]; fused form: [states (double)][alpha/beta]
        double *rows = reinterpret_cast<double *>(ws);
        float *ab = reinterpret_cast<float *>(rows + (fused ? (size_t)m * np * nb * h
                                                            : (size_t)m * np * npx));
        const uint8_t *g0 = guide + (size_t)i0 * npx * 3;
        uint8_t *d0 = dst + (size_t)i0 * npx * src_cn;
        // row segments: enough workgroups to fill 256 CUs, but segments no shorter than 2r+1.
        // (tools/gf_seg_sweep.py: a pass is flat within 3 % between 34 and 135 rows per segment at
        // 4K - the 2r warm-up rows of a segment are cheap - and slower above; a model that picks
        // the segment count by whole rounds of resident workgroups was no better.)
        // (m_fill: the images in flight on the device, i.e. both halves of a chunk)
        int seg_rows = h;
        while ((long long)strips * ceil_div(h, seg_rows) * m_fill < 1024 &&
               seg_rows > 2 * (2 * radius + 1) && seg_rows > 32)
            seg_rows = (seg_rows + 1) / 2;
        if (debug_get(kDbgGfSegRows) > 0)
            seg_rows = std::min(h, debug_get(kDbgGfSegRows));
        const int segs = ceil_div(h, seg_rows);
        for (int it = 0; it < iterations; it++) {
            const uint8_t *s0 = (it == 0 ? src : (const uint8_t *)dst) + (size_t)i0 * npx * src_cn;
            const dim3 ga3(strips3, segs, m), ga1(strips1, segs, m);
            if (src_cn == 3) {
                hipLaunchKernelGGL((gf_stage1_kernel<3, 3>), ga3, dim3(stage1_threads(3)), 0, st, g0, s0,
                                   ab, h, w, radius, eps_f, eps_small, seg_rows, colour);
                hipLaunchKernelGGL((gf_stage1_kernel<1, 3>), ga1, dim3(stage1_threads(1)), 0, st, g0, s0,
                                   ab, h, w, radius, eps_f, eps_small, seg_rows, colour);
            } else {
                hipLaunchKernelGGL((gf_stage1_kernel<1, 1>), ga1, dim3(stage1_threads(1)), 0, st, g0, s0,
                                   ab, h, w, radius, eps_f, eps_small, seg_rows, colour);
            }
            const int row_blocks = ceil_div(h, kBRows);
            if (fused) {
                if (radius == 45)
                    hipLaunchKernelGGL((gf_rowstate_kernel<45>), dim3((unsigned)(m * src_cn * row_blocks)),
                                       dim3(256), 0, st, ab, rows, h, w, row_blocks, np, colour, np, nb);
                else
                    hipLaunchKernelGGL((gf_rowstate_kernel<52>), dim3((unsigned)(m * src_cn * row_blocks)),
                                       dim3(256), 0, st, ab, rows, h, w, row_blocks, np, colour, np, nb);
                const int it3 = m * 3 * nb, it1 = m * nb;
                const dim3 g3(8 * (unsigned)ceil_div(it3, 8)), g1(8 * (unsigned)ceil_div(it1, 8));
#define RF_GF_WALK(R, TT)                                                                             \
    do {                                                                                           \
        if (src_cn == 3) {                                                                         \
            hipLaunchKernelGGL((gf_colwalk_kernel<R, TT, 3, 3>), g3, dim3(64), 0, st, ab, rows, g0, d0, \
                               h, w, nb, it3, colour);                                             \
            hipLaunchKernelGGL((gf_colwalk_kernel<R, TT, 1, 3>), g1, dim3(64), 0, st, ab, rows, g0, d0, \
                               h, w, nb, it1, colour);                                             \
        } else {                                                                                   \
            hipLaunchKernelGGL((gf_colwalk_kernel<R, TT, 1, 1>), g1, dim3(64), 0, st, ab, rows, g0, d0, \
                               h, w, nb, it1, colour);                                             \
        }                                                                                          \
    } while (0)
                if (radius == 45)
                    RF_GF_WALK(45, 15);
                else
                    RF_GF_WALK(52, 13);
#undef RF_GF_WALK
                continue;
            }
            hipLaunchKernelGGL(gf_rowsum_kernel<4>, dim3((unsigned)(m * np * row_blocks)), dim3(64), 0,
                               st, ab, rows, h, w, radius, row_blocks, np, colour, np);
            dim3 gc(ceil_div(w, 64), 1, m);
            if (src_cn == 3) {
                hipLaunchKernelGGL((gf_colsum_apply_kernel<3, 3>), gc, dim3(64, 12), 0, st, rows,
                                   g0, d0, h, w, radius, colour);
                hipLaunchKernelGGL((gf_colsum_apply_kernel<1, 3>), gc, dim3(64, 4), 0, st, rows,
                                   g0, d0, h, w, radius, colour);
            } else {
                hipLaunchKernelGGL((gf_colsum_apply_kernel<1, 1>), gc, dim3(64, 4), 0, st, rows,
                                   g0, d0, h, w, radius, colour);
            }
        }
    };

    // A chunk of two or more images runs as two halves on two streams - the caller's and a side
    // stream of the library, forked and joined with events, so the call still looks stream-ordered
    // to the caller and can be captured into a graph.  The kernels of a pass are bound by different
    // things (stage 1: latency at 3-4 waves per SIMD; row states: the latency of a chunk; column
    // walk: the issue rate of its one wave per SIMD, which leaves half of a SIMD's registers
    // free), and every launch ends in a tail of partly filled CUs: the other half's kernels fill
    // both.  Measured (3 passes at 4K): 8 images grey 6.13 -> 6.08 ms, colour 14.0 -> 12.2 ms; 13
    // images grey 10.8 -> 9.4 ms, colour 24.1 -> 19.9 ms.  The debug option "gf_one_stream" keeps
    // everything on the caller's stream (cross-check; identical bytes).
    char *ws0 = static_cast<char *>(workspace) + header;
    hipStream_t side = nullptr;
    if (fused && chunk >= 2 && !debug_get(kDbgGfOneStream))
        side = gf_side_stream(stream);
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        if (side == nullptr || m < 2) {
            run_part(i0, m, m, ws0, stream);
            continue;
        }
        const int ma = (m + 1) / 2;
        struct Events {  // destroyed on every path out (a pending event is released on completion)
            hipEvent_t fork = nullptr, join = nullptr;
            ~Events()
            {
                if (fork)
                    (void)hipEventDestroy(fork);
                if (join)
                    (void)hipEventDestroy(join);
            }
        } ev;
        RF_HIP_CHECK(hipEventCreateWithFlags(&ev.fork, hipEventDisableTiming));
        RF_HIP_CHECK(hipEventCreateWithFlags(&ev.join, hipEventDisableTiming));
        RF_HIP_CHECK(hipEventRecord(ev.fork, stream));
        RF_HIP_CHECK(hipStreamWaitEvent(side, ev.fork, 0));
        run_part(i0, ma, m, ws0, stream);
        run_part(i0 + ma, m - ma, m, ws0 + (size_t)ma * per_img_used, side);
        RF_HIP_CHECK(hipEventRecord(ev.join, side));
        RF_HIP_CHECK(hipStreamWaitEvent(stream, ev.join, 0));
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
