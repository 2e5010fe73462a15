// rf_gf_fused.hpp -- fused stage 2 of the guided filter (box means of the alpha/beta planes) for
// any radius, gfx950.  Included by rf_gf.hip (shared helpers) and by the rf_gf_fused_*.hip units
// that instantiate the kernels per radius.
//
// Contract (opencv/modules/imgproc/src/smooth.cpp boxFilter on CV_32F planes, called 12 times by
// opencv_contrib/modules/ximgproc/src/guided_filter.cpp; reference call site
// /root/reference/filter_reflectance.py:67-70): RowSum<float,double> is a running sum along the
// border-extended row from its left end, ColumnSum<double,float> a running sum down the image
// from 2r rows above the top.  These float sums are not exact, so the order is part of the
// contract; both chains are kept sequential and the double row sums never reach HBM:
//
//   gf_rowstate_kernel<R>  walks every row of every alpha/beta plane once (one lane per row) and
//                          stores the running sum only at every 16th column.
//   gf_colwalk_kernel<R>   one workgroup of TWO waves owns 16 columns x the 4 planes (alpha_0..2,
//                          beta) of one src channel and walks down the image in sub-tiles of at most
//                          16 rows; the waves take the sub-tiles alternately.
//
// The radius is a template parameter because the value that leaves a running window is the value
// that entered it 2r (+1) steps earlier IN THE SAME LANE: it is kept in a register FIFO whose slots
// must be named statically.  The radius-dependent code is small (the FIFO steps); the kernels are
// instantiated for every radius 1..kGfFusedMaxRadius and picked by a table (rf_gf_fused_*.hip).
#pragma once
#include "rf_common.hpp"

#include <algorithm>
#include <type_traits>

namespace rf {

constexpr int kGfFusedMaxRadius = 128;  // (the 8-bit stage 1's uint32 window sums reach radius 128; round 6: the fused stage 2 follows it)
constexpr int kGfFusedSmallMax = 96;    // radii above this one are compiled apart (rf_gf_fused_inst.hip, -DRF_GF_LARGE)
constexpr int kSB = 16;     // columns per state block and per column-walk workgroup
constexpr int kBRows = 64;  // rows per row-walk workgroup (one lane per row)
constexpr int kGsFloatsPublic = 9;  // floats per pixel of the guide record kept by iterated calls

// ------------------------------------------------------------------------------------------
// "Exact rows" (round 6).  RowSum<float,double> is a chain of double additions whose ORDER is part of
// the contract only as far as it rounds.  Sufficient test for a row of one plane (or of several
// planes taken together): let 2^e be the weight of the last mantissa bit of its smallest non-zero
// magnitude (every value of the row is then a multiple of 2^e, zeros included) and M its largest
// magnitude; if ks * M <= 2^(e + 53), then every partial sum of at most ks of the row's values
// (border-reflected repeats included) is a multiple of 2^e below 2^(e + 53) in magnitude, i.e. a
// double: the chain's first ks additions, every difference `entering - leaving` (|.| <= 2 M) and
// every `s += difference` (a window sum again) are exact, the chain holds the mathematically exact
// window sum at every column, and ANY order of exact additions over subsets of a window gives the
// same doubles.  In exponent fields (E = bits >> 23, a denormal's taken as 1):
//     max(Emax, 1) - max(Emin, 1) <= 29 - ceil(log2 ks)         (gf_exact_limit)
// because M < 2^(Emax - 126) and e = Emin - 150.
//
// For such rows stage 2 needs no row walk: stage 1 leaves the sum of every aligned 16-column block
// (xf, see rf_gf.hip), and the column walk rebuilds the row sum at its block's first column 16 b as
//     prefix of block b + q up to column 16 b + R   (= F[b + q] - the c0' entering operands behind it)
//   + F[b - q] + ... + F[b + q - 1]                  (block indices reflected like columns)
//   + suffix of block b - q - 1 from column 16 b - R (= its first c0 leaving operands)
// with q = R / 16, c0 = R % 16, c0' = 15 - c0 - every intermediate a sum over a subset of the window.
// Rows that fail the test are listed by gf_exact_rows_kernel (from the per-half-wave statistics
// stage 1 leaves), walked sequentially by gf_rowstate_kernel in its list mode, which writes the
// chain's true value into slot b of the same array, and the column walk takes that slot as it is.
// Needs w % 16 == 0 (a reflected block is a block); other widths take the row walk for every row.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int gf_exact_limit(int radius)
{
    const int ks = 2 * radius + 1;
    int lim = 29;
    while ((1 << (29 - lim)) < ks)
        lim--;
    return lim;
}
// Layout of the block sums / row states: element (plane, block, row) of a plane group sits at
// plane * sp + block * sb + row * sr doubles.  Row walk: [plane][nb][h]; exact rows: [row][plane][nb].
struct GfStateLayout {
    int sp, sb, sr;
};
constexpr int kGfExactMaxH = 16384;  // rows of an image the column walk's flag bitmask (LDS) holds
struct GfExact {
    const unsigned *rowmask;  // [img * groups + group][mask_words]: bit r = row r takes its state from the row walk
    const int *list;          // [img * groups + group][h]: the flagged rows, count[..] of them
    const int *count;
    int mask_words;           // ceil(h / 32)
    int on;                   // 0: every row takes the row walk (rowmask, list, count unused)
};

// One lane per (plane group, row): the row's exactness flag from the statistics of stage 1.
//   stats [img * groups + group][row][slots] {alpha word, beta word}: (max16 << 16) | (0xffff - min16)
// slots = stride of a row's statistics, used1 / used3 = how many of them the one- / three-channel
// stage 1 writes.  grid (ceil(h / 256), images x groups); count[] and rowmask[] zeroed beforehand.
template <int kHeaderOnly = 0>
__global__ __launch_bounds__(256) void gf_exact_rows_kernel(const uint2 *__restrict__ stats, int h,
                                                            int slots, int used1, int used3,
                                                            int groups, int limit,
                                                            const int *__restrict__ colour,
                                                            unsigned *__restrict__ rowmask,
                                                            int mask_words, int *__restrict__ list,
                                                            int *__restrict__ count)
{
    const int ig = blockIdx.y, img = ig / groups, g = ig - img * groups;
    if (colour != nullptr && g >= 1 && colour[img] == 0)
        return;  // grey 3-channel image: only its first channel's planes exist
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= h)
        return;
    // (the slots the image's stage-1 instantiation wrote: its strips x 8)
    const int used = (colour != nullptr && colour[img] != 0) ? used3 : used1;
    const uint2 *S = stats + ((size_t)ig * h + row) * slots;
    uint32_t wa = 0, wb = 0;
    for (int i = 0; i < used; i++) {
        const uint2 v = S[i];
        wa = max(wa & 0xffff0000u, v.x & 0xffff0000u) | max(wa & 0xffffu, v.x & 0xffffu);
        wb = max(wb & 0xffff0000u, v.y & 0xffff0000u) | max(wb & 0xffffu, v.y & 0xffffu);
    }
    auto fails = [&](uint32_t wd) {
        const int mx16 = (int)(wd >> 16), mn16 = 0xffff - (int)(wd & 0xffffu);
        if (mx16 == 0)
            return false;  // an all-zero row
        const int emax = max(mx16 >> 8, 1), emin = max(mn16 >> 8, 1);
        return emax == 255 || emax - emin > limit;  // (255: an infinity or a NaN in the row)
    };
    if (fails(wa) || fails(wb)) {
        atomicOr(&rowmask[(size_t)ig * mask_words + (row >> 5)], 1u << (row & 31));
        list[(size_t)ig * h + atomicAdd(&count[ig], 1)] = row;
    }
}

// Does this workgroup's instantiation apply to image img?  (colour == nullptr: no choice to make)
template <int SCN>
__device__ inline bool wrong_variant(const int *__restrict__ colour, int img)
{
    return colour != nullptr && (colour[img] != 0) != (SCN == 3);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Orders the LDS traffic of ONE wave: the LDS executes a wave's instructions in issue order, so a
// later read of another lane's earlier write needs no barrier, only that the compiler keeps the
// order and that the data has landed before it is used (lgkmcnt).  Global loads stay in flight.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// Row walk.  planes: [img][src_np / 4][h][w][4] (the four planes of a src channel interleaved per
// pixel); states: [img * np + plane][nb][h], nb = ceil(w / 16); states[..][b][row] = RowSum at
// column 16 b.  grid: (plane groups of the chunk) x (64-row blocks); one workgroup = 4 waves = the
// 4 planes of a group, lane = row, walking the border-extended row ext[i] = S[bi(i - r)] from its
// left end.  The workgroup fetches a chunk of 16 columns x 64 rows as float4 pixels (256
// contiguous bytes per image row and load) and hands each wave its plane through LDS (20 KB).  The
// value leaving the window, ext[i - ks], is kept in a register FIFO of F >= ks floats, F a
// multiple of 16 (slot = step mod F, static because the loop body is one period of F steps, fully
// unrolled).  The stream is prefixed with PAD dummy steps so that every chunk of 16 steps is an
// aligned run of 16 source columns.  A workgroup walks its rows alone from end to end, so the time
// of the kernel is the time of a chunk: up to four chunks of loads are in flight (a ring of buffers,
// the buffer of every chunk static), and a full chunk forms its 16 operand differences before the
// chain of dependent adds (one LDS round trip per chunk, not per step).
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void gf_rowstate_kernel(const float *__restrict__ planes,
                                                          double *__restrict__ states, int h, int w,
                                                          int row_blocks, int np,
                                                          const int *__restrict__ colour, int src_np,
                                                          int nb, int streaming, const GfStateLayout lay,
                                                          const GfExact xr)
{
    // List mode (xr.on; exact rows): the workgroup's 64 lanes are the entries 64 rb .. 64 rb + 63 of the
    // plane group's list of flagged rows instead of 64 consecutive rows; workgroups past the end of
    // the list exit at once - on typical images a fraction of a percent of the rows is listed.
    constexpr int KS = 2 * R + 1;
    constexpr int F = (KS + 15) & ~15;
    constexpr int NCH = F / 16;
    constexpr int PAD = (16 - R % 16) % 16;  // step t <-> extended index i = t - PAD, column i - R
    // chunks of loads in flight per workgroup (the kernel is bound by the latency of its loads: with
    // two, 64 KB per CU are in flight - 4.75 TB/s; registers are free up to 256 at two waves per SIMD),
    // and periods per loop body: the body covers a whole number of turns of the buffer ring so that
    // every chunk's buffer is static
    constexpr int NBUF = (NCH % 2 == 0) ? 4 : (NCH % 3 == 0) ? 3 : 2;
    constexpr int NPER = (NCH % NBUF == 0) ? 1 : 2;
    static_assert((NPER * NCH) % NBUF == 0, "the loop body is whole turns of the buffer ring");
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
    const int ig = img * ng + gq;
    const int listed = xr.on ? xr.count[ig] - row0 : 0;  // list entries from this workgroup's first one on
    if (xr.on && listed <= 0)
        return;
    const int *rlist = xr.list + (size_t)ig * h + row0;
    // image row of the workgroup's row slot r (loads of slots past the end repeat the last one)
    auto slot_row = [&](int r) __attribute__((always_inline)) {
        return xr.on ? rlist[min(r, listed - 1)] : min(row0 + r, h - 1);
    };
    const float4 *S4 =
        reinterpret_cast<const float4 *>(planes + ((size_t)img * src_np + gq * 4) * h * w);
    double *ST = states + ((size_t)img * np + gq * 4) * nb * h + (size_t)wv * lay.sp +
                 (size_t)slot_row(lane) * lay.sr;
    const int total = w + 2 * R + PAD;  // steps
    // loader role: the workgroup fetches the chunk's 64 rows x 16 columns as 1024 float4 (all
    // four planes of a pixel), thread t the pixels t, t + 256, ...: 16 consecutive threads read
    // 256 contiguous bytes of one image row
    const int cc = tid & 15;
    const bool row_ok = xr.on ? lane < listed : row0 + lane < h;
    uint32_t srow[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        srow[k] = (uint32_t)slot_row((tid + 256 * k) >> 4) * (uint32_t)w;

    float4 pre[NBUF][4];
    // (a chunk away from both ends of the row needs no border arithmetic; pixel offsets fit 32
    //  bits - the host admits images below 2^28 pixels here - so a load is SGPR base + byte offset)
    auto fetch = [&](int t0, float4(&buf)[4]) __attribute__((always_inline)) {
        const int x0 = t0 - PAD - R;
        int sx = x0 + cc;
        if (x0 < 0 || x0 + 15 >= w || t0 + 15 >= total)
            sx = border_interpolate(min(t0 + cc, total - 1) - PAD - R, w, RF_BORDER_REFLECT);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            // (non-temporal for batches far beyond the L2: the row walk reads every alpha/beta value
            //  exactly once, and keeping them out of the L2 leaves it to the column walk of the other
            //  half of the batch - C5 step 67.9 -> 67.0 ms.  A small batch, whose alpha/beta stage 1
            //  has just left in the L2, wants them from there: one 256x256 image 0.47 ms against 0.71.
            //  The same hint on stage 1's alpha/beta STORES costs 35 %.)
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 *p_ = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(S4) +
                                                              ((srow[k] + (uint32_t)sx) << 4));
            const f32x4 t_ = streaming ? __builtin_nontemporal_load(p_) : *p_;
            buf[k] = make_float4(t_.x, t_.y, t_.z, t_.w);
        }
    };
    double s = 0.0;
    float fifo[F];
    // one chunk of 16 steps; PER = period (0, 1: peeled, window still filling; 2: steady state),
    // KCH = chunk of the period: step t = t0 + c with (t mod F) = KCH*16 + c static
    auto chunk = [&](auto per_c, auto kch_c, int t0, float4(&buf)[4]) __attribute__((always_inline)) {
        constexpr int PER = decltype(per_c)::value, KCH = decltype(kch_c)::value;
        if (t0 >= total)
            return;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r_ = (tid + 256 * k) >> 4;
            tE[0][r_][cc] = buf[k].x;
            tE[1][r_][cc] = buf[k].y;
            tE[2][r_][cc] = buf[k].z;
            tE[3][r_][cc] = buf[k].w;
        }
        __syncthreads();
        if (t0 + 16 * NBUF < total)
            fetch(t0 + 16 * NBUF, buf);
        if (PER == 2 && KS >= 16 && t0 + 16 <= total) {
            // full chunk in the steady state: the 16 operands first (16-byte LDS reads), the
            // differences against the FIFO (every value that leaves during the chunk entered in
            // an earlier chunk because ks >= 16; radii below 8 take the step-by-step path), then
            // the dependent adds: one LDS round trip per chunk, no per-step branch
            float4 e4[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++)
                e4[c4] = *reinterpret_cast<const float4 *>(&tE[wv][lane][4 * c4]);
            const float e[16] = {e4[0].x, e4[0].y, e4[0].z, e4[0].w, e4[1].x, e4[1].y,
                                 e4[1].z, e4[1].w, e4[2].x, e4[2].y, e4[2].z, e4[2].w,
                                 e4[3].x, e4[3].y, e4[3].z, e4[3].w};
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                double d[8];
#pragma unroll
                for (int c8 = 0; c8 < 8; c8++)
                    d[c8] = (double)e[8 * hh + c8] -
                            (double)fifo[(KCH * 16 + 8 * hh + c8 + F - (KS % F)) % F];
#pragma unroll
                for (int c8 = 0; c8 < 8; c8++) {
                    const int c = 8 * hh + c8;
                    s += d[c8];
                    if (((c - PAD - KS + 1) & (kSB - 1)) == 0 && row_ok)
                        ST[(size_t)((t0 + c - PAD - KS + 1) >> 4) * lay.sb] = s;
                }
            }
#pragma unroll
            for (int c = 0; c < 16; c++)
                fifo[KCH * 16 + c] = e[c];
        } else {
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int tp = KCH * 16 + c;         // t mod F
                const int ip = PER * F + tp - PAD;    // i (exact in the peeled periods)
                if (PER < 2 && ip < 0) {
                    // dummy step in front of the row
                } else if (t0 + c < total) {
                    const float e = tE[wv][lane][c];
                    if (PER < 2 && ip < KS)
                        s += (double)e;
                    else
                        s += (double)e - (double)fifo[(tp + F - (KS % F)) % F];
                    fifo[tp] = e;
                    const int o = t0 + c - PAD - KS + 1;  // output column of this RowSum
                    if (o >= 0 && (o & (kSB - 1)) == 0 && row_ok)
                        ST[(size_t)(o >> 4) * lay.sb] = s;
                }
            }
        }
    };
    // one period of NCH chunks starting at step t0; FIRST = buffer of its first chunk (chunk n of the
    // row uses buffer n mod NBUF)
    auto period = [&](auto per_c, auto first_c, int t0) __attribute__((always_inline)) {
        constexpr int FIRST = decltype(first_c)::value;
        static_for<0, NCH>([&](auto k) __attribute__((always_inline)) {
            constexpr int K = decltype(k)::value;
            chunk(per_c, k, t0 + 16 * K, pre[(FIRST + K) % NBUF]);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    static_for<0, NBUF>([&](auto k) __attribute__((always_inline)) {
        constexpr int K = decltype(k)::value;
        fetch(16 * K, pre[K]);
    });
    period(I0{}, I0{}, 0);
    period(I1{}, std::integral_constant<int, NCH % NBUF>{}, F);
    for (int t0 = 2 * F; t0 < total; t0 += NPER * F) {
        period(I2{}, std::integral_constant<int, (2 * NCH) % NBUF>{}, t0);
        if constexpr (NPER == 2)
            period(I2{}, std::integral_constant<int, (3 * NCH) % NBUF>{}, t0 + F);
    }
}

// ------------------------------------------------------------------------------------------
// Column walk.
//
// Geometry of the walk for radius R.  The image rows are padded to i = 0 .. h + 2R - 1 (padded row
// i <-> image row bi(i - R)).  A period of 2R padded rows is cut into 2M sub-tiles, M = ceil(R/16)
// per half period, of T or T - 1 rows each (T = ceil(R/M) <= 16; both halves are cut alike).  The
// two waves of a workgroup take the sub-tiles alternately (wave = sub-tile index mod 2).  A value
// that enters the column window in sub-tile j leaves it exactly one period (2M sub-tiles, an even
// number) later, i.e. in a sub-tile OF THE SAME WAVE at the same position: each wave keeps only its
// own half of the 2R-deep FIFO in registers (R doubles, +-1), all slots static.
// ------------------------------------------------------------------------------------------
template <int R>
struct WalkGeom {
    static constexpr int M = (R + 15) / 16;
    static constexpr int T = (R + M - 1) / M;
    static constexpr int BIG = R - M * (T - 1);  // the first BIG sub-tiles of a half have T rows
    static_assert(R >= 1 && T >= 1 && T <= 16 && BIG >= 1 && BIG <= M, "walk geometry");
    // c = position in the period, 0 .. 2M-1
    __host__ __device__ static constexpr int size(int c) { return (c % M) < BIG ? T : T - 1; }
    __host__ __device__ static constexpr int start(int c)
    {
        const int p = c % M;
        return (c >= M ? R : 0) + p * (T - 1) + (p < BIG ? p : BIG);
    }
    // first FIFO slot of position c in its wave's FIFO
    __host__ __device__ static constexpr int fifo_off(int c)
    {
        int o = 0;
        for (int k = c & 1; k < c; k += 2)
            o += size(k);
        return o;
    }
    __host__ __device__ static constexpr int fifo_len()
    {
        int a = 0, b = 0;
        for (int k = 0; k < 2 * M; k += 2)
            a += size(k);
        for (int k = 1; k < 2 * M; k += 2)
            b += size(k);
        return a > b ? a : b;
    }
};

// Diagnostic build only (-DRF_GF_STAMP, tools/gf_stamp_build.py): per-phase shader cycles of the
// column walk, summed over waves, in a buffer of their own; no output depends on them.
#ifdef RF_GF_STAMP
__device__ unsigned long long g_cw_stamps[16];
#define RF_STAMP(i)                                                     \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        st_acc[i] += now_ - st_t;                                       \
        st_t = now_;                                                    \
    } while (0)
#else
#define RF_STAMP(i) do { } while (0)
#endif

template <int T>
struct alignas(16) WalkLds {
    static constexpr int TP = T | 1;  // odd pitch: 64-bit column reads are conflict-free
    double Rt[64][TP];                // row sums [plane*16 + col][row]
    union {
        struct {
            float E[4 * T][kSB + 1], L[4 * T][kSB + 1];  // operands [plane*T + row][col]
        } st;
        float xch[T][80];  // means of the sub-tile's rows [row][plane*16 + col] (pitch 80: the
                           // four rows a flush instruction reads sit on different banks)
    } u;
    int rowtab[2][T + 1];     // image row of each padded row
    uint32_t rowoff[2][T + 1];  // ... times 16 w (byte offset of the row in a plane group)
};

// grid: 8 * ceil(pairs / 8) * spx workgroups of two waves; pairs = images x nb column blocks, walked
// so that each XCD (workgroup id mod 8) owns a contiguous run of column blocks: a block's "leaving"
// operands are the "entering" operands of the block ~(2r+1)/16 places to its left.
//   row phase     lane = (plane, row): 4 x T chains of 15 steps from the stored states -> Rt (LDS)
//   column phase  lane = (plane, column): up to T steps of ColumnSum<double,float>; the running
//                 SUM passes from wave to wave through LDS (sumx, turn)
//   flush         the finished pixels: q = beta + a0 I0 + a1 I1 + a2 I2 -> uint8
// Each wave requests the operands and states of its next sub-tile before the chains of the current
// one run; nothing but the SUM hand-off couples the two waves, so one wave's loads, row chains and
// flush overlap the other's (two waves per SIMD up to radius ~64).
// Vector-memory instructions are the scarce resource of this kernel (a sub-tile of dword loads and
// byte stores spent 4 of its 9 thousand cycles issuing them): operands are fetched as float4 pixels,
// four image rows per instruction (lane = row of the group x column), the guide bytes of a lane's
// four output pixels come as one 12-byte load straight into registers, and a lane stores its four
// pixels with one instruction where the layout allows (12 bytes for a grey 3-channel image, 4 for a
// 1-channel one) - 11 to 14 vector-memory instructions per sub-tile instead of 46.
// spx = src bytes per pixel (1 or 3).  With spx = 3 an image whose flag colour[img] is 0 has three
// equal channels: only its channel 0 is computed and the result byte is written three times.
//
// CHAINED form (xc.xst != nullptr; debug option "gf_chained", round 4; bit-identical but measured
// slower than the row walk, see the end of this comment): there is no row-walk kernel.  The
// row sum a block's chains start from is the one its LEFT neighbour's chains end with, so block b
// takes the 64 (plane, row) sums of every sub-tile from block b - 1 and hands its own to block
// b + 1 through memory; block 0 starts from the sums of the row's first 2R extended pixels
// (gf_rowhead_kernel).  Every alpha/beta value is then read by the column walk only (entering and
// leaving operand) instead of once more by a row walk: 16 B per pixel and pass less traffic.
//   * Work items are handed out by ticket (one atomic counter per queue, queue = workgroup id mod 8
//     = the XCD the hardware deals the workgroup to): queue q owns the images i = q (mod 8) of the
//     launch and walks block-major through them, so a workgroup's left neighbour always holds an
//     EARLIER ticket of the same queue - it is resident or finished, never waiting for a place:
//     the chain cannot deadlock whatever the dispatch order.  Neighbours sit on one XCD (one L2).
//   * Hand-off without flags or fences: a sum travels as two 8-byte words {low half, tag},
//     {high half, tag'} - 8-byte accesses are single-copy atomic, so each word validates itself.
//     The tags mix the launch's epoch (a device-side counter gf_rowhead_kernel increments, so a
//     replayed graph gets fresh ones), the block, the sub-tile and the lane: what an earlier launch
//     or another use of the workspace left in a slot does not match (2^-64 for arbitrary bytes).
//     Stores and loads are agent-scope atomics: coherent across the XCDs' L2s by themselves.
//     The consumer requests a sub-tile's words one iteration ahead and polls only if they are not
//     there yet.
//   * Measured at the C5 shard (128 x 4K, 3 passes; profiles/r04_gf_chained.md): the chained walk
//     moves 63.4 B per pixel and pass (54.5 fetched) against 52.6 for row walk + column walk, and
//     the step takes 79.5 ms against 72.5: a block runs two to three sub-tiles behind its left
//     neighbour (hand-off latency + the one-iteration prefetch), so the 128-byte lines that
//     neighbouring blocks' misaligned 256-byte operand runs share - and the leaving operands, which
//     are the entering operands of the blocks 5 and 6 places to the left - are no longer found in
//     the XCD's L2, which the lock-step walk of the row-walk form gets for free.
//   * Polls are bounded: a consumer whose words never arrive goes on with what it has (wrong bytes
//     instead of a hung device) and ORs 1 into xc.sync[8].  Nothing reads or clears that word - the
//     form is an experiment behind a debug switch, and the byte comparison of
//     test_gf_switches_keep_the_bytes is what would notice a stalled hand-off.
// sub-tile slots per block of the chained hand-off, and the bytes of one image's hand-off buffer
__host__ __device__ inline int gf_chain_nsub(int h, int radius)
{
    const int m = (radius + 15) / 16;
    return ((h + 4 * radius - 1) / (2 * radius)) * 2 * m;
}
struct GfChain {
    const double *head;        // [img * np + plane][h]: sum of the row's first 2R extended pixels
    unsigned long long *xst;   // [(img * spx + ch) * nb + b][nsub_all][64]: 16-byte slots {low, tag0, high, tag1}
    unsigned *sync;            // [0..7] tickets, [8] error flag, [9] epoch
};

typedef unsigned gf_u32x4 __attribute__((ext_vector_type(4)));

// The two tags of a hand-off slot: independent 32-bit mixes (bijective finalisers over differently
// combined inputs), so that a slot written for any other (epoch, block, sub-tile, lane) - an earlier
// launch, another layout of the same workspace - matches with probability 2^-64, and a slot of the
// same place from an earlier launch (only the epoch differs) never does.
__device__ __forceinline__ unsigned gf_chain_tag0(unsigned seed, unsigned j, unsigned lane)
{
    unsigned x = seed ^ (j * 0x85ebca77u) ^ (lane * 0xc2b2ae3du);
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    x *= 0x297a2d39u;
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ unsigned gf_chain_tag1(unsigned seed, unsigned j, unsigned lane)
{
    unsigned x = seed + j * 0x27d4eb2fu + lane * 0x165667b1u;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// (CHAINED is a template parameter: with the choice made at run time the row-walk form of the very
//  same kernel fetched 41 instead of 29 B/px at the C5 shard - profiles/r04_c5_traffic.md - so the
//  experiment is compiled apart, and only for the reference's two radii, kGfChainedRadii.)
constexpr bool gf_chained_radius(int r) { return r == 45 || r == 52; }
// (EXACT - the exact-row start of the chains - is a template parameter for the same reason: compiled
//  into the one kernel, its registers and loads cost the row-walk form 3.6 % alone and 4 ms of the C5
//  step under two streams, profiles/r06_gf_exact.md; compiled for the same two radii.)
constexpr bool gf_exact_radius(int r) { return gf_chained_radius(r); }

template <int R, bool CHAINED = false, bool EXACT = false>
__global__ __launch_bounds__(128, (R <= 64 ? 2 : 1)) void gf_colwalk_kernel(
    const float *__restrict__ ab, const double *__restrict__ states,
    const uint8_t *__restrict__ guide, uint8_t *__restrict__ dst, int h, int w, int nb,
    int n_pairs, int spx, const int *__restrict__ colour, const GfChain xc,
    uint8_t *__restrict__ compact, const GfStateLayout lay, const GfExact xr,
    uint8_t *__restrict__ compact3, int chan_group)
{
    using G = WalkGeom<R>;
    constexpr int M = G::M, T = G::T;
    constexpr int KS = 2 * R + 1;
    // exact rows (xr.on): the row chains start from block sums instead of stored states, see the top
    constexpr int XQ = R / 16, XC0 = R % 16, NF = 2 * XQ + 1;
    constexpr int FL = G::fifo_len();
    constexpr int NL = (T + 3) / 4;  // operand load instructions per sub-tile and side (4 rows each)
    // work item = (image, column block) pair x src channel.  Workgroups are dealt round-robin to
    // the 8 XCDs, so workgroup id -> (XCD, index on it); every XCD owns a contiguous run of pairs
    // and walks pair by pair through the channels: the channels of a block share their guide rows
    // in that XCD's L2, and the channel items of grey 3-channel images that exit at once are spread
    // evenly over the XCDs
    __shared__ WalkLds<T> lds[2];
    __shared__ double sumx[64];
    __shared__ int turn;  // next sub-tile whose column phase may run
    __shared__ int ticket;
    __shared__ unsigned xmask[EXACT ? kGfExactMaxH / 32 : 1];  // exact rows: the image's flagged rows

    constexpr bool chained = CHAINED;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    WalkLds<T> &L = lds[wv];
    if (threadIdx.x == 0) {
        turn = 0;
        if (chained)
            ticket = (int)atomicAdd(&xc.sync[blockIdx.x & 7], 1u);
    }
    static_assert(!(CHAINED && EXACT), "one or the other");
    if constexpr (EXACT) {
        // (the work item is known from blockIdx alone in this form: see below)
        const int per_xcd_ = (n_pairs + 7) >> 3, q_ = (int)(blockIdx.x >> 3);
        int pl_ = q_ / spx, ch_ = q_ % spx;
        if (chan_group > 0) {
            const int r_ = q_ % (chan_group * spx);
            ch_ = r_ / chan_group;
            pl_ = q_ / (chan_group * spx) * chan_group + r_ % chan_group;
        }
        const int pair_ = min((int)(blockIdx.x & 7) * per_xcd_ + pl_, n_pairs - 1);
        const unsigned *gm = xr.rowmask + (size_t)((pair_ / nb) * spx + ch_) * xr.mask_words;
        for (int i = threadIdx.x; i < xr.mask_words; i += 128)
            xmask[i] = gm[i];
    }
    __syncthreads();  // the only barrier: both waves see turn = 0 (and the ticket, and the flags)
    int s_ch, b, img;
    if (chained) {
        const int qx = (int)(blockIdx.x & 7), m_img = n_pairs / nb;
        const int items = ((m_img - qx + 7) >> 3) * spx;  // (image, channel) items of this queue
        const int t = ticket;
        if (items <= 0 || t >= items * nb)
            return;
        b = t / items;
        const int k = t - b * items;
        img = qx + 8 * (k / spx);
        s_ch = k % spx;
    } else {
        const int per_xcd = (n_pairs + 7) >> 3;
        const int q = (int)(blockIdx.x >> 3);
        // order of an XCD's (pair, channel) items: channel fastest (chan_group = 0: the channels of a
        // block run side by side and share its guide rows), or runs of chan_group pairs per channel
        // (a channel's neighbouring blocks stay neighbours in time, as on a grey batch)
        int pl = q / spx;
        s_ch = q % spx;
        if (chan_group > 0) {
            const int r = q % (chan_group * spx);
            s_ch = r / chan_group;
            pl = q / (chan_group * spx) * chan_group + r % chan_group;
        }
        const int pair = (int)(blockIdx.x & 7) * per_xcd + pl;
        if (pl >= per_xcd || pair >= n_pairs)
            return;
        b = pair % nb;
        img = pair / nb;
    }
    const bool grey3 = spx == 3 && colour[img] == 0;  // three equal channels: channel 0 stands for all
    if (grey3 && s_ch > 0)
        return;  // (chained: nobody waits for a skipped channel's blocks, its neighbours skip too)

    const int cc = lane & 15;                    // column role: lane = plane * 16 + column
    const int cp = lane / T, cl = lane - cp * T;  // chain role (lane < 4T): plane, row
    const bool chain = lane < 4 * T;
    const size_t npx = (size_t)h * w;
    const int np = 4 * spx;
    const float *abg = ab + ((size_t)img * np + 4 * s_ch) * npx;          // planes 4s .. 4s+3
    const double *stg = states + ((size_t)img * np + 4 * s_ch) * nb * h;  // their states
    const uint8_t *gimg = guide + (size_t)img * npx * 3;
    uint8_t *dimg = dst + (size_t)img * npx * spx;
    // a grey 3-channel image of an iterated call hands its result on as one byte per pixel
    uint8_t *cimg = (grey3 && compact != nullptr) ? compact + (size_t)img * npx : nullptr;
    // ... and a colour image as three planes, [img][3][h][w]: this channel's plane
    if (spx == 3 && !grey3 && compact3 != nullptr)
        cimg = compact3 + ((size_t)img * 3 + s_ch) * npx;
    // RowSum at output column o = 16 b + cc (cc >= 1):  + ext[o + 2r] - ext[o - 1], ext[i] = S[bi(i - r)]
    // per-lane BYTE offsets of the float4 pixel from the wave-uniform base abg of the channel's
    // plane group ([h][w][4] floats; 32 bits: the host admits images below 2^28 pixels here)
    const uint32_t oe = 16u * (uint32_t)border_interpolate(b * kSB + cc + R, w, RF_BORDER_REFLECT);
    const uint32_t ol =
        16u * (uint32_t)border_interpolate(b * kSB + cc - 1 - R, w, RF_BORDER_REFLECT);
    const char *abgb = reinterpret_cast<const char *>(abg);
    // start of the lane's row chain: the stored state at column 16 b (row-walk form), or - chained -
    // the sum at column 16 b - 1: for block 0 the head sums, else what block b - 1 published
    // (row-walk form: [plane][nb][h], the layout rf_gf_u8 passes as `lay` for it; exact rows: `lay`)
    const double *Ps = chained ? xc.head + ((size_t)img * np + 4 * s_ch + min(cp, 3)) * h
                       : EXACT ? stg + (size_t)min(cp, 3) * lay.sp + (size_t)b * lay.sb
                               : stg + ((size_t)min(cp, 3) * nb + b) * h;
    constexpr bool exact = EXACT;
    // (an interior block's 2q + 1 sums are consecutive doubles: fetched 16 bytes at a time)
    const bool xinner = b - XQ >= 0 && b + XQ < nb && lay.sb == 1;
    // exact rows: the 2q + 1 blocks b - q .. b + q whose sums a chain starts from (reflected like
    // columns; slot b - the middle one - holds the true state of a row the row walk had to take)
    int xkb[NF];
#pragma unroll
    for (int i = 0; i < NF; i++)
        xkb[i] = (border_interpolate(b - XQ + i, nb, RF_BORDER_REFLECT) - b) * lay.sb;
    const double scale = 1.0 / (double)(KS * KS);
    const int total = h + 2 * R;  // padded rows
    const int jmax = total - 1;
    int nsub;                     // sub-tiles that start inside the padded image
    {
        const int full = total / (2 * R), rem = total - full * 2 * R;
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < 2 * M; c++)
            cnt += G::start(c) < rem ? 1 : 0;
        nsub = full * 2 * M + cnt;
    }
    const uint32_t gbytes = (uint32_t)(npx * 3);
    // chained hand-off: sums in from block b - 1, out to block b + 1; nsub_all = sub-tile slots per block
    const int nsub_all = gf_chain_nsub(h, R);
    const size_t item_blk = ((size_t)img * spx + s_ch) * nb + b;
    const bool take = chained && b > 0, give = chained && b + 1 < nb;
    // (buffer descriptors over one block's slots each: 16-byte sc1 loads / stores, L1-bypassing and
    //  write-through; the base must be wave-uniform registers)
    const unsigned slot_bytes = (unsigned)nsub_all * 64u * 16u;
    auto block_rsrc = [&](size_t blk) __attribute__((always_inline)) {
        const unsigned long long a_ =
            (unsigned long long)(uintptr_t)xc.xst + (unsigned long long)blk * slot_bytes;
        const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a_);
        const unsigned hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a_ >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(
            (void *)(uintptr_t)(((unsigned long long)hi_ << 32) | lo_), 0, (int)slot_bytes, 0x00020000);
    };
    const auto rs_in = block_rsrc(take ? item_blk - 1 : item_blk);
    const auto rs_out = block_rsrc(item_blk);
    const unsigned epoch = chained ? xc.sync[9] : 0u;
    const unsigned sa_in = (epoch * 0x9e3779b1u) ^ ((unsigned)(item_blk - 1) * 0x7feb352du);
    const unsigned sb_in = (epoch * 0x632be5abu) + ((unsigned)(item_blk - 1) * 0x9e3779b1u);
    const unsigned sa_out = (epoch * 0x9e3779b1u) ^ ((unsigned)item_blk * 0x7feb352du);
    const unsigned sb_out = (epoch * 0x632be5abu) + ((unsigned)item_blk * 0x9e3779b1u);
    gf_u32x4 xw = {0u, 0u, 0u, 0u};  // the prefetched slot {low, tag0, high, tag1} of the next sub-tile

    const int r4 = lane >> 4;            // loader role: row of a group of four, column cc
    const int fr = lane >> 2, qd = lane & 3;  // flush role: row of the sub-tile, quad of columns
    float4 pe[NL], pl[NL];
    double pst = 0.0;
    double pxf[EXACT ? NF : 1];  // exact rows: the prefetched block sums (slot XQ: the state of a flagged row)
    int pxflag = 0;
    if constexpr (EXACT) {
#pragma unroll
        for (int i = 0; i < NF; i++)
            pxf[i] = 0.0;
    }
    uint32_t gpre[3];  // guide bytes of the lane's four output pixels
    double SUM = 0.0;
    double fifo[FL];

    // padded row -> image row (BORDER_REFLECT) of the sub-tile starting at padded row i0
    auto rowtab = [&](int slot, int i0) __attribute__((always_inline)) {
        if (lane < T) {
            const int row = border_interpolate(min(i0 + lane, jmax) - R, h, RF_BORDER_REFLECT);
            L.rowtab[slot][lane] = row;
            L.rowoff[slot][lane] = 16u * (uint32_t)row * (uint32_t)w;
        }
    };
    auto fetch = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            // (timing experiments of tools/gf_stamp_build.py, results WRONG: RF_GF_EXP_ALIAS makes
            //  the leaving operand the entering one - half the distinct cache lines -, RF_GF_EXP_HOT
            //  also takes every row from image row 0 - no memory traffic to speak of)
#if defined(RF_GF_EXP_HOT)
            const uint32_t ro = 0, ol_ = oe;
#elif defined(RF_GF_EXP_ALIAS)
            const uint32_t ro = L.rowoff[slot][min(4 * i + r4, T - 1)], ol_ = oe;
#else
            const uint32_t ro = L.rowoff[slot][min(4 * i + r4, T - 1)], ol_ = ol;
#endif
            pe[i] = *reinterpret_cast<const float4 *>(abgb + (oe + ro));
            pl[i] = *reinterpret_cast<const float4 *>(abgb + (ol_ + ro));
        }
        if (chain && !take) {
            const int row = L.rowtab[slot][cl];
            if constexpr (exact) {
                pxflag = (int)((xmask[row >> 5] >> (row & 31)) & 1u);
                const double *pr = Ps + (size_t)row * lay.sr;
                if (xinner) {
                    typedef double d2_t __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
                    for (int i = 0; i + 1 < NF; i += 2) {
                        const d2_t v = *reinterpret_cast<const d2_t *>(pr + (i - XQ));
                        pxf[i] = v.x;
                        pxf[i + 1] = v.y;
                    }
                    pxf[NF - 1] = pr[XQ];
                } else {
#pragma unroll
                    for (int i = 0; i < NF; i++)
                        pxf[i] = pr[xkb[i]];
                }
            } else {
                pst = Ps[row];
            }
        }
    };
    // chained, b > 0: request the left neighbour's slot of sub-tile jj (two self-validating words)
    auto take_issue = [&](int jj) __attribute__((always_inline)) {
        xw = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (jj * 64 + lane) * 16, 0, /*sc1*/ 16);
    };
    // ... and use it: poll until both tags are the ones block b - 1 writes for sub-tile jj
    auto take_sums = [&](int jj) __attribute__((always_inline)) -> double {
        const unsigned t0 = gf_chain_tag0(sa_in, (unsigned)jj, (unsigned)lane),
                       t1 = gf_chain_tag1(sb_in, (unsigned)jj, (unsigned)lane);
        int polls = 0;
        while (__builtin_amdgcn_ballot_w64(chain && (xw.y != t0 || xw.w != t1)) != 0ull) {
            if (++polls > (1 << 20)) {  // seconds: the neighbour is gone - flag it, do not hang
                if (lane == 0)
                    atomicOr(&xc.sync[8], 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(16);
            xw = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (jj * 64 + lane) * 16, 0, 16);
        }
        return __longlong_as_double((long long)(((unsigned long long)xw.z << 32) | xw.x));
    };
    auto give_sums = [&](int jj, double v) __attribute__((always_inline)) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        const gf_u32x4 o = {(unsigned)bits, gf_chain_tag0(sa_out, (unsigned)jj, (unsigned)lane),
                            (unsigned)(bits >> 32), gf_chain_tag1(sb_out, (unsigned)jj, (unsigned)lane)};
        __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, (jj * 64 + lane) * 16, 0, 16);
    };
    // guide bytes of the lane's output pixels: row y0 + fr, columns 16 b + 4 qd .. + 3 (12 bytes)
    auto guide_fetch = [&](int y0) __attribute__((always_inline)) {
        const int gy = min(y0 + fr, h - 1);
        const uint32_t off = ((uint32_t)gy * w + b * kSB + 4 * qd) * 3;
        gpre[0] = gpre[1] = gpre[2] = 0;
        if (off + 12 <= gbytes) {
            __builtin_memcpy(gpre, gimg + off, 12);
        } else {  // the last bytes of the image (or columns past its right edge)
            for (int q = 0; q < 12; q++)
                if (off + q < gbytes)
                    gpre[q >> 2] |= (uint32_t)gimg[off + q] << (8 * (q & 3));
        }
    };

    // position of this wave's current sub-tile: j = index, jp = j mod 2M, pbase = first padded row
    // of its period
    int j = wv, jp = wv % (2 * M), pbase = (wv / (2 * M)) * 2 * R;
    auto advance = [&](int &jj, int &jjp, int &pb) __attribute__((always_inline)) {
        jj += 2;
        jjp += 2;
        if (jjp >= 2 * M) {
            jjp -= 2 * M;
            pb += 2 * R;
        }
    };

    // the finished pixels of the previous sub-tile: their stores are issued at the top of the NEXT
    // iteration, after the wait for its operands - vmcnt retires in order, so stores issued last in
    // an iteration would put their whole latency in front of that wait
    uint32_t o4 = 0;           // the lane's four result bytes (columns 16 b + 4 qd .. + 3 of row fr)
    int st_y0 = 0, st_tj = 0;  // pending stores: rows st_y0 .. st_y0 + st_tj - 1 (st_tj = 0: none)
    auto store_pending = [&]() __attribute__((always_inline)) {
        const int y = st_y0 + fr, x = b * kSB + 4 * qd;
        if (fr >= st_tj || y >= h || x >= w)
            return;
        const uint32_t pix = (uint32_t)y * w + x;
        if (x + 3 < w && (grey3 || spx == 1 || cimg != nullptr)) {
            if (cimg != nullptr) {
                __builtin_memcpy(cimg + pix, &o4, 4);
            } else if (grey3) {  // every result byte three times: 12 contiguous bytes
                const uint32_t b0 = o4 & 0xff, b1 = (o4 >> 8) & 0xff, b2 = (o4 >> 16) & 0xff,
                               b3 = o4 >> 24;
                const uint32_t d[3] = {b0 * 0x010101u | (b1 << 24), b1 * 0x0101u | (b2 * 0x0101u << 16),
                                       b2 | (b3 * 0x010101u << 8)};
                __builtin_memcpy(dimg + (size_t)pix * 3, d, 12);
            } else {
                __builtin_memcpy(dimg + pix, &o4, 4);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (x + i >= w)
                break;
            const uint8_t v = (uint8_t)(o4 >> (8 * i));
            if (cimg != nullptr) {
                cimg[pix + i] = v;
            } else if (grey3) {
                dimg[(size_t)(pix + i) * 3 + 0] = v;
                dimg[(size_t)(pix + i) * 3 + 1] = v;
                dimg[(size_t)(pix + i) * 3 + 2] = v;
            } else {
                dimg[(size_t)(pix + i) * spx + s_ch] = v;
            }
        }
    };
#ifdef RF_GF_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
    unsigned long long st_n = 0;
#endif

    if (j < nsub) {
        rowtab(0, pbase + G::start(jp));
        wave_lds_fence();
        fetch(0);
        if (take)
            take_issue(j);
    }
    int slot = 0;
    for (; j < nsub;) {
        const int i0 = pbase + G::start(jp);
        const int tj = G::size(jp);
        const bool fill = j < 2 * M;  // first period: padded rows 0 .. 2R-1 only fill the window
        int jn = j, jpn = jp, pbn = pbase;
        advance(jn, jpn, pbn);
        rowtab(slot ^ 1, pbn + G::start(jpn));
        RF_STAMP(0);
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int r = 4 * i + r4;  // row of the sub-tile this lane fetched (rows >= T: repeats)
            if (4 * i + 3 < T || r < T) {
                L.u.st.E[r][cc] = pe[i].x;
                L.u.st.E[T + r][cc] = pe[i].y;
                L.u.st.E[2 * T + r][cc] = pe[i].z;
                L.u.st.E[3 * T + r][cc] = pe[i].w;
                L.u.st.L[r][cc] = pl[i].x;
                L.u.st.L[T + r][cc] = pl[i].y;
                L.u.st.L[2 * T + r][cc] = pl[i].z;
                L.u.st.L[3 * T + r][cc] = pl[i].w;
            }
        }
        double s = take ? take_sums(j) : pst;
        double xfc[EXACT ? NF : 1];
        if constexpr (EXACT) {
#pragma unroll
            for (int i = 0; i < NF; i++)
                xfc[i] = pxf[i];
        }
        const int xflg = pxflag;
        RF_STAMP(1);  // wait for the operands (and every older store)
        store_pending();
        st_tj = 0;
        wave_lds_fence();
        if (!fill)
            guide_fetch(i0 - 2 * R);
        if (jn < nsub) {
            fetch(slot ^ 1);
            if (take)
                take_issue(jn);
        }
        RF_STAMP(2);
        if (chain) {
            // all operand differences first (independent LDS reads and conversions), then the
            // chain of dependent adds
            double d[kSB];
#pragma unroll
            for (int c = 1; c < kSB; c++)
                d[c] = (double)L.u.st.E[lane][c] - (double)L.u.st.L[lane][c];
            if constexpr (exact) {
                // the window sum at column 16 b from block sums, every intermediate a sum over a
                // subset of that window: prefix of block b + q, the 2q blocks in front of it, the
                // suffix of block b - q - 1
                double se = xfc[NF - 1];
#pragma unroll
                for (int c = 1; c <= 15 - XC0; c++)
                    se -= (double)L.u.st.E[lane][c];
#pragma unroll
                for (int i = NF - 2; i >= 0; i--)
                    se += xfc[i];
#pragma unroll
                for (int c = 1; c <= XC0; c++)
                    se += (double)L.u.st.L[lane][c];
                s = xflg ? xfc[XQ] : se;
            }
            if (chained) {
                // s is the sum at column 16 b - 1: one more step to column 16 b.  At the row's
                // first output column nothing leaves the window yet (block 0: s = sum of ext[0 .. 2R-1])
                const double d0 = (double)L.u.st.E[lane][0] - (b == 0 ? 0.0 : (double)L.u.st.L[lane][0]);
                s += d0;
            }
            L.Rt[cp * kSB][cl] = s;
#pragma unroll
            for (int c = 1; c < kSB; c++) {
                s += d[c];
                L.Rt[cp * kSB + c][cl] = s;
            }
            if (give)
                give_sums(j, s);
        }
        wave_lds_fence();
        RF_STAMP(3);
        // --- column phase: wait for the SUM of sub-tile j - 1 (the other wave's)
        if (j > 0) {
            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(
                       &turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < j)
                __builtin_amdgcn_s_sleep(1);
            wave_lds_fence();
            SUM = __hip_atomic_load(&sumx[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        RF_STAMP(4);
        static_for<0, 2 * M>([&](auto cpos) __attribute__((always_inline)) {
            constexpr int C = decltype(cpos)::value;
            constexpr int TJ = G::size(C), OFF = G::fifo_off(C);
            if (jp == C) {
                double v[TJ];
#pragma unroll
                for (int q = 0; q < TJ; q++)
                    v[q] = L.Rt[lane][q];
                if (fill) {
#pragma unroll
                    for (int q = 0; q < TJ; q++) {
                        SUM += v[q];
                        fifo[OFF + q] = v[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < TJ; q++) {
                        const double s0 = SUM + v[q];
                        L.u.xch[q][lane] = (float)(s0 * scale);
                        SUM = s0 - fifo[OFF + q];
                        fifo[OFF + q] = v[q];
                    }
                }
            }
        });
        __hip_atomic_store(&sumx[lane], SUM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        wave_lds_fence();
        if (lane == 0)
            __hip_atomic_store(&turn, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        RF_STAMP(5);
        // --- flush: the bytes of the finished pixels (stored at the top of the next iteration)
        if (!fill) {
            // the lane's four pixels: means as 16-byte LDS reads, guide bytes from registers
            const int frc = min(fr, T - 1);
            const float4 a0 = *reinterpret_cast<const float4 *>(&L.u.xch[frc][4 * qd]);
            const float4 a1 = *reinterpret_cast<const float4 *>(&L.u.xch[frc][16 + 4 * qd]);
            const float4 a2 = *reinterpret_cast<const float4 *>(&L.u.xch[frc][32 + 4 * qd]);
            const float4 bt = *reinterpret_cast<const float4 *>(&L.u.xch[frc][48 + 4 * qd]);
            const float A0[4] = {a0.x, a0.y, a0.z, a0.w}, A1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float A2[4] = {a2.x, a2.y, a2.z, a2.w}, BT[4] = {bt.x, bt.y, bt.z, bt.w};
            o4 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float i0g = (float)((gpre[(3 * i) >> 2] >> (8 * ((3 * i) & 3))) & 0xff);
                const float i1g = (float)((gpre[(3 * i + 1) >> 2] >> (8 * ((3 * i + 1) & 3))) & 0xff);
                const float i2g = (float)((gpre[(3 * i + 2) >> 2] >> (8 * ((3 * i + 2) & 3))) & 0xff);
                float q = BT[i];
                q = __fadd_rn(q, __fmul_rn(A0[i], i0g));
                q = __fadd_rn(q, __fmul_rn(A1[i], i1g));
                q = __fadd_rn(q, __fmul_rn(A2[i], i2g));
                o4 |= (uint32_t)saturate_u8(q) << (8 * i);
            }
            st_y0 = i0 - 2 * R;
            st_tj = tj;
            wave_lds_fence();
        }
        RF_STAMP(6);
#ifdef RF_GF_STAMP
        st_n++;
#endif
        j = jn;
        jp = jpn;
        pbase = pbn;
        slot ^= 1;
    }
    store_pending();
#ifdef RF_GF_STAMP
    if (lane == 0) {
        for (int i = 0; i < 7; i++)
            atomicAdd(&g_cw_stamps[i], st_acc[i]);
        atomicAdd(&g_cw_stamps[15], st_n);
    }
#endif
}

// Chained column walk, block 0: head[img * np + plane][row] = ext[0] + ... + ext[2R-1] added in this
// order from 0.0 (RowSum<float,double>'s first 2R steps; ext[i] = S[bi(i - R)]), i.e. the running sum
// one step before the row's first output.  lane = row, the four planes of a src channel together
// (float4 pixels).  grid (ceil(h / 64), images x src channels).  Thread 0 of the launch also opens
// the launch's ticket queues and advances the epoch of the hand-off tags (sync: see GfChain).
template <int kHeaderOnly = 0>  // (a template only so that every unit including this header may hold it)
__global__ __launch_bounds__(64) void gf_rowhead_kernel(const float *__restrict__ ab,
                                                        double *__restrict__ head, int h, int w,
                                                        int radius, int spx,
                                                        const int *__restrict__ colour,
                                                        unsigned *__restrict__ sync)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 8; q++)
            sync[q] = 0u;  // (word 8, the error flag, stays: it is sticky until the caller clears it)
        sync[9] = sync[9] + 1u;
    }
    const int img = blockIdx.y / spx, ch = blockIdx.y - img * spx;
    if (spx == 3 && ch > 0 && colour[img] == 0)
        return;  // grey 3-channel image: channel 0 stands for all
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= h)
        return;
    const int np = 4 * spx;
    const float4 *S4 = reinterpret_cast<const float4 *>(ab + ((size_t)img * np + 4 * ch) * h * w) +
                       (size_t)row * w;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = 0; i < 2 * radius; i++) {
        const float4 v = S4[border_interpolate(i - radius, w, RF_BORDER_REFLECT)];
        s0 += (double)v.x;
        s1 += (double)v.y;
        s2 += (double)v.z;
        s3 += (double)v.w;
    }
    double *H = head + ((size_t)img * np + 4 * ch) * h + row;
    H[0] = s0;
    H[(size_t)h] = s1;
    H[(size_t)2 * h] = s2;
    H[(size_t)3 * h] = s3;
}

// Launches of the fused stage 2 for one radius (defined per radius in rf_gf_fused_*.hip).
struct GfFusedArgs {
    const float *ab;
    double *states;
    const uint8_t *guide;
    uint8_t *dst;
    int m, h, w, nb, src_cn;
    const int *colour;
    hipStream_t stream;
    GfChain chain;  // xst == nullptr: row-walk form (gf_rowstate_kernel + stored states)
    int exp_skip;  // timing experiments only (debug option "gf_exp_skip"): bit 1 no row states, bit 2 no column walk
    uint8_t *compact;  // not the last pass of an iterated call: grey 3-channel images go here, 1 B per pixel
    GfStateLayout lay;  // where a (plane, block, row) state / block sum sits
    GfExact xr;         // exact rows: flags and list of the rows that take the row walk (on = 0: all do)
    uint8_t *compact3;  // not the last pass of an iterated call: colour images go here as three planes
    int chan_group;     // column walk: pairs per channel run in an XCD's item order (0: channel fastest)
};
typedef void (*GfFusedLaunch)(const GfFusedArgs &);
GfFusedLaunch gf_fused_launcher(int radius);  // nullptr outside 1 .. kGfFusedMaxRadius

template <int R>
void gf_fused_launch(const GfFusedArgs &a)
{
    const int row_blocks = (a.h + kBRows - 1) / kBRows;
    const int np = 4 * a.src_cn;
    const int pairs = a.m * a.nb;
    if (a.chain.xst != nullptr) {
        // chained column walk: head sums (which also resets the tickets and advances the epoch),
        // then 8 queues of ceil(m / 8) * src_cn * nb tickets each
        const int hb = (a.h + 63) / 64;
        hipLaunchKernelGGL(gf_rowhead_kernel<0>, dim3((unsigned)hb, (unsigned)(a.m * a.src_cn)), dim3(64), 0,
                           a.stream, a.ab, const_cast<double *>(a.chain.head), a.h, a.w, R, a.src_cn,
                           a.colour, a.chain.sync);
        if constexpr (gf_chained_radius(R)) {
            if (!(a.exp_skip & 4))
                hipLaunchKernelGGL((gf_colwalk_kernel<R, true>),
                                   dim3(8u * (unsigned)((a.m + 7) / 8) * a.src_cn * a.nb), dim3(128),
                                   0, a.stream, a.ab, a.states, a.guide, a.dst, a.h, a.w, a.nb, pairs,
                                   a.src_cn, a.colour, a.chain, a.compact, a.lay, GfExact{nullptr, nullptr, nullptr, 0, 0},
                                   a.compact3, 0);
        }
        return;
    }
    const int streaming = (size_t)a.m * np * a.h * a.w * sizeof(float) > ((size_t)256 << 20) ? 1 : 0;
    if (!(a.exp_skip & 2))
        hipLaunchKernelGGL((gf_rowstate_kernel<R>), dim3((unsigned)(a.m * a.src_cn * row_blocks)),
                           dim3(256), 0, a.stream, a.ab, a.states, a.h, a.w, row_blocks, np, a.colour,
                           np, a.nb, streaming, a.lay, a.xr);
    if (a.exp_skip & 4)
        return;
    // items per XCD: its pairs x channels, rounded up to whole channel runs
    const int per_xcd = (pairs + 7) / 8;
    const int cgrp = a.src_cn == 1 ? 0 : std::min(a.chan_group, per_xcd);
    const unsigned cw_grid =
        8u * (unsigned)(cgrp > 0 ? (per_xcd + cgrp - 1) / cgrp * cgrp : per_xcd) * a.src_cn;
    if constexpr (gf_exact_radius(R)) {
        if (a.xr.on) {
            hipLaunchKernelGGL((gf_colwalk_kernel<R, false, true>),
                               dim3(cw_grid), dim3(128), 0, a.stream,
                               a.ab, a.states, a.guide, a.dst, a.h, a.w, a.nb, pairs, a.src_cn, a.colour,
                               GfChain{nullptr, nullptr, nullptr}, a.compact, a.lay, a.xr, a.compact3, cgrp);
            return;
        }
    }
    hipLaunchKernelGGL((gf_colwalk_kernel<R>), dim3(cw_grid),
                       dim3(128), 0, a.stream, a.ab, a.states, a.guide, a.dst, a.h, a.w, a.nb, pairs,
                       a.src_cn, a.colour, GfChain{nullptr, nullptr, nullptr}, a.compact, a.lay, a.xr,
                       a.compact3, cgrp);
}

}  // namespace rf
