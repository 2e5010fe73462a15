// rf_jbf.hip -- joint bilateral filter, uint8, for gfx950 (MI355X).
//
// Replaces cv2.ximgproc.jointBilateralFilter as called at
// /root/reference/filter_reflectance.py:60-64.  Arithmetic contract (DESIGN.md "JBF"):
// per output pixel the taps of the radius-r disk are visited row-major, the weight is
// spaceW[k] * colorLUT[L1(joint0, jointTap)] in float32, and sum[c] += weight * src[c] is a
// separately rounded multiply then add -- the order of the 8u path of
// opencv_contrib/modules/ximgproc/src/joint_bilateral_filter.cpp on a non-FMA build.
//
// Kernels (selection in rf_jbf_u8 at the end of the file):
//   jbf_tile64_kernel  default for radius <= 52: one workgroup = 64x64 output tile (32x128,
//                      16x256 or 128x32 for the image's remainder rows / columns), 1024 threads
//                      (4 waves/SIMD), LDS-staged texel tile, LUT at the end of LDS.
//   jbf_slab_kernel    radius 53..468: the same 64x64 outputs with the disk's tap rows taken in slabs
//                      (row pitch 208 .. 1008; the grey loop; a colour src one pass per channel).
//   jbf_tiled2_kernel  64 x TH tiles with 8-byte texels and a clamped/full LUT: used when the
//                      LDS out-of-range probe fails, and by the tuning harness.
//   jbf_generic_kernel untiled, any radius, global-memory gathers (fallback + cross-check).
//   jbf_f32_kernel     the CV_32F variant (rf_jbf_f32), untiled.
// Parameter tables (colour LUT, tap tables): one device arena per parameter set, uploaded once when
// the set is first seen (get_tables); entries a captured graph points into are pinned.
// Shared pieces: jbf_tap_loop (the software-pipelined tap loop, compiler-scheduled VALU) and
// jbf_tap_loop_grey4 (its hand-interleaved form for grey tiles; J1 = single-channel joint whose
// pre-scaled texels make v_sad_u32 produce the gather address).
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "rf_common.hpp"

namespace rf {
namespace {

constexpr int kTileW = 64;
constexpr int kPix = 4;        // outputs per lane (horizontal)
constexpr int kMaxLds = 160 * 1024;
constexpr int kTlw2 = 144;     // tile row pitch in texels of jbf_tiled2_kernel (radius <= 36)
constexpr int kJbfMaxTiledR4 = 468;  // radius (rounded up to 4) the slab kernel's widest row pitch (1008) holds
// private flag bits above the public RF_JBF_* ones: the test / benchmark switches of
// rf_debug_option() as the kernels see them
constexpr int kJbfStageOnly = 0x1000, kJbfCompilerLoop = 0x2000, kJbfTile64Only = 0x4000;
constexpr int kJbfLookahead1 = 0x8000;  // grey asm loop with its gathers one column step ahead (round-4 form)

typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

struct JbfTables {
    int device = -1;
    int radius = 0;
    int joint_cn = 0;
    double sigma_color = 0, sigma_space = 0;
    int maxk = 0;
    int lut_len = 0;   // entries kept: indices >= lut_len-1 are clamped (LUT value exactly 0)
    float *d_lut = nullptr;     // [256*joint_cn]
    int *d_di = nullptr;        // [maxk]
    int *d_dj = nullptr;        // [maxk]
    float *d_sw = nullptr;      // [maxk]
    int *d_hw = nullptr;        // [2r+1] half-width of the disk on tap row i
    // weight rows |i| = 0..r, each sw_len = 2*(r4+8) floats, centre at index r4+8, zeros outside
    // the disk (the weights are symmetric in i and in j)
    int r4 = 0, sw_len = 0;
    float *d_swsym = nullptr;
    std::shared_ptr<void> keep;  // JbfTableOwner of the arrays above
};

// While alive, this thread may make the "unsafe" runtime calls (allocation, creation of events,
// synchronisation of OTHER streams) although one of its streams is capturing: the first use of a
// parameter set inside a graph capture allocates its tables (hipStreamCaptureModeRelaxed for this
// thread only; the mode is put back on the way out).
struct CaptureRelax {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    bool ok;
    CaptureRelax() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; }
    ~CaptureRelax()
    {
        if (ok)
            (void)hipThreadExchangeStreamCaptureMode(&mode);
    }
};

// Owns the arrays of one cache entry: ONE device arena, uploaded once when the entry is built - on a
// stream of its own, waited for there, before the entry is published (the one host wait of a
// parameter set's first use, tens of microseconds; every later call only enqueues kernels).  The
// caller's stream may be capturing meanwhile: nothing of the upload enters its graph.
// rf_jbf_u8 keeps a reference for the duration of the call, so an eviction (or rf_shutdown) on
// another thread cannot free tables that a call has looked up but not launched yet; the last
// reference frees them on their own device (hipFree waits for the work queued there).  A graph that
// captured a call bakes in pointers into the arena: an entry that was ever looked up on a capturing
// stream is PINNED - never evicted, freed by rf_shutdown only (destroy such graphs
// before rf_shutdown).
struct JbfTableOwner {
    int device = 0;
    void *d_arena = nullptr;
    size_t bytes = 0;
    std::atomic<bool> pinned{false};  // referenced by a captured graph
    ~JbfTableOwner()
    {
        int cur = 0;
        const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device &&
                              hipSetDevice(device) == hipSuccess;
        if (d_arena)
            (void)hipFree(d_arena);
        if (switched)
            (void)hipSetDevice(cur);
    }
};

std::mutex g_mu;
std::vector<JbfTables> g_tables;
// Owners that left the cache (evicted entries, entries a finishing call held the last reference
// to): their arrays are freed by the next call whose stream is NOT capturing - hipFree inside a
// capture invalidates it - or by rf_shutdown.
std::vector<std::shared_ptr<void>> g_retired;

bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

void drain_retired(hipStream_t stream)
{
    if (stream_is_capturing(stream))
        return;
    std::vector<std::shared_ptr<void>> bin;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        bin.swap(g_retired);
    }
    if (!bin.empty()) {
        CaptureRelax relax;  // (another thread of the process may be capturing in the global mode)
        bin.clear();
    }
}

// A call's reference to its tables: if it turns out to be the last one (the entry was evicted while
// the call was being enqueued), the arrays go to g_retired instead of being freed under the call.
struct TablesHold {
    JbfTables &t;
    ~TablesHold()
    {
        if (t.keep && t.keep.use_count() == 1) {
            std::lock_guard<std::mutex> lock(g_mu);
            g_retired.push_back(std::move(t.keep));
        }
    }
};

// The entry is about to be used by work enqueued on `stream`: if that stream is capturing, the graph
// will hold pointers into the entry's arena for as long as it lives - pin the entry.
void note_use_on(const JbfTables &t, hipStream_t stream)
{
    if (stream_is_capturing(stream))
        static_cast<JbfTableOwner *>(t.keep.get())->pinned.store(true, std::memory_order_relaxed);
}

// Host-side parameter tables, computed in double exactly like jointBilateralFilter_8u does.
int get_tables(int radius, int joint_cn, double sigma_color, double sigma_space, hipStream_t stream,
               JbfTables *out)
{
    int dev = 0;
    RF_HIP_CHECK(hipGetDevice(&dev));
    drain_retired(stream);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (const JbfTables &t : g_tables)
            if (t.device == dev && t.radius == radius && t.joint_cn == joint_cn &&
                t.sigma_color == sigma_color && t.sigma_space == sigma_space) {
                *out = t;
                break;
            }
    }
    if (out->keep) {
        note_use_on(*out, stream);
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
        }
    t.maxk = (int)di.size();
    t.r4 = (radius + 3) & ~3;
    t.sw_len = 2 * (t.r4 + 8);
    std::vector<float> swsym((size_t)(radius + 1) * t.sw_len, 0.0f);
    for (size_t k = 0; k < di.size(); k++)
        if (di[k] >= 0)
            swsym[(size_t)di[k] * t.sw_len + (t.r4 + 8) + dj[k]] = sw[k];
    // one arena: [swsym][lut][di][dj][sw][hw], every part 256-byte aligned
    const size_t part[6] = {sizeof(float) * swsym.size(), sizeof(float) * (size_t)nlut,
                            sizeof(int) * (size_t)t.maxk, sizeof(int) * (size_t)t.maxk,
                            sizeof(float) * (size_t)t.maxk, sizeof(int) * (size_t)d};
    const void *from[6] = {swsym.data(), lut.data(), di.data(), dj.data(), sw.data(), hw.data()};
    size_t off[6], total = 0;
    for (int k = 0; k < 6; k++) {
        off[k] = total;
        total += (part[k] + 255) & ~(size_t)255;
    }
    auto owner = std::make_shared<JbfTableOwner>();
    owner->device = dev;
    owner->bytes = total;
    t.keep = owner;
    {
        // allocation, upload on a private stream and the wait for it: "unsafe" calls while the
        // caller's stream may be capturing - admitted for this thread by CaptureRelax; the tables
        // are resident before anybody can find the entry
        std::vector<char> image(total, 0);
        for (int k = 0; k < 6; k++)
            std::memcpy(image.data() + off[k], from[k], part[k]);
        CaptureRelax relax;
        hipStream_t ps = nullptr;
        RF_HIP_CHECK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
        struct StreamGuard {
            hipStream_t s;
            ~StreamGuard() { (void)hipStreamDestroy(s); }
        } guard{ps};
        RF_HIP_CHECK(hipMalloc(&owner->d_arena, total));
        RF_HIP_CHECK(hipMemcpyAsync(owner->d_arena, image.data(), total, hipMemcpyHostToDevice, ps));
        RF_HIP_CHECK(hipStreamSynchronize(ps));
    }
    char *db = static_cast<char *>(owner->d_arena);
    t.d_swsym = reinterpret_cast<float *>(db + off[0]);
    t.d_lut = reinterpret_cast<float *>(db + off[1]);
    t.d_di = reinterpret_cast<int *>(db + off[2]);
    t.d_dj = reinterpret_cast<int *>(db + off[3]);
    t.d_sw = reinterpret_cast<float *>(db + off[4]);
    t.d_hw = reinterpret_cast<int *>(db + off[5]);
    {
        std::lock_guard<std::mutex> lock(g_mu);
        // (another thread may have built the same entry meanwhile: use that one, ours is freed)
        bool found = false;
        for (const JbfTables &e : g_tables)
            if (e.device == dev && e.radius == radius && e.joint_cn == joint_cn &&
                e.sigma_color == sigma_color && e.sigma_space == sigma_space) {
                g_retired.push_back(std::move(t.keep));  // ours: freed later, outside any capture
                t = e;
                found = true;
                break;
            }
        if (!found) {
            // bounded cache (parameter sweeps must not accumulate device memory): drop the oldest
            // entry that no captured graph refers to, of this device if there is one; its arrays are
            // freed when the last call using them returns.  Pinned entries stay: a cache of nothing
            // but pinned entries grows.
            if (g_tables.size() >= 64) {
                size_t victim = g_tables.size();
                for (size_t i = 0; i < g_tables.size(); i++) {
                    if (static_cast<JbfTableOwner *>(g_tables[i].keep.get())->pinned.load(
                            std::memory_order_relaxed))
                        continue;
                    if (victim == g_tables.size())
                        victim = i;
                    if (g_tables[i].device == dev) {
                        victim = i;
                        break;
                    }
                }
                if (victim != g_tables.size()) {
                    g_retired.push_back(std::move(g_tables[victim].keep));
                    g_tables.erase(g_tables.begin() + victim);
                }
            }
            g_tables.push_back(t);
        }
    }
    *out = t;
    note_use_on(*out, stream);
    return RF_OK;
}

// Packs up to 3 interleaved bytes into the low bytes of a dword (byte 3 = 0), so that
// v_sad_u8 on two such dwords is the L1 colour distance.
// cn = -1: single-channel image treated as three equal channels (RF_JBF_GREY_AS_BGR).
__device__ inline uint32_t load_packed(const uint8_t *img, size_t pix, int cn)
{
    if (cn < 0)
        return (uint32_t)img[pix] * 0x010101u;
    const uint8_t *p = img + pix * cn;
    uint32_t v = p[0];
    if (cn == 3)
        v |= ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    return v;
}

// Four consecutive pixels (cn interleaved bytes each, any alignment) -> four packed dwords.
__device__ inline void load_packed4(const uint8_t *img, size_t pix, int cn, uint32_t (&out)[4])
{
    const uint8_t *p = img + pix * (cn < 0 ? 1 : cn);
    if (cn == 3) {
        uint32_t d0, d1, d2;  // one (unaligned) 12-byte load
        __builtin_memcpy(&d0, p, 4);
        __builtin_memcpy(&d1, p + 4, 4);
        __builtin_memcpy(&d2, p + 8, 4);
        out[0] = d0 & 0x00ffffffu;
        out[1] = (d0 >> 24) | ((d1 & 0xffffu) << 8);
        out[2] = (d1 >> 16) | ((d2 & 0xffu) << 16);
        out[3] = d2 >> 8;
    } else {
        uint32_t d0;
        __builtin_memcpy(&d0, p, 4);
        const uint32_t rep = cn < 0 ? 0x010101u : 1u;
        out[0] = (d0 & 0xffu) * rep;
        out[1] = ((d0 >> 8) & 0xffu) * rep;
        out[2] = ((d0 >> 16) & 0xffu) * rep;
        out[3] = (d0 >> 24) * rep;
    }
}

// Packed joint/src values of the four tile columns X = 4k .. 4k+3 of one tile row (image row
// gy >= 0 already border-interpolated, or gy < 0 = outside under BORDER_CONSTANT).  Interior
// columns take the 12-byte path, columns that need border handling go pixel by pixel.
__device__ inline void load_tile_quad(const uint8_t *joint, const uint8_t *src, size_t img, int gy,
                                      int x_first, int w, int jcn, int scn, int border,
                                      uint32_t (&jv)[4], uint32_t (&sv)[4])
{
    if (gy < 0) {
#pragma unroll
        for (int u = 0; u < 4; u++)
            jv[u] = sv[u] = 0u;
        return;
    }
    const size_t row = img + (size_t)gy * w;
    if (x_first >= 0 && x_first + 3 < w) {
        load_packed4(joint, row + x_first, jcn, jv);
        load_packed4(src, row + x_first, scn, sv);
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int gx = border_interpolate(x_first + u, w, border);
        jv[u] = gx < 0 ? 0u : load_packed(joint, row + gx, jcn);
        sv[u] = gx < 0 ? 0u : load_packed(src, row + gx, scn);
    }
}

// Workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share an L2).  This maps the
// launch index to a tile index such that every XCD works through one contiguous range of tiles,
// so the halo a tile shares with its neighbours is served by that XCD's own L2.  Speed only.
__device__ inline int xcd_contiguous_tile(int b, int nblocks)
{
    const int per = nblocks >> 3;
    return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
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

__device__ inline float finish_value(float sum, float wsum_or_inv, int flags)
{
    return (flags & RF_JBF_TRUE_DIVISION) ? __fdiv_rn(sum, wsum_or_inv) : __fmul_rn(sum, wsum_or_inv);
}

// Stores a lane's 4 horizontally adjacent outputs.  sums[p][c]; NCH_IN = 1 replicates the single
// accumulated channel.  Interior quads of 3-channel images go out as one 12-byte store.
template <int NCH_IN, int SCN>
__device__ inline void store_quad(uint8_t *dst, size_t img, int oy, int ox0, int h, int w,
                                  const float (&sum)[kPix][NCH_IN], const float (&wsum)[kPix],
                                  int flags)
{
    if (oy >= h || ox0 >= w)
        return;
    uint8_t px[kPix][3];
#pragma unroll
    for (int p = 0; p < kPix; p++) {
        const float d = (flags & RF_JBF_TRUE_DIVISION) ? wsum[p] : __fdiv_rn(1.0f, wsum[p]);
#pragma unroll
        for (int c = 0; c < SCN; c++)
            px[p][c] = saturate_u8(finish_value(sum[p][NCH_IN == 1 ? 0 : c], d, flags));
    }
    uint8_t *o = dst + (img + (size_t)oy * w + ox0) * SCN;
    if (SCN == 3 && ox0 + 3 < w) {
        uint32_t d[3];
        d[0] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | ((uint32_t)px[1][0] << 24);
        d[1] = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | ((uint32_t)px[2][1] << 24);
        d[2] = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | ((uint32_t)px[3][2] << 24);
        __builtin_memcpy(o, d, 12);
    } else if (SCN == 1 && ox0 + 3 < w) {
        const uint32_t d = px[0][0] | (px[1][0] << 8) | (px[2][0] << 16) | ((uint32_t)px[3][0] << 24);
        __builtin_memcpy(o, &d, 4);
    } else {
#pragma unroll
        for (int p = 0; p < kPix; p++)
            if (ox0 + p < w)
#pragma unroll
                for (int c = 0; c < SCN; c++)
                    o[p * SCN + c] = px[p][c];
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
// Tiled kernels: the software-pipelined tap loop.
//
// One workgroup stages its output tile plus halo into LDS (border handling happens there, so the
// tap loop is branch-free); each lane owns 4 horizontally adjacent outputs and slides over the tap
// row, so every texel read and its byte->float conversions feed 4 outputs.  In detail:
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
// jbf_tiled2_kernel LDS: [flag][lutrep lut_len*LUTREP f32][sw (r+1)*sw_len f32][tile (TH+2r) x TLW]
// Tile column X <-> image x = tile_x0 - r4 + X, stored at (X&3)*(TLW/4) + (X>>2): the lane that
// owns outputs 4t..4t+3 reads X = 4t + const, i.e. consecutive lanes read consecutive texels
// (conflict-free) although each lane's own outputs are adjacent; TLW % 32 == 16 keeps the two
// 16-lane rows of a 32-lane LDS group on disjoint banks.
// ------------------------------------------------------------------------------------------
// Block-wide AND of a predicate through one word of the caller's dynamic LDS (HIP's
// __syncthreads_and brings 256 bytes of static LDS with it, which the kernels below cannot spare
// and which would move the end of the allocation the LUT is aligned to).  `word` must have been
// set to 1 before an earlier barrier.  Contains a barrier.
__device__ inline int block_all(int pred, volatile int *word)
{
    if (!pred)
        *word = 0;
    __syncthreads();
    return *word;
}

// Two predicates at once: bit 0 / bit 1 of the result are set iff every thread passed pred0 /
// pred1.  `word` must have been set to 3 before an earlier barrier.
__device__ inline int block_all2(int pred0, int pred1, int *word)
{
    const int keep = (pred0 ? 1 : 0) | (pred1 ? 2 : 0);
    if (keep != 3)
        atomicAnd(word, keep);
    __syncthreads();
    return *reinterpret_cast<volatile int *>(word);
}

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
#define RF_LDS_READ_B32_OFF(dst, addr, off) \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RF_LDS_READ_U16_OFF(dst, addr, off) \
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

// Accumulates all taps of one lane's 4 outputs.  NCH = channels accumulated (3, or 1 when the
// src is single-channel or every src texel of the tile is grey: identical bits, a third of the
// multiply-adds).  TB = bytes per LDS texel: 8 = {BGRx joint, BGRx src}; 4 = {B,G,R joint, grey
// src} (NCH = 1 only).  CLAMP = clamp the LUT index with v_min (otherwise the caller guarantees
// that every reachable index is either inside the staged table or beyond the end of the
// workgroup's LDS allocation, where ds_read returns 0).  sum/wsum must be zero on entry.
template <int NCH, int LUTREP, bool CLAMP, int TLW, int TB>
__device__ __forceinline__ void jbf_tap_loop(uint32_t lut_lane_addr, uint32_t sw_addr0,
                                             uint32_t tile_lane_addr, uint32_t plane_b_lane_addr,
                                             const uint32_t (&jc)[kPix], uint32_t amax, int ty,
                                             int radius, int r4, int sw_len,
                                             const int *__restrict__ hwtab, float (&sum)[kPix][NCH],
                                             float (&wsum)[kPix])
{
    static_assert((TB == 8) || (TB == 4 && NCH == 1) || (TB == 6 && NCH == 3),
                  "4-byte texels carry one src channel, 6-byte ones three");
    constexpr int TA = TB == 6 ? 4 : TB;  // bytes per texel in the main plane
    constexpr int Q4 = TLW / 4;
    using texel_t = typename std::conditional<TB == 8, uint2v, uint32_t>::type;
    auto joint_of = [](const texel_t &t) -> uint32_t {
        if constexpr (TB == 8)
            return t.x;
        else
            return t & 0x00ffffffu;
    };
    // all four addresses first, then the four reads back to back: LDS instructions issued in
    // a cluster disturb the VALU stream less than reads interleaved with their address math
    // (+2.8 % on the hand-scheduled grey loop, in-process A/B)
    auto issue_gathers = [&](uint32_t jtex, float *g) {
        uint32_t a[kPix];
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            uint32_t alpha = __builtin_amdgcn_sad_u8(jtex, jc[p], 0u);
            if (CLAMP)
                alpha = min(alpha, amax);
            a[p] = alpha * (LUTREP * 4u) + lut_lane_addr;
        }
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
        for (int p = 0; p < kPix; p++)
            RF_LDS_READ_B32(g[p], a[p]);
    };
    // TB == 6: second plane of 2-byte texels {G src, R src}; U = ring slot
#define RF_READ_TEXEL(U, off_texels)                                   \
    if constexpr (TB == 8) {                                           \
        RF_LDS_READ_B64(tq[U], ta, (off_texels) * 8);                  \
    } else {                                                           \
        RF_LDS_READ_B32_OFF(tq[U], ta, (off_texels) * 4);              \
        if constexpr (TB == 6)                                         \
            RF_LDS_READ_U16_OFF(tqb[U], tb, (off_texels) * 2);         \
    }

    for (int i = -radius; i <= radius; i++) {
        const int hw = hwtab[i + radius];
        const int hw4 = (hw + 3) & ~3;
        const int ai = i < 0 ? -i : i;
        // column c = 4*gq + u - hw4 (gq = 0 .. hw4/2): tile column X = c + r4 + 4*tx, i.e. texel
        // address = ta + (u*Q4 + gq)*TB with ta the per-lane address of (row, group 0, u = 0)
        const uint32_t texel0 = (uint32_t)((ty + i + radius) * TLW + ((r4 - hw4) >> 2));
        uint32_t ta = tile_lane_addr + texel0 * TA;
        uint32_t tb = plane_b_lane_addr + texel0 * 2;
        // weight of tap (i, j) = swc[j] = swc[-j]; group gq needs swc[hw4 - 4*gq - 4 .. +3]
        uint32_t wa_addr = sw_addr0 + (uint32_t)((ai * sw_len + (r4 + 8) + hw4 - 4) * 4);
        const int ngroups = (hw4 >> 1) + 1;

        // Register rings with compile-time indices only: column 4*gq+u lives in tq[u], its
        // gathers in gg[u & 1].  Every read issued in a step is released by the wait at the END
        // of that step, so nothing is in flight across the loop back-edge (a value in flight
        // there would be copied by the compiler's phi moves before it has landed).
        texel_t tq[4];
        uint32_t tqb[4] = {0u, 0u, 0u, 0u};  // second plane (TB == 6 only)
        float4v wna, wnb;
        float gg[2][kPix];
        RF_READ_TEXEL(0, 0)
        RF_READ_TEXEL(1, Q4)
        RF_LDS_READ_B128(wna, wa_addr, 0);
        RF_LDS_READ_B128(wnb, wa_addr, 16);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(tq[0]), "+v"(tq[1]), "+v"(tqb[0]), "+v"(tqb[1]), "+v"(wna), "+v"(wnb));
        issue_gathers(joint_of(tq[0]), gg[0]);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(gg[0][0]), "+v"(gg[0][1]), "+v"(gg[0][2]), "+v"(gg[0][3]));

#define RF_ACCUM(U)                                                                  \
    {                                                                                \
        float s[NCH];                                                                \
        if constexpr (TB == 8) {                                                     \
            const uint32_t sv = tq[(U)].y;                                           \
            s[0] = (float)(sv & 0xff);                                               \
            if constexpr (NCH == 3) {                                                \
                s[1] = (float)((sv >> 8) & 0xff);                                    \
                s[2] = (float)((sv >> 16) & 0xff);                                   \
            }                                                                        \
        } else {                                                                     \
            s[0] = (float)(tq[(U)] >> 24);                                           \
            if constexpr (TB == 6) {                                                 \
                s[1] = (float)(tqb[(U)] & 0xff);                                     \
                s[2] = (float)((tqb[(U)] >> 8) & 0xff);                              \
            }                                                                        \
        }                                                                            \
        _Pragma("unroll") for (int p = 0; p < kPix; p++)                             \
        {                                                                            \
            const float wgt = __fmul_rn(wv[4 + p - (U)], gg[(U) & 1][p]);            \
            _Pragma("unroll") for (int ch = 0; ch < NCH; ch++) sum[p][ch] =          \
                __fadd_rn(sum[p][ch], __fmul_rn(wgt, s[ch]));                        \
            wsum[p] = __fadd_rn(wsum[p], wgt);                                       \
        }                                                                            \
    }
#define RF_TEXEL_OFF(U) ((((U) + 2) & 3) * Q4 + (((U) + 2) >> 2))
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
    RF_READ_TEXEL(((U) + 2) & 3, RF_TEXEL_OFF(U))                                             \
    issue_gathers(joint_of(tq[((U) + 1) & 3]), gg[((U) + 1) & 1]);                            \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    RF_ACCUM(U)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    RF_PIN_ACC()                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                       \
                 : "+v"(tq[((U) + 2) & 3]), "+v"(tqb[((U) + 2) & 3]),                         \
                   "+v"(gg[((U) + 1) & 1][0]), "+v"(gg[((U) + 1) & 1][1]),                    \
                   "+v"(gg[((U) + 1) & 1][2]), "+v"(gg[((U) + 1) & 1][3]));                   \
    __builtin_amdgcn_sched_barrier(0);

        for (int gq = 0; gq < ngroups; gq++) {
            float wv[8];
            wv[0] = wna.x; wv[1] = wna.y; wv[2] = wna.z; wv[3] = wna.w;
            wv[4] = wnb.x; wv[5] = wnb.y; wv[6] = wnb.z; wv[7] = wnb.w;
            RF_STEP(0)
            RF_STEP(1)
            RF_STEP(2)
            // u = 3 also fetches the next group's weight window
            RF_READ_TEXEL(1, RF_TEXEL_OFF(3))
            issue_gathers(joint_of(tq[0]), gg[0]);
            wa_addr -= 16;
            RF_LDS_READ_B128(wna, wa_addr, 0);
            RF_LDS_READ_B128(wnb, wa_addr, 16);
            __builtin_amdgcn_sched_barrier(0);
            RF_ACCUM(3)
            __builtin_amdgcn_sched_barrier(0);
            RF_PIN_ACC()
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(tq[1]), "+v"(tqb[1]), "+v"(wna), "+v"(wnb), "+v"(gg[0][0]),
                           "+v"(gg[0][1]), "+v"(gg[0][2]), "+v"(gg[0][3]));
            __builtin_amdgcn_sched_barrier(0);
            ta += TA;
            tb += 2;
        }
#undef RF_STEP
#undef RF_PIN_ACC
#undef RF_ACCUM
#undef RF_TEXEL_OFF
    }
#undef RF_READ_TEXEL
}

// Hand-scheduled tap loop for grey tiles (4-byte texels, one accumulated channel), same
// arithmetic and the same pipeline as jbf_tap_loop<1, LUTREP, false, TLW, 4>.
//
// Why asm: a gfx950 SIMD retires two wave-instructions per 4 cycles only if at most one of them
// is a "full-pipe" opcode (v_sad_u8, v_lshl_add_u32, v_cvt_*, anything with an SGPR/constant
// operand ...; 4 cycles each back to back) and the other a "simple" one (v_mul_f32 / v_add_f32 /
// v_and_b32 on VGPRs; 2 cycles each) -- tools/microbench/valu_rates2.hip.  Per column this loop
// needs 9 full-pipe and 17 simple instructions; hipcc emits them as an 8-instruction full-pipe
// burst followed by the simple ones, the blocks below interleave them one for one.
// J1: the joint has one channel and its texel field holds the value pre-multiplied by the LUT's
// byte stride (3x that for RF_JBF_GREY_AS_BGR), so v_sad_u32(texel, centre, lane address) IS the
// gather address: no v_lshl_add_u32, 22 instead of 26 VALU instructions per column step.
template <int LUTREP, int TLW, bool J1 = false>
__device__ __forceinline__ void jbf_tap_loop_grey4(uint32_t lut_lane_addr, uint32_t sw_addr0,
                                                   uint32_t tile_lane_addr,
                                                   const uint32_t (&jc)[kPix], int ty, int radius,
                                                   int r4, int sw_len,
                                                   const int *__restrict__ hwtab,
                                                   float (&sum)[kPix][1], float (&wsum)[kPix])
{
    constexpr int Q4 = TLW / 4;
    constexpr int SHIFT = LUTREP == 32 ? 7 : LUTREP == 16 ? 6 : LUTREP == 8 ? 5 : 4;
    static_assert(LUTREP == 32 || LUTREP == 16 || LUTREP == 8 || LUTREP == 4, "LUT replicas");
    uint32_t mask = 0x00ffffffu;
    asm volatile("" : "+v"(mask));  // keep the mask in a VGPR (a literal operand is full-pipe)

    // Row geometry: the lane's first texel address and the address of the first weight window.
    auto row_addr = [&](int i, uint32_t &ta_out, uint32_t &wa_out, int &ngroups_out) {
        const int hw = hwtab[i + radius];
        const int hw4 = (hw + 3) & ~3;
        const int ai = i < 0 ? -i : i;
        ta_out = tile_lane_addr + (uint32_t)(((ty + i + radius) * TLW + ((r4 - hw4) >> 2)) * 4);
        wa_out = sw_addr0 + (uint32_t)((ai * sw_len + (r4 + 8) + hw4 - 4) * 4);
        ngroups_out = (hw4 >> 1) + 1;
    };

    // look-ahead texels of columns 0..3 of a group as two register pairs: columns (0, 1) and
    // (2, 3) are each fetched by one ds_read2_b32 (their tile addresses differ by Q4 texels)
    uint2v tp[2];
    float4v wna, wnb;
    float gg[2][kPix];
    uint32_t ta, wa_addr;
    int ngroups;
    row_addr(-radius, ta, wa_addr, ngroups);
    // prologue of the first tap row: texels of columns 0 and 1, weight window of group 0,
    // gathers of column 0.  Every later row gets these from the last group of the row before.
    asm volatile("ds_read2_b32 %0, %1 offset1:%2" : "=&v"(tp[0]) : "v"(ta), "n"(Q4));
    asm volatile("ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %2 offset:16"
                 : "=&v"(wna), "=&v"(wnb)
                 : "v"(wa_addr));
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(tp[0]));
    {
        const uint32_t tj = tp[0].x & mask;
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            const uint32_t a =
                J1 ? (tj > jc[p] ? tj - jc[p] : jc[p] - tj) + lut_lane_addr
                   : __builtin_amdgcn_sad_u8(tj, jc[p], 0u) * (LUTREP * 4u) + lut_lane_addr;
            asm volatile("ds_read_b32 %0, %1" : "=v"(gg[0][p]) : "v"(a));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(tp[0]), "+v"(wna), "+v"(wnb), "+v"(gg[0][0]), "+v"(gg[0][1]),
                   "+v"(gg[0][2]), "+v"(gg[0][3]));

    // even steps fetch the texels of two columns (this step's look-ahead and the next step's);
    // the pair is an output of the even steps only
#define RF_TQ(U) tp[((U) & 3) >> 1][(U) & 1]
#define RF_G4_TNOUT(U) RF_G4_TNOUT_##U
#define RF_G4_TNOUT_0 [tn] "=&v"(tp[1]),
#define RF_G4_TNOUT_2 [tn] "=&v"(tp[0]),
#define RF_G4_TNOUT_1
#define RF_G4_TNOUT_3
#define RF_G4_READ(U) RF_G4_READ_##U
#define RF_G4_READ_0 "ds_read2_b32 %[tn], %[ta] offset0:%[o0] offset1:%[o1]\n\t"
#define RF_G4_READ_2 "ds_read2_b32 %[tn], %[ta] offset0:%[o0] offset1:%[o1]\n\t"
#define RF_G4_READ_1 ""
#define RF_G4_READ_3 ""
#define RF_TEXEL_OFF4(U) (((((U) + 2) & 3) * Q4 + (((U) + 2) >> 2)) * 4)
    // Column step U of a group: texel of column +2 (from address TA + OFF), SAD + gathers of
    // column +1, accumulation of column +0.  GA = gathers being consumed, GB = gathers being
    // issued (their registers first hold alpha, then the LDS address, then the LUT value).
#define RF_G4_PART1(U, GA, GB, TA, OFF, OFF1)                                                          \
    float w0_, w1_, w2_, w3_, s_;                                                                \
    uint32_t tj_;                                                                                \
    asm volatile(RF_G4_READ(U)                                                                   \
                 "v_and_b32 %[tj], %[mask], %[t1]\n\t"                                           \
                 "v_sad_u8 %[a0], %[tj], %[jc0], 0\n\t"                                          \
                 "v_mul_f32 %[w0], %[wv0], %[g0]\n\t"                                            \
                 "v_sad_u8 %[a1], %[tj], %[jc1], 0\n\t"                                          \
                 "v_mul_f32 %[w1], %[wv1], %[g1]\n\t"                                            \
                 "v_sad_u8 %[a2], %[tj], %[jc2], 0\n\t"                                          \
                 "v_mul_f32 %[w2], %[wv2], %[g2]\n\t"                                            \
                 "v_sad_u8 %[a3], %[tj], %[jc3], 0\n\t"                                          \
                 "v_mul_f32 %[w3], %[wv3], %[g3]\n\t"                                            \
                 "v_cvt_f32_ubyte3 %[s], %[t0]"                                                  \
                 : RF_G4_TNOUT(U)[tj] "=&v"(tj_), [a0] "=&v"(GB[0]),                             \
                   [a1] "=&v"(GB[1]), [a2] "=&v"(GB[2]), [a3] "=&v"(GB[3]), [w0] "=&v"(w0_),     \
                   [w1] "=&v"(w1_), [w2] "=&v"(w2_), [w3] "=&v"(w3_), [s] "=&v"(s_)              \
                 : [ta] "v"(TA), [o0] "n"((OFF) / 4), [o1] "n"((OFF1) / 4), [mask] "v"(mask),    \
                   [t1] "v"(RF_TQ((U) + 1)), [t0] "v"(RF_TQ(U)), [jc0] "v"(jc[0]),               \
                   [jc1] "v"(jc[1]), [jc2] "v"(jc[2]), [jc3] "v"(jc[3]), [wv0] "v"(wv[4 - (U)]), \
                   [wv1] "v"(wv[5 - (U)]), [wv2] "v"(wv[6 - (U)]), [wv3] "v"(wv[7 - (U)]),       \
                   [g0] "v"(GA[0]), [g1] "v"(GA[1]), [g2] "v"(GA[2]), [g3] "v"(GA[3]));          \
    asm volatile("v_lshl_add_u32 %[a0], %[a0], %[sh], %[la]\n\t"                                 \
                 "v_add_f32 %[ws0], %[ws0], %[w0]\n\t"                                           \
                 "v_lshl_add_u32 %[a1], %[a1], %[sh], %[la]\n\t"                                 \
                 "v_add_f32 %[ws1], %[ws1], %[w1]\n\t"                                           \
                 "v_lshl_add_u32 %[a2], %[a2], %[sh], %[la]\n\t"                                 \
                 "v_add_f32 %[ws2], %[ws2], %[w2]\n\t"                                           \
                 "v_lshl_add_u32 %[a3], %[a3], %[sh], %[la]\n\t"                                 \
                 "v_add_f32 %[ws3], %[ws3], %[w3]\n\t"                                           \
                 "ds_read_b32 %[a0], %[a0]\n\t"                                                  \
                 "ds_read_b32 %[a1], %[a1]\n\t"                                                  \
                 "ds_read_b32 %[a2], %[a2]\n\t"                                                  \
                 "ds_read_b32 %[a3], %[a3]"                                                      \
                 : [a0] "+v"(GB[0]), [a1] "+v"(GB[1]), [a2] "+v"(GB[2]), [a3] "+v"(GB[3]),       \
                   [ws0] "+v"(wsum[0]), [ws1] "+v"(wsum[1]), [ws2] "+v"(wsum[2]),                \
                   [ws3] "+v"(wsum[3])                                                           \
                 : [sh] "n"(SHIFT), [la] "v"(lut_lane_addr), [w0] "v"(w0_), [w1] "v"(w1_),       \
                   [w2] "v"(w2_), [w3] "v"(w3_));
    // the same step for a single-channel joint (J1): the SAD of the pre-scaled values plus the
    // lane's LUT address is the gather address
#define RF_G4_PART1_J1(U, GA, GB, TA, OFF, OFF1)                                                       \
    float w0_, w1_, w2_, w3_, s_;                                                                \
    uint32_t tj_;                                                                                \
    asm volatile(RF_G4_READ(U)                                                                   \
                 "v_and_b32 %[tj], %[mask], %[t1]\n\t"                                           \
                 "v_sad_u32 %[a0], %[tj], %[jc0], %[la]\n\t"                                     \
                 "v_mul_f32 %[w0], %[wv0], %[g0]\n\t"                                            \
                 "v_sad_u32 %[a1], %[tj], %[jc1], %[la]\n\t"                                     \
                 "v_mul_f32 %[w1], %[wv1], %[g1]\n\t"                                            \
                 "v_sad_u32 %[a2], %[tj], %[jc2], %[la]\n\t"                                     \
                 "v_mul_f32 %[w2], %[wv2], %[g2]\n\t"                                            \
                 "v_sad_u32 %[a3], %[tj], %[jc3], %[la]\n\t"                                     \
                 "v_mul_f32 %[w3], %[wv3], %[g3]\n\t"                                            \
                 "v_cvt_f32_ubyte3 %[s], %[t0]"                                                  \
                 : RF_G4_TNOUT(U)[tj] "=&v"(tj_), [a0] "=&v"(GB[0]),                             \
                   [a1] "=&v"(GB[1]), [a2] "=&v"(GB[2]), [a3] "=&v"(GB[3]), [w0] "=&v"(w0_),     \
                   [w1] "=&v"(w1_), [w2] "=&v"(w2_), [w3] "=&v"(w3_), [s] "=&v"(s_)              \
                 : [ta] "v"(TA), [o0] "n"((OFF) / 4), [o1] "n"((OFF1) / 4), [mask] "v"(mask),    \
                   [la] "v"(lut_lane_addr),                                                      \
                   [t1] "v"(RF_TQ((U) + 1)), [t0] "v"(RF_TQ(U)), [jc0] "v"(jc[0]),               \
                   [jc1] "v"(jc[1]), [jc2] "v"(jc[2]), [jc3] "v"(jc[3]), [wv0] "v"(wv[4 - (U)]), \
                   [wv1] "v"(wv[5 - (U)]), [wv2] "v"(wv[6 - (U)]), [wv3] "v"(wv[7 - (U)]),       \
                   [g0] "v"(GA[0]), [g1] "v"(GA[1]), [g2] "v"(GA[2]), [g3] "v"(GA[3]));          \
    asm volatile("v_add_f32 %[ws0], %[ws0], %[w0]\n\t"                                           \
                 "v_add_f32 %[ws1], %[ws1], %[w1]\n\t"                                           \
                 "v_add_f32 %[ws2], %[ws2], %[w2]\n\t"                                           \
                 "v_add_f32 %[ws3], %[ws3], %[w3]\n\t"                                           \
                 "ds_read_b32 %[a0], %[a0]\n\t"                                                  \
                 "ds_read_b32 %[a1], %[a1]\n\t"                                                  \
                 "ds_read_b32 %[a2], %[a2]\n\t"                                                  \
                 "ds_read_b32 %[a3], %[a3]"                                                      \
                 : [a0] "+v"(GB[0]), [a1] "+v"(GB[1]), [a2] "+v"(GB[2]), [a3] "+v"(GB[3]),       \
                   [ws0] "+v"(wsum[0]), [ws1] "+v"(wsum[1]), [ws2] "+v"(wsum[2]),                \
                   [ws3] "+v"(wsum[3])                                                           \
                 : [w0] "v"(w0_), [w1] "v"(w1_), [w2] "v"(w2_), [w3] "v"(w3_));
#define RF_G4_PART2(TN, GB, EXTRA_OPERANDS)                                                      \
    asm volatile("v_mul_f32 %[w0], %[w0], %[s]\n\t"                                              \
                 "v_mul_f32 %[w1], %[w1], %[s]\n\t"                                              \
                 "v_mul_f32 %[w2], %[w2], %[s]\n\t"                                              \
                 "v_mul_f32 %[w3], %[w3], %[s]\n\t"                                              \
                 "v_add_f32 %[s0], %[s0], %[w0]\n\t"                                             \
                 "v_add_f32 %[s1], %[s1], %[w1]\n\t"                                             \
                 "v_add_f32 %[s2], %[s2], %[w2]\n\t"                                             \
                 "v_add_f32 %[s3], %[s3], %[w3]\n\t"                                             \
                 "s_waitcnt lgkmcnt(0)"                                                          \
                 : [w0] "+v"(w0_), [w1] "+v"(w1_), [w2] "+v"(w2_), [w3] "+v"(w3_),               \
                   [s0] "+v"(sum[0][0]), [s1] "+v"(sum[1][0]), [s2] "+v"(sum[2][0]),             \
                   [s3] "+v"(sum[3][0]), "+v"(TN), "+v"(GB[0]), "+v"(GB[1]), "+v"(GB[2]),        \
                   "+v"(GB[3]) EXTRA_OPERANDS                                                    \
                 : [s] "v"(s_));
#define RF_COMMA_W , "+v"(wna), "+v"(wnb)
#define RF_LOAD_WINDOW(ADDR)                                                                     \
    asm volatile("ds_read_b128 %0, %2\n\t"                                                       \
                 "ds_read_b128 %1, %2 offset:16"                                                 \
                 : "=&v"(wna), "=&v"(wnb)                                                        \
                 : "v"(ADDR));

#define RF_ROW_LOOP(P1)                                                            \
    for (int i = -radius; i <= radius; i++) {                                                       \
        uint32_t ta_next, wa_next;                                                                  \
        int ngroups_next;                                                                           \
        row_addr(i < radius ? i + 1 : i, ta_next, wa_next, ngroups_next);                           \
        for (int gq = 0; gq < ngroups - 1; gq++) {                                                  \
            float wv[8];                                                                            \
            wv[0] = wna.x; wv[1] = wna.y; wv[2] = wna.z; wv[3] = wna.w;                             \
            wv[4] = wnb.x; wv[5] = wnb.y; wv[6] = wnb.z; wv[7] = wnb.w;                             \
            {                                                                                       \
                P1(0, gg[0], gg[1], ta, RF_TEXEL_OFF4(0), RF_TEXEL_OFF4(1))                         \
                RF_G4_PART2(tp[1], gg[1], )                                                         \
            }                                                                                       \
            {                                                                                       \
                P1(1, gg[1], gg[0], ta, 0, 0)                                                       \
                RF_G4_PART2(tp[1], gg[0], )                                                         \
            }                                                                                       \
            {                                                                                       \
                P1(2, gg[0], gg[1], ta, RF_TEXEL_OFF4(2), RF_TEXEL_OFF4(3))                         \
                RF_G4_PART2(tp[0], gg[1], )                                                         \
            }                                                                                       \
            {                                                                                       \
                P1(3, gg[1], gg[0], ta, 0, 0)                                                       \
                wa_addr -= 16;                                                                      \
                RF_LOAD_WINDOW(wa_addr)                                                             \
                RF_G4_PART2(tp[0], gg[0], RF_COMMA_W)                                               \
            }                                                                                       \
            ta += 4;                                                                                \
        }                                                                                           \
        {                                                                                           \
            float wv[8];                                                                            \
            wv[0] = wna.x; wv[1] = wna.y; wv[2] = wna.z; wv[3] = wna.w;                             \
            wv[4] = wnb.x; wv[5] = wnb.y; wv[6] = wnb.z; wv[7] = wnb.w;                             \
            {                                                                                       \
                P1(0, gg[0], gg[1], ta, RF_TEXEL_OFF4(0), RF_TEXEL_OFF4(1))                         \
                RF_G4_PART2(tp[1], gg[1], )                                                         \
            }                                                                                       \
            {                                                                                       \
                P1(1, gg[1], gg[0], ta, 0, 0)                                                       \
                RF_G4_PART2(tp[1], gg[0], )                                                         \
            }                                                                                       \
            {                                                                                       \
                P1(2, gg[0], gg[1], ta_next, 0, Q4 * 4)                                             \
                RF_G4_PART2(tp[0], gg[1], )                                                         \
            }                                                                                       \
            {                                                                                       \
                P1(3, gg[1], gg[0], ta_next, 0, 0)                                                  \
                RF_LOAD_WINDOW(wa_next)                                                             \
                RF_G4_PART2(tp[0], gg[0], RF_COMMA_W)                                               \
            }                                                                                       \
        }                                                                                           \
        ta = ta_next;                                                                               \
        wa_addr = wa_next;                                                                          \
        ngroups = ngroups_next;                                                                     \
    }
    // (per tap row: all groups but the last run the plain steps; in the last group the two
    //  look-ahead reads fetch columns 0 and 1 of the NEXT row - the columns past the end of this
    //  one carry no weight - step 3 issues that row's first gathers and loads its first weight
    //  window, so the next row starts with a full pipe.  The last row prefetches itself again.)
    if constexpr (J1) {
        RF_ROW_LOOP(RF_G4_PART1_J1)
    } else {
        RF_ROW_LOOP(RF_G4_PART1)
    }
#undef RF_ROW_LOOP
#undef RF_G4_PART1_J1
#undef RF_LOAD_WINDOW
#undef RF_COMMA_W
#undef RF_G4_PART1
#undef RF_G4_PART2
#undef RF_TEXEL_OFF4
#undef RF_TQ
#undef RF_G4_READ
#undef RF_G4_READ_0
#undef RF_G4_READ_1
#undef RF_G4_READ_2
#undef RF_G4_READ_3
#undef RF_G4_TNOUT
#undef RF_G4_TNOUT_0
#undef RF_G4_TNOUT_1
#undef RF_G4_TNOUT_2
#undef RF_G4_TNOUT_3
}

// jbf_tap_loop_grey4 with the gathers TWO column steps ahead of their use (round 5).  In the form above
// a column's four LUT gathers are issued in the step before the one that multiplies by them, and the
// step ends in s_waitcnt lgkmcnt(0): a wave has 8 instructions of its own between issue and wait, the
// rest of the LDS latency has to come from the SIMD's other three waves - with the CU's one LDS pipeline
// 62 % busy that is not enough, and VALU and LDS each sit at 0.68 of their floors.  Here step c issues
// the gathers of column c + 2 and waits with lgkmcnt(4): everything but those four gathers - i.e. the
// gathers of column c + 1, the texel pair and the weight window - has landed (LDS operations of a wave
// return in order, so the window is issued BEFORE the step's gathers).  Four gather buffers (indexed by
// the step within the group), the src byte converted when its texel is at hand for the SAD (two steps
// before use) so that a texel pair is free for re-use after its second SAD: two pairs still suffice.
// Same instructions per step, same arithmetic and order: identical bytes.
// SLAB (round 6, jbf_slab_kernel): the loop runs the tap rows i_first .. i_last only - a slab of the disk's
// rows whose texels are what the LDS tile holds at the moment, tile row of tap row i for the lane's output
// row = ty + i + row_bias - and ADDS to sum / wsum: slabs taken in increasing i keep every pixel's taps in
// row-major order, so any radius runs through this loop with the bytes of one pass over the whole disk.
template <int LUTREP, int TLW, bool J1 = false, bool SLAB = false>
__device__ __forceinline__ void jbf_tap_loop_grey4_la2(uint32_t lut_lane_addr,
                                                       const float *__restrict__ swsym,
                                                       uint32_t tile_lane_addr,
                                                       const uint32_t (&jc)[kPix], int ty, int radius,
                                                       int r4, int sw_len,
                                                       const int *__restrict__ hwtab,
                                                       float (&sum)[kPix][1], float (&wsum)[kPix],
                                                       int i_first = 0, int i_last = 0, int row_bias = 0)
{
    const int i_lo = SLAB ? i_first : -radius, i_hi = SLAB ? i_last : radius;
    const int bias = SLAB ? row_bias : radius;
    constexpr int Q4 = TLW / 4;
    constexpr int SHIFT = LUTREP == 32 ? 7 : LUTREP == 16 ? 6 : LUTREP == 8 ? 5 : 4;
    static_assert(LUTREP == 32 || LUTREP == 16 || LUTREP == 8 || LUTREP == 4, "LUT replicas");
    static_assert(Q4 + 1 <= 255, "ds_read2_b32 offsets are 8 bits (the largest one here: Q4 + 1)");
    static_assert(TLW % 4 == 0, "column-interleaved planes");
    uint32_t mask = 0x00ffffffu;
    asm volatile("" : "+v"(mask));  // keep the mask in a VGPR (a literal operand is full-pipe)

    // A tap row of half-width hw serves the lane's four outputs from the 2 hw + 4 columns -hw .. hw + 3.
    // The row starts at the EVEN column -hws (hws = hw rounded up to even) and runs whole groups of four
    // from there: at most two columns of zero weight per row (round 4 started at a multiple of four, the
    // alignment its ds_read_b128 weight windows needed: up to six; 3,664 column steps instead of 3,764
    // per output quad at radius 33).  A row that starts in the middle of a quad of the column-
    // interleaved tile (phase 1) finds columns (0, 1) of a group in planes 2, 3 and columns (2, 3) in
    // planes 0, 1 of the next quad: each of the two pair reads has its own address register.
    const uint32_t lane_row0 = tile_lane_addr + (uint32_t)(ty * TLW * 4);  // the lane's part of an address
    auto row_addr = [&](int i, int hw, uint32_t &ta_out, uint32_t &tb_out, uint32_t &wa_out,
                        int &ngroups_out) {
        const int hws = (hw + 1) & ~1;
        const int ai = i < 0 ? -i : i;
        const int c0 = r4 - hws;  // first column, relative to the lane's quad origin
        const int quad = (i + bias) * TLW + (c0 >> 2);
        const int phase = (c0 >> 1) & 1;
        ta_out = lane_row0 + (uint32_t)((quad + (phase ? 2 * Q4 : 0)) * 4);
        tb_out = lane_row0 + (uint32_t)((quad + (phase ? 1 : 2 * Q4)) * 4);
        wa_out = (uint32_t)(ai * sw_len + (r4 + 8) + hws - 4);  // index of the first window's first weight
        ngroups_out = (hws + hw + 4 + 3) >> 2;
    };

    uint2v tp[2];        // texel pairs: tp[0] = columns (0, 1), tp[1] = columns (2, 3) of a group
    // The weight window of a group (8 wave-uniform floats) lives in SGPRs: a v_mul_f32 with an SGPR
    // operand hides behind its neighbours like the SADs do (a pair with a v_sad_u8 or a v_add_f32 issues
    // in 4.2 cycles, tools/microbench/pipe_overlap.hip), and two broadcast ds_read_b128 per group - 2 of
    // the 7 dwords an LDS return carried per step, 5 % of the launch - are gone.  Scalar loads share
    // lgkmcnt with the LDS and return out of order, so the window of the NEXT group, requested in step 0,
    // needs a full wait: RF_L2_WAIT_WINDOW, placed where it is free.
    typedef float float8v __attribute__((ext_vector_type(8)));
    float8v ws8, wn8;    // this group's window, the next group's
    float gg[4][kPix];   // gg[u]: LUT values of the group's column u (in flight, then consumed at step u)
    float sv[4];         // sv[u]: src value of column u as float
    uint32_t ta, tb, wa_addr;
    int ngroups;
    if constexpr (SLAB) {
        int hw0;
        const int *hp0 = hwtab + (i_lo + radius);
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hw0) : "s"(hp0));
        row_addr(i_lo, __builtin_amdgcn_readfirstlane(hw0), ta, tb, wa_addr, ngroups);
    } else {
        row_addr(-radius, 0, ta, tb, wa_addr, ngroups);  // (the top row of the disk: half-width 0)
    }
    // The half-width of row i + 1 is needed during row i (its last group reads ahead into row i + 1).
    // A load the compiler issues gets its wait - a full one - at the first use, in the middle of a row
    // with four gathers in flight: one pipeline drain per row.  So the value is requested a row early
    // by hand (hw_ahead, at the top of row i - 1) and taken over at the top of row i: every group has a
    // full wait in the middle of its step 3 (RF_L2_WAIT_WINDOW), a row at least one group.
    int hw_ahead;
    {
        const int *hp = SLAB ? hwtab + ((i_lo + 1 < i_hi ? i_lo + 1 : i_hi) + radius) : hwtab + 1;
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(hw_ahead) : "s"(hp));  // (waited for below)
    }
    // prologue: both texel pairs and the weight window of the first group, gathers and src values of
    // its columns 0 and 1
    asm volatile("ds_read2_b32 %0, %2 offset1:%4\n\t"
                 "ds_read2_b32 %1, %3 offset1:%4"
                 : "=&v"(tp[0]), "=&v"(tp[1])
                 : "v"(ta), "v"(tb), "n"(Q4));
    {
        const float *wp = swsym + wa_addr;
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(ws8) : "s"(wp));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tp[0]), "+v"(tp[1]), "+s"(ws8));
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const uint32_t tx = c == 0 ? tp[0].x : tp[0].y;
        const uint32_t tj = tx & mask;
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            const uint32_t a =
                J1 ? (tj > jc[p] ? tj - jc[p] : jc[p] - tj) + lut_lane_addr
                   : __builtin_amdgcn_sad_u8(tj, jc[p], 0u) * (LUTREP * 4u) + lut_lane_addr;
            asm volatile("ds_read_b32 %0, %1" : "=v"(gg[c][p]) : "v"(a));
        }
        sv[c] = (float)(tx >> 24);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(gg[0][0]), "+v"(gg[0][1]), "+v"(gg[0][2]),
                   "+v"(gg[0][3]), "+v"(gg[1][0]), "+v"(gg[1][1]), "+v"(gg[1][2]), "+v"(gg[1][3]));

#define RF_L2_TQ(U) tp[((U) & 3) >> 1][(U) & 1]
    // even steps read a texel pair of the NEXT group (or of the next row's first group): step 0 its
    // columns (0, 1) into tp[0], step 2 its columns (2, 3) into tp[1].  Where a pair sits depends on the
    // phase of its row: each of the two reads has its own address register (ta: columns (0, 1), tb:
    // columns (2, 3)), set per row, and the ds_read2 offsets are the same for both phases - one v_add
    // per group more, and no branch in the loop (at a join the compiler may move registers, and
    // gathers are in flight at every group end; issuing the read twice under complementary EXEC masks
    // was measured too: the three EXEC writes per read cost more than the shorter rows return)
#define RF_L2_TNOUT(U) RF_L2_TNOUT_##U
#define RF_L2_TNOUT_0 [tn] "=&v"(tp[0]),
#define RF_L2_TNOUT_2 [tn] "=&v"(tp[1]),
#define RF_L2_TNOUT_1
#define RF_L2_TNOUT_3
#define RF_L2_READ(U) RF_L2_READ_##U
#define RF_L2_READ_0 "ds_read2_b32 %[tn], %[ta] offset0:%[o0] offset1:%[o1]\n\t"
#define RF_L2_READ_2 RF_L2_READ_0
#define RF_L2_READ_1 ""
#define RF_L2_READ_3 ""
    // Step U of a group, ONE statement (hipcc puts an s_nop at every boundary between asm statements:
    // three per step cost 0.5 %): [texel pair]; SADs of column U + 2 interleaved with the weights of column
    // U (GA = its gathered LUT values); src value of column U + 2; [MID: the group's full wait, step 3];
    // gather addresses of column U + 2 interleaved with the weight sums of column U; the four gathers
    // (clustered: an LDS instruction between VALU instructions costs their pairing); accumulation of
    // column U (its src value converted two steps ago); the wait that leaves this step's four gathers in
    // flight.  SADn / ADRn: the instruction that forms output n's table index and the one that turns it
    // into an LDS address (J1: one v_sad_u32 does both).
#define RF_L2_STEP_X(U, GA, GB, TA, O0, O1, MID, WAITTXT, SAD0, SAD1, SAD2, SAD3, ADR0, ADR1, ADR2, \
                     ADR3)                                                                       \
    {                                                                                            \
        float w0_, w1_, w2_, w3_;                                                                \
        uint32_t tj_;                                                                            \
        asm volatile(RF_L2_READ(U)                                                               \
                     "v_and_b32 %[tj], %[mask], %[t2]\n\t"                                       \
                     SAD0 "v_mul_f32 %[w0], %[wv0], %[g0]\n\t"                                   \
                     SAD1 "v_mul_f32 %[w1], %[wv1], %[g1]\n\t"                                   \
                     SAD2 "v_mul_f32 %[w2], %[wv2], %[g2]\n\t"                                   \
                     SAD3 "v_mul_f32 %[w3], %[wv3], %[g3]\n\t"                                   \
                     "v_cvt_f32_ubyte3 %[s2], %[t2]\n\t"                                         \
                     MID                                                                         \
                     ADR0 "v_add_f32 %[ws0], %[ws0], %[w0]\n\t"                                  \
                     ADR1 "v_add_f32 %[ws1], %[ws1], %[w1]\n\t"                                  \
                     ADR2 "v_add_f32 %[ws2], %[ws2], %[w2]\n\t"                                  \
                     ADR3 "v_add_f32 %[ws3], %[ws3], %[w3]\n\t"                                  \
                     "ds_read_b32 %[a0], %[a0]\n\t"                                              \
                     "ds_read_b32 %[a1], %[a1]\n\t"                                              \
                     "ds_read_b32 %[a2], %[a2]\n\t"                                              \
                     "ds_read_b32 %[a3], %[a3]\n\t"                                              \
                     "v_mul_f32 %[w0], %[w0], %[s]\n\t"                                          \
                     "v_mul_f32 %[w1], %[w1], %[s]\n\t"                                          \
                     "v_mul_f32 %[w2], %[w2], %[s]\n\t"                                          \
                     "v_mul_f32 %[w3], %[w3], %[s]\n\t"                                          \
                     "v_add_f32 %[s0], %[s0], %[w0]\n\t"                                         \
                     "v_add_f32 %[s1], %[s1], %[w1]\n\t"                                         \
                     "v_add_f32 %[s2_], %[s2_], %[w2]\n\t"                                       \
                     "v_add_f32 %[s3], %[s3], %[w3]\n\t"                                         \
                     WAITTXT                                                                     \
                     : RF_L2_TNOUT(U)[tj] "=&v"(tj_), [a0] "=&v"(GB[0]), [a1] "=&v"(GB[1]),      \
                       [a2] "=&v"(GB[2]), [a3] "=&v"(GB[3]), [w0] "=&v"(w0_), [w1] "=&v"(w1_),   \
                       [w2] "=&v"(w2_), [w3] "=&v"(w3_), [s2] "=&v"(sv[((U) + 2) & 3]),          \
                       [ws0] "+v"(wsum[0]), [ws1] "+v"(wsum[1]), [ws2] "+v"(wsum[2]),            \
                       [ws3] "+v"(wsum[3]), [s0] "+v"(sum[0][0]), [s1] "+v"(sum[1][0]),          \
                       [s2_] "+v"(sum[2][0]), [s3] "+v"(sum[3][0])                               \
                     : [ta] "v"(TA), [o0] "n"(O0), [o1] "n"(O1), [mask] "v"(mask),               \
                       [t2] "v"(RF_L2_TQ((U) + 2)), [jc0] "v"(jc[0]), [jc1] "v"(jc[1]),          \
                       [jc2] "v"(jc[2]), [jc3] "v"(jc[3]), [wv0] "s"(wv[4 - (U)]),               \
                       [wv1] "s"(wv[5 - (U)]), [wv2] "s"(wv[6 - (U)]), [wv3] "s"(wv[7 - (U)]),   \
                       [g0] "v"(GA[0]), [g1] "v"(GA[1]), [g2] "v"(GA[2]), [g3] "v"(GA[3]),       \
                       [sh] "n"(SHIFT), [la] "v"(lut_lane_addr), [s] "v"(sv[(U)]));              \
    }
#define RF_L2_STEP(U, GA, GB, TA, O0, O1, MID, WAITTXT)                                           \
    RF_L2_STEP_X(U, GA, GB, TA, O0, O1, MID, WAITTXT, "v_sad_u8 %[a0], %[tj], %[jc0], 0\n\t",      \
                 "v_sad_u8 %[a1], %[tj], %[jc1], 0\n\t", "v_sad_u8 %[a2], %[tj], %[jc2], 0\n\t",  \
                 "v_sad_u8 %[a3], %[tj], %[jc3], 0\n\t",                                         \
                 "v_lshl_add_u32 %[a0], %[a0], %[sh], %[la]\n\t",                                \
                 "v_lshl_add_u32 %[a1], %[a1], %[sh], %[la]\n\t",                                \
                 "v_lshl_add_u32 %[a2], %[a2], %[sh], %[la]\n\t",                                \
                 "v_lshl_add_u32 %[a3], %[a3], %[sh], %[la]\n\t")
#define RF_L2_STEP_J1(U, GA, GB, TA, O0, O1, MID, WAITTXT)                                        \
    RF_L2_STEP_X(U, GA, GB, TA, O0, O1, MID, WAITTXT, "v_sad_u32 %[a0], %[tj], %[jc0], %[la]\n\t", \
                 "v_sad_u32 %[a1], %[tj], %[jc1], %[la]\n\t",                                    \
                 "v_sad_u32 %[a2], %[tj], %[jc2], %[la]\n\t",                                    \
                 "v_sad_u32 %[a3], %[tj], %[jc3], %[la]\n\t", "", "", "", "")
    // (The last step of a row waits like any other: four gathers stay in flight across the row loop's
    //  back edge, where the compiler writes code of its own - the next row's addresses.  That it moves
    //  none of the registers in flight there is checked on the machine code, along every path of the
    //  control-flow graph: tests/test_cabi.py.  A variant of this loop once got such a v_mov; a full wait
    //  at the row end, which this loop had until the check walked branches, costs 0.3 %.)
    // the window changes hands after the group: the empty statement is ordered behind step 3 and its full
    // wait (both are volatile) and keeps the copy behind itself.  (Load and hand-over are statements of
    // their own, outside any branch: where an SGPR tuple written by an asm statement meets a value of
    // another origin at a join, the backend merges them in VGPRs and cannot give the result back to an
    // "s" operand.)
#define RF_L2_HAND_OVER                                                                          \
    asm volatile("" : "+s"(wn8));                                                                \
    ws8 = wn8;
    // the next group's window: a scalar load of 8 floats from the table in global memory (scalar cache)
#define RF_L2_LOAD_WINDOW(IDX)                                                                   \
    {                                                                                            \
        const float *wp_ = swsym + (IDX);                                                        \
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(wn8) : "s"(wp_));                      \
    }
    // one group of four steps; NA / NB = address registers of the texel pairs read ahead (this row's next
    // group or the next row's first one), PA / PB their ds_read2 offsets; the group's full wait - for the
    // weight window requested before step 0 - sits in the middle of step 3, where nothing is in flight but
    // the gathers of step 2, a whole step old (step 3 reads no texel pair)
#define RF_L2_GROUP(STEP, NA, NB, PA, PB)                                                         \
    {                                                                                            \
        STEP(0, gg[0], gg[2], NA, PA, PB, "", "s_waitcnt lgkmcnt(4)")                            \
        STEP(1, gg[1], gg[3], NA, 0, 0, "", "s_waitcnt lgkmcnt(4)")                              \
        STEP(2, gg[2], gg[0], NB, PA, PB, "", "s_waitcnt lgkmcnt(4)")                            \
        STEP(3, gg[3], gg[1], NB, 0, 0, "s_waitcnt lgkmcnt(0)\n\t", "s_waitcnt lgkmcnt(4)")      \
    }
#define RF_L2_ROW_LOOP(STEP)                                                                      \
    for (int i = i_lo; i <= i_hi; i++) {                                                          \
        uint32_t ta_next, tb_next, wa_next;                                                       \
        int ngroups_next;                                                                         \
        asm volatile("" : "+s"(hw_ahead)); /* behind the full waits of the row before */           \
        row_addr(i < i_hi ? i + 1 : i, __builtin_amdgcn_readfirstlane(hw_ahead), ta_next,         \
                 tb_next, wa_next, ngroups_next);                                                 \
        {                                                                                         \
            const int *hp_ = hwtab + ((i + 2 < i_hi ? i + 2 : i_hi) + radius);                    \
            asm volatile("s_load_dword %0, %1, 0x0" : "=s"(hw_ahead) : "s"(hp_));                 \
        }                                                                                         \
        for (int gq = 0; gq < ngroups - 1; gq++) {                                                \
            float wv[8];                                                                          \
            wv[0] = ws8[0]; wv[1] = ws8[1]; wv[2] = ws8[2]; wv[3] = ws8[3];                       \
            wv[4] = ws8[4]; wv[5] = ws8[5]; wv[6] = ws8[6]; wv[7] = ws8[7];                       \
            wa_addr -= 4;                                                                         \
            RF_L2_LOAD_WINDOW(wa_addr)                                                            \
            RF_L2_GROUP(STEP, ta, tb, 1, Q4 + 1)                                                 \
            RF_L2_HAND_OVER                                                                       \
            ta += 4;                                                                              \
            tb += 4;                                                                              \
        }                                                                                         \
        {                                                                                         \
            float wv[8];                                                                          \
            wv[0] = ws8[0]; wv[1] = ws8[1]; wv[2] = ws8[2]; wv[3] = ws8[3];                       \
            wv[4] = ws8[4]; wv[5] = ws8[5]; wv[6] = ws8[6]; wv[7] = ws8[7];                       \
            RF_L2_LOAD_WINDOW(wa_next)                                                            \
            RF_L2_GROUP(STEP, ta_next, tb_next, 0, Q4)                                           \
            RF_L2_HAND_OVER                                                                       \
        }                                                                                         \
        ta = ta_next;                                                                             \
        wa_addr = wa_next;                                                                        \
        ngroups = ngroups_next;                                                                   \
        tb = tb_next;                                                                             \
    }
    if constexpr (J1) {
        RF_L2_ROW_LOOP(RF_L2_STEP_J1)
    } else {
        RF_L2_ROW_LOOP(RF_L2_STEP)
    }
    // the last steps' gathers (of a row that does not exist) are still in flight: nothing may re-use
    // their registers before they have landed
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(gg[0][0]), "+v"(gg[0][1]), "+v"(gg[0][2]), "+v"(gg[0][3]), "+v"(gg[1][0]),
                   "+v"(gg[1][1]), "+v"(gg[1][2]), "+v"(gg[1][3]), "+v"(tp[0]), "+v"(tp[1]),
                   "+s"(ws8), "+s"(wn8), "+s"(hw_ahead));
#undef RF_L2_ROW_LOOP
#undef RF_L2_GROUP
#undef RF_L2_LOAD_WINDOW
#undef RF_L2_HAND_OVER
#undef RF_L2_STEP_J1
#undef RF_L2_STEP
#undef RF_L2_STEP_X
#undef RF_L2_READ
#undef RF_L2_READ_0
#undef RF_L2_READ_1
#undef RF_L2_READ_2
#undef RF_L2_READ_3
#undef RF_L2_TNOUT
#undef RF_L2_TNOUT_0
#undef RF_L2_TNOUT_1
#undef RF_L2_TNOUT_2
#undef RF_L2_TNOUT_3
#undef RF_L2_TQ
}

// Hand-scheduled tap loop for colour tiles with 6-byte texels (main plane {B,G,R joint, B src},
// second plane {G src, R src}): the arithmetic and the pipeline of jbf_tap_loop<3, LUTREP, false,
// TLW, 6>, the row-carried prologue of jbf_tap_loop_grey4.  Per column step 44 VALU instructions,
// 12 of them on the full pipe (v_and with the literal mask kept in a VGPR is not): 4 v_sad_u8,
// 4 v_lshl_add_u32, 3 v_cvt_f32_ubyte*; hipcc emits the SADs and address computations as one
// burst of nine, here each full-pipe instruction is followed by a simple one (v_mul/v_add).
// SLAB: tap rows i_first .. i_last only, tile row of tap row i = ty + i + row_bias, sums ADDED to (see
// jbf_tap_loop_grey4_la2).
template <int LUTREP, int TLW, bool SLAB = false>
__device__ __forceinline__ void jbf_tap_loop_rgb6(uint32_t lut_lane_addr,
                                                  const float *__restrict__ swsym,
                                                  uint32_t tile_lane_addr,
                                                  uint32_t plane_b_lane_addr,
                                                  const uint32_t (&jc)[kPix], int ty, int radius,
                                                  int r4, int sw_len,
                                                  const int *__restrict__ hwtab,
                                                  float (&sum)[kPix][3], float (&wsum)[kPix],
                                                  int i_first = 0, int i_last = 0, int row_bias = 0)
{
    const int i_lo = SLAB ? i_first : -radius, i_hi = SLAB ? i_last : radius;
    const int bias = SLAB ? row_bias : radius;
    constexpr int Q4 = TLW / 4;
    constexpr int SHIFT = LUTREP == 32 ? 7 : LUTREP == 16 ? 6 : LUTREP == 8 ? 5 : 4;
    static_assert(LUTREP == 32 || LUTREP == 16 || LUTREP == 8 || LUTREP == 4, "LUT replicas");
    uint32_t mask = 0x00ffffffu;
    asm volatile("" : "+v"(mask));  // keep the mask in a VGPR (a literal operand is full-pipe)

    // Rows start at the even column -hws and run whole groups of four, as in jbf_tap_loop_grey4_la2.
    // Steps 0, 1 of a group read columns 2, 3 of the group through the address pair (ta, tb), steps 2, 3
    // read columns 0, 1 of the NEXT group through (ta2, tb2) - the row's phase decides which planes of
    // the column-interleaved tile those are, the offsets in the instructions do not change.
    auto row_addr = [&](int i, uint32_t &ta_out, uint32_t &tb_out, uint32_t &ta2_out,
                        uint32_t &tb2_out, uint32_t &wa_out, int &ngroups_out) {
        const int hw = hwtab[i + radius];
        const int hws = (hw + 1) & ~1;
        const int ai = i < 0 ? -i : i;
        const int c0 = r4 - hws;  // first column, relative to the lane's quad origin
        const uint32_t quad = (uint32_t)((ty + i + bias) * TLW + (c0 >> 2));
        const bool phase = ((c0 >> 1) & 1) != 0;
        const uint32_t q23 = quad + (phase ? 1u : 2u * Q4);  // columns 2, 3 of the row's first group
        const uint32_t q01 = quad + (phase ? 2u * Q4 : 0u);  // columns 0, 1 of the row's first group
        ta_out = tile_lane_addr + q23 * 4;
        tb_out = plane_b_lane_addr + q23 * 2;
        ta2_out = tile_lane_addr + q01 * 4;
        tb2_out = plane_b_lane_addr + q01 * 2;
        wa_out = (uint32_t)(ai * sw_len + (r4 + 8) + hws - 4);  // index of the first window's first weight
        ngroups_out = (hws + hw + 4 + 3) >> 2;
    };

    uint32_t tq[4], tqb[4];
    // the weight window of a group in SGPRs, as in jbf_tap_loop_grey4_la2: every step ends with a full
    // wait here, so the scalar load of the next window (issued in step 3, after that step's gathers)
    // needs no wait of its own
    typedef float float8v __attribute__((ext_vector_type(8)));
    float8v ws8, wn8;
    float gg[2][kPix];
    uint32_t ta, tb, ta2, tb2, wa_addr;
    int ngroups;
    row_addr(i_lo, ta, tb, ta2, tb2, wa_addr, ngroups);
    // prologue of the first tap row (later rows get theirs from the last group of the row before)
    asm volatile("ds_read_b32 %0, %4\n\t"
                 "ds_read_b32 %1, %4 offset:%6\n\t"
                 "ds_read_u16 %2, %5\n\t"
                 "ds_read_u16 %3, %5 offset:%7"
                 : "=&v"(tq[0]), "=&v"(tq[1]), "=&v"(tqb[0]), "=&v"(tqb[1])
                 : "v"(ta2), "v"(tb2), "n"(Q4 * 4), "n"(Q4 * 2));
    {
        const float *wp = swsym + wa_addr;
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(ws8) : "s"(wp));
    }
    // (a scalar load returns out of order: the first texel needs a full wait as well)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(tq[0]), "+v"(tq[1]), "+v"(tqb[0]), "+v"(tqb[1]), "+s"(ws8));
    {
        const uint32_t tj = tq[0] & mask;
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            const uint32_t a = __builtin_amdgcn_sad_u8(tj, jc[p], 0u) * (LUTREP * 4u) + lut_lane_addr;
            asm volatile("ds_read_b32 %0, %1" : "=v"(gg[0][p]) : "v"(a));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(gg[0][0]), "+v"(gg[0][1]), "+v"(gg[0][2]), "+v"(gg[0][3]));

    // Column step U: texel (both planes) of column +2 from TA/TB + offset, SAD + gathers of column
    // +1, accumulation of column +0.  GA = gathers consumed, GB = gathers issued (their registers
    // hold alpha, then the LDS address, then the LUT value).
#define RF_C6_STEP(U, GA, GB, TA, TB, OFFT, EXTRA_ASM, EXTRA_OPERANDS)                           \
    {                                                                                            \
        float w0_, w1_, w2_, w3_, s0_, s1_, s2_, m0_, m1_, m2_, m3_;                             \
        uint32_t tj_;                                                                            \
        EXTRA_ASM /* (step 3: the next window's scalar load; the step ends in a full wait) */    \
        asm volatile("ds_read_b32 %[tn], %[ta] offset:%[off4]\n\t"                               \
                     "ds_read_u16 %[tnb], %[tb] offset:%[off2]\n\t"                              \
                     "v_and_b32 %[tj], %[mask], %[t1]\n\t"                                       \
                     "v_sad_u8 %[a0], %[tj], %[jc0], 0\n\t"                                      \
                     "v_mul_f32 %[w0], %[wv0], %[g0]\n\t"                                        \
                     "v_sad_u8 %[a1], %[tj], %[jc1], 0\n\t"                                      \
                     "v_mul_f32 %[w1], %[wv1], %[g1]\n\t"                                        \
                     "v_sad_u8 %[a2], %[tj], %[jc2], 0\n\t"                                      \
                     "v_mul_f32 %[w2], %[wv2], %[g2]\n\t"                                        \
                     "v_sad_u8 %[a3], %[tj], %[jc3], 0\n\t"                                      \
                     "v_mul_f32 %[w3], %[wv3], %[g3]\n\t"                                        \
                     "v_cvt_f32_ubyte3 %[s0], %[t0]\n\t"                                         \
                     "v_add_f32 %[ws0], %[ws0], %[w0]\n\t"                                       \
                     "v_cvt_f32_ubyte0 %[s1], %[tb0]\n\t"                                        \
                     "v_add_f32 %[ws1], %[ws1], %[w1]\n\t"                                       \
                     "v_cvt_f32_ubyte1 %[s2], %[tb0]\n\t"                                        \
                     "v_add_f32 %[ws2], %[ws2], %[w2]\n\t"                                       \
                     "v_lshl_add_u32 %[a0], %[a0], %[sh], %[la]\n\t"                             \
                     "v_add_f32 %[ws3], %[ws3], %[w3]\n\t"                                       \
                     "v_lshl_add_u32 %[a1], %[a1], %[sh], %[la]\n\t"                             \
                     "v_mul_f32 %[m0], %[w0], %[s0]\n\t"                                         \
                     "v_lshl_add_u32 %[a2], %[a2], %[sh], %[la]\n\t"                             \
                     "v_mul_f32 %[m1], %[w1], %[s0]\n\t"                                         \
                     "v_lshl_add_u32 %[a3], %[a3], %[sh], %[la]\n\t"                             \
                     "v_mul_f32 %[m2], %[w2], %[s0]\n\t"                                         \
                     "ds_read_b32 %[a0], %[a0]\n\t"                                              \
                     "ds_read_b32 %[a1], %[a1]\n\t"                                              \
                     "ds_read_b32 %[a2], %[a2]\n\t"                                              \
                     "ds_read_b32 %[a3], %[a3]\n\t"                                              \
                     "v_mul_f32 %[m3], %[w3], %[s0]\n\t"                                         \
                     "v_add_f32 %[c00], %[c00], %[m0]\n\t"                                       \
                     "v_add_f32 %[c10], %[c10], %[m1]\n\t"                                       \
                     "v_add_f32 %[c20], %[c20], %[m2]\n\t"                                       \
                     "v_add_f32 %[c30], %[c30], %[m3]\n\t"                                       \
                     "v_mul_f32 %[m0], %[w0], %[s1]\n\t"                                         \
                     "v_mul_f32 %[m1], %[w1], %[s1]\n\t"                                         \
                     "v_mul_f32 %[m2], %[w2], %[s1]\n\t"                                         \
                     "v_mul_f32 %[m3], %[w3], %[s1]\n\t"                                         \
                     "v_add_f32 %[c01], %[c01], %[m0]\n\t"                                       \
                     "v_add_f32 %[c11], %[c11], %[m1]\n\t"                                       \
                     "v_add_f32 %[c21], %[c21], %[m2]\n\t"                                       \
                     "v_add_f32 %[c31], %[c31], %[m3]\n\t"                                       \
                     "v_mul_f32 %[m0], %[w0], %[s2]\n\t"                                         \
                     "v_mul_f32 %[m1], %[w1], %[s2]\n\t"                                         \
                     "v_mul_f32 %[m2], %[w2], %[s2]\n\t"                                         \
                     "v_mul_f32 %[m3], %[w3], %[s2]\n\t"                                         \
                     "v_add_f32 %[c02], %[c02], %[m0]\n\t"                                       \
                     "v_add_f32 %[c12], %[c12], %[m1]\n\t"                                       \
                     "v_add_f32 %[c22], %[c22], %[m2]\n\t"                                       \
                     "v_add_f32 %[c32], %[c32], %[m3]\n\t"                                       \
                     "s_waitcnt lgkmcnt(0)"                                                      \
                     : [tn] "=&v"(tq[((U) + 2) & 3]), [tnb] "=&v"(tqb[((U) + 2) & 3]),           \
                       [tj] "=&v"(tj_), [a0] "=&v"(GB[0]), [a1] "=&v"(GB[1]), [a2] "=&v"(GB[2]), \
                       [a3] "=&v"(GB[3]), [w0] "=&v"(w0_), [w1] "=&v"(w1_), [w2] "=&v"(w2_),     \
                       [w3] "=&v"(w3_), [s0] "=&v"(s0_), [s1] "=&v"(s1_), [s2] "=&v"(s2_),       \
                       [m0] "=&v"(m0_), [m1] "=&v"(m1_), [m2] "=&v"(m2_), [m3] "=&v"(m3_),       \
                       [ws0] "+v"(wsum[0]), [ws1] "+v"(wsum[1]), [ws2] "+v"(wsum[2]),            \
                       [ws3] "+v"(wsum[3]),                                                      \
                       [c00] "+v"(sum[0][0]), [c10] "+v"(sum[1][0]), [c20] "+v"(sum[2][0]),      \
                       [c30] "+v"(sum[3][0]), [c01] "+v"(sum[0][1]), [c11] "+v"(sum[1][1]),      \
                       [c21] "+v"(sum[2][1]), [c31] "+v"(sum[3][1]), [c02] "+v"(sum[0][2]),      \
                       [c12] "+v"(sum[1][2]), [c22] "+v"(sum[2][2]), [c32] "+v"(sum[3][2])       \
                       EXTRA_OPERANDS                                                            \
                     : [ta] "v"(TA), [tb] "v"(TB), [off4] "n"((OFFT) * 4), [off2] "n"((OFFT) * 2), \
                       [mask] "v"(mask), [t1] "v"(tq[((U) + 1) & 3]), [t0] "v"(tq[(U)]),         \
                       [tb0] "v"(tqb[(U)]), [jc0] "v"(jc[0]), [jc1] "v"(jc[1]), [jc2] "v"(jc[2]), \
                       [jc3] "v"(jc[3]), [wv0] "s"(wv[4 - (U)]), [wv1] "s"(wv[5 - (U)]),         \
                       [wv2] "s"(wv[6 - (U)]), [wv3] "s"(wv[7 - (U)]), [g0] "v"(GA[0]),          \
                       [g1] "v"(GA[1]), [g2] "v"(GA[2]), [g3] "v"(GA[3]), [sh] "n"(SHIFT),       \
                       [la] "v"(lut_lane_addr));                                                 \
    }
#define RF_C6_NOASM
#define RF_C6_COMMA_W , "+s"(wn8)
#define RF_C6_LOAD_WINDOW(IDX)                                                                   \
    {                                                                                            \
        const float *wp_ = swsym + (IDX);                                                        \
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(wn8) : "s"(wp_));                      \
    }

    for (int i = i_lo; i <= i_hi; i++) {
        uint32_t ta_next, tb_next, ta2_next, tb2_next, wa_next;
        int ngroups_next;
        row_addr(i < i_hi ? i + 1 : i, ta_next, tb_next, ta2_next, tb2_next, wa_next, ngroups_next);
        for (int gq = 0; gq < ngroups - 1; gq++) {
            float wv[8];
            wv[0] = ws8[0]; wv[1] = ws8[1]; wv[2] = ws8[2]; wv[3] = ws8[3];
            wv[4] = ws8[4]; wv[5] = ws8[5]; wv[6] = ws8[6]; wv[7] = ws8[7];
            RF_C6_STEP(0, gg[0], gg[1], ta, tb, 0, RF_C6_NOASM, )
            RF_C6_STEP(1, gg[1], gg[0], ta, tb, Q4, RF_C6_NOASM, )
            RF_C6_STEP(2, gg[0], gg[1], ta2, tb2, 1, RF_C6_NOASM, )
            wa_addr -= 4;
            RF_C6_STEP(3, gg[1], gg[0], ta2, tb2, Q4 + 1, RF_C6_LOAD_WINDOW(wa_addr),
                       RF_C6_COMMA_W)
            ws8 = wn8;
            ta += 4;
            tb += 2;
            ta2 += 4;
            tb2 += 2;
        }
        {
            float wv[8];
            wv[0] = ws8[0]; wv[1] = ws8[1]; wv[2] = ws8[2]; wv[3] = ws8[3];
            wv[4] = ws8[4]; wv[5] = ws8[5]; wv[6] = ws8[6]; wv[7] = ws8[7];
            RF_C6_STEP(0, gg[0], gg[1], ta, tb, 0, RF_C6_NOASM, )
            RF_C6_STEP(1, gg[1], gg[0], ta, tb, Q4, RF_C6_NOASM, )
            // the columns past the end of this row carry no weight: fetch the next row's first two
            RF_C6_STEP(2, gg[0], gg[1], ta2_next, tb2_next, 0, RF_C6_NOASM, )
            RF_C6_STEP(3, gg[1], gg[0], ta2_next, tb2_next, Q4, RF_C6_LOAD_WINDOW(wa_next),
                       RF_C6_COMMA_W)
            ws8 = wn8;
        }
        ta = ta_next;
        tb = tb_next;
        ta2 = ta2_next;
        tb2 = tb2_next;
        wa_addr = wa_next;
        ngroups = ngroups_next;
    }
#undef RF_C6_LOAD_WINDOW
#undef RF_C6_COMMA_W
#undef RF_C6_NOASM
#undef RF_C6_STEP
}
#undef RF_LDS_READ_B64
#undef RF_LDS_READ_B128
#undef RF_LDS_READ_B32
#undef RF_LDS_READ_B32_OFF
#undef RF_LDS_READ_U16_OFF

template <int SCN, int TH, int LUTREP, bool CLAMP>
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
    volatile int *flag_word = reinterpret_cast<volatile int *>(smem);
    float *lutrep = reinterpret_cast<float *>(smem + 16);
    const int lut_bytes = (lut_len * LUTREP * 4 + 15) & ~15;
    float *swl = reinterpret_cast<float *>(smem + 16 + lut_bytes);
    const int sw_bytes = ((radius + 1) * sw_len * 4 + 15) & ~15;
    uint2 *tile = reinterpret_cast<uint2 *>(smem + 16 + lut_bytes + sw_bytes);
    if (threadIdx.x == 0)
        *flag_word = 1;
    __syncthreads();

    const int tid = threadIdx.x;
    const int tile_id = xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x);
    const int img_idx = tile_id / tiles_per_img;
    const int t_in_img = tile_id - img_idx * tiles_per_img;
    const int tile_y0 = (t_in_img / tiles_x) * TH;
    const int tile_x0 = (t_in_img % tiles_x) * kTileW;
    const size_t img = (size_t)img_idx * h * w;
    const int r4 = (radius + 3) & ~3;
    const int tlh = TH + 2 * radius;

    for (int i = tid; i < lut_len * LUTREP; i += NT)
        lutrep[i] = lut[i / LUTREP];
    for (int i = tid; i < (radius + 1) * sw_len; i += NT)
        swl[i] = swsym[i];
    int grey = 1;  // every src texel staged by this thread has B == G == R
    for (int item = tid; item < tlh * Q4; item += NT) {
        const int ry = item / Q4, k = item - ry * Q4;
        const int gy = border_interpolate(tile_y0 - radius + ry, h, border);
        uint32_t jv[4], sv[4];
        load_tile_quad(joint, src, img, gy, tile_x0 - r4 + 4 * k, w, jcn, SCN, border, jv, sv);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (SCN == 3)
                grey &= (int)(((sv[u] ^ (sv[u] >> 8)) & 0xffffu) == 0u);
            tile[ry * TLW + u * Q4 + k] = make_uint2(jv[u], sv[u]);
        }
    }
    const int all_grey = block_all(grey, flag_word);  // also the barrier that publishes the tile
    if (flags & kJbfStageOnly)  // benchmark aid: staging only (tools/jbf_tune.py --stage-only)
        return;

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
        jbf_tap_loop<SCN, LUTREP, CLAMP, TLW, 8>(lut_lane_addr, sw_addr0, tile_lane_addr, 0u, jc, amax,
                                              ty, radius, r4, sw_len, hwtab, sum, wsum);
    } else {
        // single-channel accumulation; for a grey 3-channel src the three sums are the same
        // sequence of float operations, so replicating one of them is bit-identical
        float sum1[kPix][1];
#pragma unroll
        for (int p = 0; p < kPix; p++)
            sum1[p][0] = 0.f;
        jbf_tap_loop<1, LUTREP, CLAMP, TLW, 8>(lut_lane_addr, sw_addr0, tile_lane_addr, 0u, jc, amax, ty,
                                            radius, r4, sw_len, hwtab, sum1, wsum);
#pragma unroll
        for (int p = 0; p < kPix; p++)
#pragma unroll
            for (int c = 0; c < SCN; c++)
                sum[p][c] = sum1[p][0];
    }

    store_quad<SCN, SCN>(dst, img, tile_y0 + ty, tile_x0 + 4 * tx, h, w, sum, wsum, flags);
}

// ------------------------------------------------------------------------------------------
// 64x64-tile kernel (1024 threads = 4 waves/SIMD), the default for radius <= 36.
//
// One launch shape serves both kinds of src; each tile decides for itself while staging:
//   * grey src (single-channel src, or every src texel of the tile has B = G = R -- the case of
//     the reference's `-r.png`): the tile is staged as 4-byte texels {B,G,R joint, grey src}.
//     That halves the tile (130 rows x 144 x 4 B = 73 KB), which is what lets a 64-row tile, a
//     32x replicated (conflict-free) LUT and - for the test aids - the weight table share 160 KB
//     of LDS, i.e. what buys 4 waves per SIMD.  All 1024 threads run the single-channel loop.
//   * colour src: a second plane of 2-byte texels {G src, R src} beside the grey-packed one (6 bytes
//     per texel), one pass on all 1024 threads where that fits with at least 8 LUT replicas (radius
//     33: 32 replicas; the reference's c15 s28 at pitch 176: 8); otherwise passes of 32 / 16 / 8
//     rows of the same two planes, run by threads 0 .. 16 * rows - 1 while all threads stage.
// The colour LUT sits at the END of the workgroup's 163,840-byte LDS allocation and holds only
// the entries before its zero tail: an index past the table addresses LDS beyond the
// allocation, where ds_read returns 0 -- exactly the LUT value there -- so the per-tap clamp
// disappears.  rf_jbf_u8 verifies that behaviour once per device with a probe kernel and uses
// the clamping kernel above if it ever does not hold.
// LDS: [flag][sw table: test aids only][tile ...     free ...][LUT nz*REP f32] = 163,840 B
// ------------------------------------------------------------------------------------------
constexpr int kT64Lds = 163840;

// TH = tile rows: 64 (64x64 tile), or 32 / 16 for the 32x128 and 16x256 strips that cover the
// last h % 64 rows (same 1024 lanes, same tap loops, less padding; 3-channel sources only have
// the 32-row strip, whose colour tile still fits the LDS in one pass).
template <int SCN, int GREP, int CREP, int TLW, int TH = 64>
__global__ __launch_bounds__(1024) void jbf_tile64_kernel(
    const uint8_t *__restrict__ joint, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
    int h, int w, int jcn, int radius, int border, const float *__restrict__ lut, int nz,
    const int *__restrict__ hwtab, const float *__restrict__ swsym, int sw_len, int tiles_x,
    int tiles_per_img, int flags, int crows, int y_base, int x_base)
{
    constexpr int NT = 1024;
    constexpr int QW = NT / TH;   // lanes (4-pixel quads) per tile row
    constexpr int TW = 4 * QW;    // tile width
    static_assert(TH == 64 || TH == 32 || ((TH == 16 || TH == 128) && SCN == 1), "tile shapes");
    // a half-wave covers 32/QW tile rows of QW consecutive words each: the pitch must spread
    // those rows over disjoint banks
    static_assert(QW >= 32 || (QW == 16 && TLW % 32 == 16) || (QW == 8 && TLW % 32 == 8),
                  "row pitch keeps the rows of a half-wave on disjoint banks");
    constexpr int Q4 = TLW / 4;
    extern __shared__ __align__(16) unsigned char smem[];
    volatile int *flag_word = reinterpret_cast<volatile int *>(smem);
    // (the asm tap loops take their weights through scalar loads: the LDS copy of the weight table is
    //  staged only for the compiler-scheduled loop and the round-4 loop, the test / A-B aids)
    const bool need_sw = (flags & (kJbfCompilerLoop | kJbfLookahead1)) != 0;
    float *swl = reinterpret_cast<float *>(smem + 16);
    const int sw_bytes = need_sw ? ((radius + 1) * sw_len * 4 + 15) & ~15 : 0;
    unsigned char *tile_raw = smem + 16 + sw_bytes;
    if (threadIdx.x == 0)
        *flag_word = 3;
    __syncthreads();

    const int tid = threadIdx.x;
    const int tile_id = xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x);
    const int img_idx = tile_id / tiles_per_img;
    const int t_in_img = tile_id - img_idx * tiles_per_img;
    const int tile_y0 = y_base + (t_in_img / tiles_x) * TH;
    const int tile_x0 = x_base + (t_in_img % tiles_x) * TW;
    const size_t img = (size_t)img_idx * h * w;
    const int r4 = (radius + 3) & ~3;
    const int tx = tid % QW;
    const int ty = tid / QW;

    // ---- weight table, grey LUT (optimistic), grey-packed tile ----
    if (need_sw)
        for (int i = tid; i < (radius + 1) * sw_len; i += NT)
            swl[i] = swsym[i];
    float *lut_g = reinterpret_cast<float *>(smem + kT64Lds - nz * GREP * 4);
    for (int i = tid; i < nz * GREP; i += NT)
        lut_g[i] = lut[i / GREP];
    uint32_t *tile4 = reinterpret_cast<uint32_t *>(tile_raw);
    int grey = 1, grey_joint = 1;
    // Single-channel (or grey 3-channel) joint together with a single-channel (or grey) src: the
    // joint field of a texel holds the value times the LUT's byte stride (3x unless the joint
    // really is one channel: the colour distance of three equal channels is 3|d|), which turns
    // the SAD of the tap loop into the gather address (jbf_tap_loop_grey4<.., J1 = true>).
    // Known up front for 1-channel buffers (j1): staged in that form.  For 3-channel buffers
    // it is found out per tile, and the staged tile is then rewritten in LDS.
    const bool j1_ok = !(flags & kJbfCompilerLoop);
    const bool j1 = SCN == 1 && jcn != 3 && j1_ok;
    const uint32_t j1_scale = (uint32_t)(jcn == 1 ? 1 : 3) * (GREP * 4u);
    const int tlh = TH + 2 * radius;
    // one work item = 4 consecutive tile columns (4k..4k+3) of one tile row
    for (int item = tid; item < tlh * Q4; item += NT) {
        const int ry = item / Q4, k = item - ry * Q4;
        const int gy = border_interpolate(tile_y0 - radius + ry, h, border);
        uint32_t jv[4], sv[4];
        load_tile_quad(joint, src, img, gy, tile_x0 - r4 + 4 * k, w, jcn, SCN, border, jv, sv);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (SCN == 3)
                grey &= (int)(((sv[u] ^ (sv[u] >> 8)) & 0xffffu) == 0u);
            if (jcn == 3)
                grey_joint &= (int)(((jv[u] ^ (jv[u] >> 8)) & 0xffffu) == 0u);
            const uint32_t jfield = j1 ? (jv[u] & 0xffu) * j1_scale : jv[u];
            tile4[ry * TLW + u * Q4 + k] = jfield | (sv[u] << 24);
        }
    }
    // (the barrier inside also publishes sw table, LUT and tile)
    const int grey_bits = block_all2(grey, grey_joint, const_cast<int *>(flag_word));
    const int all_grey = grey_bits & 1;
    if (flags & kJbfStageOnly)  // benchmark aid: staging only (tools/jbf_tune.py --stage-only)
        return;
    // grey src and grey joint found out only now: rewrite the joint fields in place
    const bool j1_late = !j1 && j1_ok && (SCN == 1 || all_grey) && (grey_bits & 2);
    if (j1_late) {
        for (int idx = tid; idx < tlh * TLW; idx += NT) {
            const uint32_t t = tile4[idx];
            tile4[idx] = (t & 0xffu) * j1_scale | (t & 0xff000000u);
        }
        __syncthreads();
    }

    const uint32_t sw_addr0 = lds_addr(swl);
    if (SCN == 1 || all_grey) {
        uint32_t jc[kPix];
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            const int X = 4 * tx + p + r4;
            jc[p] = tile4[(ty + radius) * TLW + (X & 3) * Q4 + (X >> 2)] & 0x00ffffffu;
        }
        float sum1[kPix][1], wsum[kPix];
#pragma unroll
        for (int p = 0; p < kPix; p++) {
            sum1[p][0] = 0.f;
            wsum[p] = 0.f;
        }
        const uint32_t lut_lane_addr = lds_addr(lut_g) + (uint32_t)(tid & (GREP - 1)) * 4u;
        const uint32_t tile_lane_addr = lds_addr(tile4) + (uint32_t)tx * 4u;
        if (flags & kJbfCompilerLoop)  // benchmark aid: compiler-scheduled loop instead of the asm one
            jbf_tap_loop<1, GREP, false, TLW, 4>(lut_lane_addr, sw_addr0, tile_lane_addr, 0u, jc, 0u,
                                                 ty, radius, r4, sw_len, hwtab, sum1, wsum);
        else if (flags & kJbfLookahead1) {  // A/B aid: the round-4 pipeline depth
            if (j1 || j1_late)
                jbf_tap_loop_grey4<GREP, TLW, true>(lut_lane_addr, sw_addr0, tile_lane_addr, jc, ty,
                                                    radius, r4, sw_len, hwtab, sum1, wsum);
            else
                jbf_tap_loop_grey4<GREP, TLW>(lut_lane_addr, sw_addr0, tile_lane_addr, jc, ty, radius,
                                              r4, sw_len, hwtab, sum1, wsum);
        } else if (j1 || j1_late)
            jbf_tap_loop_grey4_la2<GREP, TLW, true>(lut_lane_addr, swsym, tile_lane_addr, jc, ty,
                                                    radius, r4, sw_len, hwtab, sum1, wsum);
        else
            jbf_tap_loop_grey4_la2<GREP, TLW>(lut_lane_addr, swsym, tile_lane_addr, jc, ty, radius,
                                              r4, sw_len, hwtab, sum1, wsum);
        store_quad<1, SCN>(dst, img, tile_y0 + ty, tile_x0 + 4 * tx, h, w, sum1, wsum, flags);
        return;
    }

    if constexpr (SCN == 3) {
        if (crows == TH) {
            // ---- colour src, one pass: the grey-packed plane already holds {B,G,R joint, B src};
            //      a second plane of 2-byte texels adds {G src, R src}: 6 bytes per texel, all
            //      1024 threads stay busy (4 waves/SIMD) ----
            uint16_t *plane_b = reinterpret_cast<uint16_t *>(tile_raw + (size_t)tlh * TLW * 4);
            float *lut_c = reinterpret_cast<float *>(smem + kT64Lds - nz * CREP * 4);
            __syncthreads();  // the grey LUT region is about to be overwritten
            for (int i = tid; i < nz * CREP; i += NT)
                lut_c[i] = lut[i / CREP];
            for (int item = tid; item < tlh * Q4; item += NT) {
                const int ry = item / Q4, k = item - ry * Q4;
                const int gy = border_interpolate(tile_y0 - radius + ry, h, border);
                uint32_t jv[4], sv[4];
                load_tile_quad(joint, src, img, gy, tile_x0 - r4 + 4 * k, w, jcn, 3, border, jv, sv);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    plane_b[ry * TLW + u * Q4 + k] = (uint16_t)(sv[u] >> 8);
            }
            __syncthreads();
            uint32_t jc[kPix];
#pragma unroll
            for (int p = 0; p < kPix; p++) {
                const int X = 4 * tx + p + r4;
                jc[p] = tile4[(ty + radius) * TLW + (X & 3) * Q4 + (X >> 2)] & 0x00ffffffu;
            }
            float sum[kPix][3], wsum[kPix];
#pragma unroll
            for (int p = 0; p < kPix; p++) {
                wsum[p] = 0.f;
                sum[p][0] = sum[p][1] = sum[p][2] = 0.f;
            }
            const uint32_t lut_lane_addr = lds_addr(lut_c) + (uint32_t)(tid & (CREP - 1)) * 4u;
            if (flags & kJbfCompilerLoop)  // test aid: compiler-scheduled loop instead of the asm one
                jbf_tap_loop<3, CREP, false, TLW, 6>(lut_lane_addr, sw_addr0,
                                                     lds_addr(tile4) + (uint32_t)tx * 4u,
                                                     lds_addr(plane_b) + (uint32_t)tx * 2u, jc, 0u,
                                                     ty, radius, r4, sw_len, hwtab, sum, wsum);
            else
                jbf_tap_loop_rgb6<CREP, TLW>(lut_lane_addr, swsym,
                                             lds_addr(tile4) + (uint32_t)tx * 4u,
                                             lds_addr(plane_b) + (uint32_t)tx * 2u, jc, ty, radius,
                                             r4, sw_len, hwtab, sum, wsum);
            store_quad<3, 3>(dst, img, tile_y0 + ty, tile_x0 + 4 * tx, h, w, sum, wsum, flags);
            return;
        }
        if constexpr (TH != 64)
            return;  // strips are only launched when the one-pass colour tile fits
        // ---- colour src whose one-pass tile does not fit (row pitch 176: radius 37..52, e.g. the
        //      reference's c15 s28 -> radius 42 on a colour reflectance, /root/reference/README.md:64):
        //      64/crows passes of crows rows, the same two planes of 6 bytes per texel and the same
        //      asm tap loop as the one-pass form, run by threads 0 .. 16*crows-1 while all threads
        //      stage.  (Round 3 had 8-byte texels here, which only left room for 16-row passes at
        //      radius 42 and ran the compiler-scheduled loop: 3.9x the time per tap of the radius-33
        //      colour loop, profiles/r04_bench_other.json.) ----
        float *lut_c = reinterpret_cast<float *>(smem + kT64Lds - nz * CREP * 4);
        const int tlh8 = crows + 2 * radius;
        uint16_t *plane_b = reinterpret_cast<uint16_t *>(tile_raw + (size_t)tlh8 * TLW * 4);
        for (int half = 0; half * crows < 64; half++) {
            const int y0 = tile_y0 + crows * half;
            if (y0 >= h)
                break;
            __syncthreads();  // everyone is done with the previous contents of the tile
            if (half == 0)
                for (int i = tid; i < nz * CREP; i += NT)
                    lut_c[i] = lut[i / CREP];
            for (int item = tid; item < tlh8 * Q4; item += NT) {
                const int ry = item / Q4, k = item - ry * Q4;
                const int gy = border_interpolate(y0 - radius + ry, h, border);
                uint32_t jv[4], sv[4];
                load_tile_quad(joint, src, img, gy, tile_x0 - r4 + 4 * k, w, jcn, 3, border, jv, sv);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    tile4[ry * TLW + u * Q4 + k] = jv[u] | (sv[u] << 24);
                    plane_b[ry * TLW + u * Q4 + k] = (uint16_t)(sv[u] >> 8);
                }
            }
            __syncthreads();
            if (tid < 16 * crows) {
                uint32_t jc[kPix];
#pragma unroll
                for (int p = 0; p < kPix; p++) {
                    const int X = 4 * tx + p + r4;
                    jc[p] = tile4[(ty + radius) * TLW + (X & 3) * Q4 + (X >> 2)] & 0x00ffffffu;
                }
                float sum[kPix][3], wsum[kPix];
#pragma unroll
                for (int p = 0; p < kPix; p++) {
                    wsum[p] = 0.f;
                    sum[p][0] = sum[p][1] = sum[p][2] = 0.f;
                }
                const uint32_t lut_lane_addr = lds_addr(lut_c) + (uint32_t)(tid & (CREP - 1)) * 4u;
                if (flags & kJbfCompilerLoop)  // test aid: compiler-scheduled loop instead of the asm one
                    jbf_tap_loop<3, CREP, false, TLW, 6>(lut_lane_addr, sw_addr0,
                                                         lds_addr(tile4) + (uint32_t)tx * 4u,
                                                         lds_addr(plane_b) + (uint32_t)tx * 2u, jc, 0u,
                                                         ty, radius, r4, sw_len, hwtab, sum, wsum);
                else
                    jbf_tap_loop_rgb6<CREP, TLW>(lut_lane_addr, swsym,
                                                 lds_addr(tile4) + (uint32_t)tx * 4u,
                                                 lds_addr(plane_b) + (uint32_t)tx * 2u, jc, ty, radius,
                                                 r4, sw_len, hwtab, sum, wsum);
                store_quad<3, 3>(dst, img, y0 + ty, tile_x0 + 4 * tx, h, w, sum, wsum, flags);
            }
        }
    }
}

// One channel of a 3-channel result: the lane's 4 horizontally adjacent outputs of channel c.
__device__ inline void store_quad_channel(uint8_t *dst, size_t img, int oy, int ox0, int h, int w,
                                          int c, const float (&sum)[kPix][1],
                                          const float (&wsum)[kPix], int flags)
{
    if (oy >= h || ox0 >= w)
        return;
    uint8_t *o = dst + (img + (size_t)oy * w + ox0) * 3 + c;
#pragma unroll
    for (int p = 0; p < kPix; p++) {
        if (ox0 + p >= w)
            break;
        const float d = (flags & RF_JBF_TRUE_DIVISION) ? wsum[p] : __fdiv_rn(1.0f, wsum[p]);
        o[p * 3] = saturate_u8(finish_value(sum[p][0], d, flags));
    }
}

// Probe for the "LDS reads beyond the allocation return 0" behaviour the 64x64 kernel relies on.
__global__ void lds_oob_probe_kernel(uint32_t *out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *w = reinterpret_cast<uint32_t *>(smem);
    for (int i = threadIdx.x; i < kT64Lds / 4; i += blockDim.x)
        w[i] = 0xdeadbeefu;
    __syncthreads();
    uint32_t bad = 0;
    const uint32_t addrs[4] = {(uint32_t)kT64Lds + 4u * threadIdx.x, (uint32_t)kT64Lds + 61056u,
                               262140u, (uint32_t)kT64Lds - 4u};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addrs[k]));
        bad |= (k < 3) ? (v != 0u) : (v != 0xdeadbeefu);
    }
    if (bad)
        atomicOr(out, 1u);
}

struct Tiled2Config {
    int th, lutrep;
    bool full_lut;  // stage all 256*jcn LUT entries and skip the per-tap clamp
};

size_t tiled2_lds_bytes(const JbfTables &t, int th, int lutrep, bool full_lut)
{
    const int lut_len = full_lut ? 256 * t.joint_cn : t.lut_len;
    const size_t lut_bytes = ((size_t)lut_len * lutrep * 4 + 15) & ~(size_t)15;
    const size_t sw_bytes = ((size_t)(t.radius + 1) * t.sw_len * 4 + 15) & ~(size_t)15;
    return 16 + lut_bytes + sw_bytes + (size_t)kTlw2 * (th + 2 * t.radius) * sizeof(uint2);
}

template <int SCN, int TH, int LUTREP>
int launch_tiled2(const JbfTables &t, bool full_lut, const uint8_t *joint, const uint8_t *src,
                  uint8_t *dst, int n, int h, int w, int jcn, int border, int flags,
                  hipStream_t stream)
{
    const int lut_len = full_lut ? 256 * t.joint_cn : t.lut_len;
    const bool clamp = lut_len < 256 * t.joint_cn;
    const size_t lds = tiled2_lds_bytes(t, TH, LUTREP, full_lut);
    const int tiles_x = ceil_div(w, kTileW), tiles_y = ceil_div(h, TH);
    const long long blocks = (long long)tiles_x * tiles_y * n;
    if (blocks > 0x7fffffffLL)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: batch too large for one launch");
    auto kern = clamp ? jbf_tiled2_kernel<SCN, TH, LUTREP, true>
                      : jbf_tiled2_kernel<SCN, TH, LUTREP, false>;
    RF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(16 * TH), lds, stream, joint, src, dst, h,
                       w, jcn, t.radius, border, t.d_lut, lut_len, t.d_hw, t.d_swsym, t.sw_len,
                       tiles_x, tiles_x * tiles_y, flags);
    return RF_OK;
}


std::vector<int> g_oob_ok;  // per device: -1 unknown, 0 no, 1 yes (guarded by g_mu)

// Runs the probe once per device.  Synchronises the device (only on the first JBF call).
int lds_oob_reads_zero(int dev, bool *ok)
{
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if ((int)g_oob_ok.size() <= dev)
            g_oob_ok.resize(dev + 1, -1);
        if (g_oob_ok[dev] >= 0) {
            *ok = g_oob_ok[dev] == 1;
            return RF_OK;
        }
    }
    // once per device and process, on a stream of its own, waited for here: the answer selects
    // kernels.  (The caller's stream may be capturing: CaptureRelax admits the allocation and the
    // wait on this OTHER stream; nothing of the probe enters the caller's graph.)
    CaptureRelax relax;
    uint32_t *d_flag = nullptr;
    hipStream_t ps = nullptr;
    RF_HIP_CHECK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } guard{ps};
    RF_HIP_CHECK(hipMalloc(&d_flag, sizeof(uint32_t)));
    RF_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(uint32_t), ps));
    RF_HIP_CHECK(hipFuncSetAttribute((const void *)lds_oob_probe_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kT64Lds));
    hipLaunchKernelGGL(lds_oob_probe_kernel, dim3(8), dim3(256), kT64Lds, ps, d_flag);
    uint32_t flag = 1;
    RF_HIP_CHECK(hipMemcpyAsync(&flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, ps));
    RF_HIP_CHECK(hipStreamSynchronize(ps));
    (void)hipFree(d_flag);
    std::lock_guard<std::mutex> lock(g_mu);
    g_oob_ok[dev] = flag == 0 ? 1 : 0;
    *ok = flag == 0;
    return RF_OK;
}

// LDS needed by jbf_tile64_kernel for grey / colour tiles with the given LUT replication.
// Rows per colour pass that fit (64 = one pass with 6-byte texels; 32, 16, 8 = passes with
// 8-byte texels), or 0.
int tile64_fits(const JbfTables &t, int nz, int grep, int crep, int scn, int tlw, int th, int flags)
{
    if (2 * t.r4 + 4 * (1024 / th) + 8 > tlw)
        return 0;
    const bool need_sw = (flags & (kJbfCompilerLoop | kJbfLookahead1)) != 0;  // as in the kernel
    const size_t sw_bytes =
        16 + (need_sw ? ((size_t)(t.radius + 1) * t.sw_len * 4 + 15) & ~(size_t)15 : 0);
    const size_t grey = sw_bytes + (size_t)tlw * (th + 2 * t.radius) * 4 + (size_t)nz * grep * 4;
    if (grey > (size_t)kT64Lds)
        return 0;
    if (scn == 1)
        return 32;
    // colour tiles in one pass: a 4-byte and a 2-byte plane of the full tile
    if (sw_bytes + (size_t)tlw * (th + 2 * t.radius) * 6 + (size_t)nz * crep * 4 <= (size_t)kT64Lds)
        return th;
    if (th != 64)
        return 0;
    for (int crows = 32; crows >= 8; crows >>= 1) {
        const size_t col =
            sw_bytes + (size_t)tlw * (crows + 2 * t.radius) * 6 + (size_t)nz * crep * 4;
        if (col <= (size_t)kT64Lds)
            return crows;
    }
    return 0;
}

// Rows [y_base, y_base + rows) of every image, in tiles of TH rows.
template <int SCN, int GREP, int CREP, int TLW, int TH = 64>
int launch_tile64(const JbfTables &t, int nz, int crows, const uint8_t *joint, const uint8_t *src,
                  uint8_t *dst, int n, int h, int w, int jcn, int border, int flags,
                  hipStream_t stream, int y_base = 0, int rows = -1, int x_base = 0, int cols = -1)
{
    if (rows < 0)
        rows = h;
    if (cols < 0)
        cols = w;
    const int tiles_x = ceil_div(cols, 4 * (1024 / TH)), tiles_y = ceil_div(rows, TH);
    const long long blocks = (long long)tiles_x * tiles_y * n;
    if (blocks > 0x7fffffffLL)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: batch too large for one launch");
    auto kern = jbf_tile64_kernel<SCN, GREP, CREP, TLW, TH>;
    RF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kT64Lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(1024), kT64Lds, stream, joint, src, dst,
                       h, w, jcn, t.radius, border, t.d_lut, nz, t.d_hw, t.d_swsym, t.sw_len,
                       tiles_x, tiles_x * tiles_y, flags, crows, y_base, x_base);
    return RF_OK;
}

// Radius <= 36: 64x64 tiles, except that the last h % 64 rows go to one 32x128 and/or one 16x256
// strip of tiles, and the last w % 64 columns to a column of 128x32 tiles, whenever that takes
// fewer workgroups than another row / column of 64x64 tiles (every workgroup costs the same
// 1024 lanes x all taps).  3-channel sources only have the 32x128 strip, whose colour tile still
// fits the LDS in one pass.  Strips may overlap at the bottom right corner: both write the same
// bytes.
template <int SCN, int GREP, int CREP>
int launch_tile64_rows(const JbfTables &t, int nz, int crows, const uint8_t *joint,
                       const uint8_t *src, uint8_t *dst, int n, int h, int w, int jcn, int border,
                       int flags, hipStream_t stream)
{
    const bool only64 = (flags & kJbfTile64Only) != 0;  // benchmark / test aid: 64x64 tiles only
    // ---- right strip (single-channel sources)
    int sx = 0;
    if (SCN == 1 && !only64 && (w & 63) > 0 && (w & 63) <= 32 &&
        tile64_fits(t, nz, GREP, CREP, SCN, 136, 128, flags) > 0 && ceil_div(h, 128) < ceil_div(h, 64))
        sx = w & 63;
    const int cols_main = w - sx;
    // ---- bottom strips
    const int rem = h & 63;
    const bool ok32 = tile64_fits(t, nz, GREP, CREP, SCN, 208, 32, flags) > 0;
    const bool ok16 = SCN == 1 && tile64_fits(t, nz, GREP, CREP, SCN, 336, 16, flags) > 0;
    int s32 = 0, s16 = 0;  // rows given to each strip
    if (rem > 0 && rem <= 16 && ok16)
        s16 = rem;
    else if (rem > 0 && rem <= 32 && ok32)
        s32 = rem;
    else if (rem > 32 && rem <= 48 && ok32 && ok16)
        s32 = 32, s16 = rem - 32;
    const int strip_blocks =
        (s32 ? ceil_div(cols_main, 128) : 0) + (s16 ? ceil_div(cols_main, 256) : 0);
    if (only64 || cols_main == 0 || ((s32 || s16) && strip_blocks >= ceil_div(cols_main, 64)))
        s32 = s16 = 0;
    const int rows_main = h - s32 - s16;
    int rc = RF_OK;
    if (rows_main > 0 && cols_main > 0)
        rc = launch_tile64<SCN, GREP, CREP, 144>(t, nz, crows, joint, src, dst, n, h, w, jcn,
                                                 border, flags, stream, 0, rows_main, 0, cols_main);
    if (rc == RF_OK && s32)
        rc = launch_tile64<SCN, GREP, CREP, 208, 32>(t, nz, 32, joint, src, dst, n, h, w, jcn,
                                                     border, flags, stream, rows_main, s32, 0,
                                                     cols_main);
    if constexpr (SCN == 1) {
        if (rc == RF_OK && s16)
            rc = launch_tile64<1, GREP, CREP, 336, 16>(t, nz, 32, joint, src, dst, n, h, w, jcn,
                                                       border, flags, stream, rows_main + s32, s16,
                                                       0, cols_main);
        if (rc == RF_OK && sx)
            rc = launch_tile64<1, GREP, CREP, 136, 128>(t, nz, 32, joint, src, dst, n, h, w, jcn,
                                                        border, flags, stream, 0, h, cols_main, sx);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------
// Radius 53..468 (--sigma_spatial is a free float of the reference's tool,
// /root/reference/filter_reflectance.py:117-119: sigma 36 -> radius 54, 47 -> 70, 66 -> 99, 88 -> 132):
// the 64x64 tile with its halo no longer fits the LDS.  The workgroup covers its 64x64 outputs in bands of
// `crows` rows (all 64 - every lane busy - wherever that leaves room for a slab of 24 rows) and takes
// the disk's 2r + 1 tap rows in SLABS of `slab_rows`: the LDS holds the band's rows plus one slab of
// halo at a time, 4-byte texels {B,G,R joint, ONE src byte}, the accumulators stay in registers from
// slab to slab (jbf_tap_loop_grey4_la2<.., SLAB>; the weights come through scalar loads, so the LDS
// holds only tile and LUT), and every pixel's taps still arrive in row-major order - the bytes of
// the one-pass kernels.  The row pitch (64 outputs + 2 r4 + 8 columns) is bounded by the 8-bit offsets
// of the loop's ds_read2: 1008 texels, r4 <= 468 (pitches in steps of 32 up to 336, coarser beyond: a
// wider pitch than needed only costs slab rows); beyond that the untiled kernel remains.  A 3-channel
// src whose channels differ takes ONE pass of the colour loop on 6-byte texels
// (jbf_tap_loop_rgb6<.., SLAB>; one pass per channel of the grey loop only where the second plane
// leaves no room for a slab); a scan of the tile's src bytes up front sends a grey 3-channel tile to
// the grey loop.  Round 5's row-band kernel (radius 53..72: bands of 32 / 16 / 8 rows with the
// whole halo staged, i.e. a half to an eighth of the lanes busy) is gone: slabs run 7.3 G taps/s at
// every radius (0.95 of the radius-33 rate) where the bands ran 6.7 at radius 54, 3.7 at radius 70.
// ------------------------------------------------------------------------------------------
template <int GREP, int TLW>
__global__ __launch_bounds__(1024) void jbf_slab_kernel(
    const uint8_t *__restrict__ joint, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
    int h, int w, int jcn, int scn, int radius, int border, const float *__restrict__ lut, int nz,
    const int *__restrict__ hwtab, const float *__restrict__ swsym, int sw_len, int tiles_x,
    int tiles_per_img, int flags, int crows_g, int slab_g, int crows_c, int slab_c)
{
    // crows_g / slab_g: rows per band and per slab with 4-byte texels (grey src, or a 3-channel src
    // whose tile turns out grey); crows_c / slab_c with 6-byte texels (colour src, one pass: main
    // plane {B,G,R joint, B src} + a plane of 2-byte texels {G src, R src}, jbf_tap_loop_rgb6);
    // crows_c = 0: no room for the second plane - one grey pass per channel
    constexpr int NT = 1024, Q4 = TLW / 4, QW = 16;
    static_assert(TLW % 32 == 16, "row pitch keeps the rows of a half-wave on disjoint banks");
    extern __shared__ __align__(16) unsigned char smem[];
    int *flag_word = reinterpret_cast<int *>(smem);
    uint32_t *tile4 = reinterpret_cast<uint32_t *>(smem + 16);
    const int tid = threadIdx.x;
    if (tid == 0)
        *flag_word = 3;
    const int tile_id = xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x);
    const int img_idx = tile_id / tiles_per_img;
    const int t_in_img = tile_id - img_idx * tiles_per_img;
    const int tile_y0 = (t_in_img / tiles_x) * 64;
    const int tile_x0 = (t_in_img % tiles_x) * 64;
    const size_t img = (size_t)img_idx * h * w;
    const int r4 = (radius + 3) & ~3;
    const int tx = tid % QW, ty = tid / QW;
    float *lut_g = reinterpret_cast<float *>(smem + kT64Lds - nz * GREP * 4);
    for (int i = tid; i < nz * GREP; i += NT)
        lut_g[i] = lut[i / GREP];
    const uint32_t lut_lane_addr = lds_addr(lut_g) + (uint32_t)(tid & (GREP - 1)) * 4u;
    const uint32_t tile_lane_addr = lds_addr(tile4) + (uint32_t)tx * 4u;
    // 3-channel src: is every src texel the tile's 64 rows will ever stage grey (B = G = R)?  One scan
    // of the src bytes of tile + halo up front (a texel is then used by thousands of taps): a grey tile
    // - the reference filters the CNN's grey map as a 3-channel image - takes ONE pass of the grey loop
    int all_grey = scn == 1;
    if (scn == 3) {
        int grey = 1;
        const int rows_all = min(64, h - tile_y0) + 2 * radius;
        for (int item = tid; item < rows_all * Q4; item += NT) {
            const int ry = item / Q4, k = item - ry * Q4;
            const int gy = border_interpolate(tile_y0 - radius + ry, h, border);
            uint32_t jv[4], sv[4];
            load_tile_quad(joint, src, img, gy, tile_x0 - r4 + 4 * k, w, jcn, scn, border, jv, sv);
#pragma unroll
            for (int u = 0; u < 4; u++)
                grey &= (int)(((sv[u] ^ (sv[u] >> 8)) & 0xffffu) == 0u);
        }
        // (the same word in every lane; said so, or the slab bounds - scalar-load addresses - count as divergent)
        all_grey = __builtin_amdgcn_readfirstlane(block_all2(grey, 1, flag_word) & 1);  // (barrier inside)
    }
    const bool rgb6 = !all_grey && crows_c > 0;
    const int crows = rgb6 ? crows_c : crows_g, slab_rows = rgb6 ? slab_c : slab_g;
    const bool active = tid < QW * crows;
    for (int y0 = tile_y0; y0 < tile_y0 + 64 && y0 < h; y0 += crows) {
        for (int c = 0; c < (all_grey || rgb6 ? 1 : scn); c++) {
            // the lane's four centre pixels (their joint values; border arithmetic as in the tile)
            uint32_t jc[kPix];
            float sum1[kPix][1], sum3[kPix][3], wsum[kPix];
            {
                uint32_t jv[4], sv[4];
                load_tile_quad(joint, src, img, min(y0 + ty, h - 1), tile_x0 + 4 * tx, w, jcn, scn, border,
                               jv, sv);
#pragma unroll
                for (int p = 0; p < kPix; p++) {
                    jc[p] = jv[p] & 0x00ffffffu;
                    sum1[p][0] = sum3[p][0] = sum3[p][1] = sum3[p][2] = 0.f;
                    wsum[p] = 0.f;
                }
            }
            for (int i0 = -radius; i0 <= radius; i0 += slab_rows) {
                const int i1 = min(i0 + slab_rows - 1, radius);
                const int tlh = crows + (i1 - i0);
                uint16_t *plane_b = reinterpret_cast<uint16_t *>(tile4 + (size_t)tlh * TLW);
                __syncthreads();  // everyone is done with the previous contents of the tile
                for (int item = tid; item < tlh * Q4; item += NT) {
                    const int ry = item / Q4, k = item - ry * Q4;
                    const int gy = border_interpolate(y0 + i0 + ry, h, border);
                    uint32_t jv[4], sv[4];
                    load_tile_quad(joint, src, img, gy, tile_x0 - r4 + 4 * k, w, jcn, scn, border, jv, sv);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        tile4[ry * TLW + u * Q4 + k] = jv[u] | (((sv[u] >> (8 * c)) & 0xffu) << 24);
                        if (rgb6)
                            plane_b[ry * TLW + u * Q4 + k] = (uint16_t)(sv[u] >> 8);
                    }
                }
                __syncthreads();
                if (active) {
                    if (rgb6)
                        jbf_tap_loop_rgb6<GREP, TLW, true>(lut_lane_addr, swsym, tile_lane_addr,
                                                           lds_addr(plane_b) + (uint32_t)tx * 2u, jc, ty,
                                                           radius, r4, sw_len, hwtab, sum3, wsum, i0, i1, -i0);
                    else
                        jbf_tap_loop_grey4_la2<GREP, TLW, false, true>(lut_lane_addr, swsym, tile_lane_addr,
                                                                       jc, ty, radius, r4, sw_len, hwtab,
                                                                       sum1, wsum, i0, i1, -i0);
                }
            }
            if (active) {
                if (rgb6)
                    store_quad<3, 3>(dst, img, y0 + ty, tile_x0 + 4 * tx, h, w, sum3, wsum, flags);
                else if (scn == 1)
                    store_quad<1, 1>(dst, img, y0 + ty, tile_x0 + 4 * tx, h, w, sum1, wsum, flags);
                else if (all_grey)
                    store_quad<1, 3>(dst, img, y0 + ty, tile_x0 + 4 * tx, h, w, sum1, wsum, flags);
                else
                    store_quad_channel(dst, img, y0 + ty, tile_x0 + 4 * tx, h, w, c, sum1, wsum, flags);
            }
        }
    }
}

// rows per band and per slab of jbf_slab_kernel at row pitch tlw for a LUT replicated grep times
// (crows = 0: does not fit): all 64 rows of the tile in one band - every lane busy - as long as a
// slab is at least 24 rows, else 32, else 16
// (texel_bytes: 4 for the grey-packed plane alone, 6 with the colour plane beside it)
void slab_fits(const JbfTables &t, int nz, int grep, int tlw, int texel_bytes, int *crows, int *slab_rows)
{
    *crows = *slab_rows = 0;
    if (2 * t.r4 + 64 + 8 > tlw)
        return;
    const long long rows =
        ((long long)kT64Lds - 16 - (long long)nz * grep * 4) / ((long long)tlw * texel_bytes);
    for (int cr : {64, 32, 16}) {
        const long long sl = rows - cr + 1;
        if (sl >= (cr == 16 ? 8 : 24)) {
            *crows = cr;
            *slab_rows = (int)std::min<long long>(sl, 2 * t.radius + 1);
            return;
        }
    }
}

template <int GREP, int TLW>
int launch_slab(const JbfTables &t, int nz, int crows, int slab_rows, int crows_c, int slab_c,
                const uint8_t *joint, const uint8_t *src, uint8_t *dst, int n, int h, int w, int jcn,
                int scn, int border, int flags, hipStream_t stream)
{
    const int tiles_x = ceil_div(w, 64), tiles_y = ceil_div(h, 64);
    const long long blocks = (long long)tiles_x * tiles_y * n;
    if (blocks > 0x7fffffffLL)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: batch too large for one launch");
    auto kern = jbf_slab_kernel<GREP, TLW>;
    RF_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kT64Lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(1024), kT64Lds, stream, joint, src, dst, h,
                       w, jcn, scn, t.radius, border, t.d_lut, nz, t.d_hw, t.d_swsym, t.sw_len,
                       tiles_x, tiles_x * tiles_y, flags, crows, slab_rows, crows_c, slab_c);
    return RF_OK;
}

// ------------------------------------------------------------------------------------------
// CV_32F variant: jointBilateralFilter_32f.  The colour weight is linearly
// interpolated in a per-image table of 4096 bins per joint channel over the joint's value
// range; one thread per output pixel, untiled (correctness first: no BASELINE config uses it).
// ------------------------------------------------------------------------------------------
constexpr int kF32BinsPerChannel = 1 << 12;

// order-preserving map of a float's bits to uint32
__device__ inline uint32_t ordered_bits(float v)
{
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float from_ordered_bits(uint32_t k)
{
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float v;
    __builtin_memcpy(&v, &b, 4);
    return v;
}

// minmax[2*img] = min, [2*img+1] = max (ordered bits); initialised to 0xffffffff / 0
__global__ __launch_bounds__(256) void jbf_f32_minmax_kernel(const float *__restrict__ joint,
                                                             uint32_t *__restrict__ minmax,
                                                             size_t count)
{
    const float *p = joint + (size_t)blockIdx.y * count;
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = ordered_bits(p[i]);
        lo = min(lo, k);
        hi = max(hi, k);
    }
    atomicMin(&minmax[2 * blockIdx.y], lo);
    atomicMax(&minmax[2 * blockIdx.y + 1], hi);
}

template <int JCN, int SCN>
__global__ __launch_bounds__(256) void jbf_f32_kernel(
    const float *__restrict__ joint, const float *__restrict__ src, float *__restrict__ dst, int h,
    int w, int border, const float *__restrict__ luts, int lut_stride,
    const float *__restrict__ scales, const int *__restrict__ di, const int *__restrict__ dj,
    const float *__restrict__ sw, int maxk)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    const size_t img = (size_t)blockIdx.z * h * w;
    const float *lut = luts + (size_t)blockIdx.z * lut_stride;
    const float scale_index = scales[blockIdx.z];
    float j0[JCN];
#pragma unroll
    for (int c = 0; c < JCN; c++)
        j0[c] = joint[(img + (size_t)y * w + x) * JCN + c];
    float sum[SCN];
#pragma unroll
    for (int c = 0; c < SCN; c++)
        sum[c] = 0.f;
    float wsum = 0.f;
    for (int k = 0; k < maxk; k++) {
        const int yy = border_interpolate(y + di[k], h, border);
        const int xx = border_interpolate(x + dj[k], w, border);
        float jt[JCN], st[SCN];
#pragma unroll
        for (int c = 0; c < JCN; c++)
            jt[c] = 0.f;
#pragma unroll
        for (int c = 0; c < SCN; c++)
            st[c] = 0.f;
        if (yy >= 0 && xx >= 0) {
            const size_t q = img + (size_t)yy * w + xx;
#pragma unroll
            for (int c = 0; c < JCN; c++)
                jt[c] = joint[q * JCN + c];
#pragma unroll
            for (int c = 0; c < SCN; c++)
                st[c] = src[q * SCN + c];
        }
        float alpha = 0.f;
#pragma unroll
        for (int c = 0; c < JCN; c++)
            alpha = __fadd_rn(alpha, fabsf(__fsub_rn(j0[c], jt[c])));
        alpha = __fmul_rn(alpha, scale_index);
        const int idx = (int)alpha;
        alpha = __fsub_rn(alpha, (float)idx);
        const float l0 = lut[idx], l1 = lut[idx + 1];
        const float wgt = __fmul_rn(sw[k], __fadd_rn(l0, __fmul_rn(alpha, __fsub_rn(l1, l0))));
#pragma unroll
        for (int c = 0; c < SCN; c++)
            sum[c] = __fadd_rn(sum[c], __fmul_rn(wgt, st[c]));
        wsum = __fadd_rn(wsum, wgt);
    }
    const float inv = __fdiv_rn(1.0f, wsum);
#pragma unroll
    for (int c = 0; c < SCN; c++)
        dst[(img + (size_t)y * w + x) * SCN + c] = __fmul_rn(sum[c], inv);
}


// Register-tiled CV_32F kernel.  A float texel is 4*(JCN+SCN) bytes, so the (tile + 2r)^2 halo
// tile of the 8-bit kernels does not fit LDS for 3-channel images at the reference's radius; the
// texels therefore come through the vector cache, but each lane owns 4 horizontally adjacent
// outputs (one texel load feeds 4 outputs, lanes of a wave cover 64 contiguous pixels of 4 rows),
// the interpolated colour table and the spatial weight rows live in LDS, the spatial weights of a
// lane's 4 outputs slide through registers (one LDS read per column step), columns outside the
// disk carry weight 0 (adds +0 to the sums: the tap order per output is OpenCV's), and tiles
// away from the image border skip borderInterpolate.  Same float operations per tap as
// jbf_f32_kernel, so the values are identical.
// Tile height: 32 rows for every texel size.  (8 x 1080p, radius 33, with the four-column loop:
// 3/3-channel joint/src 328 / 449 / 459 MP/s with 16 / 32 / 64 rows, 3/1 342 / 560 / 574, 1/1 615 /
// 618 / 621.  With one column per iteration 16 rows had been the fastest for 6-float texels: the
// loads of more waves thrashed the vector cache without overlapping.)
constexpr int kF32TileW = 64;
constexpr int f32_tile_h(int, int) { return 32; }

// PAIR: the colour table arrives as pairs {lut[i], lut[i+1] - lut[i]} (lut_stride floats = lut_stride/2
// pairs per image, the last pair {0, 0}): one 8-byte LDS read per tap and output instead of two
// 4-byte ones at random addresses, and the table ends where its values reach zero (indices past
// the end are clamped to the last pair: weight 0 either way).  The difference is the float
// subtraction the plain form does per tap, done once per entry.
template <int JCN, int SCN, bool PAIR>
__global__ __launch_bounds__(16 * f32_tile_h(JCN, SCN)) void jbf_f32_quad_kernel(
    const float *__restrict__ joint, const float *__restrict__ src, float *__restrict__ dst, int h,
    int w, int border, const float *__restrict__ luts, int lut_stride,
    const float *__restrict__ scales, const float *__restrict__ swsym, int sw_len, int r4,
    int radius, const int *__restrict__ hwtab)
{
    extern __shared__ __align__(16) float f32_smem[];
    float *lut_s = f32_smem;                       // [lut_stride]
    float *sw_s = f32_smem + ((lut_stride + 3) & ~3);  // [(radius + 1) * sw_len]
    const int last_pair = lut_stride / 2 - 1;
    constexpr int kF32TileH = f32_tile_h(JCN, SCN), kF32Threads = 16 * kF32TileH;
    const int tid = threadIdx.x;
    {
        const float *lut = luts + (size_t)blockIdx.z * lut_stride;
        for (int i = tid; i < lut_stride; i += kF32Threads)
            lut_s[i] = lut[i];
        for (int i = tid; i < (radius + 1) * sw_len; i += kF32Threads)
            sw_s[i] = swsym[i];
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int tx0 = blockIdx.x * kF32TileW, ty0 = blockIdx.y * kF32TileH;
    const int x0 = tx0 + 4 * lx, y = min(ty0 + ly, h - 1);
    const size_t img = (size_t)blockIdx.z * h * w;
    const float scale_index = scales[blockIdx.z];
    // a tile whose taps all fall inside the image needs no border handling
    const bool interior = tx0 - r4 - 4 >= 0 && tx0 + kF32TileW + r4 + 8 <= w && ty0 - radius >= 0 &&
                          ty0 + kF32TileH + radius <= h;
    float j0[4][JCN];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int xc = min(x0 + p, w - 1);
#pragma unroll
        for (int c = 0; c < JCN; c++)
            j0[p][c] = joint[(img + (size_t)y * w + xc) * JCN + c];
    }
    float sum[4][SCN], wsum[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        wsum[p] = 0.f;
#pragma unroll
        for (int c = 0; c < SCN; c++)
            sum[p][c] = 0.f;
    }
    // One group of four columns c4 .. c4+3 of tap row (jrow, srow, wrow): the four texels are
    // requested together (their vector-cache latencies overlap), then consumed in tap order.
    // EDGE: the group straddles the end of some output's disk.  Columns of the span that lie off
    // output p's disk are not taps of p: OpenCV never reads them, so a NaN / Inf texel there must
    // not reach p (0 * Inf is NaN, and a NaN distance would index the table out of range); c, p
    // and hw are wave-uniform, the test is a scalar branch.  Groups inside every output's disk
    // (all but the first and last one or two of a row) run without it.
    auto group = [&](auto interior_c, auto edge_c, const float *jrow, const float *srow,
                     const float *wrow, int c4, int hw, float &w0, float &w1, float &w2, float &w3)
                     __attribute__((always_inline)) {
        constexpr bool INTERIOR = decltype(interior_c)::value, EDGE = decltype(edge_c)::value;
        float jt[4][JCN], st[4][SCN], wn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int xx = INTERIOR ? x0 + c4 + u : border_interpolate(x0 + c4 + u, w, border);
#pragma unroll
            for (int ch = 0; ch < JCN; ch++)
                jt[u][ch] = jrow[(size_t)xx * JCN + ch];
#pragma unroll
            for (int ch = 0; ch < SCN; ch++)
                st[u][ch] = srow[(size_t)xx * SCN + ch];
            wn[u] = wrow[c4 + u + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c4 + u;
            const float ws[4] = {w0, w1, w2, w3};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if (EDGE && (c - p < -hw || c - p > hw))
                    continue;
                float alpha = 0.f;
#pragma unroll
                for (int ch = 0; ch < JCN; ch++)
                    alpha = __fadd_rn(alpha, fabsf(__fsub_rn(j0[p][ch], jt[u][ch])));
                alpha = __fmul_rn(alpha, scale_index);
                const int idx = (int)alpha;
                alpha = __fsub_rn(alpha, (float)idx);
                float l0, dl;
                if (PAIR) {
                    const float2 e = reinterpret_cast<const float2 *>(lut_s)[min(idx, last_pair)];
                    l0 = e.x;
                    dl = e.y;
                } else {
                    l0 = lut_s[idx];
                    dl = __fsub_rn(lut_s[idx + 1], l0);
                }
                const float wgt = __fmul_rn(ws[p], __fadd_rn(l0, __fmul_rn(alpha, dl)));
#pragma unroll
                for (int ch = 0; ch < SCN; ch++)
                    sum[p][ch] = __fadd_rn(sum[p][ch], __fmul_rn(wgt, st[u][ch]));
                wsum[p] = __fadd_rn(wsum[p], wgt);
            }
            w3 = w2;
            w2 = w1;
            w1 = w0;
            w0 = wn[u];
        }
    };
    auto rows = [&](auto interior_c) __attribute__((always_inline)) {
        constexpr bool INTERIOR = decltype(interior_c)::value;
        for (int i = -radius; i <= radius; i++) {
            const int hw = hwtab[i + radius];
            const int hw4 = (hw + 3) & ~3;
            const int yy = INTERIOR ? y + i : border_interpolate(y + i, h, border);
            const float *jrow = joint + (img + (size_t)yy * w) * JCN;
            const float *srow = src + (img + (size_t)yy * w) * SCN;
            // wrow[j], zero off the disk; the weights of outputs 0..3 at column step c are
            // wrow[c], wrow[c-1], wrow[c-2], wrow[c-3] and slide through registers
            const float *wrow = sw_s + (i < 0 ? -i : i) * sw_len + (r4 + 8);
            float w0 = wrow[-hw4], w1 = wrow[-hw4 - 1], w2 = wrow[-hw4 - 2], w3 = wrow[-hw4 - 3];
            for (int c4 = -hw4; c4 <= hw4; c4 += 4) {
                if (c4 - 3 >= -hw && c4 + 3 <= hw)
                    group(interior_c, std::false_type{}, jrow, srow, wrow, c4, hw, w0, w1, w2, w3);
                else
                    group(interior_c, std::true_type{}, jrow, srow, wrow, c4, hw, w0, w1, w2, w3);
            }
        }
    };
    if (interior)
        rows(std::true_type{});
    else
        rows(std::false_type{});
    if (ty0 + ly >= h)
        return;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        if (x0 + p >= w)
            continue;
        const float inv = __fdiv_rn(1.0f, wsum[p]);
#pragma unroll
        for (int c = 0; c < SCN; c++)
            dst[(img + (size_t)y * w + x0 + p) * SCN + c] = __fmul_rn(sum[p][c], inv);
    }
}

}  // namespace

void jbf_shutdown()
{
    std::lock_guard<std::mutex> lock(g_mu);
    g_tables.clear();  // arrays are freed by their owners (calls in flight keep theirs alive)
    g_retired.clear();
}

}  // namespace rf

extern "C" int rf_jbf_u8(const uint8_t *joint, const uint8_t *src, uint8_t *dst, int n, int h,
                         int w, int joint_cn, int src_cn, int d, double sigma_color,
                         double sigma_space, int border, int flags, void *stream_)
{
    using namespace rf;
    if (n == 0)  // an empty batch is valid whatever the (possibly NULL) pointers are
        return RF_OK;
    if (!joint || !src || !dst)
        return fail(RF_E_BADARG, "rf_jbf_u8: NULL image pointer");
    if (n < 0 || h <= 0 || w <= 0)
        return fail(RF_E_BADARG, "rf_jbf_u8: bad size n=%d h=%d w=%d", n, h, w);
    if ((joint_cn != 1 && joint_cn != 3) || (src_cn != 1 && src_cn != 3))
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: channels must be 1 or 3 (joint %d, src %d)",
                    joint_cn, src_cn);
    if (border < 0 || border > 4)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: border type %d", border);
    {
        const size_t px = (size_t)n * h * w;
        if (ranges_overlap(dst, px * src_cn, joint, px * joint_cn) ||
            ranges_overlap(dst, px * src_cn, src, px * src_cn))
            return fail(RF_E_BADARG, "rf_jbf_u8: dst must not overlap an input");
    }
    if (flags & ~(RF_JBF_TRUE_DIVISION | RF_JBF_FORCE_GENERIC | RF_JBF_GREY_AS_BGR))
        return fail(RF_E_BADARG, "rf_jbf_u8: unknown flag bits 0x%x", flags);
    // test / benchmark switches (rf_debug_option) travel to the kernels as private flag bits
    flags |= (debug_get(kDbgJbfStageOnly) ? kJbfStageOnly : 0) |
             (debug_get(kDbgJbfCompilerLoop) ? kJbfCompilerLoop : 0) |
             (debug_get(kDbgJbfTile64Only) ? kJbfTile64Only : 0) |
             (debug_get(kDbgJbfLookahead1) ? kJbfLookahead1 : 0);
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
    // RF_JBF_GREY_AS_BGR: a 1-channel joint counts as 3 equal channels (colour distance 3*|d|,
    // 766-entry LUT); the kernels get joint_cn = -1 and replicate the byte while staging
    const int joint_cn_arg = joint_cn;
    if ((flags & RF_JBF_GREY_AS_BGR) && joint_cn == 1)
        joint_cn = 3;
    const int jcn_kernel = joint_cn_arg == 1 && joint_cn == 3 ? -1 : joint_cn;
    JbfTables t;
    TablesHold hold{t};
    int rc = get_tables(radius, joint_cn, sigma_color, sigma_space, stream, &t);
    if (rc != RF_OK)
        return rc;

    // ---- kernel selection -------------------------------------------------------------
    // tune = 0: automatic; 1..6 force a (tile height, LUT replicas, full LUT) configuration of
    // the tiled kernel (benchmark aid, tools/jbf_tune.py).
    const int tune = debug_get(kDbgJbfTune) & 0xf;
    static const Tiled2Config kCfg[] = {{48, 16, false}, {32, 8, true},  {32, 32, false},
                                        {32, 16, false}, {64, 16, false}, {48, 8, false}};
    constexpr int kNumCfg = (int)(sizeof(kCfg) / sizeof(kCfg[0]));
    // measured on MI355X at 1080p: 3 waves/SIMD with a 16x replicated, clamped LUT wins for grey
    // src; the clamp-free 8x table is next
    // tune 7 forces the 64x64 kernel, tune 1..6 the 64xTH kernel
    bool done = false;
    if (!(flags & RF_JBF_FORCE_GENERIC) && t.r4 <= kJbfMaxTiledR4 && (tune == 0 || tune == 7)) {
        bool oob_ok = false;
        rc = lds_oob_reads_zero(t.device, &oob_ok);
        if (rc != RF_OK)
            return rc;
        // table entries before the zero tail (the whole table if it has none)
        const int nz = t.lut_len < 256 * joint_cn ? t.lut_len - 1 : t.lut_len;
        if (oob_ok) {
            // row pitch 144 serves radius <= 36, 176 radius <= 52 (colour tiles of the wide
            // pitch run in 16- or 8-row passes)
#define RF_T64(G_, C_, W_)                                                                        \
    if (!done && (min_rows < 64 || C_ >= 8) &&                                                    \
        tile64_fits(t, nz, G_, C_, src_cn, W_, 64, flags) >= min_rows) {                          \
        const int crows_ = tile64_fits(t, nz, G_, C_, src_cn, W_, 64, flags);                     \
        if (W_ == 144)                                                                            \
            rc = src_cn == 3 ? launch_tile64_rows<3, G_, C_>(t, nz, crows_, joint, src, dst, n,   \
                                                             h, w, jcn_kernel, border, flags,     \
                                                             stream)                              \
                             : launch_tile64_rows<1, G_, C_>(t, nz, crows_, joint, src, dst, n,   \
                                                             h, w, jcn_kernel, border, flags,     \
                                                             stream);                             \
        else                                                                                      \
            rc = src_cn == 3 ? launch_tile64<3, G_, C_, W_>(t, nz, crows_, joint, src, dst, n, h, \
                                                            w, jcn_kernel, border, flags, stream) \
                             : launch_tile64<1, G_, C_, W_>(t, nz, crows_, joint, src, dst, n, h, \
                                                            w, jcn_kernel, border, flags, stream);\
        if (rc != RF_OK)                                                                          \
            return rc;                                                                            \
        done = true;                                                                              \
    }
            // A colour src first looks for a shape whose colour tile fits in ONE pass over the 64 rows,
            // on all 1024 lanes, even with fewer LUT replicas - down to 8 - (the reference's c15 s28,
            // README.md:64, at pitch 176: 8 replicas, +3.5 % over two 32-row passes on half the lanes
            // with 16), then for one that needs passes.
            for (int min_rows = src_cn == 3 ? 64 : 1; min_rows >= 1 && !done;
                 min_rows = min_rows == 64 ? 1 : 0) {
                RF_T64(32, 32, 144)
                RF_T64(32, 16, 144)
                RF_T64(16, 8, 144)
                RF_T64(8, 4, 144)
                RF_T64(32, 16, 176)
                RF_T64(32, 8, 176)
                RF_T64(32, 4, 176)
                RF_T64(16, 8, 176)
                RF_T64(8, 4, 176)
            }
#undef RF_T64
            // radius 53..468: tap-row slabs (jbf_slab_kernel); row pitches in steps of 32 up to 336,
            // coarser beyond (a wider pitch than needed only costs slab rows)
            if (!done && t.r4 > 52 && t.r4 <= kJbfMaxTiledR4) {
                const int tlw = t.r4 <= 68    ? 208
                                : t.r4 <= 84  ? 240
                                : t.r4 <= 100 ? 272
                                : t.r4 <= 116 ? 304
                                : t.r4 <= 132 ? 336
                                : t.r4 <= 164 ? 400
                                : t.r4 <= 212 ? 496
                                : t.r4 <= 276 ? 624
                                : t.r4 <= 372 ? 816
                                              : 1008;
                int crows = 0, slab = 0, rep = 0, crows_c = 0, slab_c = 0;
                for (int g : {16, 8}) {
                    int cr, sl;
                    slab_fits(t, nz, g, tlw, 4, &cr, &sl);
                    if (cr > crows)
                        crows = cr, slab = sl, rep = g;
                }
                if (crows > 0 && src_cn == 3)  // a colour tile in one pass: 6-byte texels, same replicas
                    slab_fits(t, nz, rep, tlw, 6, &crows_c, &slab_c);
                if (crows > 0) {
#define RF_SLAB(REP_, TLW_)                                                                       \
    launch_slab<REP_, TLW_>(t, nz, crows, slab, crows_c, slab_c, joint, src, dst, n, h, w, jcn_kernel, \
                            src_cn, border, flags, stream)
                    if (tlw == 208)
                        rc = rep == 16 ? RF_SLAB(16, 208) : RF_SLAB(8, 208);
                    else if (tlw == 240)
                        rc = rep == 16 ? RF_SLAB(16, 240) : RF_SLAB(8, 240);
                    else if (tlw == 272)
                        rc = rep == 16 ? RF_SLAB(16, 272) : RF_SLAB(8, 272);
                    else if (tlw == 304)
                        rc = rep == 16 ? RF_SLAB(16, 304) : RF_SLAB(8, 304);
                    else if (tlw == 336)
                        rc = rep == 16 ? RF_SLAB(16, 336) : RF_SLAB(8, 336);
                    else if (tlw == 400)
                        rc = rep == 16 ? RF_SLAB(16, 400) : RF_SLAB(8, 400);
                    else if (tlw == 496)
                        rc = rep == 16 ? RF_SLAB(16, 496) : RF_SLAB(8, 496);
                    else if (tlw == 624)
                        rc = rep == 16 ? RF_SLAB(16, 624) : RF_SLAB(8, 624);
                    else if (tlw == 816)
                        rc = rep == 16 ? RF_SLAB(16, 816) : RF_SLAB(8, 816);
                    else
                        rc = rep == 16 ? RF_SLAB(16, 1008) : RF_SLAB(8, 1008);
#undef RF_SLAB
                    if (rc != RF_OK)
                        return rc;
                    done = true;
                }
            }
        }
    }
    int cfg = -1;
    if (!done && !(flags & RF_JBF_FORCE_GENERIC) && t.r4 <= 36 && tune != 7) {
        for (int ci = 0; ci < kNumCfg; ci++) {
            if (tune != 0 && tune != ci + 1)
                continue;
            const Tiled2Config &c = kCfg[ci];
            if (tiled2_lds_bytes(t, c.th, c.lutrep, c.full_lut) <= (size_t)kMaxLds) {
                cfg = ci;
                break;
            }
        }
    }
    if (cfg >= 0) {
#define RF_T2(TH_, REP_)                                                                        \
    rc = src_cn == 3 ? launch_tiled2<3, TH_, REP_>(t, kCfg[cfg].full_lut, joint, src, dst, n, h, \
                                                   w, jcn_kernel, border, flags, stream)        \
                     : launch_tiled2<1, TH_, REP_>(t, kCfg[cfg].full_lut, joint, src, dst, n, h, \
                                                   w, jcn_kernel, border, flags, stream)
        switch (cfg) {
        case 0: RF_T2(48, 16); break;
        case 1: RF_T2(32, 8); break;
        case 2: RF_T2(32, 32); break;
        case 3: RF_T2(32, 16); break;
        case 4: RF_T2(64, 16); break;
        default: RF_T2(48, 8); break;
        }
#undef RF_T2
        if (rc != RF_OK)
            return rc;
    } else if (!done) {
        dim3 grid(ceil_div(w, 64), ceil_div(h, 4), n);
        if (n > 65535)
            return fail(RF_E_UNSUPPORTED, "rf_jbf_u8: generic path supports n <= 65535");
        hipLaunchKernelGGL(jbf_generic_kernel, grid, dim3(256), 0, stream, joint, src, dst, h, w,
                           jcn_kernel, src_cn, border, t.d_lut, t.d_di, t.d_dj, t.d_sw, t.maxk,
                           flags);
    }
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

extern "C" size_t rf_jbf_f32_workspace_bytes(int n, int joint_cn)
{
    if (n <= 0 || (joint_cn != 1 && joint_cn != 3))
        return 0;
    // room for the pair form of the table (jbf_f32_quad_kernel): two floats per entry
    const size_t lut = (size_t)(rf::kF32BinsPerChannel * joint_cn + 3) * 2 * sizeof(float);
    return (size_t)n * (lut + 2 * sizeof(uint32_t) + sizeof(float)) + 256;
}

extern "C" int rf_jbf_f32(const float *joint, const float *src, float *dst, int n, int h, int w,
                          int joint_cn, int src_cn, int d, double sigma_color, double sigma_space,
                          int border, void *workspace, size_t workspace_bytes, void *stream_)
{
    using namespace rf;
    if (n == 0)
        return RF_OK;
    if (!joint || !src || !dst || !workspace)
        return fail(RF_E_BADARG, "rf_jbf_f32: NULL pointer");
    if (n < 0 || h <= 0 || w <= 0)
        return fail(RF_E_BADARG, "rf_jbf_f32: bad size n=%d h=%d w=%d", n, h, w);
    if ((joint_cn != 1 && joint_cn != 3) || (src_cn != 1 && src_cn != 3))
        return fail(RF_E_UNSUPPORTED, "rf_jbf_f32: channels must be 1 or 3 (joint %d, src %d)",
                    joint_cn, src_cn);
    // BORDER_CONSTANT: the zero padding lies outside the joint's value range, so OpenCV's own
    // 32F code indexes its table out of bounds there (undefined); not offered
    if (border < 1 || border > 4)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_f32: border type %d", border);
    {
        const size_t px = (size_t)n * h * w * sizeof(float);
        if (ranges_overlap(dst, px * src_cn, joint, px * joint_cn) ||
            ranges_overlap(dst, px * src_cn, src, px * src_cn))
            return fail(RF_E_BADARG, "rf_jbf_f32: dst must not overlap an input");
    }
    if (n > 65535)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_f32: n <= 65535 per call");
    if (workspace_bytes < rf_jbf_f32_workspace_bytes(n, joint_cn))
        return fail(RF_E_WORKSPACE, "rf_jbf_f32: workspace %zu B < %zu B", workspace_bytes,
                    rf_jbf_f32_workspace_bytes(n, joint_cn));
    if (sigma_color <= 0)
        sigma_color = 1;
    if (sigma_space <= 0)
        sigma_space = 1;
    int radius = d <= 0 ? (int)std::lrint(sigma_space * 1.5) : d / 2;
    if (radius < 1)
        radius = 1;
    if (radius > 4096)
        return fail(RF_E_UNSUPPORTED, "rf_jbf_f32: radius %d too large", radius);
    hipStream_t stream = (hipStream_t)stream_;
    JbfTables t;  // tap offsets and spatial weights are those of the 8-bit path
    TablesHold hold{t};
    int rc = get_tables(radius, joint_cn, sigma_color, sigma_space, stream, &t);
    if (rc != RF_OK)
        return rc;
    // workspace: [n] (min,max) ordered bits | [n] scale_index | [n] tables
    const int bins = kF32BinsPerChannel * joint_cn;
    uint32_t *d_minmax = reinterpret_cast<uint32_t *>(workspace);
    float *d_scale = reinterpret_cast<float *>(d_minmax + 2 * (size_t)n);
    float *d_luts = reinterpret_cast<float *>(
        static_cast<char *>(workspace) + (((size_t)n * 12 + 255) & ~(size_t)255));
    std::vector<uint32_t> mm(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        mm[2 * i] = 0xffffffffu;
        mm[2 * i + 1] = 0u;
    }
    RF_HIP_CHECK(hipMemcpyAsync(d_minmax, mm.data(), mm.size() * 4, hipMemcpyHostToDevice, stream));
    const size_t count = (size_t)h * w * joint_cn;
    const int mb = (int)std::min<size_t>(256, (count + 1023) / 1024);
    hipLaunchKernelGGL(jbf_f32_minmax_kernel, dim3(mb, n), dim3(256), 0, stream, joint, d_minmax,
                       count);
    // The table depends on the joint's value range and is built with the host's exp (the same
    // libm the CPU path uses), so the range comes back to the host: this entry point
    // synchronises the stream.
    RF_HIP_CHECK(hipMemcpyAsync(mm.data(), d_minmax, mm.size() * 4, hipMemcpyDeviceToHost, stream));
    RF_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<float> luts((size_t)n * (bins + 2)), scales(n);
    const double gauss_color_coeff = -0.5 / (sigma_color * sigma_color);
    for (int i = 0; i < n; i++) {
        const double minv = from_ordered_bits(mm[2 * i]), maxv = from_ordered_bits(mm[2 * i + 1]);
        if (std::fabs(minv - maxv) < FLT_EPSILON)
            return fail(RF_E_UNSUPPORTED, "rf_jbf_f32: image %d has a constant joint (OpenCV falls "
                        "back to a Gaussian blur there, which is not implemented)", i);
        const float len = (float)(maxv - minv) * joint_cn;
        const float scale_index = bins / len;
        scales[i] = scale_index;
        float *lut = luts.data() + (size_t)i * (bins + 2);
        float last = 1.f;
        for (int b = 0; b < bins + 2; b++) {
            if (last > 0.f) {
                const double val = b / scale_index;
                lut[b] = (float)std::exp(val * val * gauss_color_coeff);
                last = lut[b];
            } else {
                lut[b] = 0.f;
            }
        }
    }
    RF_HIP_CHECK(hipMemcpy(d_scale, scales.data(), scales.size() * 4, hipMemcpyHostToDevice));
    // register-tiled kernel when its LDS tables fit (always at the reference's radius); the
    // one-thread-per-pixel kernel otherwise, and as the cross-check (debug option jbf_f32_untiled)
    size_t quad_lds = (size_t)(((bins + 2 + 3) & ~3) + (radius + 1) * t.sw_len) * sizeof(float);
    const bool quad = quad_lds <= 64 * 1024 && !debug_get(kDbgJbfF32Untiled);
    // Pair form of the table (see jbf_f32_quad_kernel) when it fits LDS beside the weight rows: a
    // table that reaches zero ends there; z = first zero entry (every entry after it is zero by
    // construction)
    int lut_stride = bins + 2;
    bool pair = false;
    if (quad) {
        int zmax = 0;
        for (int i = 0; i < n; i++) {
            const float *lut = luts.data() + (size_t)i * (bins + 2);
            int z = 0;
            while (z < bins + 2 && lut[z] > 0.f)
                z++;
            zmax = std::max(zmax, z);
        }
        const int npair = zmax + 1;  // pairs 0 .. zmax; pair zmax = {0, 0}
        const size_t pair_lds =
            (size_t)(((2 * npair + 3) & ~3) + (radius + 1) * t.sw_len) * sizeof(float);
        if (npair <= bins + 3 && pair_lds <= 64 * 1024) {
            pair = true;
            lut_stride = 2 * npair;
            std::vector<float> pairs((size_t)n * lut_stride);
            for (int i = 0; i < n; i++) {
                const float *lut = luts.data() + (size_t)i * (bins + 2);
                float *pp = pairs.data() + (size_t)i * lut_stride;
                for (int b = 0; b < npair; b++) {
                    const float l0 = b < bins + 2 ? lut[b] : 0.f;
                    const float l1 = b + 1 < bins + 2 ? lut[b + 1] : 0.f;
                    pp[2 * b] = l0;
                    pp[2 * b + 1] = l1 - l0;
                }
            }
            luts.swap(pairs);
            quad_lds = (size_t)(((lut_stride + 3) & ~3) + (radius + 1) * t.sw_len) * sizeof(float);
        }
    }
    RF_HIP_CHECK(hipMemcpy(d_luts, luts.data(), luts.size() * 4, hipMemcpyHostToDevice));
    dim3 grid(ceil_div(w, 64), ceil_div(h, 4), n);
#define RF_F32(J_, S_)                                                                         \
    do {                                                                                       \
        if (quad && pair)                                                                      \
            hipLaunchKernelGGL((jbf_f32_quad_kernel<J_, S_, true>),                            \
                               dim3(ceil_div(w, kF32TileW), ceil_div(h, f32_tile_h(J_, S_)), n), \
                               dim3(16 * f32_tile_h(J_, S_)), quad_lds, stream, joint, src, dst, \
                               h, w, border, d_luts, lut_stride, d_scale, t.d_swsym, t.sw_len, \
                               t.r4, radius, t.d_hw);                                          \
        else if (quad)                                                                         \
            hipLaunchKernelGGL((jbf_f32_quad_kernel<J_, S_, false>),                           \
                               dim3(ceil_div(w, kF32TileW), ceil_div(h, f32_tile_h(J_, S_)), n), \
                               dim3(16 * f32_tile_h(J_, S_)), quad_lds, stream, joint, src, dst, \
                               h, w, border, d_luts, bins + 2, d_scale, t.d_swsym, t.sw_len,   \
                               t.r4, radius, t.d_hw);                                          \
        else                                                                                   \
            hipLaunchKernelGGL((jbf_f32_kernel<J_, S_>), grid, dim3(256), 0, stream, joint, src, \
                               dst, h, w, border, d_luts, bins + 2, d_scale, t.d_di, t.d_dj,   \
                               t.d_sw, t.maxk);                                                \
    } while (0)
    if (joint_cn == 3 && src_cn == 3)
        RF_F32(3, 3);
    else if (joint_cn == 3)
        RF_F32(3, 1);
    else if (src_cn == 3)
        RF_F32(1, 3);
    else
        RF_F32(1, 1);
#undef RF_F32
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}
