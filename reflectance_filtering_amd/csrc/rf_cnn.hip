// rf_cnn.hip -- the shipped 1x1 reflectance CNN for gfx950 (MI355X).
//
// Replaces caffe.Net.forward() on /root/reference/network_definition.prototxt:9-165 with
// /root/reference/learned_weights.caffemodel, fed by imgCV2_to_caffeBlob
// (/root/reference/decompose_with_trained_CNN.py:57-69,82-95):
//   x = srgb_lut[R,G,B]                       (uint8 BGR in; the LUT is the reference's float64
//                                              sRGB->linear curve rounded to float32)
//   h0 = relu(W0 x + b0); h_l = relu(W_l h_{l-1} + b_l), l=1..4      (32 channels each)
//   z  = wf . [h0|h1|h2|h3|h4] + bf ;  r = 1/(1+exp(-z))
// Each dot product is a float32 FMA chain over k ascending starting from 0, then + bias (the
// bias is a second rank-1 gemm in Caffe).
//
// Two kernels with identical values: cnn_reflectance_regs_kernel (default; see the comment block
// above it) and cnn_reflectance_kernel, the LDS-column form described next (cross-check).
//
// Mapping: one lane = two pixels; the 32 activations of a pixel live in VGPRs; every FMA is a
// v_pk_fma_f32 that advances TWO output channels of one pixel by one k, its weight pair
// {W[o][k], W[o+1][k]} a wave-uniform 64-bit SGPR operand streamed with s_load_dwordx16 from a
// pair-interleaved copy of the weights (cnn_pack_weights_kernel, 18 KB, rebuilt per call).  K is
// 3/32/160 per pixel, so this is not a dense contraction worth MFMA tiles; packed FMAs double
// the plain v_fma_f32 rate, which on gfx950 issues on the 4-cycle "full" pipe
// (tools/microbench/valu_rates2.hip).  The channel-pair loop is kept rolled: fully unrolled,
// hipcc hoists all 4,513 weights into SGPRs at once and spills them to VGPR lanes.
// Scalar loads are software-pipelined by hand: the weight stream of a 32-input layer is cut into
// chunks of 32 floats (16 k-steps of one channel pair) that alternate between two SGPR buffers;
// each chunk's s_load is issued one chunk (32 packed FMAs) before the s_waitcnt that releases it
// (SMEM returns out of order, so only lgkmcnt(0) is usable and nothing may be issued right
// before a wait).  Left to hipcc the loads were issued and waited for back to back.
#include <mutex>
#include <vector>

#include "rf_common.hpp"

namespace rf {
namespace {

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kPxPerLane = 2;
constexpr int kPackedFloats = 2 * 16 * 3 + 32 + 4 * (2 * 16 * 32 + 32) + 2 * 160 + 1;
constexpr int kWfOff = 128 + 4 * 1056;  // the fuse weights' place in the packed stream
constexpr int kRec = 66;  // floats per channel-pair record of a 32-input layer
// packed layout (floats):
//   [0, 96)          layer 0 pairs: for op in 0..15, k in 0..2: {W0[2op][k], W0[2op+1][k]}
//   [96, 128)        b0
//   then 4 x 16 records of 66 floats: k in 0..31: {W[2op][k], W[2op+1][k]}, then {b[2op], b[2op+1]}
//   then the fuse weights twice each, {wf[k], wf[k]} (one 64-bit operand of a packed FMA that
//   advances both pixels of a lane), then bf

__global__ void cnn_pack_weights_kernel(const float *__restrict__ w, float *__restrict__ packed)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 96) {  // layer 0
        const int op = t / 6, rem = t % 6, k = rem / 2, half = rem % 2;
        packed[t] = w[(2 * op + half) * 3 + k];
    } else if (t < 128) {
        packed[t] = w[t];
    } else if (t < 128 + 4 * 1056) {
        const int l = (t - 128) / 1056, q = (t - 128) % 1056;
        const float *wl = w + 128 + l * 1056;
        const int op = q / kRec, rem = q % kRec, k = rem / 2, half = rem % 2;
        packed[t] = k < 32 ? wl[(2 * op + half) * 32 + k] : wl[1024 + 2 * op + half];
    } else if (t < kWfOff + 320) {
        packed[t] = w[kWfOff + (t - kWfOff) / 2];
    } else if (t < kPackedFloats) {
        packed[t] = w[RF_CNN_NPARAMS - 1];
    }
}

constexpr int kCnnThreads = 128;

__device__ __forceinline__ float2v splat(float v) { return float2v{v, v}; }

// One layer for the lane's two pixels: 16 channel pairs x K steps of v_pk_fma_f32.  The channel
// pair loop is a real loop (so only one pair's 64 weights are live in SGPRs at a time); its
// results go through a lane-private column of LDS because registers cannot be indexed by the
// loop counter: act[ch][lane] holds {pixel 0, pixel 1}.
template <int K>
__device__ __forceinline__ void layer_pairs(const float *__restrict__ wp,
                                            const float *__restrict__ bias,
                                            const float (&in)[kPxPerLane][K == 3 ? 3 : 32],
                                            float2v (*act)[kCnnThreads])
{
    const int lane = threadIdx.x;
#pragma unroll 1
    for (int op = 0; op < 16; op++) {
        const float2v *wq = reinterpret_cast<const float2v *>(wp) + op * K;
        float2v acc[kPxPerLane];
#pragma unroll
        for (int p = 0; p < kPxPerLane; p++)
            acc[p] = float2v{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; k++) {
            const float2v w2 = wq[k];
#pragma unroll
            for (int p = 0; p < kPxPerLane; p++)
                acc[p] = __builtin_elementwise_fma(w2, splat(in[p][k]), acc[p]);
        }
        const float b0 = bias[2 * op], b1 = bias[2 * op + 1];
        act[2 * op][lane] = float2v{fmaxf(__fadd_rn(acc[0].x, b0), 0.f),
                                    fmaxf(__fadd_rn(acc[1].x, b0), 0.f)};
        act[2 * op + 1][lane] = float2v{fmaxf(__fadd_rn(acc[0].y, b1), 0.f),
                                        fmaxf(__fadd_rn(acc[1].y, b1), 0.f)};
    }
}

typedef float float16v __attribute__((ext_vector_type(16)));

// A 32-input layer with hand-pipelined scalar loads (see the file header).  rec = the layer's
// 16 records of kRec floats.  Buffer A holds k = 0..15 of a channel pair, buffer B k = 16..31
// and the two biases.
__device__ __forceinline__ void layer32_pipelined(const float *__restrict__ rec,
                                                  const float (&in)[kPxPerLane][32],
                                                  float2v (*act)[kCnnThreads])
{
    const int lane = threadIdx.x;
    float16v a0, a1, b0, b1;
    float2v bias;
#define RF_SLOAD16(DST, PTR, OFF) \
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(DST) : "s"(PTR), "n"(OFF))
#define RF_SLOAD2(DST, PTR, OFF) \
    asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(DST) : "s"(PTR), "n"(OFF))
#define RF_FMA16(W0, W1, KBASE)                                                              \
    _Pragma("unroll") for (int k = 0; k < 8; k++)                                            \
    {                                                                                        \
        const float2v w2 = float2v{W0[2 * k], W0[2 * k + 1]};                                \
        _Pragma("unroll") for (int p = 0; p < kPxPerLane; p++) acc[p] =                      \
            __builtin_elementwise_fma(w2, splat(in[p][(KBASE) + k]), acc[p]);                \
    }                                                                                        \
    _Pragma("unroll") for (int k = 0; k < 8; k++)                                            \
    {                                                                                        \
        const float2v w2 = float2v{W1[2 * k], W1[2 * k + 1]};                                \
        _Pragma("unroll") for (int p = 0; p < kPxPerLane; p++) acc[p] =                      \
            __builtin_elementwise_fma(w2, splat(in[p][(KBASE) + 8 + k]), acc[p]);            \
    }
    RF_SLOAD16(a0, rec, 0);
    RF_SLOAD16(a1, rec, 64);
#pragma unroll 1
    for (int op = 0; op < 16; op++) {
        const float *cur_rec = rec + op * kRec;
        const float *next_rec = cur_rec + (op < 15 ? kRec : 0);
        float2v acc[kPxPerLane];
#pragma unroll
        for (int p = 0; p < kPxPerLane; p++)
            acc[p] = float2v{0.f, 0.f};
        // buffer A has had a chunk of FMAs to arrive; request B, then work on A
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
        RF_SLOAD16(b0, cur_rec, 128);
        RF_SLOAD16(b1, cur_rec, 192);
        RF_SLOAD2(bias, cur_rec, 256);
        RF_FMA16(a0, a1, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+s"(bias));
        RF_SLOAD16(a0, next_rec, 0);
        RF_SLOAD16(a1, next_rec, 64);
        RF_FMA16(b0, b1, 16)
        act[2 * op][lane] = float2v{fmaxf(__fadd_rn(acc[0].x, bias.x), 0.f),
                                    fmaxf(__fadd_rn(acc[1].x, bias.x), 0.f)};
        act[2 * op + 1][lane] = float2v{fmaxf(__fadd_rn(acc[0].y, bias.y), 0.f),
                                        fmaxf(__fadd_rn(acc[1].y, bias.y), 0.f)};
    }
    // the last iteration's look-ahead load must land before its registers are reused
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
#undef RF_FMA16
#undef RF_SLOAD2
#undef RF_SLOAD16
}

// Reads the layer's activations back into registers and adds its 32 terms of the fuse dot product
__device__ __forceinline__ void collect(float2v (*act)[kCnnThreads], const float *__restrict__ wf32,
                                        float (&cur)[kPxPerLane][32], float (&z)[kPxPerLane])
{
    const int lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const float2v v = act[k][lane];
        cur[0][k] = v.x;
        cur[1][k] = v.y;
        z[0] = __fmaf_rn(wf32[2 * k], v.x, z[0]);
        z[1] = __fmaf_rn(wf32[2 * k], v.y, z[1]);
    }
}

__global__ __launch_bounds__(kCnnThreads) void cnn_reflectance_kernel(
    const uint8_t *__restrict__ bgr, float *__restrict__ r_out, uint8_t *__restrict__ r_u8_out,
    size_t npix, const float *__restrict__ packed, const float *__restrict__ srgb_lut)
{
    __shared__ float lut[256];
    __shared__ float2v act[32][kCnnThreads];
    for (int i = threadIdx.x; i < 256; i += kCnnThreads)
        lut[i] = srgb_lut[i];
    __syncthreads();
    const float *wf = packed + kWfOff;
    const size_t stride = (size_t)gridDim.x * kCnnThreads;
    // lane handles pixels i and i + half (both halves stay coalesced)
    const size_t half = (npix + 1) / 2;
    for (size_t i = (size_t)blockIdx.x * kCnnThreads + threadIdx.x; i < half; i += stride) {
        size_t idx[kPxPerLane] = {i, i + half};
        float x[kPxPerLane][3];
#pragma unroll
        for (int p = 0; p < kPxPerLane; p++) {
            const size_t q = idx[p] < npix ? idx[p] : npix - 1;
            const uint8_t *px = bgr + q * 3;
            x[p][0] = lut[px[2]];  // blob channel order is RGB
            x[p][1] = lut[px[1]];
            x[p][2] = lut[px[0]];
        }
        float cur[kPxPerLane][32];
        float z[kPxPerLane] = {0.f, 0.f};
        layer_pairs<3>(packed, packed + 96, x, act);
        collect(act, wf, cur, z);
#pragma unroll 1
        for (int l = 0; l < 4; l++) {
            layer32_pipelined(packed + 128 + l * 1056, cur, act);
            collect(act, wf + 64 * (l + 1), cur, z);
        }
#pragma unroll
        for (int p = 0; p < kPxPerLane; p++) {
            if (idx[p] >= npix)
                continue;
            const float zz = __fadd_rn(z[p], wf[320]);
            // caffe: 1. / (1. + exp(-x)) with a float exp; expf modelled as round(exp in double)
            const float e = (float)exp((double)(-zz));
            const float r = (float)(1.0 / (1.0 + (double)e));
            if (r_out)
                r_out[idx[p]] = r;
            if (r_u8_out)
                r_u8_out[idx[p]] = (uint8_t)__fmul_rn(r, 255.0f);  // astype(uint8): truncation
        }
    }
}

// ------------------------------------------------------------------------------------------
// Register-resident form (default): the same FMA chains, no LDS for activations.
//
// Activations are kept as one register pair per channel holding {pixel 0, pixel 1}.  The packed FMA
// of output-channel pair (2o, 2o+1) and pixel p reads its input through op_sel on the VGPR
// operand (both halves take pixel p's value) - op_sel on a VGPR source is sound on gfx950, only an
// SGPR source with re-routed halves is not - so inputs are not duplicated; the
// channel-pair loop is fully unrolled with statically named outputs, and bias + ReLU write the
// next layer's {pixel 0, pixel 1} pairs directly.  Against the LDS-column form this drops the 32
// KB of LDS per workgroup, the write/read of every activation through it and the duplicating
// moves: ~170 VGPRs and nothing else limiting occupancy, i.e. three waves per SIMD instead of two.
// The layer loop stays rolled, two layers per iteration ping-ponging between the pair sets A and
// B: fully unrolled, hipcc gives every layer's outputs registers of their own (+64 per layer) and
// the kernel drops to one wave per SIMD.  Every scalar load is an asm statement at an immediate offset from a
// per-layer base: plain loads are hoisted together and spill the SGPR file.
// ------------------------------------------------------------------------------------------
#define RF_PKFMA_P0(ACC, W2, IN)                                                              \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]"               \
                 : "+v"(ACC)                                                                  \
                 : "s"(W2), "v"(IN))
#define RF_PKFMA_P1(ACC, W2, IN)                                                              \
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"               \
                 : "+v"(ACC)                                                                  \
                 : "s"(W2), "v"(IN))

// bias + ReLU of one channel pair for both pixels, written as next-layer pairs {px0, px1}
__device__ __forceinline__ void finish_pair(const float2v &acc0, const float2v &acc1, float bx,
                                            float by, float2v &out_even, float2v &out_odd)
{
    out_even = float2v{fmaxf(__fadd_rn(acc0.x, bx), 0.f), fmaxf(__fadd_rn(acc1.x, bx), 0.f)};
    out_odd = float2v{fmaxf(__fadd_rn(acc0.y, by), 0.f), fmaxf(__fadd_rn(acc1.y, by), 0.f)};
}

// Four k-steps of both pixels' chains (eight packed FMAs) as ONE asm statement.  hipcc's hazard
// recogniser does not count an inline-asm statement as a wait state and assumes it may write
// with a destination select, so between single-instruction statements of a dependent chain it
// inserts an s_nop per k-step (944 of them in the two-layer loop body); inside a statement the
// chain is the hardware's business (plain dependent VALU, interlocked).
// k-steps 1..3 of a statement (operands: %0 %1 accumulators, %2..%5 weight pairs, %6..%9 inputs)
#define RF_PK4_TAIL                                                                           \
    "v_pk_fma_f32 %0, %3, %7, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"                          \
    "v_pk_fma_f32 %1, %3, %7, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"                          \
    "v_pk_fma_f32 %0, %4, %8, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"                          \
    "v_pk_fma_f32 %1, %4, %8, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"                          \
    "v_pk_fma_f32 %0, %5, %9, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"                          \
    "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
#define RF_PKFMA4(ACC0, ACC1, WA, WB, WC, WD, IA, IB, IC, ID)                                 \
    asm volatile("v_pk_fma_f32 %0, %2, %6, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"             \
                 "v_pk_fma_f32 %1, %2, %6, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n" RF_PK4_TAIL \
                 : "+v"(ACC0), "+v"(ACC1)                                                     \
                 : "s"(WA), "s"(WB), "s"(WC), "s"(WD), "v"(IA), "v"(IB), "v"(IC), "v"(ID))
// the first four k-steps of a chain: fma(w, x, +0) == w * x, so it starts with a packed multiply
#define RF_PKMUL4(ACC0, ACC1, WA, WB, WC, WD, IA, IB, IC, ID)                                 \
    asm volatile("v_pk_mul_f32 %0, %2, %6 op_sel:[0,0] op_sel_hi:[1,0]\n"                     \
                 "v_pk_mul_f32 %1, %2, %6 op_sel:[0,1] op_sel_hi:[1,1]\n" RF_PK4_TAIL         \
                 : "=&v"(ACC0), "=&v"(ACC1)                                                   \
                 : "s"(WA), "s"(WB), "s"(WC), "s"(WD), "v"(IA), "v"(IB), "v"(IC), "v"(ID))

// A 32-input layer: in[k] = {px0, px1} of input channel k, out[c] likewise.  rec = the layer's 16
// records of kRec floats (buffer A: k = 0..15 of a channel pair, buffer B: k = 16..31 + biases).
// Bias + ReLU of channel pair o (2 packed adds, 4 max: "simple" 2-cycle instructions) are spread
// over the FMA stream of pair o + 1, where each rides in the issue window of a packed FMA
// (tools/microbench/valu_rates2.hip) instead of costing 16 cycles in a row after every pair;
// accumulators and biases alternate between two register sets by the parity of o.
__device__ __forceinline__ void layer32_regs(const float *__restrict__ rec, const float2v (&in)[32],
                                             float2v (&out)[32])
{
    float16v a0, a1, b0, b1;
    float2v accE0, accE1, accO0, accO1;  // {channel 2o, 2o+1} of pixel 0 / pixel 1, even / odd o
    float2v biasE, biasO;
    float2v tf0, tf1;                    // accumulators + bias of the pair being finished
#define RF_SLOAD16(DST, PTR, OFF) \
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(DST) : "s"(PTR), "n"(OFF))
#define RF_SLOAD2(DST, PTR, OFF) \
    asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(DST) : "s"(PTR), "n"(OFF))
    // one sixth of the finish of pair PO (accumulators PA0 / PA1, biases PB)
#define RF_FIN(STEP, PO, PA0, PA1, PB)                                                       \
    do {                                                                                     \
        if ((STEP) == 0)                                                                     \
            tf0 = PA0 + PB;                                                                  \
        if ((STEP) == 1)                                                                     \
            tf1 = PA1 + PB;                                                                  \
        if ((STEP) == 2)                                                                     \
            out[2 * (PO)].x = fmaxf(tf0.x, 0.f);                                             \
        if ((STEP) == 3)                                                                     \
            out[2 * (PO)].y = fmaxf(tf1.x, 0.f);                                             \
        if ((STEP) == 4)                                                                     \
            out[2 * (PO) + 1].x = fmaxf(tf0.y, 0.f);                                         \
        if ((STEP) == 5)                                                                     \
            out[2 * (PO) + 1].y = fmaxf(tf1.y, 0.f);                                         \
    } while (0)
    // 16 inputs of a pair in four statements of four k-steps; FIRST: the chain starts here; FIN0:
    // first finish step placed in this half (one after each of the first three statements)
#define RF_W2(W, K) float2v{W[2 * (K)], W[2 * (K) + 1]}
#define RF_FMA16R(ACC0, ACC1, W0, W1, KBASE, FIRST, DOFIN, FIN0, PO, PA0, PA1, PB)           \
    if (FIRST)                                                                               \
        RF_PKMUL4(ACC0, ACC1, RF_W2(W0, 0), RF_W2(W0, 1), RF_W2(W0, 2), RF_W2(W0, 3),        \
                  in[(KBASE)], in[(KBASE) + 1], in[(KBASE) + 2], in[(KBASE) + 3]);           \
    else                                                                                     \
        RF_PKFMA4(ACC0, ACC1, RF_W2(W0, 0), RF_W2(W0, 1), RF_W2(W0, 2), RF_W2(W0, 3),        \
                  in[(KBASE)], in[(KBASE) + 1], in[(KBASE) + 2], in[(KBASE) + 3]);           \
    if (DOFIN)                                                                               \
        RF_FIN((FIN0), PO, PA0, PA1, PB);                                                    \
    RF_PKFMA4(ACC0, ACC1, RF_W2(W0, 4), RF_W2(W0, 5), RF_W2(W0, 6), RF_W2(W0, 7),            \
              in[(KBASE) + 4], in[(KBASE) + 5], in[(KBASE) + 6], in[(KBASE) + 7]);           \
    if (DOFIN)                                                                               \
        RF_FIN((FIN0) + 1, PO, PA0, PA1, PB);                                                \
    RF_PKFMA4(ACC0, ACC1, RF_W2(W1, 0), RF_W2(W1, 1), RF_W2(W1, 2), RF_W2(W1, 3),            \
              in[(KBASE) + 8], in[(KBASE) + 9], in[(KBASE) + 10], in[(KBASE) + 11]);         \
    if (DOFIN)                                                                               \
        RF_FIN((FIN0) + 2, PO, PA0, PA1, PB);                                                \
    RF_PKFMA4(ACC0, ACC1, RF_W2(W1, 4), RF_W2(W1, 5), RF_W2(W1, 6), RF_W2(W1, 7),            \
              in[(KBASE) + 12], in[(KBASE) + 13], in[(KBASE) + 14], in[(KBASE) + 15]);
    RF_SLOAD16(a0, rec, 0);
    RF_SLOAD16(a1, rec, 64);
    // pair OP accumulates in CUR*, pair OP - 1 (PRV*) is finished meanwhile
#define RF_OP(OP, CUR0, CUR1, CURB, PRV0, PRV1, PRVB)                                        \
    {                                                                                        \
        constexpr int cur_ = (OP) * kRec * 4;                                                \
        constexpr int nxt_ = ((OP) < 15 ? (OP) + 1 : (OP)) * kRec * 4;                       \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));                           \
        RF_SLOAD16(b0, rec, cur_ + 128);                                                     \
        RF_SLOAD16(b1, rec, cur_ + 192);                                                     \
        RF_SLOAD2(CURB, rec, cur_ + 256);                                                    \
        RF_FMA16R(CUR0, CUR1, a0, a1, 0, true, (OP) > 0, 0, (OP) - 1, PRV0, PRV1, PRVB)      \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b0), "+s"(b1), "+s"(CURB));               \
        RF_SLOAD16(a0, rec, nxt_);                                                           \
        RF_SLOAD16(a1, rec, nxt_ + 64);                                                      \
        RF_FMA16R(CUR0, CUR1, b0, b1, 16, false, (OP) > 0, 3, (OP) - 1, PRV0, PRV1, PRVB)    \
    }
#define RF_OP_E(OP) RF_OP(OP, accE0, accE1, biasE, accO0, accO1, biasO)
#define RF_OP_O(OP) RF_OP(OP, accO0, accO1, biasO, accE0, accE1, biasE)
    RF_OP_E(0) RF_OP_O(1) RF_OP_E(2) RF_OP_O(3) RF_OP_E(4) RF_OP_O(5) RF_OP_E(6) RF_OP_O(7)
    RF_OP_E(8) RF_OP_O(9) RF_OP_E(10) RF_OP_O(11) RF_OP_E(12) RF_OP_O(13) RF_OP_E(14) RF_OP_O(15)
#undef RF_OP_O
#undef RF_OP_E
#undef RF_OP
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a0), "+s"(a1));
#pragma unroll
    for (int st = 0; st < 6; st++)
        RF_FIN(st, 15, accO0, accO1, biasO);
#undef RF_FMA16R
#undef RF_W2
#undef RF_FIN
#undef RF_SLOAD2
#undef RF_SLOAD16
}

// 32 terms of the fuse dot product for both pixels: one packed FMA per term, its weight operand
// the pair {wf[k], wf[k]} of the packed stream (no operand re-routing: an SGPR source with op_sel is
// the form that is wrong on gfx950, DESIGN.md 3.3, which is why the stream holds these twice).
// Eight terms per asm statement (see RF_PKFMA4 for why), volatile so that hipcc does not spread
// them over the neighbouring layer and spill the SGPR file.
__device__ __forceinline__ void fuse32(const float2v (&act)[32], const float *__restrict__ wf64,
                                       float2v &z)
{
    float16v w0, w1, w2, w3;
    asm volatile("s_load_dwordx16 %0, %1, 0" : "=s"(w0) : "s"(wf64));
    asm volatile("s_load_dwordx16 %0, %1, 64" : "=s"(w1) : "s"(wf64));
    asm volatile("s_load_dwordx16 %0, %1, 128" : "=s"(w2) : "s"(wf64));
    asm volatile("s_load_dwordx16 %0, %1, 192" : "=s"(w3) : "s"(wf64));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3));
#define RF_W2(W, K) float2v{W[2 * (K)], W[2 * (K) + 1]}
#define RF_FUSE8(W, KBASE)                                                                    \
    asm volatile("v_pk_fma_f32 %0, %1, %9, %0\n"                                              \
                 "v_pk_fma_f32 %0, %2, %10, %0\n"                                             \
                 "v_pk_fma_f32 %0, %3, %11, %0\n"                                             \
                 "v_pk_fma_f32 %0, %4, %12, %0\n"                                             \
                 "v_pk_fma_f32 %0, %5, %13, %0\n"                                             \
                 "v_pk_fma_f32 %0, %6, %14, %0\n"                                             \
                 "v_pk_fma_f32 %0, %7, %15, %0\n"                                             \
                 "v_pk_fma_f32 %0, %8, %16, %0"                                               \
                 : "+v"(z)                                                                    \
                 : "s"(RF_W2(W, 0)), "s"(RF_W2(W, 1)), "s"(RF_W2(W, 2)), "s"(RF_W2(W, 3)),    \
                   "s"(RF_W2(W, 4)), "s"(RF_W2(W, 5)), "s"(RF_W2(W, 6)), "s"(RF_W2(W, 7)),    \
                   "v"(act[(KBASE)]), "v"(act[(KBASE) + 1]), "v"(act[(KBASE) + 2]),           \
                   "v"(act[(KBASE) + 3]), "v"(act[(KBASE) + 4]), "v"(act[(KBASE) + 5]),       \
                   "v"(act[(KBASE) + 6]), "v"(act[(KBASE) + 7]))
    RF_FUSE8(w0, 0);
    RF_FUSE8(w1, 8);
    RF_FUSE8(w2, 16);
    RF_FUSE8(w3, 24);
#undef RF_FUSE8
#undef RF_W2
}

__global__ __launch_bounds__(kCnnThreads) void cnn_reflectance_regs_kernel(
    const uint8_t *__restrict__ bgr, float *__restrict__ r_out, uint8_t *__restrict__ r_u8_out,
    size_t npix, const float *__restrict__ packed, const float *__restrict__ srgb_lut)
{
    __shared__ float lut[256];
    for (int i = threadIdx.x; i < 256; i += kCnnThreads)
        lut[i] = srgb_lut[i];
    __syncthreads();
    const float *wf = packed + kWfOff;
    const size_t stride = (size_t)gridDim.x * kCnnThreads;
    const size_t half = (npix + 1) / 2;  // lane handles pixels i and i + half (both stay coalesced)
    for (size_t i = (size_t)blockIdx.x * kCnnThreads + threadIdx.x; i < half; i += stride) {
        const size_t idx[kPxPerLane] = {i, i + half};
        float2v x[3];
        {
            const uint8_t *p0 = bgr + idx[0] * 3;
            const uint8_t *p1 = bgr + (idx[1] < npix ? idx[1] : npix - 1) * 3;
            x[0] = float2v{lut[p0[2]], lut[p1[2]]};  // blob channel order is RGB
            x[1] = float2v{lut[p0[1]], lut[p1[1]]};
            x[2] = float2v{lut[p0[0]], lut[p1[0]]};
        }
        float2v A[32], B[32];
        float2v z = float2v{0.f, 0.f};  // the fuse dot product of {pixel 0, pixel 1}
        // layer 0: 16 channel pairs x 3 inputs, weights {W0[2o][k], W0[2o+1][k]} at packed[6 o + 2 k],
        // biases at packed[96 + c]; two halves of 8 pairs, each with its own explicit scalar loads
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            float16v w0, w1, w2v, bb;
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(w0) : "s"(packed + 48 * hh), "n"(0));
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(w1) : "s"(packed + 48 * hh), "n"(64));
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(w2v) : "s"(packed + 48 * hh), "n"(128));
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(bb) : "s"(packed + 96 + 16 * hh), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w0), "+s"(w1), "+s"(w2v), "+s"(bb));
#pragma unroll
            for (int o8 = 0; o8 < 8; o8++) {
                float2v acc0 = float2v{0.f, 0.f}, acc1 = float2v{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int e = 6 * o8 + 2 * k;  // element of the 48-float half
                    const float2v wp = e < 16   ? float2v{w0[e & 15], w0[(e & 15) + 1]}
                                       : e < 32 ? float2v{w1[e & 15], w1[(e & 15) + 1]}
                                                : float2v{w2v[e & 15], w2v[(e & 15) + 1]};
                    RF_PKFMA_P0(acc0, wp, x[k]);
                    RF_PKFMA_P1(acc1, wp, x[k]);
                }
                const int op = 8 * hh + o8;
                finish_pair(acc0, acc1, bb[2 * o8], bb[2 * o8 + 1], A[2 * op], A[2 * op + 1]);
            }
        }
        fuse32(A, wf, z);
        // the four 32 -> 32 layers ping-pong between A and B: no copies, and one rolled loop of two
        // iterations keeps hipcc from giving every layer's outputs registers of their own
#pragma unroll 1
        for (int l = 0; l < 4; l += 2) {
            layer32_regs(packed + 128 + l * 1056, A, B);
            fuse32(B, wf + 64 * (l + 1), z);
            layer32_regs(packed + 128 + (l + 1) * 1056, B, A);
            fuse32(A, wf + 64 * (l + 2), z);
        }
#pragma unroll
        for (int p = 0; p < kPxPerLane; p++) {
            if (idx[p] >= npix)
                continue;
            const float zz = __fadd_rn(p == 0 ? z.x : z.y, wf[320]);
            // caffe: 1. / (1. + exp(-x)) with a float exp; expf modelled as round(exp in double)
            const float e = (float)exp((double)(-zz));
            const float r = (float)(1.0 / (1.0 + (double)e));
            if (r_out)
                r_out[idx[p]] = r;
            if (r_u8_out)
                r_u8_out[idx[p]] = (uint8_t)__fmul_rn(r, 255.0f);  // astype(uint8): truncation
        }
    }
}
#undef RF_PKFMA_P0
#undef RF_PKFMA_P1
#undef RF_PKFMA4
#undef RF_PKMUL4
#undef RF_PK4_TAIL

// rf_cnn_reflectance_u8 (raw weights) keeps one packed copy per (device, stream): a call re-packs
// the caller's weights on its own stream (18 tiny workgroups; the weights may have changed since
// the last call), so calls on different streams never share a buffer and calls on one stream are
// ordered.  The table is bounded (kMaxSlots): when it is full the least recently used slot whose
// stream has drained is recycled (a stream that no longer exists counts as drained; if every slot
// is busy the oldest one's stream is waited for).  A slot is handed out with the table's mutex
// HELD and the caller enqueues its two kernels before releasing it, so no other thread can see
// the slot's stream as drained in between and take the buffer.  A stream that is being captured
// is refused: a graph would bake a slot's pointer in and replay it after the slot has gone to
// another stream - graphs use the stateless pair rf_cnn_pack_weights + rf_cnn_reflectance_packed_u8.
struct PackedSlot {
    int device;
    hipStream_t stream;
    float *buf;
    unsigned long long used;  // tick of the last call
};
constexpr size_t kMaxSlots = 16;
std::mutex g_cnn_mu;
std::vector<PackedSlot> g_packed;
unsigned long long g_tick = 0;

int packed_buffer(hipStream_t stream, float **out, std::unique_lock<std::mutex> &lock)
{
    int dev = 0;
    RF_HIP_CHECK(hipGetDevice(&dev));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess)
        (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone)
        return fail(RF_E_UNSUPPORTED,
                    "rf_cnn_reflectance_u8: the stream is being captured; this entry point keeps its "
                    "packed weights in a per-stream slot of the library that a later call may hand "
                    "to another stream - capture rf_cnn_reflectance_packed_u8 on weights packed "
                    "beforehand with rf_cnn_pack_weights");
    lock = std::unique_lock<std::mutex>(g_cnn_mu);  // held until the caller has enqueued its work
    for (PackedSlot &s : g_packed)
        if (s.device == dev && s.stream == stream) {
            s.used = ++g_tick;
            *out = s.buf;
            return RF_OK;
        }
    if (g_packed.size() >= kMaxSlots) {
        // recycle: least recently used slot whose stream has nothing pending
        size_t pick = g_packed.size(), oldest = 0;
        for (size_t i = 0; i < g_packed.size(); i++) {
            if (g_packed[i].used < g_packed[oldest].used)
                oldest = i;
            const hipError_t q = hipStreamQuery(g_packed[i].stream);
            if (q == hipErrorNotReady)
                continue;
            if (q != hipSuccess)
                (void)hipGetLastError();  // the stream is gone: nothing of it can be in flight
            if (pick == g_packed.size() || g_packed[i].used < g_packed[pick].used)
                pick = i;
        }
        if (pick == g_packed.size()) {
            pick = oldest;
            if (hipStreamSynchronize(g_packed[pick].stream) != hipSuccess)
                (void)hipGetLastError();
        }
        PackedSlot &s = g_packed[pick];
        if (s.device != dev) {
            (void)hipFree(s.buf);
            s.buf = nullptr;
            float *buf = nullptr;
            const hipError_t e = hipMalloc(&buf, sizeof(float) * kPackedFloats);
            if (e != hipSuccess) {
                g_packed.erase(g_packed.begin() + (long)pick);
                return fail(RF_E_HIP, "hipMalloc of the packed weights failed: %s",
                            hipGetErrorString(e));
            }
            s.buf = buf;
        }
        s.device = dev;
        s.stream = stream;
        s.used = ++g_tick;
        *out = s.buf;
        return RF_OK;
    }
    float *buf = nullptr;
    RF_HIP_CHECK(hipMalloc(&buf, sizeof(float) * kPackedFloats));
    g_packed.push_back({dev, stream, buf, ++g_tick});
    *out = buf;
    return RF_OK;
}

int launch_forward(const uint8_t *bgr, float *r_out, uint8_t *r_u8_out, int n, int h, int w,
                   const float *packed, const float *srgb_lut, hipStream_t stream)
{
    const size_t npix = (size_t)n * h * w;
    size_t blocks = ((npix + 1) / 2 + kCnnThreads - 1) / kCnnThreads;
    // grid-stride loop over at most 128 workgroups per CU: a grid that just fills the chip (6 per
    // CU) is 8 % slower than 20 per CU, 80-160 per CU another 1.5 % faster (workgroups that end at
    // different times keep the SIMDs' waves out of phase); one iteration per workgroup pays the
    // sRGB table's load every time
    if (blocks > 256 * 128)
        blocks = 256 * 128;
    if (debug_get(kDbgCnnLdsColumns))  // cross-check: the LDS-column form of round 1
        hipLaunchKernelGGL(cnn_reflectance_kernel, dim3((unsigned)blocks), dim3(kCnnThreads), 0,
                           stream, bgr, r_out, r_u8_out, npix, packed, srgb_lut);
    else
        hipLaunchKernelGGL(cnn_reflectance_regs_kernel, dim3((unsigned)blocks), dim3(kCnnThreads),
                           0, stream, bgr, r_out, r_u8_out, npix, packed, srgb_lut);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

int check_forward_args(const char *who, const void *bgr, const void *weights, const void *srgb_lut,
                       const void *r_out, const void *r_u8_out, int n, int h, int w)
{
    if (!bgr || !weights || !srgb_lut || (!r_out && !r_u8_out))
        return fail(RF_E_BADARG, "%s: NULL pointer", who);
    if (n < 0 || h <= 0 || w <= 0)
        return fail(RF_E_BADARG, "%s: bad size n=%d h=%d w=%d", who, n, h, w);
    return RF_OK;
}

}  // namespace

void cnn_shutdown()
{
    std::lock_guard<std::mutex> lock(g_cnn_mu);
    for (const PackedSlot &s : g_packed)
        (void)hipFree(s.buf);
    g_packed.clear();
}

}  // namespace rf

extern "C" int rf_cnn_pack_weights(const float *weights, float *packed, void *stream_)
{
    using namespace rf;
    static_assert(kPackedFloats == RF_CNN_NPACKED, "the size the header promises");
    if (!weights || !packed)
        return fail(RF_E_BADARG, "rf_cnn_pack_weights: NULL pointer");
    if (ranges_overlap(weights, sizeof(float) * RF_CNN_NPARAMS, packed,
                       sizeof(float) * RF_CNN_NPACKED))
        return fail(RF_E_BADARG, "rf_cnn_pack_weights: packed must not overlap weights");
    hipLaunchKernelGGL(cnn_pack_weights_kernel, dim3((RF_CNN_NPACKED + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream_, weights, packed);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

extern "C" int rf_cnn_reflectance_packed_u8(const uint8_t *bgr, float *r_out, uint8_t *r_u8_out,
                                            int n, int h, int w, const float *packed,
                                            const float *srgb_lut, void *stream_)
{
    using namespace rf;
    if (n == 0)  // an empty batch is valid whatever the (possibly NULL) pointers are
        return RF_OK;
    const int rc = check_forward_args("rf_cnn_reflectance_packed_u8", bgr, packed, srgb_lut, r_out,
                                      r_u8_out, n, h, w);
    if (rc != RF_OK)
        return rc;
    return launch_forward(bgr, r_out, r_u8_out, n, h, w, packed, srgb_lut, (hipStream_t)stream_);
}

extern "C" int rf_cnn_reflectance_u8(const uint8_t *bgr, float *r_out, uint8_t *r_u8_out, int n,
                                     int h, int w, const float *weights, const float *srgb_lut,
                                     void *stream_)
{
    using namespace rf;
    if (n == 0)
        return RF_OK;
    int rc = check_forward_args("rf_cnn_reflectance_u8", bgr, weights, srgb_lut, r_out, r_u8_out,
                                n, h, w);
    if (rc != RF_OK)
        return rc;
    hipStream_t stream = (hipStream_t)stream_;
    float *packed = nullptr;
    std::unique_lock<std::mutex> slot_lock;  // released when both kernels are enqueued
    rc = packed_buffer(stream, &packed, slot_lock);
    if (rc != RF_OK)
        return rc;
    // (the packed copy is rebuilt on the caller's stream every call: the weights may have changed)
    hipLaunchKernelGGL(cnn_pack_weights_kernel, dim3((RF_CNN_NPACKED + 255) / 256), dim3(256), 0,
                       stream, weights, packed);
    return launch_forward(bgr, r_out, r_u8_out, n, h, w, packed, srgb_lut, stream);
}
