// pipe_overlap.hip -- do gfx950's four-cycle ("full-pipe") and two-cycle ("simple") VALU instructions overlap,
// and does it take interleaving WITHIN a wave or do different waves of a SIMD fill each other's gaps?
// (follow-up of crosslane_rates.hip; decides whether stage 1 of the guided filter, whose compiler-scheduled
// stream clusters 78 DPP adds, 39 v_fma_f64, 27 multiply-adds ..., has issue slots to win)
// original header of the template follows:
// crosslane_rates.hip -- issue cost of the cross-lane candidates for the guided filter's stage-1 prefix
// scan (gfx950): DPP adds / moves by control, v_readlane, ds_swizzle, ds_bpermute, the gfx950 permlane
// swaps, and the fp64 / conversion instructions of the window means.  8 independent chains per wave, so
// a figure is an ISSUE cost, not a dependent latency (waves/SIMD = 1 shows the latency-bound end).
// Build: hipcc -O3 --offload-arch=gfx950 crosslane_rates.hip -o crosslane_rates.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                           \
    do {                                                                   \
        hipError_t e = (x);                                                \
        if (e != hipSuccess) {                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));         \
            exit(1);                                                       \
        }                                                                  \
    } while (0)

constexpr int kIters = 4096 * 32;

#define OPS16(OP) OP(r0) OP(r0) OP(r0) OP(r0)

#define KERNEL(NAME, ASM)                                                                  \
    __global__ void NAME(unsigned *out, int iters, unsigned long long *clk)                \
    {                                                                                      \
        extern __shared__ unsigned dyn_lds[];                                              \
        if (iters < 0)                                                                     \
            dyn_lds[threadIdx.x] = 1;                                                      \
        unsigned r0 = threadIdx.x * 2654435761u, r1 = r0 ^ 0x55, r2 = r0 + 77, r3 = r0 * 3, \
                 r4 = r0 + 5, r5 = r0 ^ 9, r6 = r0 + 11, r7 = r0 * 7;                       \
        const unsigned a = threadIdx.x | 0x01020304u, b = 0x3f800001u;                     \
        double dr0 = r0, dr1 = r1, dr2 = r2, dr3 = r3, dr4 = r4, dr5 = r5, dr6 = r6, dr7 = r7;  \
        const double db = 1.0000001;                                                       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        unsigned long long q0 = __builtin_amdgcn_s_memrealtime();                          \
        for (int it = 0; it < iters; it++) {                                               \
            OPS16(ASM)                                                                     \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        unsigned long long q1 = __builtin_amdgcn_s_memrealtime();                          \
        if (blockIdx.x == 0 && threadIdx.x == 0 && clk) {                                  \
            clk[0] = t1 - t0;                                                              \
            clk[1] = q1 - q0;                                                              \
        }                                                                                  \
        unsigned r = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 +                               \
                     (unsigned)(dr0 + dr1 + dr2 + dr3 + dr4 + dr5 + dr6 + dr7);            \
        if (r == 0x12345678u)                                                              \
            out[threadIdx.x] = r;                                                          \
    }



#define F_SAD(x) "v_sad_u8 %" #x ", %" #x ", %8, 0\n"
#define S_MUL(x) "v_mul_f32 %" #x ", %9, %" #x "\n"
#define F_DPP(x) "v_add_u32_dpp %" #x ", %" #x ", %" #x " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define S_SUB(x) "v_sub_u32 %" #x ", %8, %" #x "\n"
#define F_MAD(x) "v_mad_u32_u24 %" #x ", %" #x ", %8, %8\n"
#define F_FMA(x) "v_fma_f32 %" #x ", %" #x ", %9, %9\n"
#define REGS "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)
// 16 instructions per OP: 8 full-pipe + 8 simple, interleaved one for one or clustered 8 + 8
#define A_SADMUL_INTER(unused_) asm volatile(F_SAD(0) S_MUL(1) F_SAD(2) S_MUL(3) F_SAD(4) S_MUL(5) F_SAD(6) S_MUL(7) F_SAD(1) S_MUL(0) F_SAD(3) S_MUL(2) F_SAD(5) S_MUL(4) F_SAD(7) S_MUL(6) : REGS);
#define A_SADMUL_CLUST(unused_) asm volatile(F_SAD(0) F_SAD(1) F_SAD(2) F_SAD(3) F_SAD(4) F_SAD(5) F_SAD(6) F_SAD(7) S_MUL(0) S_MUL(1) S_MUL(2) S_MUL(3) S_MUL(4) S_MUL(5) S_MUL(6) S_MUL(7) : REGS);
#define A_DPPSUB_INTER(unused_) asm volatile("s_nop 1\n" F_DPP(0) S_SUB(1) F_DPP(2) S_SUB(3) F_DPP(4) S_SUB(5) F_DPP(6) S_SUB(7) F_DPP(1) S_SUB(0) F_DPP(3) S_SUB(2) F_DPP(5) S_SUB(4) F_DPP(7) S_SUB(6) : REGS);
#define A_DPPSUB_CLUST(unused_) asm volatile("s_nop 1\n" F_DPP(0) F_DPP(1) F_DPP(2) F_DPP(3) F_DPP(4) F_DPP(5) F_DPP(6) F_DPP(7) S_SUB(0) S_SUB(1) S_SUB(2) S_SUB(3) S_SUB(4) S_SUB(5) S_SUB(6) S_SUB(7) : REGS);
#define A_MADMUL_INTER(unused_) asm volatile(F_MAD(0) S_MUL(1) F_MAD(2) S_MUL(3) F_MAD(4) S_MUL(5) F_MAD(6) S_MUL(7) F_MAD(1) S_MUL(0) F_MAD(3) S_MUL(2) F_MAD(5) S_MUL(4) F_MAD(7) S_MUL(6) : REGS);
#define A_FMAMUL_INTER(unused_) asm volatile(F_FMA(0) S_MUL(1) F_FMA(2) S_MUL(3) F_FMA(4) S_MUL(5) F_FMA(6) S_MUL(7) F_FMA(1) S_MUL(0) F_FMA(3) S_MUL(2) F_FMA(5) S_MUL(4) F_FMA(7) S_MUL(6) : REGS);
#define A_FMAMUL_CLUST(unused_) asm volatile(F_FMA(0) F_FMA(1) F_FMA(2) F_FMA(3) F_FMA(4) F_FMA(5) F_FMA(6) F_FMA(7) S_MUL(0) S_MUL(1) S_MUL(2) S_MUL(3) S_MUL(4) S_MUL(5) S_MUL(6) S_MUL(7) : REGS);

#define F_LSHLADD(x) "v_lshl_add_u32 %" #x ", %" #x ", 1, %8\n"
#define F_ADD3(x) "v_add3_u32 %" #x ", %" #x ", %8, %8\n"
#define F_FIXUP(x) "v_div_fixup_f32 %" #x ", %" #x ", %9, %9\n"
#define F_CVTUB(x) "v_cvt_f32_ubyte0 %" #x ", %" #x "\n"
#define A_LSHLMUL_INTER(unused_) asm volatile(F_LSHLADD(0) S_MUL(1) F_LSHLADD(2) S_MUL(3) F_LSHLADD(4) S_MUL(5) F_LSHLADD(6) S_MUL(7) F_LSHLADD(1) S_MUL(0) F_LSHLADD(3) S_MUL(2) F_LSHLADD(5) S_MUL(4) F_LSHLADD(7) S_MUL(6) : REGS);
#define A_ADD3MUL_INTER(unused_) asm volatile(F_ADD3(0) S_MUL(1) F_ADD3(2) S_MUL(3) F_ADD3(4) S_MUL(5) F_ADD3(6) S_MUL(7) F_ADD3(1) S_MUL(0) F_ADD3(3) S_MUL(2) F_ADD3(5) S_MUL(4) F_ADD3(7) S_MUL(6) : REGS);
#define A_FIXUPMUL_INTER(unused_) asm volatile(F_FIXUP(0) S_MUL(1) F_FIXUP(2) S_MUL(3) F_FIXUP(4) S_MUL(5) F_FIXUP(6) S_MUL(7) F_FIXUP(1) S_MUL(0) F_FIXUP(3) S_MUL(2) F_FIXUP(5) S_MUL(4) F_FIXUP(7) S_MUL(6) : REGS);
#define A_CVTUBMUL_INTER(unused_) asm volatile(F_CVTUB(0) S_MUL(1) F_CVTUB(2) S_MUL(3) F_CVTUB(4) S_MUL(5) F_CVTUB(6) S_MUL(7) F_CVTUB(1) S_MUL(0) F_CVTUB(3) S_MUL(2) F_CVTUB(5) S_MUL(4) F_CVTUB(7) S_MUL(6) : REGS);
// fp64: 4 double chains + 8 float chains; 8 v_fma_f64 (or conversions) + 8 v_mul_f32 per OP
#define DREGS "+v"(dr0), "+v"(dr1), "+v"(dr2), "+v"(dr3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(db), "v"(b)
#define F_FMA64(x) "v_fma_f64 %" #x ", %" #x ", %8, %8\n"
#define S_MULB(x) "v_mul_f32 %" #x ", %9, %" #x "\n"
#define A_FMA64MUL_INTER(unused_) asm volatile(F_FMA64(0) S_MULB(4) F_FMA64(1) S_MULB(5) F_FMA64(2) S_MULB(6) F_FMA64(3) S_MULB(7) F_FMA64(0) S_MULB(4) F_FMA64(1) S_MULB(5) F_FMA64(2) S_MULB(6) F_FMA64(3) S_MULB(7) : DREGS);
#define A_FMA64MUL_CLUST(unused_) asm volatile(F_FMA64(0) F_FMA64(1) F_FMA64(2) F_FMA64(3) F_FMA64(0) F_FMA64(1) F_FMA64(2) F_FMA64(3) S_MULB(4) S_MULB(5) S_MULB(6) S_MULB(7) S_MULB(4) S_MULB(5) S_MULB(6) S_MULB(7) : DREGS);
#define F_CVT64(x, y) "v_cvt_f32_f64 %" #y ", %" #x "\n"
#define A_CVT64MUL_INTER(unused_) asm volatile(F_CVT64(0, 4) S_MULB(5) F_CVT64(1, 6) S_MULB(7) F_CVT64(2, 4) S_MULB(5) F_CVT64(3, 6) S_MULB(7) F_CVT64(0, 4) S_MULB(5) F_CVT64(1, 6) S_MULB(7) F_CVT64(2, 4) S_MULB(5) F_CVT64(3, 6) S_MULB(7) : DREGS);

// a multiply whose weight comes from an SGPR (wave-uniform operand) next to a two-cycle add
#define SREGS "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(b), "v"(b)
#define F_MULS(x) "v_mul_f32 %" #x ", %8, %" #x "\n"
#define S_ADDV(x) "v_add_f32 %" #x ", %9, %" #x "\n"
#define A_MULSGPR_ADD_INTER(unused_) asm volatile(F_MULS(0) S_ADDV(1) F_MULS(2) S_ADDV(3) F_MULS(4) S_ADDV(5) F_MULS(6) S_ADDV(7) F_MULS(1) S_ADDV(0) F_MULS(3) S_ADDV(2) F_MULS(5) S_ADDV(4) F_MULS(7) S_ADDV(6) : SREGS);
#define A_MULSGPR_ONLY(unused_) asm volatile(F_MULS(0) F_MULS(1) F_MULS(2) F_MULS(3) F_MULS(4) F_MULS(5) F_MULS(6) F_MULS(7) F_MULS(0) F_MULS(1) F_MULS(2) F_MULS(3) F_MULS(4) F_MULS(5) F_MULS(6) F_MULS(7) : SREGS);
// ... and next to a v_sad_u8 (two "four-cycle" instructions of the hiding kind side by side)
#define F_SADV(x) "v_sad_u8 %" #x ", %" #x ", %9, 0\n"
#define A_MULSGPR_SAD_INTER(unused_) asm volatile(F_MULS(0) F_SADV(1) F_MULS(2) F_SADV(3) F_MULS(4) F_SADV(5) F_MULS(6) F_SADV(7) F_MULS(1) F_SADV(0) F_MULS(3) F_SADV(2) F_MULS(5) F_SADV(4) F_MULS(7) F_SADV(6) : SREGS);
// 64 + 64: clusters as long as stage 1's (the 78 DPP adds of its scan)
#define A_SADMUL_CLUST64(unused_) asm volatile(F_SAD(0) F_SAD(1) F_SAD(2) F_SAD(3) F_SAD(4) F_SAD(5) F_SAD(6) F_SAD(7) : REGS);
// waves of even index run only full-pipe instructions, odd ones only simple ones (16 each per OP)
#define A_SPLIT_WAVES(unused_) if ((threadIdx.x >> 6) & 1) { asm volatile(S_MUL(0) S_MUL(1) S_MUL(2) S_MUL(3) S_MUL(4) S_MUL(5) S_MUL(6) S_MUL(7) S_MUL(0) S_MUL(1) S_MUL(2) S_MUL(3) S_MUL(4) S_MUL(5) S_MUL(6) S_MUL(7) : REGS); } else { asm volatile(F_SAD(0) F_SAD(1) F_SAD(2) F_SAD(3) F_SAD(4) F_SAD(5) F_SAD(6) F_SAD(7) F_SAD(0) F_SAD(1) F_SAD(2) F_SAD(3) F_SAD(4) F_SAD(5) F_SAD(6) F_SAD(7) : REGS); }
#define A_ONLY_SAD16(unused_) asm volatile(F_SAD(0) F_SAD(1) F_SAD(2) F_SAD(3) F_SAD(4) F_SAD(5) F_SAD(6) F_SAD(7) F_SAD(0) F_SAD(1) F_SAD(2) F_SAD(3) F_SAD(4) F_SAD(5) F_SAD(6) F_SAD(7) : REGS);
#define A_ONLY_MUL16(unused_) asm volatile(S_MUL(0) S_MUL(1) S_MUL(2) S_MUL(3) S_MUL(4) S_MUL(5) S_MUL(6) S_MUL(7) S_MUL(0) S_MUL(1) S_MUL(2) S_MUL(3) S_MUL(4) S_MUL(5) S_MUL(6) S_MUL(7) : REGS);

// 24-bit integer multiplies (stage 1's products: one v_mad_i32_i24 each, or a multiply and an add?)
#define F_MUL24(x) "v_mul_u32_u24 %" #x ", %" #x ", %8\n"
#define F_MULI24(x) "v_mul_i32_i24 %" #x ", %" #x ", %8\n"
#define S_ADDU(x) "v_add_u32 %" #x ", %8, %" #x "\n"
#define A_MUL24MUL_INTER(unused_) asm volatile(F_MUL24(0) S_MUL(1) F_MUL24(2) S_MUL(3) F_MUL24(4) S_MUL(5) F_MUL24(6) S_MUL(7) F_MUL24(1) S_MUL(0) F_MUL24(3) S_MUL(2) F_MUL24(5) S_MUL(4) F_MUL24(7) S_MUL(6) : REGS);
#define A_MULI24ADD_INTER(unused_) asm volatile(F_MULI24(0) S_ADDU(1) F_MULI24(2) S_ADDU(3) F_MULI24(4) S_ADDU(5) F_MULI24(6) S_ADDU(7) F_MULI24(1) S_ADDU(0) F_MULI24(3) S_ADDU(2) F_MULI24(5) S_ADDU(4) F_MULI24(7) S_ADDU(6) : REGS);
#define A_ONLY_MUL24(unused_) asm volatile(F_MUL24(0) F_MUL24(1) F_MUL24(2) F_MUL24(3) F_MUL24(4) F_MUL24(5) F_MUL24(6) F_MUL24(7) F_MUL24(0) F_MUL24(1) F_MUL24(2) F_MUL24(3) F_MUL24(4) F_MUL24(5) F_MUL24(6) F_MUL24(7) : REGS);
#define A_ONLY_MAD24(unused_) asm volatile(F_MAD(0) F_MAD(1) F_MAD(2) F_MAD(3) F_MAD(4) F_MAD(5) F_MAD(6) F_MAD(7) F_MAD(0) F_MAD(1) F_MAD(2) F_MAD(3) F_MAD(4) F_MAD(5) F_MAD(6) F_MAD(7) : REGS);
// the move form of a DPP step (v_mov_b32_dpp + a plain add instead of v_add_u32_dpp)
#define F_DPPMOV(x, y) "v_mov_b32_dpp %" #y ", %" #x " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define A_DPPMOVSUB_INTER(unused_) asm volatile("s_nop 1\n" F_DPPMOV(0, 1) S_SUB(2) F_DPPMOV(2, 3) S_SUB(4) F_DPPMOV(4, 5) S_SUB(6) F_DPPMOV(6, 7) S_SUB(0) F_DPPMOV(1, 0) S_SUB(3) F_DPPMOV(3, 2) S_SUB(5) F_DPPMOV(5, 4) S_SUB(7) F_DPPMOV(7, 6) S_SUB(1) : REGS);
KERNEL(k_mul24_mul_interleaved, A_MUL24MUL_INTER)
KERNEL(k_muli24_addu_interleaved, A_MULI24ADD_INTER)
KERNEL(k_only_mul24, A_ONLY_MUL24)
KERNEL(k_only_mad24, A_ONLY_MAD24)
KERNEL(k_dppmov_sub_interleaved, A_DPPMOVSUB_INTER)
KERNEL(k_sadmul_interleaved, A_SADMUL_INTER)
KERNEL(k_sadmul_clustered8, A_SADMUL_CLUST)
KERNEL(k_dppsub_interleaved, A_DPPSUB_INTER)
KERNEL(k_dppsub_clustered8, A_DPPSUB_CLUST)
KERNEL(k_madmul_interleaved, A_MADMUL_INTER)
KERNEL(k_fmamul_interleaved, A_FMAMUL_INTER)
KERNEL(k_fmamul_clustered8, A_FMAMUL_CLUST)
KERNEL(k_lshladd_mul_interleaved, A_LSHLMUL_INTER)
KERNEL(k_add3_mul_interleaved, A_ADD3MUL_INTER)
KERNEL(k_divfixup_mul_interleaved, A_FIXUPMUL_INTER)
KERNEL(k_cvtubyte_mul_interleaved, A_CVTUBMUL_INTER)
KERNEL(k_fma64_mul_interleaved, A_FMA64MUL_INTER)
KERNEL(k_fma64_mul_clustered8, A_FMA64MUL_CLUST)
KERNEL(k_cvt64_mul_interleaved, A_CVT64MUL_INTER)
KERNEL(k_mulsgpr_add_interleaved, A_MULSGPR_ADD_INTER)
KERNEL(k_mulsgpr_only, A_MULSGPR_ONLY)
KERNEL(k_mulsgpr_sad_interleaved, A_MULSGPR_SAD_INTER)
KERNEL(k_split_waves, A_SPLIT_WAVES)
KERNEL(k_only_sad, A_ONLY_SAD16)
KERNEL(k_only_mul, A_ONLY_MUL16)

typedef void (*kern_t)(unsigned *, int, unsigned long long *);

void run(const char *name, kern_t k, int instr_per_op, unsigned *d_out, unsigned long long *d_clk)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4, 8}) {
        // 256-thread blocks (one wave per SIMD each); LDS per block caps blocks/CU = waves/SIMD;
        // 16 rounds of blocks per CU so that placement imbalance averages out
        const int threads = 256;
        const int blocks = 256 * wps * 16;
        const size_t lds = (160 * 1024) / wps - (wps > 1 ? 1024 : 0);
        CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d_out, kIters / 64, nullptr);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d_out, kIters / 16, d_clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long clk[2];
        CHECK(hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost));
        const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);  // realtime ticks are 100 MHz
        const double winstr = (double)(kIters / 16) * 16 * instr_per_op;  // per wave
        // every SIMD executes 16 rounds x wps waves
        const double per_simd = winstr * 16.0 * wps;
        printf("%-26s waves/SIMD=%d %8.3f ms  clock(blk0) %.3f GHz  %.2f cycles/wave-instr/SIMD (wall, at that clock)\n",
               name, wps, ms, ghz, ms * 1e-3 * ghz * 1e9 / per_simd);
    }
}



int main()
{
    unsigned *d_out;
    unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, 1 << 16));
    CHECK(hipMalloc(&d_clk, 64));
    for (int i = 0; i < 10; i++)
        hipLaunchKernelGGL(k_only_mul, dim3(1024), dim3(256), 0, 0, d_out, kIters / 16, nullptr);
    CHECK(hipDeviceSynchronize());
#define RUN(k, n) run(#k, k, n, d_out, d_clk)
    RUN(k_sadmul_interleaved, 4);
    RUN(k_sadmul_clustered8, 4);
    RUN(k_dppsub_interleaved, 4);
    RUN(k_dppsub_clustered8, 4);
    RUN(k_madmul_interleaved, 4);
    RUN(k_fmamul_interleaved, 4);
    RUN(k_fmamul_clustered8, 4);
    RUN(k_lshladd_mul_interleaved, 4);
    RUN(k_add3_mul_interleaved, 4);
    RUN(k_divfixup_mul_interleaved, 4);
    RUN(k_cvtubyte_mul_interleaved, 4);
    RUN(k_fma64_mul_interleaved, 4);
    RUN(k_fma64_mul_clustered8, 4);
    RUN(k_cvt64_mul_interleaved, 4);
    RUN(k_mulsgpr_add_interleaved, 4);
    RUN(k_mulsgpr_only, 4);
    RUN(k_mulsgpr_sad_interleaved, 4);
    RUN(k_split_waves, 4);
    RUN(k_only_sad, 4);
    RUN(k_only_mul, 4);
    RUN(k_mul24_mul_interleaved, 4);
    RUN(k_muli24_addu_interleaved, 4);
    RUN(k_only_mul24, 4);
    RUN(k_only_mad24, 4);
    RUN(k_dppmov_sub_interleaved, 4);
    return 0;
}
