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

#define OPS16(OP) OP(r0) OP(r1) OP(r2) OP(r3) OP(r4) OP(r5) OP(r6) OP(r7) OP(r0) OP(r1) OP(r2) OP(r3) OP(r4) OP(r5) OP(r6) OP(r7)

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


#define A_DPP_SHR1(x) asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(x));
#define A_DPP_SHR8(x) asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(x));
#define A_DPP_BC15(x) asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
#define A_DPP_BC31(x) asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
#define A_DPP_MOV_SHR1(x) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(x));
#define A_DPP_WSHR1(x) asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(x));
#define A_DPP_QUAD(x) asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
#define A_NOP_ADDU(x) asm volatile("s_nop 1\n v_add_u32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_READLANE(x) asm volatile("v_readlane_b32 s20, %0, 63\n v_add_u32 %0, s20, %0" : "+v"(x) : : "s20");
#define A_SWIZZLE(x) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n s_waitcnt lgkmcnt(0)" : "+v"(x));
#define A_BPERM(x) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(a));
#define A_PLSWAP32(x) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(r7));
#define A_PLSWAP16(x) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(r7));
#define A_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d##x) : "v"(db));
#define A_ADD64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##x) : "v"(db));
#define A_CVT3264(x) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x) : "v"(d##x));
#define A_MAD_I24(x) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(x) : "v"(a));
#define A_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(a));
#define A_FMAF(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define A_SUBU(x) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define A_DIVFIX(x) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b));

KERNEL(k_dpp_shr1, A_DPP_SHR1)
KERNEL(k_dpp_shr8, A_DPP_SHR8)
KERNEL(k_dpp_bcast15, A_DPP_BC15)
KERNEL(k_dpp_bcast31, A_DPP_BC31)
KERNEL(k_dpp_mov_shr1, A_DPP_MOV_SHR1)
KERNEL(k_dpp_wave_shr1, A_DPP_WSHR1)
KERNEL(k_dpp_quad_perm, A_DPP_QUAD)
KERNEL(k_nop_addu, A_NOP_ADDU)
KERNEL(k_readlane_add, A_READLANE)
KERNEL(k_ds_swizzle, A_SWIZZLE)
KERNEL(k_ds_bpermute, A_BPERM)
KERNEL(k_permlane32_swap, A_PLSWAP32)
KERNEL(k_permlane16_swap, A_PLSWAP16)
KERNEL(k_fma_f64, A_FMA64)
KERNEL(k_add_f64, A_ADD64)
KERNEL(k_cvt_f32_f64, A_CVT3264)
KERNEL(k_mad_i32_i24, A_MAD_I24)
KERNEL(k_add3_u32, A_ADD3)
KERNEL(k_fma_f32, A_FMAF)
KERNEL(k_sub_u32, A_SUBU)
KERNEL(k_rcp_f32, A_RCP)
KERNEL(k_div_fixup_f32, A_DIVFIX)

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
        hipLaunchKernelGGL(k_fma_f32, dim3(1024), dim3(256), 0, 0, d_out, kIters, nullptr);
    CHECK(hipDeviceSynchronize());
#define RUN(k, n) run(#k, k, n, d_out, d_clk)
    RUN(k_dpp_shr1, 1);
    RUN(k_dpp_shr8, 1);
    RUN(k_dpp_bcast15, 1);
    RUN(k_dpp_bcast31, 1);
    RUN(k_dpp_mov_shr1, 1);
    RUN(k_dpp_wave_shr1, 1);
    RUN(k_dpp_quad_perm, 1);
    RUN(k_nop_addu, 1);
    RUN(k_readlane_add, 1);
    RUN(k_ds_swizzle, 1);
    RUN(k_ds_bpermute, 1);
    RUN(k_permlane32_swap, 1);
    RUN(k_permlane16_swap, 1);
    RUN(k_fma_f64, 1);
    RUN(k_add_f64, 1);
    RUN(k_cvt_f32_f64, 1);
    RUN(k_mad_i32_i24, 1);
    RUN(k_add3_u32, 1);
    RUN(k_fma_f32, 1);
    RUN(k_sub_u32, 1);
    RUN(k_rcp_f32, 1);
    RUN(k_div_fixup_f32, 1);
    return 0;
}
