// Do two VGPR source operands in the same register bank (register number mod 4) cost a
// "simple" VALU instruction extra cycles on gfx950?  Four independent chains per variant, 4
// waves per SIMD, explicit registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define BODY_DIFF \
    "v_add_f32 v40, v45, v40\n v_add_f32 v41, v46, v41\n v_add_f32 v42, v47, v42\n v_add_f32 v43, v44, v43\n"
#define BODY_SAME \
    "v_add_f32 v40, v44, v40\n v_add_f32 v41, v45, v41\n v_add_f32 v42, v46, v42\n v_add_f32 v43, v47, v43\n"
#define BODY_MUL_DIFF \
    "v_mul_f32 v40, v45, v48\n v_mul_f32 v41, v46, v49\n v_mul_f32 v42, v47, v50\n v_mul_f32 v43, v44, v51\n"
#define BODY_MUL_SAME \
    "v_mul_f32 v40, v44, v48\n v_mul_f32 v41, v45, v49\n v_mul_f32 v42, v46, v50\n v_mul_f32 v43, v47, v51\n"
#define BODY_SAD_DIFF \
    "v_sad_u8 v40, v45, v50, 0\n v_sad_u8 v41, v46, v51, 0\n v_sad_u8 v42, v47, v48, 0\n v_sad_u8 v43, v44, v49, 0\n"
#define BODY_SAD_SAME \
    "v_sad_u8 v40, v44, v48, 0\n v_sad_u8 v41, v45, v49, 0\n v_sad_u8 v42, v46, v50, 0\n v_sad_u8 v43, v47, v51, 0\n"
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51"

#define KERNEL(name, body)                                                   \
    __global__ __launch_bounds__(1024) void name(float *out, int iters)     \
    {                                                                        \
        asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n" \
                     "v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n" \
                     "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 1.0\n v_mov_b32 v51, 1.0\n" ::: CLOB); \
        for (int i = 0; i < iters; i++)                                      \
            asm volatile(REP8(REP8(body)) ::: CLOB);                         \
        float r;                                                             \
        asm volatile("v_add_f32 %0, v40, v41" : "=v"(r)::CLOB);              \
        if (r == 123.f)                                                      \
            out[threadIdx.x] = r;                                            \
    }
KERNEL(k_add_diff, BODY_DIFF)
KERNEL(k_add_same, BODY_SAME)
KERNEL(k_mul_diff, BODY_MUL_DIFF)
KERNEL(k_mul_same, BODY_MUL_SAME)
KERNEL(k_sad_diff, BODY_SAD_DIFF)
KERNEL(k_sad_same, BODY_SAD_SAME)

template <typename K> int run(const char *name, K k, float *d)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 4 waves x iters x 256 instructions
    const double instr = 4.0 * iters * 256;
    printf("%-12s %.3f ms  %.2f cycles/wave-instr/SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
    return 0;
}

int main()
{
    float *d;
    CHECK(hipMalloc(&d, 4096));
    run("add_diff", k_add_diff, d);
    run("add_same", k_add_same, d);
    run("mul_diff", k_mul_diff, d);
    run("mul_same", k_mul_same, d);
    run("sad_diff", k_sad_diff, d);
    run("sad_same", k_sad_same, d);
    return 0;
}
