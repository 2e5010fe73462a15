// Issue cost of the float64 instructions behind the guided filter's exact window means
// (v_cvt_f64_u32, v_mul_f64, v_cvt_f32_f64) and of the bit-trick alternative (v_add_f64).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
#define REP8(x) x x x x x x x x
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
#define B_CVT64 "v_cvt_f64_u32 v[40:41], v48\n v_cvt_f64_u32 v[42:43], v49\n v_cvt_f64_u32 v[44:45], v50\n v_cvt_f64_u32 v[46:47], v51\n"
#define B_MUL64 "v_mul_f64 v[40:41], v[48:49], v[52:53]\n v_mul_f64 v[42:43], v[50:51], v[52:53]\n v_mul_f64 v[44:45], v[48:49], v[54:55]\n v_mul_f64 v[46:47], v[50:51], v[54:55]\n"
#define B_ADD64 "v_add_f64 v[40:41], v[48:49], v[52:53]\n v_add_f64 v[42:43], v[50:51], v[52:53]\n v_add_f64 v[44:45], v[48:49], v[54:55]\n v_add_f64 v[46:47], v[50:51], v[54:55]\n"
#define B_CVT32 "v_cvt_f32_f64 v40, v[48:49]\n v_cvt_f32_f64 v41, v[50:51]\n v_cvt_f32_f64 v42, v[52:53]\n v_cvt_f32_f64 v43, v[54:55]\n"
#define B_CVTF32U "v_cvt_f32_u32 v40, v48\n v_cvt_f32_u32 v41, v49\n v_cvt_f32_u32 v42, v50\n v_cvt_f32_u32 v43, v51\n"
#define KERNEL(name, body)                                                   \
    __global__ __launch_bounds__(1024) void name(float *out, int iters)     \
    {                                                                        \
        asm volatile("v_mov_b32 v48, 3\n v_mov_b32 v49, 0x40080000\n v_mov_b32 v50, 5\n v_mov_b32 v51, 0x40080000\n" \
                     "v_mov_b32 v52, 0\n v_mov_b32 v53, 0x3ff00000\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0x3ff80000\n" ::: CLOB); \
        for (int i = 0; i < iters; i++)                                      \
            asm volatile(REP8(REP8(body)) ::: CLOB);                         \
        float r;                                                             \
        asm volatile("v_add_f32 %0, v40, v41" : "=v"(r)::CLOB);              \
        if (r == 123.25f)                                                    \
            out[threadIdx.x] = r;                                            \
    }
KERNEL(k_cvt_f64_u32, B_CVT64)
KERNEL(k_mul_f64, B_MUL64)
KERNEL(k_add_f64, B_ADD64)
KERNEL(k_cvt_f32_f64, B_CVT32)
KERNEL(k_cvt_f32_u32, B_CVTF32U)
template <typename K> int run(const char *name, K k, float *d)
{
    const int iters = 1000;
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
    const double instr = 4.0 * iters * 256;
    printf("%-16s %.3f ms  %.2f cycles/wave-instr/SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
    return 0;
}
int main()
{
    float *d;
    CHECK(hipMalloc(&d, 4096));
    run("cvt_f64_u32", k_cvt_f64_u32, d);
    run("mul_f64", k_mul_f64, d);
    run("add_f64", k_add_f64, d);
    run("cvt_f32_f64", k_cvt_f32_f64, d);
    run("cvt_f32_u32", k_cvt_f32_u32, d);
    return 0;
}
