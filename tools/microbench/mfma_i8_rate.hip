// mfma_i8_rate.hip -- issue rate of the int8 matrix-core instructions a banded 0/1 "box sum as a
// matrix product" would use (round-4 review, stretch item 6): v_mfma_i32_32x32x16_i8 and the gfx950
// v_mfma_i32_32x32x32_i8, back to back on independent accumulators, 1 / 2 / 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_i8_rate.hip -o mfma_i8_rate.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                           \
    do {                                                                   \
        hipError_t e = (x);                                                \
        if (e != hipSuccess) {                                             \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));         \
            exit(1);                                                       \
        }                                                                  \
    } while (0)

typedef int int16v __attribute__((ext_vector_type(16)));
typedef int int4v __attribute__((ext_vector_type(4)));

template <int K32>
__global__ void k_mfma(int *out, int iters, unsigned long long *clk)
{
    extern __shared__ unsigned dyn_lds[];
    if (iters < 0)
        dyn_lds[threadIdx.x] = 1;
    int16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const long a8 = 0x0101010101010101L * (threadIdx.x & 3), b8 = 0x0101010101010101L;
    const int4v a16 = {(int)threadIdx.x & 0x01010101, 0x01010101, 0x01010101, 0x01010101};
    const int4v b16 = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (K32) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, b16, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, b16, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, b16, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a16, b16, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x16_i8(a8, b8, c3, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) {
        clk[0] = t1 - t0;
        clk[1] = q1 - q0;
    }
    int r = 0;
    for (int i = 0; i < 16; i++)
        r += c0[i] + c1[i] + c2[i] + c3[i];
    if (r == 0x12345678)
        out[threadIdx.x] = r;
}

template <int K32>
void run(const char *name, int *d_out, unsigned long long *d_clk)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4096;
    for (int wps : {1, 2, 4}) {
        const int threads = 256, blocks = 256 * wps * 8;
        const size_t lds = (160 * 1024) / wps - (wps > 1 ? 1024 : 0);
        CHECK(hipFuncSetAttribute((const void *)k_mfma<K32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_mfma<K32>, dim3(blocks), dim3(threads), lds, 0, d_out, 64, nullptr);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma<K32>, dim3(blocks), dim3(threads), lds, 0, d_out, iters, d_clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long clk[2];
        CHECK(hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost));
        const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
        const double per_simd = (double)iters * 4 * 8.0 * wps;   // MFMAs per SIMD (8 rounds of blocks)
        const double cyc = ms * 1e-3 * ghz * 1e9 / per_simd;
        const double macs = 32.0 * 32.0 * (K32 ? 32 : 16);
        printf("%-24s waves/SIMD=%d %8.3f ms  clock %.3f GHz  %6.2f cycles per MFMA per SIMD  = %.0f MAC/clk/CU  (%.2f Pop/s dense on 256 CUs at that clock)\n",
               name, wps, ms, ghz, cyc, 4.0 * macs / cyc, 2.0 * 4.0 * macs / cyc * 256 * ghz * 1e9 / 1e15);
    }
}

int main()
{
    int *d_out;
    unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, 1 << 16));
    CHECK(hipMalloc(&d_clk, 64));
    run<0>("v_mfma_i32_32x32x16_i8", d_out, d_clk);
    run<1>("v_mfma_i32_32x32x32_i8", d_out, d_clk);
    return 0;
}
