// Does v_mfma_f32_32x32x2_f32 overlap with VALU work on gfx950?
//   mfma-only loop, valu-only loop (v_fma_f32), and both interleaved (K VALU per MFMA), one and
//   two waves per SIMD.  If the matrix instruction ran beside the VALU, "both" would cost
//   max(mfma, valu); if it occupies the VALU's issue for its whole duration, the sum.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_f32_valu_overlap.hip -o tools/microbench/mfma_f32_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float16v __attribute__((ext_vector_type(16)));

template <int MODE, int K>  // MODE 0: mfma only, 1: valu only, 2: both
__global__ __launch_bounds__(256) void kern(float *out, int iters, unsigned long long *cyc)
{
    float16v acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
    for (int i = 0; i < 8; i++)
        v[i] = a + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE != 1) {
                if (u & 1)
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                else
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            }
            if (MODE != 0) {
#pragma unroll
                for (int k = 0; k < K; k++)
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 8]) : "v"(a), "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; i++)
        s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; i++)
        s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *cyc = t1 - t0;
}

template <int MODE, int K>
void run(const char *name, int waves_per_simd, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipLaunchKernelGGL((kern<MODE, K>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE, K>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD %d: %8.3f ms, %7.1f cycles per step (1 mfma + %d valu)\n", name,
           waves_per_simd, ms, (double)c / (iters * 8.0), MODE == 0 ? 0 : K);
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float) * 4);
    hipMalloc(&cyc, 8);
    for (int w = 1; w <= 2; w++) {
        run<0, 0>("mfma_f32_32x32x2 only", w, out, cyc);
        run<1, 8>("8 v_fma_f32 only", w, out, cyc);
        run<2, 8>("mfma + 8 v_fma_f32", w, out, cyc);
        run<1, 16>("16 v_fma_f32 only", w, out, cyc);
        run<2, 16>("mfma + 16 v_fma_f32", w, out, cyc);
        run<2, 4>("mfma + 4 v_fma_f32", w, out, cyc);
    }
    return 0;
}
