// l1_gather.hip -- can the vector-memory path (TA/L1) serve LUT gathers next to the LDS?
// Each lane repeatedly loads a dword at a pseudo-random offset inside a small table (1.2 KB, like
// the joint-bilateral colour LUT), 8 loads in flight.  Prints wave-instructions per ns per CU.
// Build: hipcc -O3 --offload-arch=gfx950 l1_gather.hip -o l1_gather.bin
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void gather_kernel(const float *table, int nentries, float *out, int iters)
{
    extern __shared__ unsigned dyn[];
    if (iters < 0)
        dyn[threadIdx.x] = 1;
    unsigned idx[8];
    for (int k = 0; k < 8; k++)
        idx[k] = ((threadIdx.x * 2654435761u) >> (7 + k)) % nentries;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = __builtin_nontemporal_load(&table[idx[k]]) ;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            acc += v[k];
            idx[k] = (idx[k] * 5u + 1u + (unsigned)it) % nentries;  // keeps addresses changing
        }
    }
    if (acc == 123.456f)
        out[threadIdx.x] = acc;
}

int main()
{
    float *table, *out;
    hipMalloc(&table, 4096);
    hipMalloc(&out, 4096);
    hipMemset(table, 0, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {2, 4, 8}) {
        const size_t lds = (160 * 1024) / wps - (wps > 1 ? 1024 : 0);
        hipFuncSetAttribute((const void *)gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
        const int blocks = 256 * wps * 4;
        for (int nent : {300, 766}) {
            hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), lds, 0, table, nent, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), lds, 0, table, nent, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double winstr_per_cu = (double)iters * 8 * 4 * wps * 4;  // loads per CU (4 rounds)
            printf("waves/SIMD=%d table=%d entries: %.3f ms, %.3f gather wave-instr/ns/CU (%.1f cycles each @2.4GHz)\n",
                   wps, nent, ms, winstr_per_cu / (ms * 1e6), ms * 1e6 * 2.4 / winstr_per_cu);
        }
    }
    return 0;
}
