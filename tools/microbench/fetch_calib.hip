// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the access pattern of the
// joint-bilateral staging loads (12 contiguous bytes per lane, global_load_dwordx3) against a
// known byte count, next to the 16-B/lane pattern MI355X_MICROARCH.md documents as reading 1/2.
// Build: hipcc -O3 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib.bin
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ void read12_kernel(const uint8_t *p, size_t n12, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n12;
         i += (size_t)gridDim.x * blockDim.x) {
        uint32_t d0, d1, d2;
        __builtin_memcpy(&d0, p + i * 12, 4);
        __builtin_memcpy(&d1, p + i * 12 + 4, 4);
        __builtin_memcpy(&d2, p + i * 12 + 8, 4);
        acc ^= d0 ^ d1 ^ d2;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

__global__ void read16_kernel(const uint4 *p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

// Round 4: the guided-filter kernels load 1, 4 and 8 bytes per lane (guide / src bytes, float
// row states, double sums); the same "read N bytes once" with those widths.
template <typename T>
__global__ void readT_kernel(const T *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        uint32_t w[(sizeof(T) + 3) / 4] = {0};
        __builtin_memcpy(w, &v, sizeof(T));
        for (unsigned k = 0; k < (sizeof(T) + 3) / 4; ++k)
            acc ^= w[k];
    }
    if (acc == 0x7bu)
        sink[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)3 << 30;  // 3 GiB >> 256 MiB Infinity Cache
    uint8_t *buf;
    uint32_t *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess)
        return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(read12_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 12, sink);
    hipLaunchKernelGGL(read16_kernel, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16,
                       sink);
    // 1 B/lane reads 1 GiB (a third of the buffer: 1-byte loads are slow), 4 and 8 B/lane all of it
    hipLaunchKernelGGL(readT_kernel<uint8_t>, dim3(8192), dim3(256), 0, 0, buf, bytes / 3, sink);
    hipLaunchKernelGGL(readT_kernel<uint32_t>, dim3(4096), dim3(256), 0, 0, (const uint32_t *)buf,
                       bytes / 4, sink);
    hipLaunchKernelGGL(readT_kernel<uint2>, dim3(4096), dim3(256), 0, 0, (const uint2 *)buf,
                       bytes / 8, sink);
    hipDeviceSynchronize();
    printf("readT_kernel<uint8_t> read %zu bytes once\n", bytes / 3);
    printf("each other kernel read %zu bytes once\n", bytes);
    return 0;
}
