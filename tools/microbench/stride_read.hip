// stride_read.hip -- achievable read bandwidth of the guided filter's stage-2 access patterns
// against a plain sequential read (round 4: is the 4.8 TB/s of the row walk / column walk a
// property of their 256-byte-per-image-row pieces?).
//   seq      every wave reads 64 x 16 B contiguous (1 KB), waves walk the buffer linearly
//   rows256  a wave instruction reads 4 rows x 256 B (rows 61,440 B apart: one alpha/beta row of a
//            3840-wide image), a workgroup of 4 waves covers 64 rows x 256 B per step and walks
//            along the row (the row walk's pattern: 16 columns x 64 rows per chunk)
//   tile4k   the same bytes when 16 columns x 16 rows were one contiguous 4 KB tile
// Build: hipcc -O3 --offload-arch=gfx950 stride_read.hip -o stride_read.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

constexpr int W = 3840, H = 2160;            // one plane group: H rows x W pixels x 16 B
constexpr size_t ROW = (size_t)W * 16;

__global__ __launch_bounds__(256) void k_seq(const uint4 *p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// grid: images x (H/64) row blocks; each workgroup walks its 64 rows from left to right in chunks
// of 16 pixels (256 B per row): thread t reads pixel (row t/16 + 16*k, column chunk*16 + t%16)
__global__ __launch_bounds__(256) void k_rows256(const uint4 *p, int nimg, uint32_t *sink)
{
    const int img = blockIdx.x / (H / 64), rb = blockIdx.x % (H / 64);
    const uint4 *base = p + ((size_t)img * H + (size_t)rb * 64) * W;
    uint32_t acc = 0;
    const int cc = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    for (int x0 = 0; x0 < W; x0 += 16) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = base[(size_t)(r0 + 16 * k) * W + x0 + cc];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the same walk over a tiled layout: [row block of 16][column block of 16][16 rows][16 px]
__global__ __launch_bounds__(256) void k_tile4k(const uint4 *p, int nimg, uint32_t *sink)
{
    const int img = blockIdx.x / (H / 64), rb = blockIdx.x % (H / 64);
    const uint4 *base = p + (size_t)img * H * W;
    uint32_t acc = 0;
    for (int cb = 0; cb < W / 16; cb++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {   // four 4 KB tiles (16 rows each) of this column block
            const size_t tile = ((size_t)(rb * 4 + k) * (W / 16) + cb) * 256;
            const uint4 v = base[tile + threadIdx.x];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const int nimg = 24;                               // 24 x 133 MB = 3.2 GB >> caches
    const size_t bytes = (size_t)nimg * H * ROW;
    uint4 *buf; uint32_t *sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        float ms;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_seq, dim3(8192), dim3(256), 0, 0, buf, bytes / 16, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("seq      %.3f ms  %.2f TB/s\n", ms, bytes / (ms * 1e-3) / 1e12);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rows256, dim3(nimg * (H / 64)), dim3(256), 0, 0, buf, nimg, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("rows256  %.3f ms  %.2f TB/s  (%d workgroups)\n", ms, (double)nimg * (H / 64) * 64 * ROW / (ms * 1e-3) / 1e12, nimg * (H / 64));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_tile4k, dim3(nimg * (H / 64)), dim3(256), 0, 0, buf, nimg, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("tile4k   %.3f ms  %.2f TB/s\n", ms, (double)nimg * (H / 64) * 64 * ROW / (ms * 1e-3) / 1e12);
    }
    return 0;
}
