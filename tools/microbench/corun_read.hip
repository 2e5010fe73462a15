// corun_read.hip -- measurement aid (round 6): light streaming readers of the guided filter's
// alpha/beta planes, launched beside librf_hip.so's stage 1 by tools/gf_corun.py to see whether a
// kernel of <= 64 VGPRs and no LDS is admitted to CUs that are full of stage-1 workgroups
// (4 x 40 KB LDS, 4 x 112 VGPRs per SIMD) and what the pair then costs.
//   variant 0  coalesced float4 grid-stride read (the best a reader can do)
//   variant 1  lane = image row: every lane walks its own row with 16-byte loads (lane stride =
//              one image row), 64 rows per wave - the access pattern of a row walk without LDS
//   variant 2  lane = (row, plane): 4-byte loads, 16 rows x 4 planes per wave
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o corun_read.so corun_read.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void read_coalesced(
    const float4 *__restrict__ p, size_t n4, float *__restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    if (acc == 123.456f)
        *sink = acc;
}

// planes: [img][h][w] float4; one wave = 64 consecutive rows of one image, grid.x = images x ceil(h/64)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void read_lane_row(
    const float4 *__restrict__ p, int h, int w, float *__restrict__ sink)
{
    const int rb = (h + 63) / 64;
    const int img = blockIdx.x / rb, row = min((int)(blockIdx.x % rb) * 64 + (int)threadIdx.x, h - 1);
    const float4 *r = p + ((size_t)img * h + row) * w;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int x = 0; x < w; x += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            v[k] = r[x + k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a0 += (double)v[k].x;
            a1 += (double)v[k].y;
            a2 += (double)v[k].z;
            a3 += (double)v[k].w;
        }
    }
    if ((a0 + a1) + (a2 + a3) == 123.456)
        *sink = (float)a0;
}

// one wave = 16 consecutive rows x 4 planes; lane = row * 4 + plane
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void read_lane_row_plane(
    const float *__restrict__ p, int h, int w, float *__restrict__ sink)
{
    const int rb = (h + 15) / 16;
    const int img = blockIdx.x / rb;
    const int row = min((int)(blockIdx.x % rb) * 16 + (int)(threadIdx.x >> 2), h - 1);
    const float *r = p + (((size_t)img * h + row) * w) * 4 + (threadIdx.x & 3);
    double a = 0;
    for (int x = 0; x < w; x += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; k++)
            v[k] = r[(size_t)(x + k) * 4];
#pragma unroll
        for (int k = 0; k < 16; k++)
            a += (double)v[k];
    }
    if (a == 123.456)
        *sink = (float)a;
}

}  // namespace

extern "C" int corun_read(int variant, const void *p, int n_img, int h, int w, float *sink, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0)
        hipLaunchKernelGGL(read_coalesced, dim3(256 * 16), dim3(256), 0, st, (const float4 *)p,
                           (size_t)n_img * h * w, sink);
    else if (variant == 1)
        hipLaunchKernelGGL(read_lane_row, dim3(n_img * ((h + 63) / 64)), dim3(64), 0, st,
                           (const float4 *)p, h, w, sink);
    else
        hipLaunchKernelGGL(read_lane_row_plane, dim3(n_img * ((h + 15) / 16)), dim3(64), 0, st,
                           (const float *)p, h, w, sink);
    return (int)hipGetLastError();
}
