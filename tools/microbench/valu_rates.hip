// valu_rates.hip -- instruction-rate microbenchmarks that size the joint-bilateral tap loop on
// MI355X (gfx950).  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates.bin
// Prints, per instruction mix and waves/SIMD, wave-instructions per ns per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int kIters = 4096 * 64;

// 8 independent chains, 16 instructions per loop trip
#define REP8(OP)                                                                        \
    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

template <int KIND>
__global__ void rate_kernel(float *out, int iters)
{
    __shared__ float lds[32 * 320];
    for (int i = threadIdx.x; i < 32 * 320; i += blockDim.x)
        lds[i] = 1.0f + 1e-7f * i;
    __syncthreads();
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,
          a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f,
            p6 = p2 + 1.f, p7 = p3 + 1.f;
    const float2v pb = {b, b};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x55, u2 = u0 + 77, u3 = u0 * 3, u4 = u0 + 5,
             u5 = u0 ^ 9, u6 = u0 + 11, u7 = u0 * 7;
    // LDS addresses: conflict-free (lane-private bank) and pseudo-random
    unsigned addr_cf = (threadIdx.x & 31) * 4;
    unsigned addr_rnd = ((threadIdx.x * 2654435761u) >> 20) % (32 * 300) * 4;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {  // v_mul_f32 chains
#define OP(x) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(b));
            REP8(OP) REP8(OP)
#undef OP
        } else if (KIND == 1) {  // v_add_f32 chains
#define OP(x) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "v"(b));
            REP8(OP) REP8(OP)
#undef OP
        } else if (KIND == 2) {  // v_fma_f32
#define OP(x) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(x) : "v"(b));
            REP8(OP) REP8(OP)
#undef OP
        } else if (KIND == 3) {  // v_pk_mul_f32
#define OP(x) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(x) : "v"(pb));
            OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7) OP(p0) OP(p1) OP(p2) OP(p3) OP(p4)
                OP(p5) OP(p6) OP(p7)
#undef OP
        } else if (KIND == 4) {  // v_pk_add_f32
#define OP(x) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x) : "v"(pb));
            OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7) OP(p0) OP(p1) OP(p2) OP(p3) OP(p4)
                OP(p5) OP(p6) OP(p7)
#undef OP
        } else if (KIND == 5) {  // v_sad_u8
#define OP(x) asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(x) : "v"(u0 ^ 0x01020304u));
            OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7) OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6)
                OP(u7) OP(u1) OP(u2)
#undef OP
        } else if (KIND == 6) {  // v_cvt_f32_ubyte1
#define OP(x) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x) : "v"(u0));
            REP8(OP) REP8(OP)
#undef OP
        } else if (KIND == 7) {  // v_lshl_add_u32 + v_min_u32 pair
#define OP(x) asm volatile("v_min_u32 %0, %0, %1\n v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(u0));
            OP(u1) OP(u2) OP(u3) OP(u4) OP(u5) OP(u6) OP(u7) OP(u1)
#undef OP
        } else if (KIND == 8) {  // ds_read_b32 conflict-free, 16 per trip
#define OP(x) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(addr_cf));
            REP8(OP) REP8(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 9) {  // ds_read_b32 random addresses
#define OP(x) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(addr_rnd));
            REP8(OP) REP8(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 10) {  // ds_read_b64 conflict-free
            unsigned a64 = (threadIdx.x & 63) * 8;
#define OP(x) asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(a64));
            OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7) OP(p0) OP(p1) OP(p2) OP(p3) OP(p4)
                OP(p5) OP(p6) OP(p7)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 11) {  // the scalar JBF tap: sad,min,lshl_add,ds_read,mul,3x(mul,add),add
            asm volatile(
                "v_sad_u8 %1, %9, %10, 0\n"
                "v_min_u32 %1, %1, %11\n"
                "v_lshl_add_u32 %1, %1, 7, %12\n"
                "ds_read_b32 %1, %1\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_mul_f32 %1, %13, %1\n"
                "v_mul_f32 %2, %1, %14\n"
                "v_add_f32 %5, %5, %2\n"
                "v_mul_f32 %3, %1, %15\n"
                "v_add_f32 %6, %6, %3\n"
                "v_mul_f32 %4, %1, %16\n"
                "v_add_f32 %7, %7, %4\n"
                "v_add_f32 %8, %8, %1\n"
                : "+v"(u1), "=&v"(u2), "=&v"(a0), "=&v"(a1), "=&v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),
                  "+v"(a6)
                : "v"(u0), "v"(u3), "v"(u4 & 255), "v"(addr_cf), "v"(b), "v"(a7), "v"(b), "v"(a7));
        } else if (KIND == 12) {  // packed JBF tap: sad,min,lshl_add,ds_read,mul, 2x pk_mul, 2x pk_add
            asm volatile(
                "v_sad_u8 %1, %6, %7, 0\n"
                "v_min_u32 %1, %1, %8\n"
                "v_lshl_add_u32 %1, %1, 7, %9\n"
                "ds_read_b32 %1, %1\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_mul_f32 %1, %10, %1\n"
                "v_pk_mul_f32 %2, %11, %13\n"
                "v_pk_add_f32 %4, %4, %2\n"
                "v_pk_mul_f32 %3, %12, %13\n"
                "v_pk_add_f32 %5, %5, %3\n"
                : "+v"(u1), "=&v"(u2), "=&v"(p0), "=&v"(p1), "+v"(p2), "+v"(p3)
                : "v"(u0), "v"(u3), "v"(u4 & 255), "v"(addr_cf), "v"(b), "v"(p6), "v"(p7), "v"(p5));
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y +
              p7.y + (float)(u1 + u2 + u3 + u4 + u5 + u6 + u7);
    if (r == 12345.678f)
        out[threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, int instr_per_trip, float *d_out)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
        const int blocks_per_cu = (256 * waves_per_simd) / threads;
        const int blocks = 256 * blocks_per_cu;
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, kIters / 2);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, kIters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_instr = (double)kIters * instr_per_trip * (4.0 * waves_per_simd);  // per CU
        printf("%-28s waves/SIMD=%d  %8.3f ms  %7.3f wave-instr/ns/CU  (%.2f cycles/wave-instr/SIMD @2.4GHz)\n",
               name, waves_per_simd, ms, wave_instr / (ms * 1e6),
               (ms * 1e6 * 2.4) / (wave_instr / 4.0));
    }
}

int main()
{
    float *d_out;
    CHECK(hipMalloc(&d_out, 4096));
    // wake the clocks up
    for (int i = 0; i < 20; i++)
        hipLaunchKernelGGL(rate_kernel<2>, dim3(1024), dim3(256), 0, 0, d_out, kIters);
    CHECK(hipDeviceSynchronize());
    run<0>("v_mul_f32", 16, d_out);
    run<1>("v_add_f32", 16, d_out);
    run<2>("v_fma_f32", 16, d_out);
    run<3>("v_pk_mul_f32", 16, d_out);
    run<4>("v_pk_add_f32", 16, d_out);
    run<5>("v_sad_u8", 16, d_out);
    run<6>("v_cvt_f32_ubyte1", 16, d_out);
    run<7>("v_min_u32+v_lshl_add_u32", 16, d_out);
    run<8>("ds_read_b32 conflict-free", 16, d_out);
    run<9>("ds_read_b32 random", 16, d_out);
    run<10>("ds_read_b64 conflict-free", 16, d_out);
    run<11>("jbf tap scalar (12 valu+1 ds)", 1, d_out);
    run<12>("jbf tap packed (8 valu+1 ds)", 1, d_out);
    return 0;
}
