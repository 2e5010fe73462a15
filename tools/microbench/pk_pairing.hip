// pk_pairing.hip -- does a packed-f32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32, 4-cycle
// pipe) share its issue window with a "simple" 2-cycle instruction the way v_sad_u8 does?  If so,
// replacing two pairs of the joint-bilateral loop's simple adds by packed adds shortens a column
// step from 26 to 24 issue slots.  Streams (per step, at 4 and 8 waves/SIMD):
//   pk+mul, pk+2mul, pk+pk, sad+pk, the 26-slot step (9 full + 17 simple),
//   the 24-slot step (11 full incl. 2 packed + 13 simple), the 23-slot step (12 full + 11 simple)
// Build: hipcc -O3 --offload-arch=gfx950 pk_pairing.hip -o pk_pairing.bin
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

typedef float float2v __attribute__((ext_vector_type(2)));

// F = full-pipe op on an integer register, S = simple op on a float register, P = packed op
#define F_SAD(x) "v_sad_u8 %[" #x "], %[" #x "], %[a], 0\n\t"
#define F_LSH(x) "v_lshl_add_u32 %[" #x "], %[" #x "], 1, %[a]\n\t"
#define S_MUL(x) "v_mul_f32 %[" #x "], %[b], %[" #x "]\n\t"
#define S_ADD(x) "v_add_f32 %[" #x "], %[b], %[" #x "]\n\t"
#define P_ADD(x) "v_pk_add_f32 %[" #x "], %[" #x "], %[bb]\n\t"
#define P_MUL(x) "v_pk_mul_f32 %[" #x "], %[" #x "], %[bb]\n\t"

#define OPERANDS                                                                               \
    : [i0] "+v"(i0), [i1] "+v"(i1), [i2] "+v"(i2), [i3] "+v"(i3), [f0] "+v"(f0), [f1] "+v"(f1), \
      [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5), [f6] "+v"(f6), [f7] "+v"(f7), \
      [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3)                                 \
    : [a] "v"(a), [b] "v"(b), [bb] "v"(bb)

#define KERNEL(NAME, BODY)                                                                     \
    __global__ void NAME(float *out, int iters)                                                \
    {                                                                                          \
        extern __shared__ unsigned dyn_lds[];                                                  \
        if (iters < 0)                                                                         \
            dyn_lds[threadIdx.x] = 1;                                                          \
        unsigned i0 = threadIdx.x * 2654435761u, i1 = i0 ^ 0x55, i2 = i0 + 77, i3 = i0 * 3;    \
        float f0 = 1.f + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4,      \
              f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;                                           \
        float2v p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7};                    \
        const unsigned a = threadIdx.x | 0x01020304u;                                          \
        const float b = 1.0000001f;                                                            \
        const float2v bb = {b, b};                                                             \
        for (int it = 0; it < iters; it++) {                                                   \
            asm volatile(BODY OPERANDS);                                                       \
            asm volatile(BODY OPERANDS);                                                       \
        }                                                                                      \
        float r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p0.y + p1.x + p1.y + p2.x +   \
                  p2.y + p3.x + p3.y + (float)(i0 + i1 + i2 + i3);                             \
        if (r == 0.12345f)                                                                     \
            out[threadIdx.x] = r;                                                              \
    }

KERNEL(k_pk_mul, P_ADD(p0) S_MUL(f0) P_ADD(p1) S_MUL(f1) P_ADD(p2) S_MUL(f2) P_ADD(p3) S_MUL(f3))
KERNEL(k_pk_2mul, P_ADD(p0) S_MUL(f0) S_MUL(f4) P_ADD(p1) S_MUL(f1) S_MUL(f5) P_ADD(p2) S_MUL(f2)
                      S_MUL(f6) P_ADD(p3) S_MUL(f3) S_MUL(f7))
KERNEL(k_pk_pk, P_ADD(p0) P_MUL(p1) P_ADD(p2) P_MUL(p3) P_ADD(p0) P_MUL(p1) P_ADD(p2) P_MUL(p3))
KERNEL(k_sad_pk, F_SAD(i0) P_ADD(p0) F_SAD(i1) P_ADD(p1) F_SAD(i2) P_ADD(p2) F_SAD(i3) P_ADD(p3))
KERNEL(k_sad_mul, F_SAD(i0) S_MUL(f0) F_SAD(i1) S_MUL(f1) F_SAD(i2) S_MUL(f2) F_SAD(i3) S_MUL(f3))
// 26 slots: 9 full + 17 simple, interleaved one for one, the 8 spare simples at the end
KERNEL(k_step26, F_SAD(i0) S_MUL(f0) F_SAD(i1) S_MUL(f1) F_SAD(i2) S_MUL(f2) F_SAD(i3) S_MUL(f3)
                     F_LSH(i0) S_ADD(f4) F_LSH(i1) S_ADD(f5) F_LSH(i2) S_ADD(f6) F_LSH(i3) S_ADD(f7)
                         F_SAD(i0) S_MUL(f0) S_MUL(f1) S_MUL(f2) S_MUL(f3) S_ADD(f4) S_ADD(f5)
                             S_ADD(f6) S_ADD(f7) S_MUL(f0))
// 24 slots: 4 simple adds -> 2 packed adds (11 full + 13 simple)
KERNEL(k_step24, F_SAD(i0) S_MUL(f0) F_SAD(i1) S_MUL(f1) F_SAD(i2) S_MUL(f2) F_SAD(i3) S_MUL(f3)
                     F_LSH(i0) S_ADD(f4) F_LSH(i1) S_ADD(f5) F_LSH(i2) S_ADD(f6) F_LSH(i3) S_ADD(f7)
                         F_SAD(i0) S_MUL(f0) P_ADD(p0) S_MUL(f1) P_ADD(p1) S_MUL(f2) S_MUL(f3)
                             S_MUL(f0))
// 23 slots: 6 simple -> 3 packed (12 full + 11 simple)
KERNEL(k_step23, F_SAD(i0) S_MUL(f0) F_SAD(i1) S_MUL(f1) F_SAD(i2) S_MUL(f2) F_SAD(i3) S_MUL(f3)
                     F_LSH(i0) S_ADD(f4) F_LSH(i1) S_ADD(f5) F_LSH(i2) S_ADD(f6) F_LSH(i3) S_ADD(f7)
                         F_SAD(i0) S_MUL(f0) P_ADD(p0) S_MUL(f1) P_ADD(p1) S_MUL(f2) P_ADD(p2))
// 22 slots: 8 simple -> 4 packed (13 full + 9 simple)
KERNEL(k_step22, F_SAD(i0) S_MUL(f0) F_SAD(i1) S_MUL(f1) F_SAD(i2) S_MUL(f2) F_SAD(i3) S_MUL(f3)
                     F_LSH(i0) S_ADD(f4) F_LSH(i1) S_ADD(f5) F_LSH(i2) S_ADD(f6) F_LSH(i3) S_ADD(f7)
                         F_SAD(i0) S_MUL(f0) P_ADD(p0) P_MUL(p1) P_ADD(p2) P_MUL(p3))

typedef void (*kern_t)(float *, int);

static void run(const char *name, kern_t k, int slots, float *d_out)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4096;
    for (int wps : {4, 8}) {
        const int blocks = 256 * wps * 16;
        const size_t lds = (160 * 1024) / wps - 1024;
        CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d_out, iters / 8);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d_out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        // per SIMD: 16 rounds x wps waves x iters x 2 bodies; report ns per body per SIMD-wave slot
        const double bodies = 16.0 * wps * iters * 2;
        const double ns_per_body = best * 1e6 / bodies;
        printf("%-10s waves/SIMD=%d  %8.3f ms  %6.2f ns per step per SIMD (%d slots: %.2f cycles per "
               "slot at 2.4 GHz)\n",
               name, wps, best, ns_per_body, slots, ns_per_body * 2.4 / slots);
    }
}

int main()
{
    float *d_out;
    CHECK(hipMalloc(&d_out, 1 << 16));
#define RUN(k, n) run(#k, k, n, d_out)
    RUN(k_sad_mul, 8);
    RUN(k_pk_mul, 8);
    RUN(k_pk_2mul, 12);
    RUN(k_pk_pk, 8);
    RUN(k_sad_pk, 8);
    RUN(k_step26, 26);
    RUN(k_step24, 24);
    RUN(k_step23, 23);
    RUN(k_step22, 22);
    return 0;
}
