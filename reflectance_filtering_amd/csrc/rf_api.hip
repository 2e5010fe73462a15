// rf_api.hip -- version, error reporting and shutdown of librf_hip.so.
#include "rf_common.hpp"

#include <atomic>
#include <cstring>

#include "reflectance_filtering_debug.h"

namespace rf {

void jbf_shutdown();
void cnn_shutdown();
void gf_shutdown();

char *last_error_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static std::atomic<int> g_debug[kDbgCount];

int debug_get(int id) { return g_debug[id].load(std::memory_order_relaxed); }

}  // namespace rf

extern "C" int rf_debug_option(const char *name, int value)
{
    static const char *const names[] = {"gf_two_kernel",     "jbf_stage_only",
                                                      "jbf_compiler_loop", "jbf_tile64_only",
                                                      "jbf_tune",          "jbf_f32_untiled",
                                                      "cnn_lds_columns",   "gf_seg_rows",
                                                      "gf_one_stream",     "gf_force_two_streams",
                                                      "gf_guide_cache",    "gf_chained",        "gf_no_compact",
                                                      "gf_exp_skip",       "gf_stagger",        "gf_parts",
                                                      "gf_s1_cap",         "gf_s1_min_wgs",
                                                      "jbf_lookahead1",    "gf_s1_legacy_strips",
                                                      "gf_exact_all_flagged", "gf_cw_chan_run",   "gf_exact"};
    static_assert(sizeof(names) / sizeof(names[0]) == rf::kDbgCount, "one name per DebugOption");
    if (name)
        for (int i = 0; i < rf::kDbgCount; i++)
            if (std::strcmp(name, names[i]) == 0)
                return rf::g_debug[i].exchange(value < 0 ? 0 : value);
    return rf::fail(RF_E_BADARG, "rf_debug_option: unknown option '%s'", name ? name : "(null)");
}

// One wave that brackets `micros` microseconds of wall time (s_memrealtime: 100 MHz, constant)
// with the shader-cycle counter (s_memtime) and sleeps in between: launched on a second stream
// next to a running kernel it reads the clock the chip holds under THAT kernel's load.
__global__ void clock_probe_kernel(unsigned long long *out, unsigned long long ticks)
{
    if (threadIdx.x != 0)
        return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

extern "C" int rf_debug_clock_probe(unsigned long long *out2, int micros, void *stream)
{
    if (!out2 || micros < 1 || micros > 1000000)
        return rf::fail(RF_E_BADARG, "rf_debug_clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2,
                       (unsigned long long)micros * 100ull);
    RF_HIP_CHECK(hipGetLastError());
    return RF_OK;
}

// (a development build - make GF_DEV_ONLY=<radius>: one radius of the fused guided filter only -
//  says so in its version, and __graft_entry__.build() refuses to accept it)
#ifdef RF_GF_DEV_ONLY
extern "C" int rf_version(void) { return RF_VERSION | 0x40000000; }
#else
extern "C" int rf_version(void) { return RF_VERSION; }
#endif

#ifndef RF_TOOLCHAIN
#define RF_TOOLCHAIN "unknown toolchain"
#endif
// The hand-written asm loops keep LDS / scalar reads in flight across code the compiler writes;
// tests/test_cabi.py audits the machine code of THIS build - and prints what built it.
extern "C" const char *rf_debug_build_info(void) { return RF_TOOLCHAIN; }

extern "C" const char *rf_last_error(void) { return rf::last_error_buf(); }

extern "C" int rf_shutdown(void)
{
    rf::jbf_shutdown();
    rf::cnn_shutdown();
    rf::gf_shutdown();
    return RF_OK;
}
