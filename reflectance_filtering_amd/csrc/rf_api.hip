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
                                                      "gf_one_stream"};
    static_assert(sizeof(names) / sizeof(names[0]) == rf::kDbgCount, "one name per DebugOption");
    if (name)
        for (int i = 0; i < rf::kDbgCount; i++)
            if (std::strcmp(name, names[i]) == 0)
                return rf::g_debug[i].exchange(value < 0 ? 0 : value);
    return rf::fail(RF_E_BADARG, "rf_debug_option: unknown option '%s'", name ? name : "(null)");
}

extern "C" int rf_version(void) { return RF_VERSION; }

extern "C" const char *rf_last_error(void) { return rf::last_error_buf(); }

extern "C" int rf_shutdown(void)
{
    rf::jbf_shutdown();
    rf::cnn_shutdown();
    rf::gf_shutdown();
    return RF_OK;
}
