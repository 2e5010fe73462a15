// rf_api.hip -- version, error reporting and shutdown of librf_hip.so.
#include "rf_common.hpp"

namespace rf {

void jbf_shutdown();
void cnn_shutdown();

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

}  // namespace rf

extern "C" int rf_version(void) { return RF_VERSION; }

extern "C" const char *rf_last_error(void) { return rf::last_error_buf(); }

extern "C" int rf_shutdown(void)
{
    rf::jbf_shutdown();
    rf::cnn_shutdown();
    return RF_OK;
}
