// Error reporting and version entry points of libvings_hip.so.
#include <stdarg.h>

#include "common.hpp"

namespace vings {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vings

extern "C" const char* vings_last_error(void) { return vings::g_err; }
extern "C" int vings_abi_version(void) { return 1; }
