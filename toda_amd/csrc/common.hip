// Error plumbing shared by every translation unit of libtoda_hip.so.
#include <stdarg.h>

#include "common.h"

namespace toda {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace toda

extern "C" const char* toda_last_error(void) { return toda::g_err; }
extern "C" int toda_abi_version(void) { return 1; }
