// Error plumbing + version of the C ABI (include/geoadv.h).
#include "common.h"
#include <string.h>

namespace geoadv {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace geoadv

extern "C" int geoadv_version(void) { return 1000 * 0 + 1; }
extern "C" const char *geoadv_last_error(void) { return geoadv::g_err; }
