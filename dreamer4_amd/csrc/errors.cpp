#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace d4 {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return 3;
}
const char* last_error() { return g_err; }
}  // namespace d4
