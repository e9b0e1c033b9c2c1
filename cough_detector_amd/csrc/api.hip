// Library-wide C-ABI entry points: version / arch / error text.
#include <cstdarg>
#include <cstdio>

#include "common.h"

namespace cough {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cough

extern "C" int cough_amd_abi_version(void) { return COUGH_AMD_ABI_VERSION; }
extern "C" const char* cough_amd_arch(void) { return "gfx950"; }
extern "C" const char* cough_amd_last_error(void) { return cough::g_err; }
