// Error plumbing and version of the C ABI (include/gandanet.h).
#include "common.h"
#include "../../include/gandanet.h"

#include <string.h>

namespace {
thread_local char g_err[512] = {0};
}

extern "C" void gd_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" int gd_last_error(char* buf, int n) {
    const int len = (int)strlen(g_err);
    if (buf && n > 0) {
        strncpy(buf, g_err, (size_t)n - 1);
        buf[n - 1] = 0;
    }
    return len;
}

extern "C" int gd_version(void) { return GD_VERSION; }

// struct sizes, so a binding can verify its mirror of the descriptors
extern "C" int gd_sizeof_conv_desc(void) { return (int)sizeof(gd_conv_desc); }
extern "C" int gd_sizeof_gemm_nt_desc(void) { return (int)sizeof(gd_gemm_nt_desc); }

// Deterministic mode: kernels that combine partial sums with fp32 atomics (split-K NT GEMM, 3x3
// weight gradient) run unsplit -- one adder per output element, bitwise reproducible, slower.  The PAM backward's dQ
// form is chosen by its caller (gd_pam_flash_bwd form 1).  Process-global.
namespace {
int g_deterministic = 0;
}
extern "C" void gd_set_deterministic(int on) { g_deterministic = on ? 1 : 0; }
extern "C" int gd_get_deterministic(void) { return g_deterministic; }
