// Error reporting and version for the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/octa_hip.h"

static thread_local char g_err[512] = "";

void octa_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int octa_version(void) { return OCTA_HIP_ABI_VERSION; }
extern "C" const char* octa_last_error(void) { return g_err; }

// Deterministic mode ("deterministic variant required for parity tests"): every sum that crosses workgroups runs in a
// fixed order -- private partial tiles + an ordered fold where a scratch is passed, one workgroup per output address otherwise.
// Slower; parity tests run with it, the benchmark without.  octa_tuning_set(5, 0 / 1), or OCTA_DETERMINISTIC=1 in the environment.
static int g_det = -1;
bool octa_deterministic() {
    if (g_det < 0) { const char* e = getenv("OCTA_DETERMINISTIC"); g_det = (e && atoi(e) != 0) ? 1 : 0; }
    return g_det == 1;
}
void octa_set_deterministic(int on) { g_det = on ? 1 : 0; }

// First-pass reductions (BatchNorm statistics / backward sums, split-attention backward sums) walk their tensor END FIRST: see
// bn_reduce_kernel.  octa_tuning_set(7, 0 / 1) or OCTA_REV_WALK=1.  Default OFF: measured in situ (tools/ab_tuning.py "7=0" "7=1",
// profiles/r05_ab_rev_walk.txt) 25.62 ms per step without against 25.63 - 26.45 with -- no gain, kept as a documented experiment.
static int g_rev = -1;
int octa_rev_walk() {
    if (g_rev < 0) { const char* e = getenv("OCTA_REV_WALK"); g_rev = (e && atoi(e) != 0) ? 1 : 0; }
    return g_rev;
}
void octa_set_rev_walk(int on) { g_rev = on ? 1 : 0; }
