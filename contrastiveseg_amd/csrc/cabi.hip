// Library-wide plumbing of libcseg_hip.so: ABI version and the thread-local error message behind the reference-style
// "return 1 = ok / 0 = error" convention (lib/extensions/cc_attention/src/ca.cu:199-204 prints and returns 0; here the
// message is retrievable through cseg_last_error()).
#include "cseg_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void cseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cseg_last_error(void) { return g_err; }
extern "C" int cseg_abi_version(void) { return 6; }   // 6: grouped launches (cseg_*_group_*), CSEG_NT_GROUP; 2: bn_* entry points, lse buffer of upsample_ce; 3: cseg_*_split_* (selectable arithmetic), cseg_amax_f32; 4: count row of the BN moments / sums; 5: cseg_contrast_fwd_fused, cseg_conv3x3_split_dil_fwd, any width in the split kernels
