// Error string storage for the C-ABI (thread local; see include/ape_hip.h).
#include <stdarg.h>
#include <stdio.h>
static thread_local char g_err[512] = "";
extern "C" const char* ape_hip_last_error(void) { return g_err; }
void ape_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" int ape_hip_abi_version(void) { return 5; }   // 2: ApeGemmArgs.rope_cs, ffn_fused / head_gemv dtype arguments; 3: ape_hip_meter_*; 4: ape_hip_sdma_*; 5: ApeGemmArgs.rowstat_cols / rowstat_eps (struct grew at its end), ape_hip_sdma_d2h_multi

// layout self-check for FFI bindings: sizeof the argument structs (0 = ApeGemmArgs, 1 = ApeLayerNormArgs, 2 = ApeGroupNormArgs)
#include "../../include/ape_hip.h"
extern "C" int ape_hip_sizeof_args(int which) {
  switch (which) {
    case 0: return (int)sizeof(ApeGemmArgs);
    case 1: return (int)sizeof(ApeLayerNormArgs);
    case 2: return (int)sizeof(ApeGroupNormArgs);
    default: return -1;
  }
}
