// Library identification + per-thread last-error record for libse3ds_hip.so.
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace se3ds {
static thread_local char g_last_error[256] = "";
void set_last_error(hipError_t e, const char* where) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", where, hipGetErrorString(e));
}
}  // namespace se3ds

extern "C" {
const char* se3ds_version(void) { return "se3ds_hip abi1 gfx950"; }
const char* se3ds_last_error(void) { return se3ds::g_last_error; }
}
