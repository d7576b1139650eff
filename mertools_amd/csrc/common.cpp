// Host-side error plumbing shared by every translation unit of libmer_hip.so.
#include "common.h"
#include <string.h>

namespace mer {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MER_ELAUNCH;
  }
  return MER_OK;
}
}  // namespace mer

extern "C" const char* mer_last_error(void) { return mer::g_err; }
extern "C" const char* mer_version(void) { return "mer_hip 0.1.0 (gfx950)"; }
extern "C" const char* mer_target_arch(void) { return "gfx950"; }
