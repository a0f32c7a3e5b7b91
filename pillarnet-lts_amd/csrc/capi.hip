// C-ABI plumbing shared by every entry point: version + thread-local error text.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void pn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pn_abi_version(void) { return 1; }
extern "C" const char* pn_last_error(void) { return g_err; }
