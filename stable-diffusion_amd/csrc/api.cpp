// api.cpp - error reporting and version for libsdmi355.so
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "sdmi.h"

static thread_local char g_err[512] = "";

void sdmi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int sdmi_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sdmi_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SDMI_ELAUNCH;
  }
  return SDMI_OK;
}

static thread_local char g_kernel[160] = "";
void sdmi_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}
extern "C" const char* sdmi_last_kernel(void) { return g_kernel; }

extern "C" int sdmi_version(void) { return 101; }
#ifdef SDMI_BF16
extern "C" int sdmi_act_dtype(void) { return 1; }
#else
extern "C" int sdmi_act_dtype(void) { return 0; }
#endif
extern "C" const char* sdmi_last_error(void) { return g_err; }

// Runtime experiment switches (benchmarks / A-B runs only; defaults are the shipped behaviour): lets one process time two
// code paths interleaved (scratch/ab_step.py) instead of comparing runs on different boxes.  Table: common.h SDMI_OPTIONS.
#include <string.h>
#include "common.h"
int g_sdmi_opt[OPT_COUNT] = {
#define SDMI_OPT_DEF(name, def, doc) def,
    SDMI_OPTIONS(SDMI_OPT_DEF)
#undef SDMI_OPT_DEF
};
static const char* const g_opt_names[OPT_COUNT] = {
#define SDMI_OPT_NAME(name, def, doc) #name,
    SDMI_OPTIONS(SDMI_OPT_NAME)
#undef SDMI_OPT_NAME
};
extern "C" int sdmi_set_option(const char* name, int value) {
  for (int i = 0; i < OPT_COUNT; ++i)
    if (!strcmp(g_opt_names[i], name)) { g_sdmi_opt[i] = value; return SDMI_OK; }
  sdmi_set_error("sdmi_set_option: unknown option '%s'", name);
  return SDMI_EINVAL;
}
