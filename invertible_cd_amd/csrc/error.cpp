// error.cpp - last-error string + version of libicd_amd.so
#include <stdarg.h>
#include <stdio.h>
#include "../../include/icd_amd.h"

static thread_local char g_err[1024] = "";

void icd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* icd_last_error(void) { return g_err; }
extern "C" int icd_version(void) { return 1; }
