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

// sha1 (12 hex digits) over the kernel sources this binary was built from: build.py passes it on the command line and rebuilds this file
// whenever it changes, so the digest travels INSIDE the .so - bench.py reports it as roofline.kernels_sha, and _lib.load() refuses a
// library whose digest differs from the csrc/ next to it (an edited-but-not-rebuilt tree can no longer report a sha it did not run).
#ifndef ICD_BUILD_SHA
#define ICD_BUILD_SHA "unstamped"
#endif
extern "C" const char* icd_build_sha(void) { return ICD_BUILD_SHA; }
