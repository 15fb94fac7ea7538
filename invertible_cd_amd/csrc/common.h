// Shared device/host helpers for the gfx950 kernels of libicd_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/icd_amd.h"

typedef _Float16 half_t;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// error plumbing (host)
void icd_set_error(const char* fmt, ...);
#define ICD_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            icd_set_error(__VA_ARGS__);          \
            return ICD_ERR_INVALID_ARG;          \
        }                                        \
    } while (0)
#define ICD_CHECK_LAUNCH(what)                                                         \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            icd_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e__));  \
            return ICD_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#ifdef __HIPCC__
// 16 bytes of zeros for out-of-bounds im2col taps / ragged tiles (global_load_lds needs a real source address)
static __device__ __attribute__((aligned(256))) unsigned char icd_zero_page[256];   // zero-initialised, one per TU

// async global -> LDS copy of 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
#endif
