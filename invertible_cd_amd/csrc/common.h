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
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

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

// Cross-lane reductions on the VALU (DPP within a row of 16 lanes, v_permlane{16,32}_swap across rows): no LDS round
// trips.  group_* reduce aligned groups of W lanes (W = 4 .. 64); every lane of the group receives the result.
__device__ __forceinline__ float dpp_mov(float v, int ctrl_quad_xor1_xor2_hmirror_mirror) {
    switch (ctrl_quad_xor1_xor2_hmirror_mirror) {
    case 0: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));  // row_half_mirror
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)); // row_mirror
    }
}
template <int W, typename Op>
__device__ __forceinline__ float group_reduce(float v, Op op) {
    v = op(v, dpp_mov(v, 0));
    v = op(v, dpp_mov(v, 1));
    if (W >= 8) v = op(v, dpp_mov(v, 2));
    if (W >= 16) v = op(v, dpp_mov(v, 3));
    if (W >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    if (W >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    return v;
}
template <int W> __device__ __forceinline__ float group_sum(float v) { return group_reduce<W>(v, [](float a, float b) { return a + b; }); }
template <int W> __device__ __forceinline__ float group_max(float v) { return group_reduce<W>(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }
#endif
