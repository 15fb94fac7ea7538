// Issue cost of the softmax instructions of the flash kernels on gfx950: cycles per wave64 instruction (s_memtime around an unrolled
// block of independent instructions), one and two waves per SIMD, alone and beside MFMAs of the same wave.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip && ./valu_rate      (attention)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// KIND: 0 v_exp_f32, 1 v_fma_f32, 2 v_cvt_pk_f16_f32 (rtz builtin form is v_cvt_pkrtz; use the gfx950 v_cvt_pk_f16_f32), 3 v_max3_f32,
//       4 v_permlane16_swap, 5 v_pk_mul_f32, 6 v_exp_f16, 7: 4 MFMA 32x32x16 + 16 v_exp interleaved, 8: 4 MFMA alone, 9: 4 MFMA + 16 v_fma
template <int KIND>
__global__ __launch_bounds__(256) void k(const float* in, float* out, long long* cyc, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[(t + i * 64) & 4095];
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)x[e]; b[e] = (_Float16)x[8 + e]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float y0 = x[0], y1 = x[1];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y0), "v"(y1));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y0));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y0), "v"(y1));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[(i + 8) & 15]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 5) {
            float2* xp = reinterpret_cast<float2*>(x);
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(xp[i & 7]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_exp_f16 %0, %0" : "+v"(x[i]));
            REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
                    if constexpr (KIND == 7) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[m * 4 + i]));
                    }
                    if constexpr (KIND == 9) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[m * 4 + i]) : "v"(y0), "v"(y1));
                    }
                    if constexpr (KIND == 10) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[m * 4 + i]));
                    }
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[t] = s;
    if ((threadIdx.x & 63) == 0) cyc[t >> 6] = t1 - t0;
}

template <int KIND> void run(const char* name, int per_iter, const float* in, float* out, long long* cyc, int blocks_per_cu) {
    const int iters = 2000, grid = 256 * blocks_per_cu, nw = grid * 4;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    s /= nw;
    // s_memtime counts at a constant 100 MHz on this part; wall time gives the same figure: report both
    printf("  %-44s waves/SIMD=%d  %8.2f counter ticks / block of %d   %8.3f ns per instruction-slot (wall, per wave)\n", name, blocks_per_cu,
           s / iters, per_iter, ms * 1e6 / ((double)iters * per_iter));
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = -((i * 37) % 101) / 17.0f;
    float *in, *out; long long* cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 2 * 256 * 4); hipMalloc(&cyc, 256 * 2 * 4 * 8);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run<1>("v_fma_f32 x64", 64, in, out, cyc, w);
        run<0>("v_exp_f32 x64", 64, in, out, cyc, w);
        run<6>("v_exp_f16 x64", 64, in, out, cyc, w);
        run<2>("v_cvt_pk_f16_f32 x64", 64, in, out, cyc, w);
        run<3>("v_max3_f32 x64", 64, in, out, cyc, w);
        run<4>("v_permlane16_swap_b32 x64", 64, in, out, cyc, w);
        run<5>("v_pk_mul_f32 x64", 64, in, out, cyc, w);
        run<8>("16 MFMA 32x32x16 alone", 16, in, out, cyc, w);
        run<7>("16 MFMA 32x32x16 + 64 v_exp_f32 interleaved", 16, in, out, cyc, w);
        run<10>("16 MFMA 32x32x16 + 32 v_exp_f32 interleaved", 16, in, out, cyc, w);
        run<9>("16 MFMA 32x32x16 + 64 v_fma_f32 interleaved", 16, in, out, cyc, w);
    }
    return 0;
}
