// Sustained MFMA rate under the package power limit: 32x32x16 vs 16x16x32 f16, operands in registers, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 -o mfma_power tools/mfma_power.hip && ./mfma_power [1 = zero operands]   
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(const f16x8* a, const f16x8* b, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    f16x8 av[4], bv[4];
    for (int i = 0; i < 4; ++i) { av[i] = a[(t * 4 + i) & 4095]; bv[i] = b[(t * 4 + i) & 4095]; }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[t] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(const f16x8* a, const f16x8* b, float* out, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    f16x8 av[4], bv[4];
    for (int i = 0; i < 4; ++i) { av[i] = a[(t * 4 + i) & 4095]; bv[i] = b[(t * 4 + i) & 4095]; }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[t] = s;
}
template <typename F> double run(F launch, double flop_per_launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return flop_per_launch * reps / (ms * 1e-3) / 1e12;
}
int main(int argc, char** argv) {
    const int zero = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<_Float16> h(4096 * 8 * 2);
    srand(1);
    for (auto& v : h) v = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    f16x8 *a, *b; float* out;
    hipMalloc(&a, 4096 * 16); hipMalloc(&b, 4096 * 16); hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
    hipMemcpy(a, h.data(), 4096 * 16, hipMemcpyHostToDevice); hipMemcpy(b, h.data() + 4096 * 8, 4096 * 16, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        const int grid = 256 * blocks_per_cu;        // 256 threads = 4 waves per block
        const double waves = grid * 4.0;
        printf("data=%s waves/SIMD=%d\n", zero ? "zeros" : "random", blocks_per_cu);
        printf("  32x32x16 x8 acc: %7.0f TFLOP/s\n", run([&] { hipLaunchKernelGGL(k32<8>, dim3(grid), dim3(256), 0, 0, a, b, out, iters); }, waves * iters * 8 * 32768.0, 20));
        printf("  32x32x16 x4 acc: %7.0f TFLOP/s\n", run([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, a, b, out, iters); }, waves * iters * 4 * 32768.0, 20));
        printf("  16x16x32 x16 acc:%7.0f TFLOP/s\n", run([&] { hipLaunchKernelGGL(k16<16>, dim3(grid), dim3(256), 0, 0, a, b, out, iters); }, waves * iters * 16 * 16384.0, 20));
        printf("  16x16x32 x8 acc: %7.0f TFLOP/s\n", run([&] { hipLaunchKernelGGL(k16<8>, dim3(grid), dim3(256), 0, 0, a, b, out, iters); }, waves * iters * 8 * 16384.0, 20));
    }
    // a quarter of the chip only (clock not power-limited)
    printf("64 blocks (quarter chip), random data:\n");
    printf("  32x32x16 x8 acc: %7.0f TFLOP/s (x4 = %.0f chip-equivalent)\n", run([&] { hipLaunchKernelGGL(k32<8>, dim3(64), dim3(256), 0, 0, a, b, out, iters); }, 256.0 * iters * 8 * 32768.0, 20), 0.0);
    printf("  16x16x32 x16 acc:%7.0f TFLOP/s\n", run([&] { hipLaunchKernelGGL(k16<16>, dim3(64), dim3(256), 0, 0, a, b, out, iters); }, 256.0 * iters * 16 * 16384.0, 20));
    return 0;
}
