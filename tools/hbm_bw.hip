// Achievable HBM bandwidth of this box with trivial streaming kernels: read-only (sum), write-only (fill), copy, and a strided
// tile-row pattern like a GEMM epilogue's (512-B segments at a 2560-B pitch).  16 B per lane, grid-stride, buffers far larger than
// the 256 MB Infinity Cache.   hipcc --offload-arch=gfx950 -O3 -o hbm_bw tools/hbm_bw.hip && ./hbm_bw      (DESIGN.md sections 4 / 10)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ a, long long n, float* out) {
    f4 s = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += __builtin_nontemporal_load(a + i);
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_write(f4* __restrict__ a, long long n, int nt) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (nt) __builtin_nontemporal_store(v, a + i); else a[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, long long n, int nt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const f4 v = __builtin_nontemporal_load(a + i);
        if (nt) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
// each block writes 256 rows x 512 B of a [rows][2560 B] matrix (the 256 x 256 fp16 tile of an N = 1280 output), tiles in row-major order
__global__ __launch_bounds__(256) void k_tile_write(char* __restrict__ a, long long rows) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const long long ntile = rows / 256 * 5;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long long r0 = t / 5 * 256, c0 = t % 5 * 512;
        for (int i = threadIdx.x; i < 256 * 32; i += 256) {          // 32 chunks of 16 B per row
            const int r = i >> 5, c = (i & 31) * 16;
            *reinterpret_cast<f4*>(a + (r0 + r) * 2560 + c0 + c) = v;
        }
    }
}
template <typename F> double gbps(F launch, double bytes, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return bytes * reps / (ms * 1e-3) / 1e9;
}
int main() {
    const long long bytes = 2LL << 30;                               // 2 GiB per buffer
    f4 *a, *b; float* out;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&out, 4);
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    const long long n = bytes / 16;
    for (int g : {1024, 2048, 4096, 8192}) {
        printf("grid %5d: read %7.0f GB/s   write %7.0f   write nt %7.0f   copy %7.0f (r+w)   copy nt %7.0f\n", g,
               gbps([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, out); }, (double)bytes, 5),
               gbps([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, a, n, 0); }, (double)bytes, 5),
               gbps([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, a, n, 1); }, (double)bytes, 5),
               gbps([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n, 0); }, 2.0 * bytes, 5),
               gbps([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n, 1); }, 2.0 * bytes, 5));
    }
    const long long rows = bytes / 2560 / 256 * 256;
    for (int g : {256, 512, 1024})
        printf("tile-row writes (256 rows x 512 B per block, 2560-B pitch), grid %4d: %7.0f GB/s\n", g,
               gbps([&] { hipLaunchKernelGGL(k_tile_write, dim3(g), dim3(256), 0, 0, (char*)a, rows); }, (double)rows * 2560, 5));
    // working sets that fit the 256 MB Infinity Cache
    for (long long mb : {64LL, 128LL, 512LL}) {
        const long long nn = (mb << 20) / 16;
        printf("%4lld MB working set: read %7.0f GB/s   write %7.0f   copy %7.0f (r+w)\n", mb,
               gbps([&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, nn, out); }, (double)(mb << 20), 20),
               gbps([&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, a, nn, 0); }, (double)(mb << 20), 20),
               gbps([&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, a, b, nn, 0); }, 2.0 * (mb << 20), 20));
    }
    return 0;
}
