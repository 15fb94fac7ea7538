// norm.hip - HBM-bound normalisation kernels of the UNet path (gfx950): GroupNorm(+SiLU), LayerNorm, row softmax.
//
// All of them move fp16 NHWC / token-major activations with 16-byte per-lane accesses; statistics are fp32 and
// reduced in a fixed order (no atomics) so results are bit-reproducible run to run.
#include <algorithm>
#include "common.h"

namespace {

constexpr int GN_THREADS = 512;

// pixels per statistics block: about 768 blocks per launch (3 per CU) but never fewer than 64 pixels per block
// (round-3 sweep, profiles/r03_gn_sweep.txt: 768 statistics blocks per launch beat 1536 by 2-13 % on the C = 320 levels and tie
//  elsewhere; 3072 never wins)
__host__ __device__ inline int gn_pix_per_split(int B, int HW, int tgt = 768) {
    int target = tgt / (B > 0 ? B : 1);
    if (target < 1) target = 1;
    const int most = HW / 64 > 0 ? HW / 64 : 1;
    const int nsplit = target < most ? target : most;
    return (HW + nsplit - 1) / nsplit;
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm pass 1: per (sample, pixel-slab) partial (sum, sumsq) per group.  Thread -> fixed 8-channel chunk.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const half_t* __restrict__ x0, int C0,
                                                               const half_t* __restrict__ x1, int C1, int HW,
                                                               int groups, int pix_per_split, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* part = reinterpret_cast<float*>(smem_raw);          // [rows_per_iter][C][2]
    const int C = C0 + C1, nchunk = C >> 3;
    const int rpi = GN_THREADS / nchunk;
    const int tid = threadIdx.x;
    const int chunk = tid % nchunk, rsub = tid / nchunk;
    const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
    const int p_begin = sp * pix_per_split;
    const int p_end = min(HW, p_begin + pix_per_split);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (rsub < rpi) {
        const int ch = chunk * 8;
        const bool first = ch < C0;
        const half_t* base = first ? x0 + (long long)b * HW * C0 + ch : x1 + (long long)b * HW * C1 + (ch - C0);
        const int cs = first ? C0 : C1;
        // 4 independent 16-B loads in flight per thread per iteration (HBM-bound: keep the memory pipe full)
        int p = p_begin + rsub;
        for (; p + 3 * rpi < p_end; p += 4 * rpi) {
            f16x8 v0 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)p * cs));
            f16x8 v1 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)(p + rpi) * cs));
            f16x8 v2 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)(p + 2 * rpi) * cs));
            f16x8 v3 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)(p + 3 * rpi) * cs));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f0 = (float)v0[e], f1 = (float)v1[e], f2 = (float)v2[e], f3 = (float)v3[e];
                s[e] += (f0 + f1) + (f2 + f3);
                q[e] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
            }
        }
        for (; p < p_end; p += rpi) {
            f16x8 v = *reinterpret_cast<const f16x8*>(base + (long long)p * cs);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] += f * f; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            part[((rsub * C) + ch + e) * 2 + 0] = s[e];
            part[((rsub * C) + ch + e) * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    if (tid < groups) {
        const int cpg = C / groups;
        float ss = 0.f, qq = 0.f;
        for (int r = 0; r < rpi; ++r)
            for (int c = 0; c < cpg; ++c) {
                ss += part[((r * C) + tid * cpg + c) * 2 + 0];
                qq += part[((r * C) + tid * cpg + c) * 2 + 1];
            }
        float* o = ws + (((long long)b * nsplit + sp) * groups + tid) * 2;
        o[0] = ss; o[1] = qq;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm pass 2: finish the statistics (fixed order), normalise, affine, optional SiLU, write one NHWC tensor.
// ---------------------------------------------------------------------------------------------------------------
// CARRY (round 5): the inputs are tensors of the residual stream that travel with their bf8 error carry (value = fp16 + 2^-14 * bf8,
// gemm_common.h) - the normalisation reads fp16 + carry (the statistics pass keeps reading the fp16 part: a mean over thousands of
// elements does not see rounding noise), i.e. one byte more per element on this pass only.  `aux` (optional, needs CARRY): the operand
// the split shortcut conv of a ResnetBlock2D reads beside x0 - with two sources [x1 (C1) | lo0 (C0) | lo1 (C1)], with one source
// [lo0 (C0)], lo = fp16(2^-14 * carry), row stride ld_aux - so that conv sees x0 + lo0 and x1 + lo1 through its two-source loader
// (weights [W0 | W1 | W0 | W1]); written here because this kernel is the one that already holds x and its carry in registers.
struct GnCarry { const unsigned char* c0; const unsigned char* c1; half_t* aux; int ld_aux; };

__device__ __forceinline__ void gn_carry8(float (&f)[8], const u32x2 w) {
    const f32x2 c0 = __builtin_amdgcn_cvt_pk_f32_bf8(w[0], false), c1 = __builtin_amdgcn_cvt_pk_f32_bf8(w[0], true);
    const f32x2 c2 = __builtin_amdgcn_cvt_pk_f32_bf8(w[1], false), c3 = __builtin_amdgcn_cvt_pk_f32_bf8(w[1], true);
    const float k = 1.f / 16384.f;
    f[0] = c0[0] * k; f[1] = c0[1] * k; f[2] = c1[0] * k; f[3] = c1[1] * k;
    f[4] = c2[0] * k; f[5] = c2[1] * k; f[6] = c3[0] * k; f[7] = c3[1] * k;
}

template <bool CARRY>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const half_t* __restrict__ x0, int C0,
                                                               const half_t* __restrict__ x1, int C1, int HW,
                                                               int groups, int nsplit, int pix_per_block,
                                                               const float* __restrict__ ws,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int silu,
                                                               half_t* __restrict__ out, GnCarry cy) {
    __shared__ float s_mean[64], s_rstd[64];
    const int C = C0 + C1, nchunk = C >> 3;
    const int rpi = GN_THREADS / nchunk;
    const int tid = threadIdx.x, b = blockIdx.y;
    {   // finish the statistics: 16 lanes per group walk the partials (fixed order -> reproducible), DPP tree on top
        const int g = tid >> 4, j = tid & 15;
        float ss = 0.f, qq = 0.f;
        if (g < groups)
            for (int sp = j; sp < nsplit; sp += 16) {
                const float* o = ws + (((long long)b * nsplit + sp) * groups + g) * 2;
                ss += o[0]; qq += o[1];
            }
        ss = group_sum<16>(ss); qq = group_sum<16>(qq);
        if (g < groups && j == 0) {
            const float n = (float)HW * (float)(C / groups);
            const float mean = ss / n;
            const float var = fmaxf(qq / n - mean * mean, 0.f);
            s_mean[g] = mean;
            s_rstd[g] = rsqrtf(var + eps);
        }
    }
    __syncthreads();
    const int chunk = tid % nchunk, rsub = tid / nchunk;
    if (rsub >= rpi) return;
    const int ch = chunk * 8, cpg = C / groups;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (ch + e) / cpg;
        sc[e] = s_rstd[g] * gamma[ch + e];
        sh[e] = beta[ch + e] - s_mean[g] * sc[e];
    }
    const bool first = ch < C0;
    const half_t* base = first ? x0 + (long long)b * HW * C0 + ch : x1 + (long long)b * HW * C1 + (ch - C0);
    const int cs = first ? C0 : C1;
    half_t* ob = out + (long long)b * HW * C + ch;
    const int p_begin = blockIdx.x * pix_per_block;
    const int p_end = min(HW, p_begin + pix_per_block);
    // carry bytes of this thread's 8 channels (same [pixel][channel] layout as the fp16 tensor, one byte per element)
    const unsigned char* cbase = nullptr;
    half_t* aux_lo = nullptr; half_t* aux_hi = nullptr;
    if (CARRY) {
        const unsigned char* csrc = first ? cy.c0 : cy.c1;
        if (csrc) cbase = csrc + (long long)b * HW * cs + (first ? ch : ch - C0);
        if (cy.aux) {
            half_t* arow = cy.aux + (long long)b * HW * cy.ld_aux;
            aux_lo = arow + (C1 ? C1 : 0) + ch;                     // [x1 | lo0 | lo1]: lo of concat channel ch sits at C1 + ch
            if (!first) aux_hi = arow + (ch - C0);
        }
    }
    auto norm8 = [&](const f16x8& v, const u32x2 w, long long prow) {
        f16x8 o;
        float cf[8];
        if (CARRY) gn_carry8(cf, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e];
            if (CARRY) f += cf[e];
            f = f * sc[e] + sh[e];
            if (silu) f = silu_f(f);
            o[e] = (half_t)f;
        }
        if (CARRY && aux_lo) {
            f16x8 lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) lo[e] = (half_t)cf[e];
            *reinterpret_cast<f16x8*>(aux_lo + prow * cy.ld_aux) = lo;
            if (aux_hi) *reinterpret_cast<f16x8*>(aux_hi + prow * cy.ld_aux) = v;
        }
        return o;
    };
    auto ldc = [&](long long prow) {
        u32x2 w = {0u, 0u};
        if (CARRY && cbase) w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(cbase + prow * cs));
        return w;
    };
    int p = p_begin + rsub;
    for (; p + 3 * rpi < p_end; p += 4 * rpi) {          // 4 loads in flight per thread
        f16x8 v0 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)p * cs));
        f16x8 v1 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)(p + rpi) * cs));
        f16x8 v2 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)(p + 2 * rpi) * cs));
        f16x8 v3 = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(base + (long long)(p + 3 * rpi) * cs));
        const u32x2 w0 = ldc(p), w1 = ldc(p + rpi), w2 = ldc(p + 2 * rpi), w3 = ldc(p + 3 * rpi);
        *reinterpret_cast<f16x8*>(ob + (long long)p * C) = norm8(v0, w0, p);
        *reinterpret_cast<f16x8*>(ob + (long long)(p + rpi) * C) = norm8(v1, w1, p + rpi);
        *reinterpret_cast<f16x8*>(ob + (long long)(p + 2 * rpi) * C) = norm8(v2, w2, p + 2 * rpi);
        *reinterpret_cast<f16x8*>(ob + (long long)(p + 3 * rpi) * C) = norm8(v3, w3, p + 3 * rpi);
    }
    for (; p < p_end; p += rpi)
        *reinterpret_cast<f16x8*>(ob + (long long)p * C) = norm8(*reinterpret_cast<const f16x8*>(base + (long long)p * cs), ldc(p), p);
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm for small feature maps (H*W <= 1024: the 32x32 / 16x16 / 8x8 levels), ONE launch: a block owns one sample's
// slab of GPB whole groups (CB = GPB * C/groups channels, a multiple of 8), reads it once for the statistics, reduces in
// LDS in a fixed order, then re-reads it (L2-resident: <= 160 KiB per block) to normalise.  No workspace, no second
// launch: at small batch the two-kernel form is pure launch latency (61 GroupNorms = 20 % of a B = 1 UNet evaluation).
// ---------------------------------------------------------------------------------------------------------------
template <bool CARRY>
__global__ __launch_bounds__(GN_THREADS) void gn_small_kernel(const half_t* __restrict__ x0, int C0,
                                                               const half_t* __restrict__ x1, int C1, int HW, int groups,
                                                               int GPB, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int silu,
                                                               half_t* __restrict__ out, GnCarry cy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int C = C0 + C1, cpg = C / groups, CB = GPB * cpg, nch = CB >> 3;
    const int rpi = GN_THREADS / nch;
    float* part = reinterpret_cast<float*>(smem_raw);          // [rpi][CB][2]
    float* chan = part + rpi * CB * 2;                          // [CB][2]  per-channel totals
    float* stat = chan + CB * 2;                                // [GPB][2] mean, rstd
    const int tid = threadIdx.x, b = blockIdx.y;
    const int c_lo = blockIdx.x * CB;
    const int chunk = tid % nch, rsub = tid / nch;
    const bool live = rsub < rpi;
    const int ch = c_lo + chunk * 8;
    const bool first = ch < C0;
    const half_t* base = first ? x0 + (long long)b * HW * C0 + ch : x1 + (long long)b * HW * C1 + (ch - C0);
    const int cs = first ? C0 : C1;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (live) {
        int p = rsub;
        for (; p + 3 * rpi < HW; p += 4 * rpi) {
            f16x8 v0 = *reinterpret_cast<const f16x8*>(base + (long long)p * cs);
            f16x8 v1 = *reinterpret_cast<const f16x8*>(base + (long long)(p + rpi) * cs);
            f16x8 v2 = *reinterpret_cast<const f16x8*>(base + (long long)(p + 2 * rpi) * cs);
            f16x8 v3 = *reinterpret_cast<const f16x8*>(base + (long long)(p + 3 * rpi) * cs);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f0 = (float)v0[e], f1 = (float)v1[e], f2 = (float)v2[e], f3 = (float)v3[e];
                s[e] += (f0 + f1) + (f2 + f3);
                q[e] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
            }
        }
        for (; p < HW; p += rpi) {
            f16x8 v = *reinterpret_cast<const f16x8*>(base + (long long)p * cs);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] += f * f; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            part[((rsub * CB) + chunk * 8 + e) * 2 + 0] = s[e];
            part[((rsub * CB) + chunk * 8 + e) * 2 + 1] = q[e];
        }
    }
    __syncthreads();
    if (tid < CB) {                                             // per-channel totals over the row slots (fixed order)
        float ss = 0.f, qq = 0.f;
        for (int r = 0; r < rpi; ++r) { ss += part[(r * CB + tid) * 2]; qq += part[(r * CB + tid) * 2 + 1]; }
        chan[tid * 2] = ss; chan[tid * 2 + 1] = qq;
    }
    __syncthreads();
    if (tid < GPB) {
        float ss = 0.f, qq = 0.f;
        for (int c = 0; c < cpg; ++c) { ss += chan[(tid * cpg + c) * 2]; qq += chan[(tid * cpg + c) * 2 + 1]; }
        const float n = (float)HW * (float)cpg;
        const float mean = ss / n;
        stat[tid * 2] = mean;
        stat[tid * 2 + 1] = rsqrtf(fmaxf(qq / n - mean * mean, 0.f) + eps);
    }
    __syncthreads();
    if (!live) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (chunk * 8 + e) / cpg;
        sc[e] = stat[g * 2 + 1] * gamma[ch + e];
        sh[e] = beta[ch + e] - stat[g * 2] * sc[e];
    }
    half_t* ob = out + (long long)b * HW * C + ch;
    const unsigned char* cbase = nullptr;
    half_t* aux_lo = nullptr; half_t* aux_hi = nullptr;
    if (CARRY) {
        const unsigned char* csrc = first ? cy.c0 : cy.c1;
        if (csrc) cbase = csrc + (long long)b * HW * cs + (first ? ch : ch - C0);
        if (cy.aux) {
            half_t* arow = cy.aux + (long long)b * HW * cy.ld_aux;
            aux_lo = arow + C1 + ch;
            if (!first) aux_hi = arow + (ch - C0);
        }
    }
    auto norm8 = [&](const f16x8& v, const u32x2 w, long long prow) {
        f16x8 o;
        float cf[8];
        if (CARRY) gn_carry8(cf, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)v[e];
            if (CARRY) f += cf[e];
            f = f * sc[e] + sh[e];
            if (silu) f = silu_f(f);
            o[e] = (half_t)f;
        }
        if (CARRY && aux_lo) {
            f16x8 lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) lo[e] = (half_t)cf[e];
            *reinterpret_cast<f16x8*>(aux_lo + prow * cy.ld_aux) = lo;
            if (aux_hi) *reinterpret_cast<f16x8*>(aux_hi + prow * cy.ld_aux) = v;
        }
        return o;
    };
    auto ldc = [&](long long prow) {
        u32x2 w = {0u, 0u};
        if (CARRY && cbase) w = *reinterpret_cast<const u32x2*>(cbase + prow * cs);
        return w;
    };
    int p = rsub;
    for (; p + 3 * rpi < HW; p += 4 * rpi) {
        f16x8 v0 = *reinterpret_cast<const f16x8*>(base + (long long)p * cs);
        f16x8 v1 = *reinterpret_cast<const f16x8*>(base + (long long)(p + rpi) * cs);
        f16x8 v2 = *reinterpret_cast<const f16x8*>(base + (long long)(p + 2 * rpi) * cs);
        f16x8 v3 = *reinterpret_cast<const f16x8*>(base + (long long)(p + 3 * rpi) * cs);
        const u32x2 w0 = ldc(p), w1 = ldc(p + rpi), w2 = ldc(p + 2 * rpi), w3 = ldc(p + 3 * rpi);
        *reinterpret_cast<f16x8*>(ob + (long long)p * C) = norm8(v0, w0, p);
        *reinterpret_cast<f16x8*>(ob + (long long)(p + rpi) * C) = norm8(v1, w1, p + rpi);
        *reinterpret_cast<f16x8*>(ob + (long long)(p + 2 * rpi) * C) = norm8(v2, w2, p + 2 * rpi);
        *reinterpret_cast<f16x8*>(ob + (long long)(p + 3 * rpi) * C) = norm8(v3, w3, p + 3 * rpi);
    }
    for (; p < HW; p += rpi)
        *reinterpret_cast<f16x8*>(ob + (long long)p * C) = norm8(*reinterpret_cast<const f16x8*>(base + (long long)p * cs), ldc(p), p);
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm of the fp32-fidelity VAE path (vae.py `.to(torch.float32)`, the reference's upcast of the SDXL VAE,
// utils/generation_sdxl.py:465-466): fp32 input [B*HW, C] (the residual stream / conv outputs are kept in fp32 there because
// real SDXL-VAE activations exceed the fp16 range), statistics in fp32, output in the "split3" operand format of that
// path: every normalised value v leaves as hi = fp16(v), lo = fp16(v - hi) in a [B*HW, 3C] tensor laid out [hi | lo | hi],
// so that the unchanged fp16 MFMA GEMMs compute (a_hi + a_lo)(w_hi + w_lo) - a_lo w_lo against weights packed
// [w_hi | w_hi | w_lo]: ~2^-21 relative operand error instead of 2^-11.  Thread = 4 channels (one 16-B fp32 load).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GN_THREADS) void gn_stats_f32_kernel(const float* __restrict__ x, int C, int HW, int groups,
                                                                   int pix_per_split, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* part = reinterpret_cast<float*>(smem_raw);          // [rpi][C][2]
    const int nchunk = C >> 2, rpi = GN_THREADS / nchunk;
    const int tid = threadIdx.x, chunk = tid % nchunk, rsub = tid / nchunk;
    const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
    const int p_begin = sp * pix_per_split, p_end = min(HW, p_begin + pix_per_split);
    if (rsub < rpi) {
        // Sums of (x - pivot), pivot = the group's first element of this sample: real SDXL-VAE activations have |mean| >> std in
        // some groups (the reason this path exists), and E[x^2] - mean^2 on raw fp32 sums would cancel several digits there.
        // With the pivot inside the group's range the sums stay O(std), the variance keeps ~7 digits (torch uses Welford).
        float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, pv[4];
        const int cpg0 = C / groups;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = x[(long long)b * HW * C + ((chunk * 4 + e) / cpg0) * cpg0];
        const float* base = x + (long long)b * HW * C + chunk * 4;
        for (int p = p_begin + rsub; p < p_end; p += rpi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)p * C);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dv = v[e] - pv[e]; s[e] += dv; q[e] += dv * dv; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { part[((rsub * C) + chunk * 4 + e) * 2] = s[e]; part[((rsub * C) + chunk * 4 + e) * 2 + 1] = q[e]; }
    }
    __syncthreads();
    if (tid < groups) {
        const int cpg = C / groups;
        float ss = 0.f, qq = 0.f;
        for (int r = 0; r < rpi; ++r)
            for (int c = 0; c < cpg; ++c) { ss += part[((r * C) + tid * cpg + c) * 2]; qq += part[((r * C) + tid * cpg + c) * 2 + 1]; }
        float* o = ws + (((long long)b * nsplit + sp) * groups + tid) * 2;
        o[0] = ss; o[1] = qq;
    }
}

__device__ __forceinline__ void split_store4(half_t* row, int C, int c, const float (&v)[4]) {
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (half_t)v[e]; lo[e] = (half_t)(v[e] - (float)hi[e]); }
    *reinterpret_cast<f16x4*>(row + c) = hi;
    *reinterpret_cast<f16x4*>(row + C + c) = lo;
    *reinterpret_cast<f16x4*>(row + 2 * C + c) = hi;
}

__global__ __launch_bounds__(GN_THREADS) void gn_apply_split_kernel(const float* __restrict__ x, int C, int HW, int groups, int nsplit,
                                                                     int pix_per_block, const float* __restrict__ ws,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float eps, int silu, half_t* __restrict__ out) {
    __shared__ float s_mean[64], s_rstd[64];
    const int nchunk = C >> 2, rpi = GN_THREADS / nchunk;
    const int tid = threadIdx.x, b = blockIdx.y;
    {
        const int g = tid >> 4, j = tid & 15;
        float ss = 0.f, qq = 0.f;
        if (g < groups)
            for (int sp = j; sp < nsplit; sp += 16) {
                const float* o = ws + (((long long)b * nsplit + sp) * groups + g) * 2;
                ss += o[0]; qq += o[1];
            }
        ss = group_sum<16>(ss); qq = group_sum<16>(qq);
        if (g < groups && j == 0) {
            const float n = (float)HW * (float)(C / groups);
            const float dm = ss / n;                             // mean of (x - pivot), see gn_stats_f32_kernel
            s_mean[g] = x[(long long)b * HW * C + g * (C / groups)] + dm;
            s_rstd[g] = rsqrtf(fmaxf(qq / n - dm * dm, 0.f) + eps);
        }
    }
    __syncthreads();
    const int chunk = tid % nchunk, rsub = tid / nchunk;
    if (rsub >= rpi) return;
    const int ch = chunk * 4, cpg = C / groups;
    float sc[4], mu[4], be[4];                  // (v - mean) * sc + beta: the folded form v * sc + (beta - mean * sc) would round the
#pragma unroll                                 // shift at the magnitude of mean * sc - digits this path exists to keep
    for (int e = 0; e < 4; ++e) {
        const int g = (ch + e) / cpg;
        sc[e] = s_rstd[g] * gamma[ch + e];
        mu[e] = s_mean[g];
        be[e] = beta[ch + e];
    }
    const float* base = x + (long long)b * HW * C + ch;
    half_t* ob = out + (long long)b * HW * 3 * C;
    const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
    for (int p = p_begin + rsub; p < p_end; p += rpi) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)p * C);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float f = (v[e] - mu[e]) * sc[e] + be[e];
            if (silu) f = f / (1.0f + expf(-f));                 // exact expf: this path trades speed for fp32 fidelity
            o[e] = f;
        }
        split_store4(ob + (long long)p * 3 * C, C, ch, o);
    }
}

// x fp32 [rows, C] * scale -> split3 fp16 [rows, 3C]  (raw residual-stream tensors that feed a conv without a GroupNorm in
// between - Upsample2D / Downsample2D / conv_shortcut inputs, the packed latent / image: scaled by a power of two so that
// the fp16 range is never exceeded; the consumer multiplies by 1 / scale in its epilogue, which is exact)
__global__ __launch_bounds__(256) void split_cast_kernel(const float* __restrict__ x, long long rows, int C, float scale,
                                                          half_t* __restrict__ out) {
    const int nchunk = C >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nchunk) return;
    const long long r = idx / nchunk;
    const int c = (int)(idx - r * nchunk) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c);
    const float o[4] = {v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale};
    split_store4(out + r * 3 * C, C, c, o);
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per token row, row kept in registers (C <= 2048), exact two-pass mean / variance.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, long long rows, int C,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         half_t* __restrict__ out) {
    const int l = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = C >> 3;
    const half_t* xr = x + row * C;
    f16x8 v[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l + i * 64;
        if (c < nchunk) {
            v[i] = *reinterpret_cast<const f16x8*>(xr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[i][e];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l + i * 64;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    half_t* orow = out + row * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l + i * 64;
        if (c < nchunk) {
            f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + c * 8 + 4);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c * 8), b1 = *reinterpret_cast<const f32x4*>(beta + c * 8 + 4);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)(((float)v[i][e] - mean) * rstd * g0[e] + b0[e]);
                o[4 + e] = (half_t)(((float)v[i][4 + e] - mean) * rstd * g1[e] + b1[e]);
            }
            *reinterpret_cast<f16x8*>(orow + c * 8) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm, several rows per wave: LPR lanes own one row (CPL 16-B chunks each), so all 64 lanes carry data at
// C = 320 / 640 / 1280 (LPR = 8 / 16 / 32, CPL = 5); gamma / beta stay in registers for the LN_ITERS row groups a wave
// walks through (re-reading them per row costs 4x the row's own bytes through L1), and the next group's loads are in
// flight while the current one is reduced (DPP / permlane, no LDS).  Same exact two-pass arithmetic as above.
// ---------------------------------------------------------------------------------------------------------------
constexpr int LN_ITERS = 4;

// STATS: write (mean, rstd) per row to `stats` instead of the normalised rows - the LayerNorm itself is then applied
// inside the consuming GEMM (gamma folded into its weights, mean / rstd as a rank-1 epilogue correction; icd_gemm_desc
// ln_stats / ln_colsum): the normalised tensor is never written or re-read.
template <int LPR, int CPL, bool STATS = false>
__global__ __launch_bounds__(256) void layernorm_multi_kernel(const half_t* __restrict__ x, long long rows, int C,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               half_t* __restrict__ out, float* __restrict__ stats) {
    constexpr int RPW = 64 / LPR;
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = l % LPR, rw = l / LPR;
    const long long row0 = ((long long)blockIdx.x * 4 + wv) * (RPW * LN_ITERS) + rw;
    f32x4 g[CPL][2], bt[CPL][2];
    if (!STATS) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = (sub + i * LPR) * 8;
        g[i][0] = *reinterpret_cast<const f32x4*>(gamma + c); g[i][1] = *reinterpret_cast<const f32x4*>(gamma + c + 4);
        bt[i][0] = *reinterpret_cast<const f32x4*>(beta + c); bt[i][1] = *reinterpret_cast<const f32x4*>(beta + c + 4);
    }
    }
    const float invC = 1.0f / (float)C;
    f16x8 v[2][CPL];
    auto load = [&](int buf, long long row) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            if (row < rows) v[buf][i] = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(x + row * C + (sub + i * LPR) * 8));
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) v[buf][i][e] = (half_t)0.f;
        }
    };
    load(0, row0);
#pragma unroll
    for (int it = 0; it < LN_ITERS; ++it) {
        const long long row = row0 + (long long)it * RPW;
        const int cur = it & 1;
        if (it + 1 < LN_ITERS) load(cur ^ 1, row + RPW);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[cur][i][e];
        const float mean = group_sum<LPR>(sum) * invC;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[cur][i][e] - mean; sq += d * d; }
        const float rstd = rsqrtf(group_sum<LPR>(sq) * invC + eps);
        if (STATS) {
            if (row < rows && sub == 0) *reinterpret_cast<f32x2*>(stats + row * 2) = (f32x2){mean, rstd};
            continue;
        }
        if (row < rows) {
            half_t* orow = out + row * C;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (half_t)(((float)v[cur][i][e] - mean) * rstd * g[i][0][e] + bt[i][0][e]);
                    o[4 + e] = (half_t)(((float)v[cur][i][4 + e] - mean) * rstd * g[i][1][e] + bt[i][1][e]);
                }
                *reinterpret_cast<f16x8*>(orow + (sub + i * LPR) * 8) = o;
            }
        }
    }
}

template <int LPR, int CPL>
void launch_ln_multi(const half_t* x, long long rows, int C, const float* gamma, const float* beta, float eps, half_t* out,
                     float* stats, hipStream_t st) {
    const long long rows_per_block = 4LL * (64 / LPR) * LN_ITERS;
    const dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
    if (stats) hipLaunchKernelGGL((layernorm_multi_kernel<LPR, CPL, true>), grid, dim3(256), 0, st, x, rows, C, gamma, beta, eps, out, stats);
    else hipLaunchKernelGGL((layernorm_multi_kernel<LPR, CPL, false>), grid, dim3(256), 0, st, x, rows, C, gamma, beta, eps, out, stats);
}

// generic width: one wave per row, stats only
__global__ __launch_bounds__(256) void layernorm_stats_kernel(const half_t* __restrict__ x, long long rows, int C, float eps,
                                                               float* __restrict__ stats) {
    const int l = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = C >> 3;
    const half_t* xr = x + row * C;
    f16x8 v[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l + i * 64;
        if (c < nchunk) {
            v[i] = *reinterpret_cast<const f16x8*>(xr + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[i][e];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l + i * 64;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    if (l == 0) *reinterpret_cast<f32x2*>(stats + row * 2) = (f32x2){mean, rstd};
}

// ---------------------------------------------------------------------------------------------------------------
// Row softmax for the materialised-P path: fp32 scores -> fp16 probabilities, pad columns zeroed. One wave / row.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long long rows, int cols,
                                                            int ld_s, float scale, half_t* __restrict__ p, int ld_p) {
    const int l = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * ld_s;
    float mx = -INFINITY;
    for (int c = l; c < cols; c += 64) mx = fmaxf(mx, sr[c]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = l; c < cols; c += 64) sum += __expf((sr[c] - mx) * scale);
    const float inv = 1.0f / wave_sum(sum);
    half_t* pr = p + row * ld_p;
    for (int c = l; c < ld_p; c += 64) pr[c] = c < cols ? (half_t)(__expf((sr[c] - mx) * scale) * inv) : (half_t)0.f;
}

// Row in registers: one global read of the fp32 scores, one exp per element, 16-B loads / 8-B stores.  NV = float4
// vectors per lane (cols <= 256 * NV, ld_s % 4 == 0, ld_p % 4 == 0).
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ s, long long rows, int cols,
                                                                int ld_s, float scale, half_t* __restrict__ p, int ld_p) {
    const int l = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = s + row * ld_s;
    f32x4 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (l + i * 64) * 4;
        if (c < cols) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sr + c));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c + e >= cols) v[i][e] = -INFINITY;          // ragged tail / lanes past the row
            mx = fmaxf(mx, v[i][e]);
        }
    }
    mx = wave_max(mx);
    const float k = scale * 1.4426950408889634f, nm = -mx * k;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = __builtin_amdgcn_exp2f(fmaf(v[i][e], k, nm)); sum += v[i][e]; }
    const float inv = 1.0f / wave_sum(sum);
    half_t* pr = p + row * ld_p;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (l + i * 64) * 4;
        if (c < ld_p) {
            f16x4 o = {(half_t)(v[i][0] * inv), (half_t)(v[i][1] * inv), (half_t)(v[i][2] * inv), (half_t)(v[i][3] * inv)};
            *reinterpret_cast<f16x4*>(pr + c) = o;            // exp2(-inf) = 0 -> pad columns are written as zeros
        }
    }
}

}  // namespace

extern "C" int64_t icd_groupnorm_ws_floats(int32_t B, int32_t HW, int32_t groups) {
    const int pps = gn_pix_per_split(B, HW, 4096);
    const int nsplit = (HW + pps - 1) / pps;
    return (int64_t)B * nsplit * groups * 2;
}

static int groupnorm_run(const void* x0, int32_t C0, const void* c0, const void* x1, int32_t C1, const void* c1, int32_t B, int32_t HW,
                         int32_t groups, const float* gamma, const float* beta, float eps, int32_t silu, void* out, void* aux,
                         int32_t ld_aux, float* stats_ws, void* stream, const char* who) {
    ICD_CHECK_ARG(x0 && out && stats_ws && gamma && beta, "%s: null pointer", who);
    const int C = C0 + C1;
    ICD_CHECK_ARG(C0 > 0 && C0 % 8 == 0 && C1 >= 0 && C1 % 8 == 0, "%s: channels must be multiples of 8", who);
    ICD_CHECK_ARG((C1 == 0) == (x1 == nullptr), "%s: x1/C1 mismatch", who);
    ICD_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0, "%s: bad group count %d for C=%d", who, groups, C);
    ICD_CHECK_ARG(C / 8 <= GN_THREADS, "%s: C=%d too large", who, C);
    ICD_CHECK_ARG(B > 0 && HW > 0, "%s: empty input", who);
    ICD_CHECK_ARG(!(c1 && !x1), "%s: carry of an absent second source", who);
    ICD_CHECK_ARG(!aux || (ld_aux % 8 == 0 && ld_aux >= C0 + 2 * C1), "%s: aux needs ld_aux %% 8 == 0 and >= C0 + 2 * C1", who);
    const bool carry = c0 || c1 || aux;
    const GnCarry cy{(const unsigned char*)c0, (const unsigned char*)c1, (half_t*)aux, ld_aux};
    hipStream_t st = (hipStream_t)stream;
    {   // small maps: one launch, a block per (sample, slab of whole groups whose channel count is a multiple of 8)
        const int cpg = C / groups;
        int gpb = 1;
        while ((gpb * cpg) % 8 != 0) gpb *= 2;                  // 8 / gcd(cpg, 8): 1, 2, 4 or 8
        const int cb = gpb * cpg;
        // the slab must not straddle the two concat sources mid-chunk (C0 % 8 == 0 holds) and must tile the groups
        // measured: wins up to ~12 M elements (every level at small batch, the 16x16 / 8x8 levels at B = 32); above that the
        // two streaming kernels with their wider rows are faster
        if (HW <= 1024 && (long long)B * HW * C <= 12LL * 1024 * 1024 && groups % gpb == 0 && cb <= GN_THREADS) {
            const int rpi_s = GN_THREADS / (cb / 8);
            const size_t smem_s = ((size_t)rpi_s * cb * 2 + (size_t)cb * 2 + (size_t)gpb * 2) * sizeof(float);
            if (smem_s <= 64 * 1024) {
                if (carry)
                    hipLaunchKernelGGL(gn_small_kernel<true>, dim3(groups / gpb, B), dim3(GN_THREADS), smem_s, st, (const half_t*)x0, C0,
                                       (const half_t*)x1, C1, HW, groups, gpb, gamma, beta, eps, silu, (half_t*)out, cy);
                else
                    hipLaunchKernelGGL(gn_small_kernel<false>, dim3(groups / gpb, B), dim3(GN_THREADS), smem_s, st, (const half_t*)x0, C0,
                                       (const half_t*)x1, C1, HW, groups, gpb, gamma, beta, eps, silu, (half_t*)out, cy);
                ICD_CHECK_LAUNCH("icd_groupnorm(small)");
                return ICD_OK;
            }
        }
    }
    const int pps = gn_pix_per_split(B, HW);
    const int nsplit = (HW + pps - 1) / pps;
    const int rpi = GN_THREADS / (C / 8);
    const size_t smem = (size_t)rpi * C * 2 * sizeof(float);
    ICD_CHECK_ARG(smem <= 64 * 1024, "%s: LDS budget exceeded", who);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, B), dim3(GN_THREADS), smem, st, (const half_t*)x0, C0,
                       (const half_t*)x1, C1, HW, groups, pps, stats_ws);
    ICD_CHECK_LAUNCH("icd_groupnorm(stats)");
    // pixels per apply block: 512 - 1024 blocks per launch (same sweep: 128 at B x HW = 131072, 64 at 32768; 256 only pays on the
    // 128 x 128 maps and by < 2 %)
    const int ppb = (long long)B * HW >= 98304 ? 128 : 64;
    if (carry)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((HW + ppb - 1) / ppb, B), dim3(GN_THREADS), 0, st, (const half_t*)x0, C0,
                           (const half_t*)x1, C1, HW, groups, nsplit, ppb, stats_ws, gamma, beta, eps, silu, (half_t*)out, cy);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((HW + ppb - 1) / ppb, B), dim3(GN_THREADS), 0, st, (const half_t*)x0, C0,
                           (const half_t*)x1, C1, HW, groups, nsplit, ppb, stats_ws, gamma, beta, eps, silu, (half_t*)out, cy);
    ICD_CHECK_LAUNCH("icd_groupnorm(apply)");
    return ICD_OK;
}

extern "C" int icd_groupnorm(const void* x0, int32_t C0, const void* x1, int32_t C1, int32_t B, int32_t HW,
                             int32_t groups, const float* gamma, const float* beta, float eps, int32_t silu,
                             void* out, float* stats_ws, void* stream) {
    return groupnorm_run(x0, C0, nullptr, x1, C1, nullptr, B, HW, groups, gamma, beta, eps, silu, out, nullptr, 0, stats_ws, stream,
                         "icd_groupnorm");
}

extern "C" int icd_groupnorm_carry(const void* x0, int32_t C0, const void* carry0, const void* x1, int32_t C1, const void* carry1,
                                   int32_t B, int32_t HW, int32_t groups, const float* gamma, const float* beta, float eps,
                                   int32_t silu, void* out, void* aux, int32_t ld_aux, float* stats_ws, void* stream) {
    return groupnorm_run(x0, C0, carry0, x1, C1, carry1, B, HW, groups, gamma, beta, eps, silu, out, aux, ld_aux, stats_ws, stream,
                         "icd_groupnorm_carry");
}

// lo = fp16(2^-14 * carry): the second K segment of a split-operand GEMM (A = [hi | lo], W = [W | W]) for consumers of a carried
// tensor that no GroupNorm reads first (proj_out of a Transformer2DModel, the downsampler conv)
__global__ __launch_bounds__(256) void carry_expand_kernel(const unsigned char* __restrict__ c, long long n8, half_t* __restrict__ lo) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(c) + i);
        float cf[8];
        gn_carry8(cf, w);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)cf[e];
        *(reinterpret_cast<f16x8*>(lo) + i) = o;
    }
}

// ... and [lo | hi] rows (hi copied): the second source of a conv over [x | lo | x] against per-tap [W_hi | W_hi | W_lo] - split activations
// AND split weights (the phase form of the upsampling conv sums taps, and the sums are not fp16 numbers)
__global__ __launch_bounds__(256) void carry_expand2_kernel(const unsigned char* __restrict__ c, const half_t* __restrict__ hi, long long rows,
                                                            int C8, half_t* __restrict__ out) {
    const long long n8 = rows * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const long long r = i / C8;
        const int k = (int)(i - r * C8);
        const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(c) + i);
        const f16x8 h = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(hi) + i);
        float cf[8];
        gn_carry8(cf, w);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)cf[e];
        f16x8* row = reinterpret_cast<f16x8*>(out) + r * (2 * C8);
        row[k] = o;
        row[C8 + k] = h;
    }
}

extern "C" int icd_carry_expand2(const void* carry, const void* hi, int64_t rows, int32_t C, void* out, void* stream) {
    ICD_CHECK_ARG(carry && hi && out && rows > 0 && C > 0 && C % 8 == 0, "icd_carry_expand2: null pointer or C not a multiple of 8");
    const long long n8 = rows * (C / 8);
    const int blocks = (int)std::min<long long>((n8 + 255) / 256, 4096);
    hipLaunchKernelGGL(carry_expand2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)carry, (const half_t*)hi,
                       (long long)rows, C / 8, (half_t*)out);
    ICD_CHECK_LAUNCH("icd_carry_expand2");
    return ICD_OK;
}

extern "C" int icd_carry_expand(const void* carry, int64_t n, void* lo, void* stream) {
    ICD_CHECK_ARG(carry && lo && n > 0 && n % 8 == 0, "icd_carry_expand: null pointer or element count not a multiple of 8");
    const long long n8 = n / 8;
    const int blocks = (int)std::min<long long>((n8 + 255) / 256, 4096);
    hipLaunchKernelGGL(carry_expand_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)carry, n8, (half_t*)lo);
    ICD_CHECK_LAUNCH("icd_carry_expand");
    return ICD_OK;
}

extern "C" int icd_layernorm(const void* x, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps,
                             void* out, void* stream) {
    ICD_CHECK_ARG(x && out && gamma && beta, "icd_layernorm: null pointer");
    ICD_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 2048, "icd_layernorm: C must be a multiple of 8 and <= 2048 (got %d)", C);
    ICD_CHECK_ARG(rows > 0, "icd_layernorm: empty input");
    hipStream_t st = (hipStream_t)stream;
    const half_t* xh = (const half_t*)x;
    half_t* oh = (half_t*)out;
    const int nchunk = C / 8;
    // widths with 5 chunks per lane group (C = 320 * 2^k) take the multi-row kernel; anything else the generic one
    if (nchunk == 40) launch_ln_multi<8, 5>(xh, rows, C, gamma, beta, eps, oh, nullptr, st);
    else if (nchunk == 80) launch_ln_multi<16, 5>(xh, rows, C, gamma, beta, eps, oh, nullptr, st);
    else if (nchunk == 160) launch_ln_multi<32, 5>(xh, rows, C, gamma, beta, eps, oh, nullptr, st);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, xh, (long long)rows, C, gamma,
                           beta, eps, oh);
    ICD_CHECK_LAUNCH("icd_layernorm");
    return ICD_OK;
}

extern "C" int icd_groupnorm_f32_split(const float* x, int32_t C, int32_t B, int32_t HW, int32_t groups, const float* gamma,
                                       const float* beta, float eps, int32_t silu, void* out_split3, float* stats_ws, void* stream) {
    ICD_CHECK_ARG(x && out_split3 && stats_ws && gamma && beta, "icd_groupnorm_f32_split: null pointer");
    ICD_CHECK_ARG(C > 0 && C % 4 == 0 && C / 4 <= GN_THREADS, "icd_groupnorm_f32_split: C must be a multiple of 4, <= %d", 4 * GN_THREADS);
    ICD_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0 && B > 0 && HW > 0, "icd_groupnorm_f32_split: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int pps = gn_pix_per_split(B, HW);
    const int nsplit = (HW + pps - 1) / pps;
    const int rpi = GN_THREADS / (C / 4);
    const size_t smem = (size_t)rpi * C * 2 * sizeof(float);
    ICD_CHECK_ARG(smem <= 64 * 1024, "icd_groupnorm_f32_split: LDS budget exceeded");
    hipLaunchKernelGGL(gn_stats_f32_kernel, dim3(nsplit, B), dim3(GN_THREADS), smem, st, x, C, HW, groups, pps, stats_ws);
    ICD_CHECK_LAUNCH("icd_groupnorm_f32_split(stats)");
    const int ppb = 64;
    hipLaunchKernelGGL(gn_apply_split_kernel, dim3((HW + ppb - 1) / ppb, B), dim3(GN_THREADS), 0, st, x, C, HW, groups, nsplit, ppb,
                       stats_ws, gamma, beta, eps, silu, (half_t*)out_split3);
    ICD_CHECK_LAUNCH("icd_groupnorm_f32_split(apply)");
    return ICD_OK;
}

// max |x| over n fp32 values -> *out (device, fp32; icd_absmax zeroes it first): non-negative floats order like their bit
// patterns, so a device-wide atomicMax on the bits is exact and order independent.  A NaN input counts as +inf (fmaxf alone
// would drop it), so the host's finiteness guard (ops.split_cast_guarded) fires for NaN activations as well.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        m = (v != v) ? INFINITY : fmaxf(m, fabsf(v));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

extern "C" int icd_absmax(const float* x, int64_t n, float* out, void* stream) {
    ICD_CHECK_ARG(x && out && n > 0, "icd_absmax: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) { icd_set_error("icd_absmax: hipMemsetAsync failed"); return ICD_ERR_HIP; }
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, x, (long long)n, (unsigned*)out);
    ICD_CHECK_LAUNCH("icd_absmax");
    return ICD_OK;
}

extern "C" int icd_split_cast(const float* x, int64_t rows, int32_t C, float scale, void* out_split3, void* stream) {
    ICD_CHECK_ARG(x && out_split3 && rows > 0 && C > 0 && C % 4 == 0, "icd_split_cast: bad arguments (C must be a multiple of 4)");
    const long long n = (long long)rows * (C / 4);
    hipLaunchKernelGGL(split_cast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, C, scale,
                       (half_t*)out_split3);
    ICD_CHECK_LAUNCH("icd_split_cast");
    return ICD_OK;
}

extern "C" int icd_layernorm_stats(const void* x, int64_t rows, int32_t C, float eps, float* stats, void* stream) {
    ICD_CHECK_ARG(x && stats, "icd_layernorm_stats: null pointer");
    ICD_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 2048, "icd_layernorm_stats: C must be a multiple of 8 and <= 2048 (got %d)", C);
    ICD_CHECK_ARG(rows > 0, "icd_layernorm_stats: empty input");
    hipStream_t st = (hipStream_t)stream;
    const half_t* xh = (const half_t*)x;
    const int nchunk = C / 8;
    if (nchunk == 40) launch_ln_multi<8, 5>(xh, rows, C, nullptr, nullptr, eps, nullptr, stats, st);
    else if (nchunk == 80) launch_ln_multi<16, 5>(xh, rows, C, nullptr, nullptr, eps, nullptr, stats, st);
    else if (nchunk == 160) launch_ln_multi<32, 5>(xh, rows, C, nullptr, nullptr, eps, nullptr, stats, st);
    else hipLaunchKernelGGL(layernorm_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, xh, (long long)rows, C, eps, stats);
    ICD_CHECK_LAUNCH("icd_layernorm_stats");
    return ICD_OK;
}

extern "C" int icd_softmax_rows(const float* s, int64_t rows, int32_t cols, int32_t ld_s, float scale, void* p,
                                int32_t ld_p, void* stream) {
    ICD_CHECK_ARG(s && p, "icd_softmax_rows: null pointer");
    ICD_CHECK_ARG(rows > 0 && cols > 0 && ld_s >= cols && ld_p >= cols, "icd_softmax_rows: bad shape");
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = ld_s % 4 == 0 && ld_p % 4 == 0 && ld_p <= ld_s + 3 && scale > 0.f;
    if (vec && ld_p <= 256)
        hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, grid, dim3(256), 0, st, s, (long long)rows, cols, ld_s, scale, (half_t*)p, ld_p);
    else if (vec && ld_p <= 1024)
        hipLaunchKernelGGL(softmax_rows_reg_kernel<4>, grid, dim3(256), 0, st, s, (long long)rows, cols, ld_s, scale, (half_t*)p, ld_p);
    else if (vec && ld_p <= 4096)
        hipLaunchKernelGGL(softmax_rows_reg_kernel<16>, grid, dim3(256), 0, st, s, (long long)rows, cols, ld_s, scale, (half_t*)p, ld_p);
    else
        hipLaunchKernelGGL(softmax_rows_kernel, grid, dim3(256), 0, st, s, (long long)rows, cols, ld_s, scale, (half_t*)p, ld_p);
    ICD_CHECK_LAUNCH("icd_softmax_rows");
    return ICD_OK;
}
