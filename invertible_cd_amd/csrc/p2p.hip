// p2p.hip - the prompt-to-prompt cross-attention edit, fused and in place.
//
// Reference: AttentionControlEdit.forward, utils/p2p.py:190-207 with replace_cross_attention of AttentionReplace
// (:227, einsum('hpw,bwn->bhpn', base, mapper)), AttentionRefine (:238-241, gather by mapper * alphas + cur * (1 - alphas))
// and AttentionReweight (:254-258, per-token equalizer chained on a previous controller), followed by the time-dependent
// blend  P[1:] = a_t * edit(P[0], P[1:]) + (1 - a_t) * P[1:].  For every controller the result is, per probability row,
//     new_row[b] = base_row . A_b + D_b (*) cur_row[b]
// with a (n_tokens x n_tokens) matrix A_b and a diagonal D_b that depend only on the step (built on the host by
// invertible_cd_amd/p2p.py from the mapper / alphas / equalizer / cross_replace_alpha tensors).  One kernel applies it to
// the conditional rows of the materialised probabilities, in place, where the reference issues a reshape, an einsum or
// gather, two multiplies, an add and a strided copy per layer.
#include <algorithm>
#include <cstdint>
#include "common.h"

namespace {

// One wave = 32 probability rows.  out^T[n][row] = sum_w A^T[n][w] . base^T[w][row] on the matrix cores:
//   B operand: the base rows themselves, 16 B per lane straight from global (lane = row, k = 16ks + 8lh ..),
//   A operand: fragments of the pre-transposed fp16 operator At[e][n][w] (96 x 80 per edit, zero padded),
//   accumulator: lane = row, 4 consecutive tokens per (n-tile, g) -> 8-byte read-modify-write of the edited prompt's row.
constexpr int P2P_KS = 5;          // 16-token k-steps  (80 token slots)
constexpr int P2P_NT = 3;          // 32-token n-tiles  (96 output slots)

__global__ __launch_bounds__(256) void p2p_cross_edit_kernel(half_t* __restrict__ probs, long long rows, long long group_stride,
                                                              int ld, int nedit, const half_t* __restrict__ At,
                                                              const float* __restrict__ D) {
    const int tid = threadIdx.x, l = tid & 63, lr = l & 31, lh = l >> 5, wv = tid >> 6;
    const long long row = ((long long)blockIdx.x * 4 + wv) * 32 + lr;
    const bool ok = row < rows;
    f16x8 z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) z8[e] = (half_t)0.f;
    f16x8 bf[P2P_KS];
#pragma unroll
    for (int ks = 0; ks < P2P_KS; ++ks)
        bf[ks] = ok ? *reinterpret_cast<const f16x8*>(probs + row * ld + ks * 16 + lh * 8) : z8;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < nedit; ++e) {
        const half_t* Ae = At + (long long)e * (P2P_NT * 32) * (P2P_KS * 16);
        f32x16 acc[P2P_NT];
#pragma unroll
        for (int nt = 0; nt < P2P_NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < P2P_KS; ++ks) {
                const f16x8 af = *reinterpret_cast<const f16x8*>(Ae + (nt * 32 + lr) * (P2P_KS * 16) + ks * 16 + lh * 8);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[ks], ks == 0 ? zero16 : acc[nt], 0, 0, 0);
            }
        if (!ok) continue;
        half_t* cp = probs + (long long)(e + 1) * group_stride + row * ld;
        const float* De = D + (long long)e * (P2P_NT * 32);
#pragma unroll
        for (int nt = 0; nt < P2P_NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * lh;
                if (n + 4 > ld) continue;                           // token slots past the row (pad columns stay zero)
                const f16x4 cur = *reinterpret_cast<const f16x4*>(cp + n);
                const f32x4 d = *reinterpret_cast<const f32x4*>(De + n);
                f16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (half_t)fmaf(d[i], (float)cur[i], acc[nt][4 * g + i]);
                *reinterpret_cast<f16x4*>(cp + n) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LocalBlend (utils/p2p.py:18-44) in one launch.  Block p = prompt p; it rebuilds the base prompt's masks next to its own (the maps
// are a few hundred KB), so no block waits for another:
//   heat_q[pix] = mean over (layer, head) of sum_w maps[l][(q*H_l + h), pix, w] * alpha[q][w]          (fp32)
//   on_q = (maxpool3x3(heat_q) / max(maxpool3x3(heat_q))) > th_pool                                      (pad = -inf, stride 1)
//   mask_p = on_0 | on_p ;  with substruct words: mask_p &= ~(sub_0 | sub_p), sub_q = (heat'_q / max heat'_q) > th_sub, no pooling
//   out[p] = x[0] + float(mask_p, nearest-resized to H x W) * (x[p] - x[0])     (difference in x's dtype, the rest fp32: torch's
//                                                                               promotion of `base + mask.float() * (x_t - base)`)
struct BlendArgs {
    const half_t* maps[8];
    int heads[8];
    int n_layers, P, res, n_words, ld;
    const float* alpha;          // [P][n_words]
    const float* alpha_sub;      // [P][n_words] or null
    float th_pool, th_sub;
    const void* x;               // [P][C][H][W] fp16 or fp32
    int x_f32, C, H, W;
    float* out;                  // [P][C][H][W] fp32
};

__global__ __launch_bounds__(256) void local_blend_kernel(BlendArgs a) {
    __shared__ float heat[1024], pooled[1024], red[256];
    __shared__ unsigned char mask[1024], on_tmp[1024];
    const int p = blockIdx.x, tid = threadIdx.x, res2 = a.res * a.res;
    int total_heads = 0;
    for (int l = 0; l < a.n_layers; ++l) total_heads += a.heads[l];
    for (int i = tid; i < res2; i += 256) mask[i] = 0;
    __syncthreads();
    // pass 0 / 1: main words of prompt 0 / p (pooled, OR-ed into mask); pass 2 / 3: substruct words (cleared from mask)
    const int npass = a.alpha_sub ? 4 : 2;
    unsigned char sub_any = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int q = (pass & 1) ? p : 0;
        const bool sub = pass >= 2;
        if ((pass & 1) && p == 0) continue;                        // the base prompt's own pass is pass 0 / 2
        const float* al = (sub ? a.alpha_sub : a.alpha) + (long long)q * a.n_words;
        for (int pix = tid; pix < res2; pix += 256) {
            float tot = 0.f;
            for (int l = 0; l < a.n_layers; ++l)
                for (int h = 0; h < a.heads[l]; ++h) {
                    const half_t* row = a.maps[l] + (((long long)q * a.heads[l] + h) * res2 + pix) * a.ld;
                    float s = 0.f;
                    for (int w = 0; w < a.n_words; ++w) {
                        const float aw = al[w];
                        if (aw != 0.f) s += (float)row[w] * aw;
                    }
                    tot += s;
                }
            heat[pix] = tot / (float)total_heads;
        }
        __syncthreads();
        float mx = -INFINITY;
        for (int pix = tid; pix < res2; pix += 256) {
            float v = heat[pix];
            if (!sub) {                                             // 3x3 max pool, stride 1, padding 1 (out-of-range taps ignored)
                const int y = pix / a.res, x = pix - y * a.res;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        if (yy >= 0 && yy < a.res && xx >= 0 && xx < a.res) v = fmaxf(v, heat[yy * a.res + xx]);
                    }
            }
            pooled[pix] = v;
            mx = fmaxf(mx, v);
        }
        red[tid] = mx;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
            __syncthreads();
        }
        mx = red[0];
        const float th = sub ? a.th_sub : a.th_pool;
        for (int pix = tid; pix < res2; pix += 256) {
            const bool on = (pooled[pix] / mx) > th;
            if (!sub) mask[pix] |= on ? 1 : 0;
            else on_tmp[pix] = (pass == 2 ? 0 : on_tmp[pix]) | (on ? 1 : 0);
        }
        __syncthreads();
        if (sub) sub_any = 1;
    }
    if (sub_any) {
        for (int pix = tid; pix < res2; pix += 256) mask[pix] = mask[pix] & (on_tmp[pix] ? 0 : 1);
        __syncthreads();
    }
    // blend
    const long long per = (long long)a.C * a.H * a.W;
    const float sy = (float)a.res / (float)a.H, sx = (float)a.res / (float)a.W;
    for (long long i = tid; i < per; i += 256) {
        const int xw = (int)(i % a.W), yh = (int)((i / a.W) % a.H);
        const int my = min((int)floorf(yh * sy), a.res - 1), mxi = min((int)floorf(xw * sx), a.res - 1);
        const float m = mask[my * a.res + mxi] ? 1.f : 0.f;
        float base, diff;
        if (a.x_f32) {
            const float* xf = reinterpret_cast<const float*>(a.x);
            base = xf[i];
            diff = xf[(long long)p * per + i] - base;
        } else {
            const half_t* xh = reinterpret_cast<const half_t*>(a.x);
            const half_t b = xh[i];
            base = (float)b;
            diff = (float)(half_t)((float)xh[(long long)p * per + i] - base);      // the subtraction happens in fp16 in torch
        }
        a.out[(long long)p * per + i] = base + m * diff;
    }
}

// dst[i] += src[i] for up to 32 fp16 tensors in ONE launch (AttentionStore.between_steps, utils/p2p.py:164-170: 32 stored maps)
struct AccumArgs { half_t* dst[32]; const half_t* src[32]; long long n[32]; };
__global__ __launch_bounds__(256) void accumulate_multi_kernel(AccumArgs a) {
    const int t = blockIdx.y;
    half_t* d = a.dst[t]; const half_t* s = a.src[t];
    const long long n = a.n[t], n8 = n >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        f16x8 x = *reinterpret_cast<const f16x8*>(d + i * 8), y = *reinterpret_cast<const f16x8*>(s + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (half_t)((float)x[e] + (float)y[e]);
        *reinterpret_cast<f16x8*>(d + i * 8) = x;
    }
    if (blockIdx.x == 0)
        for (long long i = (n8 << 3) + threadIdx.x; i < n; i += 256) d[i] = (half_t)((float)d[i] + (float)s[i]);
}

}  // namespace

extern "C" int icd_p2p_cross_edit(void* probs, int32_t n_prompts, int32_t heads, int64_t nq, int32_t nk, int32_t ld,
                                  const void* At, const float* D, void* stream) {
    ICD_CHECK_ARG(probs && At && D, "icd_p2p_cross_edit: null pointer");
    ICD_CHECK_ARG(n_prompts >= 2 && heads > 0 && nq > 0, "icd_p2p_cross_edit: need a base prompt and at least one edit");
    ICD_CHECK_ARG(nk > 0 && nk <= 80 && ld >= 80 && ld % 8 == 0,
                  "icd_p2p_cross_edit: tokens <= 80, row stride >= 80 and a multiple of 8 (got %d, %d)", nk, ld);
    const long long rows = (long long)heads * nq;
    hipLaunchKernelGGL(p2p_cross_edit_kernel, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)probs, rows, rows * ld, ld, n_prompts - 1, (const half_t*)At, D);
    ICD_CHECK_LAUNCH("icd_p2p_cross_edit");
    return ICD_OK;
}

extern "C" int icd_local_blend(const void* const* maps, const int32_t* heads, int32_t n_layers, int32_t n_prompts, int32_t res,
                               int32_t n_words, int32_t ld, const float* alpha, const float* alpha_sub, float th_pool, float th_sub,
                               const void* x, int32_t x_is_f32, int32_t C, int32_t H, int32_t W, float* out, void* stream) {
    ICD_CHECK_ARG(maps && heads && alpha && x && out, "icd_local_blend: null pointer");
    ICD_CHECK_ARG(n_layers > 0 && n_layers <= 8 && n_prompts > 0 && res > 0 && res * res <= 1024 && n_words > 0 && ld >= n_words,
                  "icd_local_blend: 1..8 layers, res*res <= 1024, ld >= n_words");
    ICD_CHECK_ARG(C > 0 && H > 0 && W > 0, "icd_local_blend: empty latent");
    BlendArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        ICD_CHECK_ARG(maps[l] && heads[l] > 0, "icd_local_blend: layer %d has no maps", l);
        a.maps[l] = (const half_t*)maps[l]; a.heads[l] = heads[l];
    }
    a.n_layers = n_layers; a.P = n_prompts; a.res = res; a.n_words = n_words; a.ld = ld;
    a.alpha = alpha; a.alpha_sub = alpha_sub; a.th_pool = th_pool; a.th_sub = th_sub;
    a.x = x; a.x_f32 = x_is_f32; a.C = C; a.H = H; a.W = W; a.out = out;
    hipLaunchKernelGGL(local_blend_kernel, dim3(n_prompts), dim3(256), 0, (hipStream_t)stream, a);
    ICD_CHECK_LAUNCH("icd_local_blend");
    return ICD_OK;
}

extern "C" int icd_accumulate_multi(void* const* dst, const void* const* src, const int64_t* counts, int32_t n_tensors, void* stream) {
    ICD_CHECK_ARG(dst && src && counts && n_tensors > 0 && n_tensors <= 32, "icd_accumulate_multi: 1..32 tensors");
    AccumArgs a{};
    long long nmax = 0;
    for (int t = 0; t < n_tensors; ++t) {
        ICD_CHECK_ARG(dst[t] && src[t] && counts[t] >= 0, "icd_accumulate_multi: tensor %d", t);
        ICD_CHECK_ARG(((uintptr_t)dst[t] % 16 == 0) && ((uintptr_t)src[t] % 16 == 0), "icd_accumulate_multi: tensor %d is not 16-byte aligned", t);
        a.dst[t] = (half_t*)dst[t]; a.src[t] = (const half_t*)src[t]; a.n[t] = counts[t];
        nmax = counts[t] > nmax ? counts[t] : nmax;
    }
    const unsigned gx = (unsigned)std::min<long long>(std::max<long long>((nmax / 8 + 255) / 256, 1), 512);
    hipLaunchKernelGGL(accumulate_multi_kernel, dim3(gx, n_tensors), dim3(256), 0, (hipStream_t)stream, a);
    ICD_CHECK_LAUNCH("icd_accumulate_multi");
    return ICD_OK;
}
