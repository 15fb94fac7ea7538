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
