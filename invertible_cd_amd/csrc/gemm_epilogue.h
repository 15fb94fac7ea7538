// Epilogue shared by the 256-wide tile kernels (gemm_big.hip, gemm_pp.hip): every wave transposes its own accumulators
// through a private LDS patch and streams them out as 128-B row segments with bias / time-bias / residual / fused
// LayerNorm correction / GEGLU / transposed (V^T) output / split-K partial stores.
#pragma once
#include "gemm_common.h"

namespace icd_gemm_detail {

// Neutral operands for the branch-free fast path below: a missing bias / residual / time-bias / LayerNorm column sum reads
// zeros, a missing LayerNorm row statistic reads (mean 0, rstd 1) - so every load is issued unconditionally and early.
static __device__ __attribute__((aligned(256))) const float icd_epi_zero[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                                0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
static __device__ __attribute__((aligned(16))) const float icd_epi_ln_id[2] = {0.f, 1.f};

// acc[i][j]: 32x32 MFMA tile (i = 32-row group of the wave's rows, j = 32-column group), wave (wm, wn) of a WM x WN grid
// whose wave tile is (TM*32) x (TN*32); m0 / n0: origin of the block tile.  All waves of the block must call it.
template <int TM, int TN, bool FAST_OK = true>
__device__ __forceinline__ void wave_epilogue(const GemmK& p, f32x16 (&acc)[TM][TN], unsigned char* smem, int wv, int wm, int wn,
                                              int l, int m0, int n0, int split, unsigned long long* tl) {
    const int lr = l & 31, lh = l >> 5;
    const int tid = threadIdx.x;
    const bool geglu = p.flags & ICD_GEMM_GEGLU;
    const bool out_f32 = p.flags & ICD_GEMM_OUT_F32;
    constexpr int LDW = 68;                                      // floats per staged row (64 + 4: conflict-free b128 writes)
    constexpr int LDT = 36;                                      // transposed patch: [64 n][32 m + 4]
    const bool trans = p.flags & ICD_GEMM_OUT_TRANS;
    __syncthreads();
    if (tl && tid == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
    float* wst = reinterpret_cast<float*>(smem) + wv * (trans ? 64 * LDT : 32 * LDW);
    float* part = p.ksplit > 1 ? p.partial + (long long)split * p.M * p.N : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow0 = m0 + (wm * TM + i) * 32;
#pragma unroll
        for (int j0 = 0; j0 < TN; j0 += 2) {
            const int jn = (TN - j0) >= 2 ? 2 : 1;               // j-tiles in this group (compile-time after unrolling)
            if (trans) {
                // V^T epilogue: out[(b*N + n)*ldo + key], (b, key) = divmod(m, rows_per_sample).  The patch is staged
                // transposed ([n][m]); lane l then owns column n = l with the 32 consecutive keys of this i-tile
                // (the planner guarantees rows_per_sample % 32 == 0 and M % 32 == 0: a tile never straddles samples).
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    if (jj >= jn) break;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        wst[(jj * 32 + 8 * (e >> 2) + 4 * lh + (e & 3)) * LDT + lr] = acc[i][j0 + jj][e];
                }
                const int n = (wn * TN + j0) * 32 + l;               // column inside the block tile
                if (l < jn * 32 && n0 + n < p.N && mrow0 < p.M) {
                    const int b = mrow0 / p.rps, key0 = mrow0 - b * p.rps;
                    half_t* dst = reinterpret_cast<half_t*>(p.out) + ((long long)b * p.N + n0 + n) * p.ldo + key0;
                    const float* sp = wst + l * LDT;
                    const float sn = p.ln_stats ? p.ln_s[n0 + n] : 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 v0 = *reinterpret_cast<const f32x4*>(sp + 8 * c), v1 = *reinterpret_cast<const f32x4*>(sp + 8 * c + 4);
                        float v[8] = {v0[0] * p.alpha, v0[1] * p.alpha, v0[2] * p.alpha, v0[3] * p.alpha,
                                      v1[0] * p.alpha, v1[1] * p.alpha, v1[2] * p.alpha, v1[3] * p.alpha};
                        if (p.ln_stats) {                    // rows m = mrow0 + 8c + e: per-row (mean, rstd), this lane's column sum
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (long long)(mrow0 + 8 * c + e));
                                v[e] = st[1] * (v[e] - st[0] * sn);
                            }
                        }
                        f16x8 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7]};
                        *reinterpret_cast<f16x8*>(dst + 8 * c) = o;
                    }
                }
                continue;
            }
            // ---- fast path of the common epilogue (fp16 output, optional bias / time-bias / fp16 residual / fused LayerNorm, no
            // split-K): every global operand of the patch's 4 passes is requested BEFORE the accumulators are staged through
            // LDS, branch-free (absent operands read a neutral page).  The generic code below waits for each pass's loads in
            // turn - 16 exposed memory latencies per wave; measured per block (tools/gemm_timeline.py) that epilogue costs
            // 9 - 23 us of a 40 - 60 us tile.
            const int ncol0 = n0 + (wn * TN + j0) * 32;
            const bool fast = FAST_OK && !geglu && !part && !out_f32 && !(p.flags & ICD_GEMM_RESID_F32);
            const int f_chs = jn == 2 ? 3 : 2, f_npass = jn == 2 ? 4 : 2;
            const int f_c8 = (l & ((1 << f_chs) - 1)) * 8, f_n = ncol0 + f_c8;
            constexpr bool PF_ROWBIAS = TM * TN <= 8;            // the 160-accumulator tiles have no registers left for it,
            constexpr int PF_PASSES = TM * TN <= 8 ? 4 : 2;      // and request only the first two passes early
            f32x4 f_b0, f_b1, f_s0, f_s1;
            f16x8 f_rs[4], f_rb[4];
            f32x2 f_st[4];
            // GEGLU patches: the same idea - the four column operands (bias and LayerNorm column sums of the value and of the gate
            // columns) and the two passes' row statistics are requested before the accumulators are staged
            const int g_oc = (l & 3) * 8;
            f32x4 g_bh0, g_bh1, g_bg0, g_bg1, g_sh0, g_sh1, g_sg0, g_sg1;
            f32x2 g_st[2];
            if (geglu) {
                const bool ncol_ok = ncol0 + g_oc < p.N;
                const float* bp = (p.bias && ncol_ok) ? p.bias + ncol0 + g_oc : icd_epi_zero;
                const float* sp = (p.ln_stats && ncol_ok) ? p.ln_s + ncol0 + g_oc : icd_epi_zero;
                const int goff = (p.bias && ncol_ok) ? 32 : 0, soff = (p.ln_stats && ncol_ok) ? 32 : 0;
                g_bh0 = *reinterpret_cast<const f32x4*>(bp); g_bh1 = *reinterpret_cast<const f32x4*>(bp + 4);
                g_bg0 = *reinterpret_cast<const f32x4*>(bp + goff); g_bg1 = *reinterpret_cast<const f32x4*>(bp + goff + 4);
                g_sh0 = *reinterpret_cast<const f32x4*>(sp); g_sh1 = *reinterpret_cast<const f32x4*>(sp + 4);
                g_sg0 = *reinterpret_cast<const f32x4*>(sp + soff); g_sg1 = *reinterpret_cast<const f32x4*>(sp + soff + 4);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int m = mrow0 + ((pass * 64 + l) >> 2);
                    const float* lp = (p.ln_stats && m < p.M && ncol_ok) ? p.ln_stats + 2 * (long long)m : icd_epi_ln_id;
                    g_st[pass] = *reinterpret_cast<const f32x2*>(lp);
                }
            }
            if (fast) {
                const bool ncol_ok = f_n < p.N;
                const float* bp = (p.bias && ncol_ok) ? p.bias + f_n : icd_epi_zero;
                const float* sp = (p.ln_stats && ncol_ok) ? p.ln_s + f_n : icd_epi_zero;
                f_b0 = *reinterpret_cast<const f32x4*>(bp); f_b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                f_s0 = *reinterpret_cast<const f32x4*>(sp); f_s1 = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
                for (int pass = 0; pass < PF_PASSES; ++pass) {
                    if (pass >= f_npass) break;
                    const int m = mrow0 + ((pass * 64 + l) >> f_chs);
                    const bool okp = m < p.M && ncol_ok;
                    const half_t* zp = reinterpret_cast<const half_t*>(icd_epi_zero);
                    const half_t* rp = (p.resid && okp) ? p.resid + (long long)m * p.ldr + f_n : zp;
                    const half_t* tp = (p.rowbias && okp) ? p.rowbias + (long long)(m / p.rps) * p.ld_rowbias + f_n : zp;
                    const float* lp = (p.ln_stats && okp) ? p.ln_stats + 2 * (long long)m : icd_epi_ln_id;
                    f_rs[pass] = *reinterpret_cast<const f16x8*>(rp);
                    if (PF_ROWBIAS) f_rb[pass] = *reinterpret_cast<const f16x8*>(tp);
                    f_st[pass] = *reinterpret_cast<const f32x2*>(lp);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if (jj >= jn) break;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& a = acc[i][j0 + jj];
                    f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                    *reinterpret_cast<f32x4*>(wst + lr * LDW + jj * 32 + 8 * g + 4 * lh) = v;
                }
            }
            if (geglu) {                                         // 64 staged columns = [32 h | 32 gate] -> 32 outputs
                half_t* out = reinterpret_cast<half_t*>(p.out);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {           // 32 rows x 4 chunks of 8 outputs
                    const int r = (pass * 64 + l) >> 2;
                    const int m = mrow0 + r;
                    const float* sp = wst + r * LDW + g_oc;
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sp), h1 = *reinterpret_cast<const f32x4*>(sp + 4);
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(sp + 32), g1 = *reinterpret_cast<const f32x4*>(sp + 36);
                    const float mu = g_st[pass][0], rstd = g_st[pass][1];
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hv0 = rstd * (h0[e] * p.alpha - mu * g_sh0[e]) + g_bh0[e];
                        const float hv1 = rstd * (h1[e] * p.alpha - mu * g_sh1[e]) + g_bh1[e];
                        const float gv0 = rstd * (g0[e] * p.alpha - mu * g_sg0[e]) + g_bg0[e];
                        const float gv1 = rstd * (g1[e] * p.alpha - mu * g_sg1[e]) + g_bg1[e];
                        o[e] = (half_t)(hv0 * gelu_fast(gv0));
                        o[4 + e] = (half_t)(hv1 * gelu_fast(gv1));
                    }
                    if (m < p.M && ncol0 + g_oc < p.N) *reinterpret_cast<f16x8*>(out + (long long)m * p.ldo + (ncol0 >> 1) + g_oc) = o;
                }
            } else if (fast) {
                half_t* outp = reinterpret_cast<half_t*>(p.out);
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    if (pass >= f_npass) break;
                    const int r = (pass * 64 + l) >> f_chs;
                    const int m = mrow0 + r;
                    const float* sp = wst + r * LDW + f_c8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                    const bool okl = m < p.M && f_n < p.N;
                    if (!PF_ROWBIAS) {
                        const half_t* tp = (p.rowbias && okl) ? p.rowbias + (long long)(m / p.rps) * p.ld_rowbias + f_n
                                                              : reinterpret_cast<const half_t*>(icd_epi_zero);
                        f_rb[pass] = *reinterpret_cast<const f16x8*>(tp);
                    }
                    if (pass >= PF_PASSES) {
                        const half_t* rp = (p.resid && okl) ? p.resid + (long long)m * p.ldr + f_n : reinterpret_cast<const half_t*>(icd_epi_zero);
                        const float* lp = (p.ln_stats && okl) ? p.ln_stats + 2 * (long long)m : icd_epi_ln_id;
                        f_rs[pass] = *reinterpret_cast<const f16x8*>(rp);
                        f_st[pass] = *reinterpret_cast<const f32x2*>(lp);
                    }
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = f_st[pass][1] * (v0[e] * p.alpha - f_st[pass][0] * f_s0[e]) + f_b0[e];
                        const float a1 = f_st[pass][1] * (v1[e] * p.alpha - f_st[pass][0] * f_s1[e]) + f_b1[e];
                        o[e] = (half_t)(a0 + (float)f_rb[pass][e] + (float)f_rs[pass][e]);
                        o[4 + e] = (half_t)(a1 + (float)f_rb[pass][4 + e] + (float)f_rs[pass][4 + e]);
                    }
                    if (m < p.M && f_n < p.N) *reinterpret_cast<f16x8*>(outp + (long long)m * p.ldo + f_n) = o;
                }
            } else {
                const int chs = jn == 2 ? 3 : 2;                 // log2(8-wide chunks per staged row)
                const int npass = jn == 2 ? 4 : 2;
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    if (pass >= npass) break;
                    const int item = pass * 64 + l;
                    const int r = item >> chs, c8 = (item & ((1 << chs) - 1)) * 8;
                    const int m = mrow0 + r, n = ncol0 + c8;
                    if (m >= p.M || n >= p.N) continue;
                    const float* sp = wst + r * LDW + c8;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                    if (part) {
                        float* dst = part + (long long)m * p.N + n;
                        *reinterpret_cast<f32x4*>(dst) = v0;
                        *reinterpret_cast<f32x4*>(dst + 4) = v1;
                        continue;
                    }
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
                    if (p.ln_stats) ln_correct8(v, p.ln_stats, p.ln_s, m, n);
                    if (p.bias) {
                        f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                    }
                    if (p.rowbias) {
                        f16x8 rb = *reinterpret_cast<const f16x8*>(p.rowbias + (long long)(m / p.rps) * p.ld_rowbias + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rb[e];
                    }
                    if (p.resid) {
                        if (p.flags & ICD_GEMM_RESID_F32) {
                            const float* rp = reinterpret_cast<const float*>(p.resid) + (long long)m * p.ldr + n;
                            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                        } else {
                        f16x8 rs = *reinterpret_cast<const f16x8*>(p.resid + (long long)m * p.ldr + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rs[e];
                        }
                    }
                    if (out_f32) {
                        float* out = reinterpret_cast<float*>(p.out) + (long long)m * p.ldo + n;
                        *reinterpret_cast<f32x4*>(out) = (f32x4){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(out + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    } else {
                        f16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                        *reinterpret_cast<f16x8*>(reinterpret_cast<half_t*>(p.out) + (long long)m * p.ldo + n) = o;
                    }
                }
            }
        }
    }
}

}  // namespace icd_gemm_detail
