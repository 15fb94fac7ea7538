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

// One staged patch (32 rows x JN*32 columns of one wave) of the common epilogue - fp16 output, optional bias / time-bias /
// fp16 residual / fused LayerNorm correction, no split-K.  Every global operand of the patch's passes is requested BEFORE the
// accumulators are staged through LDS (the generic code waits for each pass's loads in turn - 16 exposed memory latencies per
// wave; measured per block with tools/gemm_timeline.py that epilogue costs 9 - 23 us of a 40 - 60 us tile).  Which operands exist
// is a template parameter chosen by a wave-uniform switch in the caller (the all-true variant serves every other mix, absent
// operands reading the neutral page): in the specialised variants an absent operand costs neither a load nor a register
// (reading a neutral page instead was measured at +7..20 % on plain GEMMs - a 16-B residual read per lane is as much L1 traffic
// as the output store).  Out-of-range lanes read the neutral page instead of branching around their loads.
// RC: the residual comes with an error carry (p.resid_c: one 8-B load per pass beside the 16-B one); OC: the carry of the output is
// stored beside it (p.out_c) - the two variants the executor's carried residual stream needs (start of a chain: OC; an add: R + RC
// + OC).  Both are compile-time so that the default variants carry neither the registers nor the branch.  (Round 3's fp32 twin of the
// stream - R32 / O32 variants, 4 + 6 more bytes per element and add - is served by the general path below since round 4.)
template <int JN, bool R, bool T, bool L, bool PF_ROWBIAS, int PF_PASSES, bool RC = false, bool OC = false>
__device__ __forceinline__ void fast_patch(const GemmK& p, const f32x16& acc0, const f32x16& acc1, float* wst, int l, int mrow0,
                                           int ncol0, const float* ln_lds, int ln_m0) {
    constexpr int LDW = 68;
    constexpr int CHS = JN == 2 ? 3 : 2, NPASS = JN == 2 ? 4 : 2;
    const int lr = l & 31, lh = l >> 5;
    const int c8 = (l & ((1 << CHS) - 1)) * 8, n = ncol0 + c8;
    const bool ncol_ok = n < p.N;
    const half_t* zp = reinterpret_cast<const half_t*>(icd_epi_zero);
    f32x4 b0, b1, s0, s1;
    f16x8 rs[NPASS], rb[NPASS];
    u32x2 rc[RC ? NPASS : 1];
    const unsigned char* zc = reinterpret_cast<const unsigned char*>(icd_epi_zero);      // (a zero byte is a zero carry)
    f32x2 st[NPASS];
    {
        const float* bp = (p.bias && ncol_ok) ? p.bias + n : icd_epi_zero;       // 2 x 16 B per patch: not worth a variant
        b0 = *reinterpret_cast<const f32x4*>(bp); b1 = *reinterpret_cast<const f32x4*>(bp + 4);
    }
    if (L) {
        const float* sp = (p.ln_stats && ncol_ok) ? p.ln_s + n : icd_epi_zero;
        s0 = *reinterpret_cast<const f32x4*>(sp); s1 = *reinterpret_cast<const f32x4*>(sp + 4);
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass >= PF_PASSES) break;
        const int m = mrow0 + ((pass * 64 + l) >> CHS);
        const bool okp = m < p.M && ncol_ok;
        if (R) rs[pass] = *reinterpret_cast<const f16x8*>((p.resid && okp) ? p.resid + (long long)m * p.ldr + n : zp);
        if (RC) rc[pass] = *reinterpret_cast<const u32x2*>((p.resid_c && okp) ? p.resid_c + (long long)m * p.ldr + n : zc);
        if (T && PF_ROWBIAS) rb[pass] = *reinterpret_cast<const f16x8*>((p.rowbias && okp) ? p.rowbias + (long long)(m / p.rps) * p.ld_rowbias + n : zp);
        if (L) {                                 // row statistics: from the kernel's own LDS table when it computed them (ln_lds)
            if (ln_lds) st[pass] = *reinterpret_cast<const f32x2*>(ln_lds + 2 * (m - ln_m0));
            else st[pass] = *reinterpret_cast<const f32x2*>((p.ln_stats && okp) ? p.ln_stats + 2 * (long long)m : icd_epi_ln_id);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(wst + lr * LDW + 8 * g + 4 * lh) = (f32x4){acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
        if (JN == 2)
            *reinterpret_cast<f32x4*>(wst + lr * LDW + 32 + 8 * g + 4 * lh) = (f32x4){acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
    }
    half_t* outp = reinterpret_cast<half_t*>(p.out);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int r = (pass * 64 + l) >> CHS;
        const int m = mrow0 + r;
        const float* sp = wst + r * LDW + c8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
        const bool okl = m < p.M && ncol_ok;
        if (T && !PF_ROWBIAS) rb[pass] = *reinterpret_cast<const f16x8*>((p.rowbias && okl) ? p.rowbias + (long long)(m / p.rps) * p.ld_rowbias + n : zp);
        if (pass >= PF_PASSES) {
            if (R) rs[pass] = *reinterpret_cast<const f16x8*>((p.resid && okl) ? p.resid + (long long)m * p.ldr + n : zp);
            if (RC) rc[pass] = *reinterpret_cast<const u32x2*>((p.resid_c && okl) ? p.resid_c + (long long)m * p.ldr + n : zc);
            if (L) {
                if (ln_lds) st[pass] = *reinterpret_cast<const f32x2*>(ln_lds + 2 * (m - ln_m0));
                else st[pass] = *reinterpret_cast<const f32x2*>((p.ln_stats && okl) ? p.ln_stats + 2 * (long long)m : icd_epi_ln_id);
            }
        }
        f16x8 o;
        float a[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a0 = v0[e] * p.alpha, a1 = v1[e] * p.alpha;
            if (L) { a0 = st[pass][1] * (a0 - st[pass][0] * s0[e]); a1 = st[pass][1] * (a1 - st[pass][0] * s1[e]); }
            a0 += b0[e]; a1 += b1[e];
            if (T) { a0 += (float)rb[pass][e]; a1 += (float)rb[pass][4 + e]; }
            if (R) { a0 += (float)rs[pass][e]; a1 += (float)rs[pass][4 + e]; }
            a[e] = a0; a[4 + e] = a1;
        }
        if (RC) carry_add8(a, rc[pass]);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)a[e];
        const long long orow = out_row(p, m) * p.ldo + n;
        if (okl) *reinterpret_cast<f16x8*>(outp + orow) = o;
        if (OC && okl) *reinterpret_cast<u32x2*>(p.out_c + orow) = carry_of8(a, o);
    }
}

// All patches of one wave through fast_patch (one operand mix per instantiation).
template <int TM, int TN, bool R, bool T, bool L, bool RC = false, bool OC = false>
__device__ __forceinline__ void wave_epilogue_fast(const GemmK& p, f32x16 (&acc)[TM][TN], float* wst, int wm, int wn, int l, int m0,
                                                   int n0, const float* ln_lds) {
    constexpr bool PF_ROWBIAS = TM * TN <= 8;                    // the 160-accumulator tiles have no registers left for it,
    constexpr int PF_PASSES = TM * TN <= 8 ? 4 : 2;              // and request only the first two passes early
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mrow0 = m0 + (wm * TM + i) * 32;
#pragma unroll
        for (int j0 = 0; j0 < TN; j0 += 2) {
            const int ncol0 = n0 + (wn * TN + j0) * 32;
            if (TN - j0 >= 2) fast_patch<2, R, T, L, PF_ROWBIAS, PF_PASSES, RC, OC>(p, acc[i][j0], acc[i][j0 + 1 < TN ? j0 + 1 : j0], wst, l, mrow0, ncol0, ln_lds, m0);
            else fast_patch<1, R, T, L, PF_ROWBIAS, PF_PASSES, RC, OC>(p, acc[i][j0], acc[i][j0], wst, l, mrow0, ncol0, ln_lds, m0);
        }
    }
}

// acc[i][j]: 32x32 MFMA tile (i = 32-row group of the wave's rows, j = 32-column group), wave (wm, wn) of a WM x WN grid
// whose wave tile is (TM*32) x (TN*32); m0 / n0: origin of the block tile.  All waves of the block must call it.
// CARRY: the instantiation that serves launches with an error carry (p.resid_c / p.out_c, ksplit == 1); the default instantiation
// carries no trace of it (as a run-time branch in ONE kernel the two extra variants cost the plain launches 1 - 3 %: registers,
// code size).
template <int TM, int TN, bool FAST_OK = true, bool CARRY = false>
__device__ __forceinline__ void wave_epilogue(const GemmK& p, f32x16 (&acc)[TM][TN], unsigned char* smem, int wv, int wm, int wn,
                                              int l, int m0, int n0, int split, unsigned long long* tl, const float* ln_lds = nullptr) {
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
    if constexpr (FAST_OK && CARRY)
    if (!trans && !geglu && !part && !out_f32 && !p.out32 && p.out_c && !p.ln_stats && !(p.flags & ICD_GEMM_RESID_F32)) {
        // the executor's carried residual stream: h <- h + f with the rounding error of the sum kept beside it, or the start of a chain;
        // round 5: conv1 of a ResnetBlock2D (+ time bias) hands its output to GroupNorm 2 with a carry too
        if (p.resid && p.resid_c && !p.rowbias) { wave_epilogue_fast<TM, TN, true, false, false, true, true>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); return; }
        if (!p.resid && !p.rowbias) { wave_epilogue_fast<TM, TN, false, false, false, false, true>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); return; }
        if (!p.resid && p.rowbias) { wave_epilogue_fast<TM, TN, false, true, false, false, true>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); return; }
    }
    if constexpr (!CARRY)
    if (FAST_OK && !trans && !geglu && !part && !out_f32 && !p.out32 && !(p.flags & ICD_GEMM_RESID_F32)) {
        // fast path of the common epilogue, specialised by which operands exist (wave-uniform switch around the whole wave tile)
        switch ((p.resid ? 1 : 0) | (p.rowbias ? 2 : 0) | (p.ln_stats ? 4 : 0)) {
            case 0: wave_epilogue_fast<TM, TN, false, false, false>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); break;   // plain / bias only
            case 1: wave_epilogue_fast<TM, TN, true, false, false>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); break;    // + residual
            case 2: wave_epilogue_fast<TM, TN, false, true, false>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); break;    // + time bias
            case 4: wave_epilogue_fast<TM, TN, false, false, true>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); break;    // LayerNorm-folded
            default: wave_epilogue_fast<TM, TN, true, true, true>(p, acc, wst, wm, wn, l, m0, n0, ln_lds); break;     // other mixes
        }
        return;
    }
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
            const int ncol0 = n0 + (wn * TN + j0) * 32;
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
                    if (ln_lds) g_st[pass] = *reinterpret_cast<const f32x2*>(ln_lds + 2 * (m - m0));
                    else g_st[pass] = *reinterpret_cast<const f32x2*>(lp);
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
                    // rstd (x alpha - mu s) + b as x (rstd alpha) + (s (-rstd mu) + b): two packed FMAs per pair of values, and the GELU
                    // on pairs (gemm_common.h) - this epilogue is VALU-bound (64 outputs per lane and tile, 5 us of a 17.6 us tile at K = 320)
                    const float ra = g_st[pass][1] * p.alpha, rm = -g_st[pass][1] * g_st[pass][0];
                    const f32x2 ra2 = {ra, ra}, rm2 = {rm, rm};
                    auto lin = [&](const f32x4& x, const f32x4& s_, const f32x4& b_, const int e) {
                        return __builtin_elementwise_fma((f32x2){x[e], x[e + 1]}, ra2,
                                                         __builtin_elementwise_fma((f32x2){s_[e], s_[e + 1]}, rm2, (f32x2){b_[e], b_[e + 1]}));
                    };
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 oa = lin(h0, g_sh0, g_bh0, e) * gelu_fast2(lin(g0, g_sg0, g_bg0, e));
                        const f32x2 ob = lin(h1, g_sh1, g_bh1, e) * gelu_fast2(lin(g1, g_sg1, g_bg1, e));
                        o[e] = (half_t)oa[0]; o[e + 1] = (half_t)oa[1];
                        o[4 + e] = (half_t)ob[0]; o[4 + e + 1] = (half_t)ob[1];
                    }
                    if (m < p.M && ncol0 + g_oc < p.N) *reinterpret_cast<f16x8*>(out + (long long)m * p.ldo + (ncol0 >> 1) + g_oc) = o;
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
                        if constexpr (CARRY) { if (p.resid_c) carry_add8(v, p.resid_c + (long long)m * p.ldr + n); }
                        }
                    }
                    if (p.out32) {
                        float* o32 = p.out32 + out_row(p, m) * p.ldo + n;
                        *reinterpret_cast<f32x4*>(o32) = (f32x4){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(o32 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    }
                    if (out_f32) {                           // (through the row map too: a phase conv with fp32 output, icd_gemm_desc.out_remap_w)
                        float* out = reinterpret_cast<float*>(p.out) + out_row(p, m) * p.ldo + n;
                        *reinterpret_cast<f32x4*>(out) = (f32x4){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(out + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    } else {
                        f16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                        const long long orow = out_row(p, m) * p.ldo + n;
                        *reinterpret_cast<f16x8*>(reinterpret_cast<half_t*>(p.out) + orow) = o;
                        if constexpr (CARRY) { if (p.out_c) *reinterpret_cast<u32x2*>(p.out_c + orow) = carry_of8(v, o); }
                    }
                }
            }
        }
    }
}

}  // namespace icd_gemm_detail
