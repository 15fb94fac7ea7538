// gemm.hip - the MFMA workhorse of the iCD U-Net path on gfx950.
//
// One kernel family covers every dense contraction diffusers issues for the UNet:
//   conv3x3 (pad 1, stride 1|2, optional nearest-2x upsample and channel concat folded into the loader) as implicit GEMM,
//   1x1 conv / Linear (proj_in/out, to_q/k/v, to_out, FF, time-embedding MLPs), and the batched attention matmuls of
//   the materialised-P path (utils/p2p.py:335-338: baddbmm -> softmax -> controller -> bmm).
//
// Design (CDNA4):
//   * block tile (64*WM) x 128 x 64, WM*2 waves (WM x 2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_f16 tiles, fp32 accum.
//     WM=4 (256x128, 8 waves, 3 LDS stages = 144 KiB, counted vmcnt: the loads of k-tile t+2 stay in flight across the
//     barrier of k-tile t+1) is the throughput configuration; WM=2 (128x128, 4 waves, 2 stages, 2 blocks/CU) serves
//     small problems.  Small-M / huge-K layers (8x8 and 16x16 feature maps, K up to 23040) are split along K over
//     grid.y with fp32 partial tiles and a fused reduce+epilogue kernel, so they still fill 256 CUs.
//   * Both operands are streamed global -> LDS with global_load_lds (16 B / lane, no VGPR round trip).  The im2col
//     gather is done on the per-lane SOURCE address (tap / upsample / stride / concat / zero padding via a zero page),
//     the LDS image stays lane-linear as the DMA requires.  The tap / channel position is tracked in SGPRs (it is
//     wave-uniform whenever Cin % 64 == 0): the steady-state loader is a 64-bit pointer bump per chunk, the pixel
//     arithmetic runs only when the tap or the concat source changes.
//   * LDS tile rows are 128 B (64 halves); 16-B chunks are XOR-swizzled with ((row>>1)&7) so the ds_read_b128
//     fragment reads of a 16-lane group hit 16 distinct (bank-row half, chunk) slots - conflict free.  The swizzle is
//     applied on the source address (which chunk a lane fetches) and on the read address (same involution).
//   * MFMA operands are swapped (weights as A, activations as B) so each lane ends up with 4 consecutive output
//     channels of one output row; the tile is staged through LDS in fp32 and written with fully coalesced 16-B
//     stores, with bias / time-bias / residual / GEGLU applied once, in fp32, before the single fp16 rounding.
//   * blockIdx -> tile mapping is XCD-aware (block b runs on XCD b%8): every XCD gets a contiguous range of tiles
//     so the n-tiles that share an activation tile hit the same private L2.
#include <algorithm>
#include <string.h>
#include <type_traits>
#include "gemm_common.h"

using namespace icd_gemm_detail;

namespace {

constexpr int BN = 128;

// MODE 0: dense A [M,K] (lda);  MODE 1: conv, Cin % 64 == 0 && C0 % 64 == 0 (tap / concat source are wave-uniform
// per k-tile and tracked in SGPRs);  MODE 2: conv, any Cin % 8 == 0 (per-lane tap tracking; reduced-width test nets).
// ---------------------------------------------------------------------------------------------------------------------
// Cross-attention as the epilogue of the query projection (the kernel BASELINE.json's north star names, in the only
// form that changes its roofline: as a stand-alone kernel it moves Q in + O out for 4*N*77*C flops, arithmetic
// intensity ~77 flop/B).  Here the 256 x 128 tile of  q = LN(h) W_q^T  stays in the accumulators: a wave owns 64 query
// rows x 64 columns = ONE head (d = 64) of two 32-query tiles, and runs  S^T = K q^T -> softmax over the <= 96 key slots
// -> O^T = V^T P^T  on them in place; only O is written.  Q is never stored or re-read and the attention launch disappears.
//   * accumulator -> MFMA B operand without any cross-lane traffic: lane (lr, lh) holds for query lr the head dims
//     16ks + 8(e >> 2) + 4lh + (e & 3), e = 0..7, of k-step ks in registers 8(ks & 1) + e of acc[i][ks >> 1]; the sum over the
//     head dim does not care about the order, so the K fragments are simply loaded in the same permuted order (two 8-B
//     loads per fragment instead of one 16-B load);
//   * K rows enter bit-swapped (bits 2 and 3 of the key index) so the 8 probabilities a lane feeds to one P^T k-slot are
//     8 consecutive keys and every V^T fragment is one 16-B load (same trick as attention.hip);
//   * fused LayerNorm correction, bias, the softmax scale and log2(e) are applied to the accumulators in fp32 before the
//     single fp16 rounding of q;
//   * O goes through a private 4.5 KiB LDS patch per wave and leaves as 128-B row segments.
// Requirements (checked by icd_gemm): head dim 64, N % 128 == 0, rows_per_sample % 256 == 0, keys <= 96.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void xattn_epilogue(const GemmK& p, f32x16 (&acc)[2][2], unsigned char* smem, int wv, int wm, int wn,
                                               int l, int m0, int n0) {
    constexpr int KT = 3, KS = 4, ST = 6, DT = 2;
    const int lr = l & 31, lh = l >> 5;
    const int key_lim = p.x_nk - 8 * lh;
    const int b = m0 / p.rps;                              // the 256 rows of a block lie inside one sample
    const int ncol = n0 + wn * 64;                         // first column of this wave's head
    const half_t* Kb = p.xk + (long long)b * p.x_nk * p.x_ldk + ncol;
    const half_t* Vb = p.xvt + (long long)b * p.x_vt_bs + (long long)ncol * p.x_ldvt;
    f16x8 z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) z8[e] = (half_t)0.f;
    const int prow = (lr & 0x13) | ((lr & 4) << 1) | ((lr & 8) >> 1);
    f16x8 kf[KT][KS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int key = kt * 32 + prow;
            if (key < p.x_nk) {
                const half_t* src = Kb + (long long)key * p.x_ldk + ks * 16 + lh * 4;
                const f16x4 lo = *reinterpret_cast<const f16x4*>(src), hi = *reinterpret_cast<const f16x4*>(src + 8);
                kf[kt][ks] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else kf[kt][ks] = z8;
        }
    f16x8 vf[ST][DT];
#pragma unroll
    for (int st = 0; st < ST; ++st)
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const int row = i * 32 + lr, key = st * 16 + lh * 8;
            vf[st][i] = key < p.x_ldvt ? *reinterpret_cast<const f16x8*>(Vb + (long long)row * p.x_ldvt + key) : z8;
        }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                       // every wave is done with the operand stages: LDS is free
    half_t* patch = reinterpret_cast<half_t*>(smem) + wv * (32 * 72);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mrow = m0 + wm * 64 + i * 32;
        // ---- q = (LN-corrected, biased) accumulators * softmax scale * log2(e), rounded once to fp16 ----
        f32x2 lst = {0.f, 1.f};
        if (p.ln_stats) lst = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (long long)(mrow + lr));
        f16x8 qf[KS];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = ncol + j * 32 + 8 * g + 4 * lh;
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, t4 = {0.f, 0.f, 0.f, 0.f};
                if (p.ln_stats) s4 = *reinterpret_cast<const f32x4*>(p.ln_s + n);
                if (p.bias) t4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][4 * g + e] * p.alpha;
                    v = lst[1] * (v - lst[0] * s4[e]) + t4[e];
                    qf[2 * j + (g >> 1)][4 * (g & 1) + e] = (half_t)(v * p.x_scale_log2);
                }
            }
        // ---- S^T[key][q] over 96 key slots, exact softmax (element e of s[kt] is key 32kt + 16(e>>3) + 8lh + (e&7)) ----
        f32x16 s[KT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][ks], qf[ks], ks == 0 ? zero16 : s[kt], 0, 0, 0);
        float mx = -INFINITY;
        int kl = key_lim;
        asm volatile("" : "+v"(kl));                        // opaque per query tile: no hoisted predicates (see xattn_epilogue_big)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (kt * 32 + 16 * (e >> 3) + (e & 7) >= kl) s[kt][e] = -INFINITY;      // key >= x_nk
                mx = fmaxf(mx, s[kt][e]);
            }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float rs = 0.f;
        f16x8 pf[ST];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f(s[kt][e] - mx);
                rs += pv;
                pf[kt * 2 + (e >> 3)][e & 7] = (half_t)pv;
            }
        f32x16 o[DT];
#pragma unroll
        for (int st = 0; st < ST; ++st)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[st][dt], pf[st], st == 0 ? zero16 : o[dt], 0, 0, 0);
        float l_tot;
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
            l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = 1.0f / l_tot;
        // ---- O tile [32 queries][64 dims] through the wave's LDS patch (row stride 72 halves), out as 128-B rows ----
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 v = {(half_t)(o[dt][4 * g] * inv), (half_t)(o[dt][4 * g + 1] * inv), (half_t)(o[dt][4 * g + 2] * inv),
                           (half_t)(o[dt][4 * g + 3] * inv)};
                *reinterpret_cast<f16x4*>(patch + lr * 72 + dt * 32 + 8 * g + 4 * lh) = v;
            }
        half_t* out = reinterpret_cast<half_t*>(p.out);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + (l >> 3), c8 = (l & 7) * 8;
            const f16x8 v = *reinterpret_cast<const f16x8*>(patch + r * 72 + c8);
            *reinterpret_cast<f16x8*>(out + (long long)(mrow + r) * p.ldo + ncol + c8) = v;
        }
    }
}

template <int MODE, bool TRANS, int WM, int NSTAGE, bool XATTN = false>
__global__ __launch_bounds__(WM * 128, 2) void gemm_kernel(GemmK p) {
    constexpr int NT = WM * 128;                 // threads
    constexpr int BM = WM * 64;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int NWJ = 1024 / NT;               // W chunks per thread per stage (A chunks per thread: always 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int wm = wv >> 1, wn = wv & 1;
    unsigned long long* tl = p.timeline ? p.timeline + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 : nullptr;
    if (tl && tid == 0) { tl[0] = __builtin_amdgcn_s_memrealtime(); tl[1] = tl[0]; }

    // ---- XCD-aware tile id ------------------------------------------------------------------------------
    int mt, nt;
    tile_of_block(blockIdx.x, p.nbm, p.nbn, p.gm, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.y;
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.kt_per_split;
    const int kt_end = min(nk_total, kt_begin + p.kt_per_split);
    const int nk = kt_end - kt_begin;            // >= 1 by construction

    const int z = blockIdx.z;
    const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
    const half_t* A0 = p.a0 + z0 * p.a_bs0 + z1 * p.a_bs1;
    const half_t* Wp = p.w + z0 * p.w_bs0 + z1 * p.w_bs1;
    const half_t* zero = reinterpret_cast<const half_t*>(icd_zero_page);

    // ---- loader state -----------------------------------------------------------------------------------
    // Chunk j of this lane always lands at LDS slot (wave base + j*1024 + lane*16).  The global_load_lds immediate offset
    // (j*1024) is added to BOTH the LDS and the global address, so pointers are kept biased by -j*512 halves and one M0
    // value serves all chunks of an operand.  A chunk that is out of range parks on the zero page with increment 0:
    // the steady-state loader is one 64-bit add per chunk, no selects.
    const int lrow = l >> 3, pchunk = l & 7;
    const int Cin = p.C0 + p.C1;
    const int ntaps = (int)(p.tapmap >> 60), pad = (p.flags & ICD_GEMM_PAD_HI) ? 0 : p.ksize >> 1;      // taps iterated (a 2 x 2 subset in the phase form)
    const int Hu = p.Hin << p.upsample, Wu = p.Win << p.upsample;
    const bool ktail = (p.K & 63) != 0;

    const half_t* a_ptr[4];      // biased source pointer of chunk j
    int a_inc[4];                // halves to advance per k-tile (BK, or 0 when parked on the zero page)
    int a_lc[4];                 // logical chunk index within the k-tile (0..7)
    int a_pix[4], a_y[4], a_x[4];   // conv: b*Hin*Win, output y, x
    bool a_ok[4];                // m < M
    int a_c[4], a_tap[4];        // MODE 2: per-lane channel / tap
    const half_t* w_ptr[NWJ]; int w_inc[NWJ]; int w_lc[NWJ]; bool w_ok[NWJ];
    const int k_begin = kt_begin * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wv * 32 + j * 8 + lrow;
        const int lc = pchunk ^ ((r >> 1) & 7);
        const int m = m0 + r;
        a_lc[j] = lc;
        a_ok[j] = m < p.M;
        a_ptr[j] = zero - j * 512; a_inc[j] = 0;
        a_pix[j] = a_y[j] = a_x[j] = 0; a_c[j] = a_tap[j] = 0;
        if (MODE == 0) {
            if (a_ok[j]) { a_ptr[j] = A0 + (long long)m * p.lda + k_begin + lc * 8 - j * 512; a_inc[j] = BK; }
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            a_y[j] = rem / p.Wout;
            a_x[j] = rem - a_y[j] * p.Wout;
            a_pix[j] = b * p.Hin * p.Win;
            const int k = k_begin + lc * 8;
            a_tap[j] = k / Cin;
            a_c[j] = k - a_tap[j] * Cin;
        }
    }
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
        const int r = wv * (8 * NWJ) + j * 8 + lrow;
        const int lc = pchunk ^ ((r >> 1) & 7);
        const int n = n0 + r;
        w_lc[j] = lc;
        w_ok[j] = n < p.Nw;
        w_ptr[j] = w_ok[j] ? Wp + (long long)n * p.ldw + k_begin + lc * 8 - j * 512 : zero - j * 512;
        w_inc[j] = w_ok[j] ? BK : 0;
    }
    // wave-uniform position of the k-tile inside the im2col K axis (MODE 1)
    int u_tap = MODE == 1 ? k_begin / Cin : 0;
    int u_c = MODE == 1 ? k_begin - u_tap * Cin : 0;
    bool u_recompute = true;

    auto conv_src = [&](int j, int tap, int c, bool& ok) -> const half_t* {
        const int t3 = (int)((p.tapmap >> (4 * min(tap, 8))) & 15u);        // iterated tap -> tap of the 3 x 3 geometry
        const int dy = (t3 * 11) >> 5, dx = t3 - dy * 3;                    // tap / 3, tap % 3 for tap < 9
        const int yu = a_y[j] * p.stride + (ntaps != 1 ? dy : 0) - pad;
        const int xu = a_x[j] * p.stride + (ntaps != 1 ? dx : 0) - pad;
        ok = a_ok[j] && tap < ntaps && (unsigned)yu < (unsigned)Hu && (unsigned)xu < (unsigned)Wu;
        const long long pix = a_pix[j] + (yu >> p.upsample) * p.Win + (xu >> p.upsample);
        return (c < p.C0) ? p.a0 + pix * p.C0 + c : p.a1 + pix * p.C1 + (c - p.C0);
    };

    // wave-uniform LDS bases (SGPR): one M0 per operand per stage
    const int wave_a = __builtin_amdgcn_readfirstlane(wv * 4096);
    const int wave_w = __builtin_amdgcn_readfirstlane(A_BYTES + wv * (1024 * NWJ));

#define ICD_GLDS4(PTRS, BASE)                                                                                             \
    do {                                                                                                                   \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTRS)[0],                         \
                                         (__attribute__((address_space(3))) void*)(BASE), 16, 0, 0);                       \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTRS)[1],                         \
                                         (__attribute__((address_space(3))) void*)(BASE), 16, 1024, 0);                    \
        if (sizeof(PTRS) / sizeof((PTRS)[0]) == 4) {                                                                       \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTRS)[2 % (sizeof(PTRS) / sizeof((PTRS)[0]))], \
                                             (__attribute__((address_space(3))) void*)(BASE), 16, 2048, 0);                \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTRS)[3 % (sizeof(PTRS) / sizeof((PTRS)[0]))], \
                                             (__attribute__((address_space(3))) void*)(BASE), 16, 3072, 0);                \
        }                                                                                                                  \
    } while (0)

    auto issue_stage = [&](int kt, int stage_off) {      // kt: absolute k-tile index; stage_off: byte offset of the stage
        unsigned char* sa = smem + stage_off + wave_a;
        unsigned char* sw = smem + stage_off + wave_w;
        if (MODE == 1) {
            if (u_recompute) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bool ok;
                    const half_t* s0 = conv_src(j, u_tap, u_c, ok) + a_lc[j] * 8;
                    a_ptr[j] = (ok ? s0 : zero) - j * 512;
                    a_inc[j] = ok ? BK : 0;
                }
            }
            u_c += BK;
            u_recompute = false;
            if (u_c == Cin) { u_c = 0; ++u_tap; u_recompute = true; }
            else if (u_c == p.C0) u_recompute = true;
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool ok;
                const half_t* s0 = conv_src(j, a_tap[j], a_c[j], ok);
                a_ptr[j] = (ok ? s0 : zero) - j * 512;
                int cn = a_c[j] + BK, tp = a_tap[j];
                while (cn >= Cin) { cn -= Cin; ++tp; }
                a_c[j] = cn; a_tap[j] = tp;
            }
        }
        if (ktail && kt == nk_total - 1) {          // ragged last k-tile (dense attention bmm): chunk-level masking
            const half_t* ta[4]; const half_t* tw[NWJ];
#pragma unroll
            for (int j = 0; j < 4; ++j) ta[j] = (kt * BK + a_lc[j] * 8 < p.K) ? a_ptr[j] : zero - j * 512;
#pragma unroll
            for (int j = 0; j < NWJ; ++j) tw[j] = (kt * BK + w_lc[j] * 8 < p.K) ? w_ptr[j] : zero - j * 512;
            ICD_GLDS4(ta, sa);
            ICD_GLDS4(tw, sw);
        } else {
            ICD_GLDS4(a_ptr, sa);
            ICD_GLDS4(w_ptr, sw);
        }
        if (MODE != 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a_ptr[j] += a_inc[j];
        }
#pragma unroll
        for (int j = 0; j < NWJ; ++j) w_ptr[j] += w_inc[j];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- LDS read addresses: row bases + swizzled chunk offsets, precomputed (the swizzle key only depends on lane) ----
    const int lr = l & 31, lh = l >> 5;
    int rd_a[4], rd_w[4];
    {
        const int x = (lr >> 1) & 7;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int off = ((s4 * 2 + lh) ^ x) << 4;
            rd_a[s4] = (wm * 64 + lr) * 128 + off;
            rd_w[s4] = A_BYTES + (wn * 64 + lr) * 128 + off;
        }
    }

    // ---- main loop: NSTAGE-deep LDS ring, loads of tile t+NSTAGE-1 issued before the MFMAs of tile t ----------
    constexpr int LOADS = 4 + NWJ;               // global_load_lds per thread per stage
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) issue_stage(kt_begin + s, s * STAGE_BYTES);

    auto k_tile = [&](auto stage_tag, int t) {
        constexpr int S = decltype(stage_tag)::value;
        constexpr int SN = (S + NSTAGE - 1) % NSTAGE;
        // wait until tile t has landed: at most (NSTAGE-2) younger stages may still be in flight
        if (NSTAGE == 2 || t + (NSTAGE - 2) >= nk) __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0f70 | (LOADS * (NSTAGE - 2)));                                 // vmcnt(LOADS)
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < nk) issue_stage(kt_begin + t + NSTAGE - 1, SN * STAGE_BYTES);
        const unsigned char* sb = smem + S * STAGE_BYTES;
        // register double-buffered fragments: the ds_reads of sub-step s+1 are in flight under the MFMAs of sub-step s
        f16x8 af[2][2], wf[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[0][i] = *reinterpret_cast<const f16x8*>(sb + rd_a[0] + i * 4096);
            wf[0][i] = *reinterpret_cast<const f16x8*>(sb + rd_w[0] + i * 4096);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int cur = s4 & 1, nxt = cur ^ 1;
            if (s4 < 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[nxt][i] = *reinterpret_cast<const f16x8*>(sb + rd_a[s4 + 1] + i * 4096);
                    wf[nxt][i] = *reinterpret_cast<const f16x8*>(sb + rd_w[s4 + 1] + i * 4096);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of this sub-step's MFMAs (distinct registers)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (TRANS) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], wf[cur][j], acc[i][j], 0, 0, 0);
                    else       acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
                }
        }
    };
    for (int t = 0; t < nk; t += NSTAGE) {
        k_tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < nk) k_tile(std::integral_constant<int, 1>{}, t + 1);
        if (NSTAGE > 2 && t + 2 < nk) k_tile(std::integral_constant<int, (NSTAGE > 2 ? 2 : 0)>{}, t + 2);
    }
#undef ICD_GLDS4

    if constexpr (XATTN) {
        static_assert(!XATTN || (MODE == 0 && !TRANS), "fused cross-attention: dense tiles (every wave owns 64 rows x one head)");
        if (tl && tid == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
        xattn_epilogue(p, acc, smem, wv, wm, wn, l, m0, n0);
        if (tl) {
            __syncthreads();
            if (tid == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    // ---- epilogue: fp32 staging through LDS, one 64-row slab (one wave row) at a time ----------------------
    if (tl && tid == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
    float* stage = reinterpret_cast<float*>(smem);
    const bool geglu = p.flags & ICD_GEMM_GEGLU;
    const bool out_f32 = p.flags & ICD_GEMM_OUT_F32;
    const long long o_off = z0 * p.o_bs0 + z1 * p.o_bs1;
    for (int h = 0; h < WM; ++h) {
        __syncthreads();
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        if (!TRANS)   // lane: row m = i*32+lr, cols n = wn*64 + j*32 + 8g + 4lh + (0..3)
                            *reinterpret_cast<f32x4*>(stage + (i * 32 + lr) * EPI_LD + wn * 64 + j * 32 + 8 * g + 4 * lh) = v;
                        else          // lane: row n = wn*64 + j*32 + lr, cols m = i*32 + 8g + 4lh + (0..3)
                            *reinterpret_cast<f32x4*>(stage + (wn * 64 + j * 32 + lr) * EPI_LD_T + i * 32 + 8 * g + 4 * lh) = v;
                    }
        }
        __syncthreads();
        if (p.ksplit > 1) {
            // split-K: raw fp32 partial tile, reduced (and bias / residual applied) by splitk_reduce_kernel
            float* part = p.partial + (long long)split * p.M * p.N;
#pragma unroll
            for (int pass = 0; pass < 1024 / NT; ++pass) {
                const int item = pass * NT + tid;
                const int r = item >> 4, c8 = (item & 15) * 8;
                const int m = m0 + h * 64 + r, n = n0 + c8;
                if (m >= p.M || n >= p.N) continue;
                const float* sp = stage + r * EPI_LD + c8;
                float* dst = part + (long long)m * p.N + n;
                *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(sp);
                *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(sp + 4);
            }
        } else if (TRANS) {
            half_t* out = reinterpret_cast<half_t*>(p.out) + o_off;
#pragma unroll
            for (int pass = 0; pass < 1024 / NT; ++pass) {
                const int item = pass * NT + tid;
                const int nl = item >> 3, mc = (item & 7) * 8;
                const int n = n0 + nl, m = m0 + h * 64 + mc;
                if (n >= p.N || m >= p.M) continue;
                const float* sp = stage + nl * EPI_LD_T + mc;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
                if (p.ln_stats) {
                    const float sn = p.ln_s[n];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m + e < p.M) {
                            const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (long long)(m + e));
                            v[e] = st[1] * (v[e] - st[0] * sn);
                        }
                }
                const int b = m / p.rps, key = m - b * p.rps;
                if (key + 8 <= p.rps && m + 8 <= p.M && (key & 7) == 0 && (p.ldo & 7) == 0) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<f16x8*>(out + ((long long)b * p.N + n) * p.ldo + key) = o;
                    if (key + 8 == p.rps)
                        for (int kk = p.rps; kk < p.ldo; ++kk) out[((long long)b * p.N + n) * p.ldo + kk] = (half_t)0.f;
                } else {
                    for (int e = 0; e < 8; ++e) {
                        const int mm = m + e;
                        if (mm >= p.M) break;
                        const int bb = mm / p.rps, kk = mm - bb * p.rps;
                        half_t* row = out + ((long long)bb * p.N + n) * p.ldo;
                        row[kk] = (half_t)v[e];
                        if (kk == p.rps - 1)
                            for (int k2 = p.rps; k2 < p.ldo; ++k2) row[k2] = (half_t)0.f;
                    }
                }
            }
        } else if (geglu) {
            half_t* out = reinterpret_cast<half_t*>(p.out) + o_off;
#pragma unroll
            for (int pass = 0; pass < 512 / NT; ++pass) {
                const int item = pass * NT + tid;
                const int r = item >> 3, oc = (item & 7) * 8;
                const int m = m0 + h * 64 + r;
                const int hcol = (oc >> 5) * 64 + (oc & 31);
                if (m >= p.M || n0 + hcol >= p.N) continue;
                const float* sp = stage + r * EPI_LD + hcol;
                f32x4 h0 = *reinterpret_cast<const f32x4*>(sp), h1 = *reinterpret_cast<const f32x4*>(sp + 4);
                f32x4 g0 = *reinterpret_cast<const f32x4*>(sp + 32), g1 = *reinterpret_cast<const f32x4*>(sp + 36);
                float hv[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                float gv[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                if (p.ln_stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hv[e] *= p.alpha; gv[e] *= p.alpha; }
                    ln_correct8(hv, p.ln_stats, p.ln_s, m, n0 + hcol);
                    ln_correct8(gv, p.ln_stats, p.ln_s, m, n0 + hcol + 32);
                    if (p.bias) {
                        const float* bp = p.bias + n0 + hcol;
#pragma unroll
                        for (int e = 0; e < 8; ++e) { hv[e] += bp[e]; gv[e] += bp[32 + e]; }
                    }
                } else
                if (p.bias) {
                    const float* bp = p.bias + n0 + hcol;
                    f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
                    f32x4 c0 = *reinterpret_cast<const f32x4*>(bp + 32), c1 = *reinterpret_cast<const f32x4*>(bp + 36);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hv[e] = hv[e] * p.alpha + b0[e]; hv[4 + e] = hv[4 + e] * p.alpha + b1[e];
                        gv[e] = gv[e] * p.alpha + c0[e]; gv[4 + e] = gv[4 + e] * p.alpha + c1[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { hv[e] *= p.alpha; gv[e] *= p.alpha; }
                }
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 ge = gelu_fast2((f32x2){gv[e], gv[e + 1]});
                    o[e] = (half_t)(hv[e] * ge[0]); o[e + 1] = (half_t)(hv[e + 1] * ge[1]);
                }
                *reinterpret_cast<f16x8*>(out + (long long)m * p.ldo + (n0 >> 1) + oc) = o;
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < 1024 / NT; ++pass) {
                const int item = pass * NT + tid;
                const int r = item >> 4, c8 = (item & 15) * 8;
                const int m = m0 + h * 64 + r, n = n0 + c8;
                if (m >= p.M || n >= p.N) continue;
                const float* sp = stage + r * EPI_LD + c8;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
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
                    if (p.resid_c) carry_add8(v, p.resid_c + (long long)m * p.ldr + n);
                    }
                }
                if (p.out32) {
                    float* o32 = p.out32 + (long long)m * p.ldo + n;
                    *reinterpret_cast<f32x4*>(o32) = (f32x4){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(o32 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                }
                const long long orow = out_row(p, m) * p.ldo + n;
                if (out_f32) {
                    float* out = reinterpret_cast<float*>(p.out) + o_off + orow;
                    *reinterpret_cast<f32x4*>(out) = (f32x4){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(out + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<f16x8*>(reinterpret_cast<half_t*>(p.out) + o_off + orow) = o;
                    if (p.out_c) *reinterpret_cast<u32x2*>(p.out_c + orow) = carry_of8(v, o);
                }
            }
        }
    }
    if (tl) {
        __syncthreads();
        if (tid == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// split-K second pass: out = alpha * sum_s partial[s] + bias + rowbias + resid   (thread = 8 consecutive columns)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmK p) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nch = p.N >> 3;
    if (idx >= (long long)p.M * nch) return;
    const int m = (int)(idx / nch), n = (int)(idx - (long long)m * nch) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.ksplit; ++s) {
        const float* src = p.partial + ((long long)s * p.M + m) * p.N + n;
        f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
    }
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
        if (p.resid_c) carry_add8(v, p.resid_c + (long long)m * p.ldr + n);
        }
    }
    if (p.out32) {
        float* o32 = p.out32 + (long long)m * p.ldo + n;
        *reinterpret_cast<f32x4*>(o32) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(o32 + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    }
    const long long orow = out_row(p, m) * p.ldo + n;
    if (p.flags & ICD_GEMM_OUT_F32) {
        float* out = reinterpret_cast<float*>(p.out) + orow;
        *reinterpret_cast<f32x4*>(out) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(out + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
        *reinterpret_cast<f16x8*>(reinterpret_cast<half_t*>(p.out) + orow) = o;
        if (p.out_c) *reinterpret_cast<u32x2*>(p.out_c + orow) = carry_of8(v, o);
    }
}

int launch_reduce(const GemmK& k, hipStream_t st) {
    const long long items = (long long)k.M * (k.N >> 3);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, k);
    ICD_CHECK_LAUNCH("icd_gemm(split-K reduce)");
    return ICD_OK;
}

template <int MODE, bool TRANS, int WM, int NSTAGE, bool XATTN = false>
int launch(const GemmK& k, int batch, hipStream_t st) {
    constexpr int smem = NSTAGE * (WM * 64 + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, TRANS, WM, NSTAGE, XATTN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    dim3 grid(k.nbm * k.nbn, k.ksplit, batch);
    hipLaunchKernelGGL((gemm_kernel<MODE, TRANS, WM, NSTAGE, XATTN>), grid, dim3(WM * 128), smem, st, k);
    ICD_CHECK_LAUNCH("icd_gemm");
    if (k.ksplit > 1) return launch_reduce(k, st);
    return ICD_OK;
}

}  // namespace

// Tile / split-K plan: 256x128 tiles when they fill the chip, otherwise 128x128; split K when the grid is still small
// and K is deep.  Pure function of the shape (bit-reproducible results for a given shape).
static void plan_gemm(int M, int N, int K, int batch, bool allow_split, long long ws_bytes, int* wm, int* ksplit) {
    const int nbn = (N + 127) / 128, nk = (K + 63) / 64;
    const long long b256 = (long long)((M + 255) / 256) * nbn * batch;
    const long long b128 = (long long)((M + 127) / 128) * nbn * batch;
    *ksplit = 1;
    if (b256 >= 200 || (M >= 256 && nk >= 32 && allow_split)) *wm = 4;
    else *wm = 2;
    const long long blocks = *wm == 4 ? b256 : b128;
    if (allow_split && blocks < 160 && nk >= 16) {
        int s = (int)((256 + blocks - 1) / blocks);
        s = s > 8 ? 8 : s;
        while (s > 1 && nk / s < 8) --s;
        while (s > 1 && (long long)s * M * N * 4 > ws_bytes) --s;
        *ksplit = s < 1 ? 1 : s;
    }
}

extern "C" int64_t icd_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    int wm, ks;
    plan_gemm(M, N, K, 1, true, 1LL << 60, &wm, &ks);
    int64_t need = ks > 1 ? (int64_t)ks * M * N * 4 : 0;
    const int nk = (K + 63) / 64;
    for (int tn = 4; tn <= 5; ++tn) {                                         // the big-tile plans may split deeper
        if ((tn == 5 && N % 320 != 0) || (tn == 4 && N % 256 != 0) || M < 256 || K % 64 != 0) continue;
        const long long blocks = (long long)((M + 255) / 256) * (N / (64 * tn));
        if (blocks < 192 && nk >= 16) {
            int s = (int)((256 + blocks - 1) / blocks);
            s = s > 8 ? 8 : s;
            while (s > 1 && nk / s < 8) --s;
            if (s > 1) need = std::max<int64_t>(need, (int64_t)s * M * N * 4);
        }
    }
    return need;
}

// One planner for icd_gemm (launch = true) and icd_gemm_plan (launch = false: reports the tile the same call would run on, nothing
// is enqueued).  All tuning / diagnostics inputs travel in the descriptor: the library keeps no process-wide GEMM state.
static int gemm_run(const icd_gemm_desc* d, void* stream, icd_gemm_plan_info* info, bool do_launch) {

    ICD_CHECK_ARG(d != nullptr, "icd_gemm: null descriptor");
    ICD_CHECK_ARG(d->a0 && d->w && d->out, "icd_gemm: a0/w/out must be non-null");
    ICD_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "icd_gemm: M,N,K must be positive (got %d,%d,%d)", d->M, d->N, d->K);
    ICD_CHECK_ARG(d->K % 8 == 0 && d->ldw % 8 == 0, "icd_gemm: K and ldw must be multiples of 8 (K=%d ldw=%d)", d->K, d->ldw);
    ICD_CHECK_ARG(d->N % 8 == 0, "icd_gemm: N must be a multiple of 8 (got %d)", d->N);
    ICD_CHECK_ARG(d->mode == 0 || d->mode == 1, "icd_gemm: mode must be 0 (dense) or 1 (conv)");
    const bool trans = d->flags & ICD_GEMM_OUT_TRANS;
    const bool geglu = d->flags & ICD_GEMM_GEGLU;
    ICD_CHECK_ARG(!(trans && (geglu || (d->flags & ICD_GEMM_OUT_F32) || d->bias || d->resid || d->rowbias)),
                  "icd_gemm: transposed output supports alpha (and the fused LayerNorm) only");
    ICD_CHECK_ARG(!(geglu && (d->resid || d->rowbias || (d->flags & ICD_GEMM_OUT_F32) || d->N % 64 != 0)),
                  "icd_gemm: GEGLU needs N %% 64 == 0 and no resid/rowbias/f32 output");
    if (d->rowbias || trans) ICD_CHECK_ARG(d->rows_per_sample > 0, "icd_gemm: rows_per_sample required");
    GemmK k;
    k.a0 = (const half_t*)d->a0; k.a1 = (const half_t*)d->a1; k.w = (const half_t*)d->w;
    k.bias = d->bias; k.rowbias = (const half_t*)d->rowbias; k.resid = (const half_t*)d->resid; k.out = d->out;
    k.partial = (float*)d->splitk_ws;
    k.M = d->M; k.N = d->N; k.K = d->K; k.Nw = d->Nw > 0 ? d->Nw : d->N;
    k.lda = d->lda; k.ldw = d->ldw; k.ldo = d->ldo; k.ldr = d->ldr; k.ld_rowbias = d->ld_rowbias;
    k.rps = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
    k.C0 = d->C0; k.C1 = d->C1; k.Hin = d->Hin; k.Win = d->Win; k.Hout = d->Hout; k.Wout = d->Wout;
    k.ksize = d->ksize; k.stride = d->stride; k.upsample = d->upsample;
    int ktaps = d->ksize * d->ksize;
    k.tapmap = 0x876543210ULL | ((unsigned long long)ktaps << 60);
    k.orm_pack = 1; k.orm_magic = 0;
    if (d->conv_ktaps || d->out_remap_w) {
        // phase form of the upsampling conv (icd_amd.h): 4 of the 9 taps on the input grid, output rows scattered to one pixel phase
        ICD_CHECK_ARG(d->mode == 1 && d->ksize == 3 && d->stride == 1 && d->upsample == 0 && d->conv_ktaps == 4 && !(d->flags & ICD_GEMM_PAD_HI),
                      "icd_gemm: conv_ktaps = 4 goes with a stride-1 3x3 geometry without upsample");
        ICD_CHECK_ARG(d->conv_tap_base == 0 || d->conv_tap_base == 1 || d->conv_tap_base == 3 || d->conv_tap_base == 4,
                      "icd_gemm: conv_tap_base must be 3 py + px, py / px in {0, 1}");
        ICD_CHECK_ARG(d->out_remap_w == 0 || (d->out_remap_w == d->Wout && !trans && !geglu && !d->resid && !d->out_f32 && d->batch <= 1),
                      "icd_gemm: out_remap_w must equal Wout (plain fp16 / fp32 output, no residual)");
        ktaps = 4;
        { const unsigned long long t0 = (unsigned long long)d->conv_tap_base; k.tapmap = t0 | ((t0 + 1) << 4) | ((t0 + 3) << 8) | ((t0 + 4) << 12) | (4ULL << 60); }
        if (d->out_remap_w) {
            const unsigned W = (unsigned)d->out_remap_w;
            ICD_CHECK_ARG(W >= 2 && W < 8192 && d->out_remap_c >= 0 && d->out_remap_c < 32768, "icd_gemm: out_remap_w must be 2 .. 8191");
            k.orm_pack = 2u | ((2u * W) << 2) | ((unsigned)d->out_remap_c << 17);
            k.orm_magic = (unsigned)((0x100000000ULL + W - 1) / W);             // floor(m / W) = umulhi(m, magic) for m * (magic W - 2^32) < 2^32
            ICD_CHECK_ARG((unsigned long long)d->M * (unsigned long long)((unsigned long long)k.orm_magic * W - 0x100000000ULL) < 0x100000000ULL,
                          "icd_gemm: out_remap_w: M too large for the row map");
        }
    }
    k.zdiv = d->zdiv > 0 ? d->zdiv : 1;
    k.a_bs0 = d->a_bs0; k.a_bs1 = d->a_bs1; k.w_bs0 = d->w_bs0; k.w_bs1 = d->w_bs1; k.o_bs0 = d->o_bs0; k.o_bs1 = d->o_bs1;
    k.alpha = d->alpha; k.flags = d->flags;
    k.timeline = (unsigned long long*)d->debug_timeline;
    k.out32 = d->out_f32;
    ICD_CHECK_ARG(!(d->out_f32 && (trans || geglu || (d->flags & ICD_GEMM_OUT_F32) || d->batch > 1 || d->xattn_k)),
                  "icd_gemm: out_f32 (second fp32 output) goes with a plain fp16 output only");
    k.resid_c = (const unsigned char*)d->resid_carry; k.out_c = (unsigned char*)d->out_carry;
    ICD_CHECK_ARG(!(d->out_carry && (trans || geglu || (d->flags & ICD_GEMM_OUT_F32) || d->batch > 1 || d->xattn_k)),
                  "icd_gemm: out_carry (error carry of the output) goes with a plain fp16 output only");
    ICD_CHECK_ARG(!(d->resid_carry && (!d->resid || (d->flags & ICD_GEMM_RESID_F32) || d->batch > 1)),
                  "icd_gemm: resid_carry needs an fp16 resid (unbatched)");
    ICD_CHECK_ARG(!((d->out_carry && d->ldo % 8 != 0) || (d->resid_carry && d->ldr % 8 != 0)),
                  "icd_gemm: carried tensors need leading dimensions that are multiples of 8");
    const int g_group_m = d->tune_group_m, g_xattn_tile = d->tune_xattn_tile;
    k.gm = g_group_m > 0 ? g_group_m : 1;        // the planner widens it below for launches with many n-tiles
    k.ln_stats = d->ln_stats; k.ln_s = d->ln_colsum;
    k.ln_stats_w = nullptr; k.ln_eps = d->ln_eps > 0.f ? d->ln_eps : 1e-5f;
    // ICD_GEMM_LN_COMPUTE: the big tiles compute the statistics in their main loop; every other path runs the statistics launch
    const bool ln_compute = (d->flags & ICD_GEMM_LN_COMPUTE) != 0;
    ICD_CHECK_ARG(!ln_compute || d->ln_stats, "icd_gemm: ICD_GEMM_LN_COMPUTE needs the ln_stats buffer (and ln_colsum)");
    auto plan_is = [&](int kernel, int tm, int tn, int xattn) {      // report the choice; true: planning only, do not launch
        if (info) {
            info->kernel = kernel; info->tile_m = tm; info->tile_n = tn; info->ksplit = k.ksplit;
            info->ln_inline = k.ln_stats_w != nullptr; info->xattn = xattn;
        }
        return !do_launch;
    };
    auto ln_stats_launch = [&]() -> int {
        if (!ln_compute || !do_launch) return ICD_OK;
        ICD_CHECK_ARG(d->lda == d->K, "icd_gemm: ICD_GEMM_LN_COMPUTE on this path needs contiguous rows (lda == K)");
        return icd_layernorm_stats(d->a0, d->M, d->K, k.ln_eps, const_cast<float*>(d->ln_stats), stream);
    };
    ICD_CHECK_ARG((d->ln_stats == nullptr) == (d->ln_colsum == nullptr), "icd_gemm: ln_stats and ln_colsum go together");
    ICD_CHECK_ARG(!(d->ln_stats && (d->mode != 0 || (d->batch > 1) || (d->flags & ICD_GEMM_OUT_F32))),
                  "icd_gemm: the fused LayerNorm applies to dense, unbatched, fp16-output GEMMs");
    const int batch = d->batch > 0 ? d->batch : 1;
    k.xk = (const half_t*)d->xattn_k; k.xvt = (const half_t*)d->xattn_vt;
    k.x_nk = d->xattn_nk; k.x_ldk = d->xattn_ldk; k.x_ldvt = d->xattn_ldvt; k.x_vt_bs = d->xattn_vt_bs;
    k.x_scale_log2 = d->xattn_scale * 1.4426950408889634f;
    if (d->xattn_k) {
        // query projection + cross-attention in one launch: out = softmax(scale (A W^T + ...) K^T) V per 64-wide head
        ICD_CHECK_ARG(d->xattn_vt && d->xattn_scale > 0.f, "icd_gemm(xattn): xattn_vt and a positive xattn_scale are required");
        ICD_CHECK_ARG(d->mode == 0 && batch == 1 && !(d->flags & (ICD_GEMM_GEGLU | ICD_GEMM_OUT_F32 | ICD_GEMM_OUT_TRANS)) &&
                      !d->resid && !d->rowbias, "icd_gemm(xattn): dense fp16 GEMM without residual / rowbias / GEGLU only");
        ICD_CHECK_ARG(d->N % 128 == 0 && d->K % 64 == 0 && d->lda % 8 == 0 && d->ldo % 8 == 0,
                      "icd_gemm(xattn): N %% 128, K %% 64 and 16-byte aligned leading dims required (head dim 64)");
        ICD_CHECK_ARG(d->rows_per_sample > 0 && d->rows_per_sample % 256 == 0 && d->M % d->rows_per_sample == 0,
                      "icd_gemm(xattn): rows_per_sample must be a multiple of 256 (a 256-row tile stays inside one sample)");
        ICD_CHECK_ARG(d->xattn_nk > 0 && d->xattn_nk <= 96 && d->xattn_ldk % 4 == 0 && d->xattn_ldvt % 8 == 0 && d->xattn_ldvt >= d->xattn_nk,
                      "icd_gemm(xattn): 1..96 keys, ldk %% 4 == 0, ldvt %% 8 == 0, ldvt >= keys");
        k.ksplit = 1; k.kt_per_split = (d->K + BK - 1) / BK;
        // 128 x 128 tiles with two resident blocks per CU (measured, SDXL 32x32 level: 63.8 us at B = 8 / 98.9 us at B = 16 against
        // 73.8 / 113.1 us with 256 x 128 tiles, whose partial last round costs a whole tile; the two launches it replaces take
        // 67.3 / 109.9 us); the 256-row tile stays available for tuning
        // ... and since the 256 x 256 tile of gemm_big.hip hosts it (a block = 256 queries x 4 heads, K / V^T staged once in the
        // idle operand stages): the fast GEMM tile under the same epilogue, LayerNorm statistics from its own main loop
        if ((g_xattn_tile == 0 || g_xattn_tile == 5 || g_xattn_tile == 6) && d->N % 256 == 0 && (long long)(d->M / 256) * (d->N / 256) >= 32) {
            // 256 or 192 query rows per block: the 192-row tile lays ceil(rps / 192) m-tiles over every sample (the last one partial), which
            // turns the 160 blocks of SDXL's 1024-token layers at 8 images per GPU into 240 - one round on 256 CUs either way, of blocks
            // with 3/4 of the work.  Whichever needs less (rounds x rows per block); tune_xattn_tile 5 / 6 force 256 / 192 (A/B)
            const long long nb = d->N / 256, b256 = (long long)(d->M / 256) * nb;
            const int bps192 = (d->rows_per_sample + 191) / 192;
            const long long b192 = (long long)(d->M / d->rows_per_sample) * bps192 * nb;
            const long long t256 = (b256 + 255) / 256 * 256, t192 = (b192 + 255) / 256 * 192;
            const bool t192_wins = g_xattn_tile == 6 || (g_xattn_tile == 0 && t192 < t256);
            k.nbm = t192_wins ? (d->M / d->rows_per_sample) * bps192 : d->M / 256; k.nbn = d->N / 256;
            if (ln_compute && k.nbn <= 12 && !(d->flags & ICD_GEMM_TUNE_NO_LN_INLINE)) k.ln_stats_w = const_cast<float*>(d->ln_stats);
            else { const int rc = ln_stats_launch(); if (rc != ICD_OK) return rc; }
            if (plan_is(1, t192_wins ? 192 : 256, 256, 1)) return ICD_OK;
            return launch_big(k, t192_wins ? 101 : 100, (hipStream_t)stream);
        }
        const bool big = g_xattn_tile == 4;
        k.nbm = d->M / (big ? 256 : 128); k.nbn = d->N / 128;
        if (g_group_m <= 0 && k.nbn >= 16) k.gm = 8;
        { const int rc = ln_stats_launch(); if (rc != ICD_OK) return rc; }
        if (plan_is(0, big ? 256 : 128, 128, 1)) return ICD_OK;
        return big ? launch<0, false, 4, 3, true>(k, 1, (hipStream_t)stream) : launch<0, false, 2, 2, true>(k, 1, (hipStream_t)stream);
    }
    int wm = 2, ks = 1;
    const bool allow_split = !trans && !geglu && batch == 1 && d->splitk_ws != nullptr && d->splitk_ws_bytes > 0;
    const int nk_total = (d->K + BK - 1) / BK;
    hipStream_t st = (hipStream_t)stream;
    // ---- high-intensity tiles (gemm_big.hip), chosen from BIG_TILES by the cost model below ------------------------
    {
        // (the big conv tiles address their sources with 32-bit buffer offsets whose bit 31 marks a zero-padding read: < 2 GiB per source
        // tensor, else the 128-wide kernels)
        // (sample count from the output geometry, as the kernel derives its descriptor size - not from rows_per_sample, which a caller
        // may leave unset; the geometry itself is validated before any tile is chosen)
        if (d->mode == 1) {
            ICD_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "icd_gemm: conv ksize must be 1 or 3");
            ICD_CHECK_ARG(d->stride == 1 || d->stride == 2, "icd_gemm: conv stride must be 1 or 2");
            ICD_CHECK_ARG(d->upsample == 0 || d->upsample == 1, "icd_gemm: upsample must be 0 or 1");
            ICD_CHECK_ARG(d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0,
                          "icd_gemm: bad conv geometry");
        }
        const long long conv_hw = d->mode == 1 ? (long long)d->Hout * d->Wout : 0;
        const long long conv_src_bytes = conv_hw > 0 ? ((d->M + conv_hw - 1) / conv_hw) * d->Hin * d->Win * std::max(d->C0, d->C1) * 2 : 0;
        const bool conv_fast = d->mode == 1 && ((d->C0 + d->C1) % 64 == 0) && (d->C0 % 64 == 0) && conv_src_bytes < (1LL << 31) - (1 << 22);
        // transposed (V^T) outputs take the big tiles when a 32-row tile never straddles two samples and no pad columns
        // have to be zeroed (ldo == rows_per_sample); otherwise the 128-wide kernel's general epilogue handles them
        const bool trans_ok = !trans || (d->rows_per_sample % 32 == 0 && d->M % 32 == 0 && d->ldo == d->rows_per_sample);
        const bool base_ok = batch == 1 && trans_ok && (d->mode == 0 || conv_fast) && d->M >= 256 && (d->Nw <= 0 || d->Nw >= d->N) &&
                             (d->K % 64 == 0) && !(d->flags & ICD_GEMM_TUNE_NO_BIG);
        // Cost model in units of "one k-tile of a 256x256 block" (calibrated with tools/gemm_bench.py, same-box A/B):
        // a block owns its CU, so a launch costs rounds x (k-tiles x tk + fixed), fixed = prologue + exposed epilogue.
        int cfg = -1, s = 1;
        double best = 1e30, best_fill = 0.0, best_ok = 1e30, fill_ok = 0.0;
        int cfg_ok = -1;
        const int forced = ((d->flags >> 24) & 15) - 1;          // ICD_GEMM_TUNE_BIG_CFG(i)
        for (int ci = 0; ci < NUM_BIG_TILES && base_ok; ++ci) {
            const BigTile& c = BIG_TILES[ci];
            if (forced >= 0 && ci != forced) continue;
            // (the 192-row ping-pong tile serves the dense M = 8192-class layers; as a conv tile it measured 7 % behind the 256 x 320 one on
            //  8192 x 1280 x 11520 / 17280 - more operand bytes per flop - and is taken for convs only when forced)
            if (ci == PP192_TILE && d->mode == 1 && forced != ci) continue;
            if ((ci == PP_TILE || ci == PP320_TILE || ci == PP192_TILE) && (!pp_operands_ok(k, d->mode == 1) || (d->flags & ICD_GEMM_TUNE_NO_PP))) continue;
            if (d->N % c.bn != 0 || (geglu && !c.geglu_ok) || ((d->flags & ICD_GEMM_TUNE_BN256) && c.bn != 256)) continue;
            const long long b0 = (long long)((d->M + c.bm - 1) / c.bm) * (d->N / c.bn);
            int smax = 1;
            if (allow_split && b0 < 192 && nk_total >= 16) {
                smax = 8;
                while (smax > 1 && nk_total / smax < 8) --smax;
                while (smax > 1 && (long long)smax * d->M * d->N * 4 > d->splitk_ws_bytes) --smax;
            }
            for (int sx = 1; sx <= smax; ++sx) {
                const long long bt = b0 * sx;
                const long long full = bt / 256, rem = bt % 256;
                // rounds are strict for the first two (all blocks in lockstep), later ones desynchronise
                const double rounds = bt <= 512 ? (double)((bt + 255) / 256) : (double)full + (rem ? 0.3 + 0.7 * rem / 256.0 : 0.0);
                const double act = bt >= 256 ? 256.0 : (double)bt;
                const double w = act <= 160 ? 0.0 : act >= 200 ? 1.0 : (act - 160) / 40.0;
                const double tk = c.tk_part + w * (c.tk_full - c.tk_part);
                const int kps = (nk_total + sx - 1) / sx;
                double cost = rounds * (kps * tk + c.fixed);
                if (sx > 1) cost += (double)(sx + 1) * d->M * d->N * 4.0 / 3.5e12 / 1.5e-6 + 4.0;   // fp32 partials + reduce launch
                // (round 6: the cheapest candidate is picked as before and the launch goes to the 128-wide kernels when it fills its last round
                //  below 45 % - UNLESS an un-split candidate fills well: 2048 x 2560 x 1280 lost its 128 x 320 tile (128 blocks) to the 128 x 128
                //  kernel the day a cheaper-looking 192 x 256 entry (110 blocks) appeared.  Split-K candidates are not rescued this way: 2048 x 1280
                //  x 1280 on 128 x 320 split 2 measured 26.2 us against 17.9 on the 128 x 128 kernel.)
                const double fill = (double)bt / (double)(((bt + 255) / 256) * 256);
                if (fill >= 0.45 && sx == 1 && cost < best_ok) { best_ok = cost; cfg_ok = ci; fill_ok = fill; }
                if (cost < best) { best = cost; cfg = ci; s = sx; best_fill = fill; }
            }
        }
        if (cfg >= 0 && best_fill < 0.45 && forced < 0 && !(d->flags & ICD_GEMM_TUNE_FORCE_BIG) && cfg_ok >= 0) { cfg = cfg_ok; s = 1; best_fill = fill_ok; }
        if (cfg >= 0 && (best_fill >= 0.45 || forced >= 0 || (d->flags & ICD_GEMM_TUNE_FORCE_BIG))) {
            const BigTile& c = BIG_TILES[cfg];
            k.nbm = (d->M + c.bm - 1) / c.bm; k.nbn = d->N / c.bn;
            if (g_group_m <= 0 && k.nbn >= 16) k.gm = 8;        // measured: +5 % on N = 10240 (40 n-tiles), nothing to gain on few n-tiles
            k.kt_per_split = (nk_total + s - 1) / s;
            k.ksplit = (nk_total + k.kt_per_split - 1) / k.kt_per_split;
            if (d->mode == 1) {
                ICD_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "icd_gemm: conv ksize must be 1 or 3");
                ICD_CHECK_ARG(d->K == ktaps * (d->C0 + d->C1), "icd_gemm: K != taps*Cin");
                ICD_CHECK_ARG((d->C1 == 0) == (d->a1 == nullptr), "icd_gemm: a1/C1 mismatch");
            } else {
                k.ksize = 0; k.Hout = 0;
                ICD_CHECK_ARG(d->lda % 8 == 0, "icd_gemm: lda must be a multiple of 8");
            }
            if (ln_compute) {
                // every n-tile recomputes the statistics of its rows (a few % of its main loop): free up to ~10 n-tiles (to_q / to_qk:
                // 65.1 vs 65.7 us at 8192 x 2560 x 1280), dearer than the 8 us statistics launch on the 40 n-tiles of a GEGLU
                // projection (243.9 vs 229.5 + 8)
                const bool inline_ok = d->mode == 0 && k.ksplit == 1 && !trans && !(d->flags & (ICD_GEMM_OUT_F32 | ICD_GEMM_RESID_F32)) &&
                                       !(d->flags & ICD_GEMM_TUNE_NO_LN_INLINE) && k.nbn <= 12 && !d->out_carry && !d->resid_carry;
                if (inline_ok) k.ln_stats_w = const_cast<float*>(d->ln_stats);
                else { const int rc0 = ln_stats_launch(); if (rc0 != ICD_OK) return rc0; }
            }
            if (plan_is(1, c.bm, c.bn, 0)) return ICD_OK;
            const int rc = launch_big(k, cfg, st);
            if (rc != ICD_OK) return rc;
            if (k.ksplit > 1) return launch_reduce(k, st);
            return ICD_OK;
        }
    }
    { const int rc = ln_stats_launch(); if (rc != ICD_OK) return rc; }
    plan_gemm(d->M, d->N, d->K, batch, allow_split, d->splitk_ws_bytes, &wm, &ks);
    if (d->flags & ICD_GEMM_TUNE_WM2) { wm = 2; ks = 1; }          // tuning overrides (tools/gemm_bench.py)
    if (d->flags & ICD_GEMM_TUNE_WM4) { wm = 4; ks = 1; }
    k.ksplit = ks;
    k.kt_per_split = (nk_total + ks - 1) / ks;
    k.ksplit = (nk_total + k.kt_per_split - 1) / k.kt_per_split;      // no empty splits
    k.nbm = (d->M + wm * 64 - 1) / (wm * 64); k.nbn = (d->N + BN - 1) / BN;
    if (g_group_m <= 0 && k.nbn >= 16) k.gm = 8;
    if (d->mode == 1) {
        ICD_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "icd_gemm: conv ksize must be 1 or 3");
        ICD_CHECK_ARG(d->stride == 1 || d->stride == 2, "icd_gemm: conv stride must be 1 or 2");
        ICD_CHECK_ARG(d->upsample == 0 || d->upsample == 1, "icd_gemm: upsample must be 0 or 1");
        ICD_CHECK_ARG(d->C0 > 0 && d->C0 % 8 == 0 && d->C1 % 8 == 0 && d->C1 >= 0, "icd_gemm: conv channels must be multiples of 8");
        ICD_CHECK_ARG((d->C1 == 0) == (d->a1 == nullptr), "icd_gemm: a1/C1 mismatch");
        ICD_CHECK_ARG(d->K == ktaps * (d->C0 + d->C1), "icd_gemm: K != taps*Cin");
        ICD_CHECK_ARG(d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0,
                      "icd_gemm: bad conv geometry");
        ICD_CHECK_ARG(batch == 1 && !trans, "icd_gemm: conv mode is not batched / transposed");
        const bool fast = ((d->C0 + d->C1) % 64 == 0) && (d->C0 % 64 == 0);
        if (!fast) k.nbm = (d->M + 127) / 128;
        if (plan_is(0, fast && wm == 4 ? 256 : 128, 128, 0)) return ICD_OK;
        if (!fast) return launch<2, false, 2, 2>(k, 1, st);
        return wm == 4 ? launch<1, false, 4, 3>(k, 1, st) : launch<1, false, 2, 2>(k, 1, st);
    }
    ICD_CHECK_ARG(d->lda % 8 == 0, "icd_gemm: lda must be a multiple of 8");
    if (plan_is(0, wm * 64, 128, 0)) return ICD_OK;
    if (trans) return wm == 4 ? launch<0, true, 4, 3>(k, batch, st) : launch<0, true, 2, 2>(k, batch, st);
    return wm == 4 ? launch<0, false, 4, 3>(k, batch, st) : launch<0, false, 2, 2>(k, batch, st);
}

extern "C" int icd_gemm(const icd_gemm_desc* d, void* stream) { return gemm_run(d, stream, nullptr, true); }

extern "C" int icd_gemm_plan(const icd_gemm_desc* d, icd_gemm_plan_info* out) {
    ICD_CHECK_ARG(out != nullptr, "icd_gemm_plan: null output");
    memset(out, 0, sizeof(*out));
    return gemm_run(d, nullptr, out, false);
}
