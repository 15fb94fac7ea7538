// attention.hip - fused (flash-style) attention for layers whose controller does not need the probabilities.
//
// Replaces the un-fused baddbmm -> softmax -> bmm the p2p hook forces in the reference (utils/p2p.py:335-338) on every
// layer where the plugin is a data no-op (N > 32^2 for every shipped controller, utils/p2p.py:147,184-188; all layers
// when no controller is registered, e.g. the whole SDXL path utils/generation_sdxl.py:445-453).
//
// CDNA4 design:
//   * one workgroup = 4 waves; each wave owns QT x 32 query rows of one (batch, head) (QT = 2 for the long-sequence
//     head dims 40 / 64: every K / V^T fragment read from LDS feeds two query tiles, and the scheduler gets two
//     independent softmax / MFMA streams to interleave).  KV tiles of 64 keys, double-buffered in LDS via
//     global_load_lds (K tile [64][dpad16], V^T tile [dpad32][64]; V arrives already transposed from the to_v GEMM
//     epilogue, so both MFMA operands are K-contiguous and no transposing LDS read is needed).
//   * S^T = K.Q^T with v_mfma_f32_32x32x16_f16 (K as A operand, Q fragments held in registers as B operand): each lane
//     owns ONE query column and 32 of the tile's 64 keys, so the online-softmax row reductions are in-lane plus one
//     cross-half shuffle.  The softmax scale is folded into the exp2 argument (one FMA + one v_exp_f32 per score);
//     key masking exists only in a peeled instantiation for a ragged last tile.
//   * O^T += V^T.P^T reuses the S^T accumulator registers directly as the B operand: K rows enter the S^T MFMA in a
//     bit-swapped order so the k-slots of a lane are 8 consecutive keys and a V^T fragment is one ds_read_b128.
//     P never leaves registers.
//   * when the padded head dim leaves a free V^T row (d = 40, 80, ...) that row is loaded with ones and the MFMA itself
//     accumulates the softmax denominator (with exactly the fp16-rounded P the numerator uses): no VALU row sums.
//   * XCD-aware block order: the q-tiles of one (b, h) run on one XCD so its K/V stay in that XCD's L2.
#include <type_traits>
#include "common.h"

namespace {

// one fp16 1.0 followed by zeros: the "ones column" of the K tile in head-dim slot d (MODE 1 of attn_fused_kernel)
__device__ __attribute__((aligned(16))) const half_t icd_e0_page[8] = {1, 0, 0, 0, 0, 0, 0, 0};
__device__ const half_t icd_ones_page[64] = {
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

template <int NCH>
__device__ __forceinline__ int k_swz(int row, int chunk) {
    if (NCH % 16 == 0) return chunk ^ (row & 15);
    if (NCH % 8 == 0) return chunk ^ ((row >> 1) & 7);
    if (NCH % 4 == 0) return chunk ^ ((row >> 2) & 3);
    return chunk;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct AttnK {
    const half_t* q; const half_t* k; const half_t* vt; half_t* out;
    int B, H, Nq, Nk, d, ldq, ldk, ldvt, ldo;
    long long vt_bs;              // elements between the V^T blocks of consecutive samples
    float scale_log2;
    int nqt;
    int causal;                   // keys after the query are masked (CLIP text encoder); Nq == Nk
};

// KS = 16-wide k-steps over the head dim (dpad16 = 16*KS); DT = 32-row tiles of the head dim; QT = query tiles per wave
//
// The loop is VALU-issue bound (PMC: VALU pipe 70 % busy with two waves per SIMD, MFMA pipe 25 %), so everything that is
// not exp / scale / convert / max is kept out of the steady state:
//   * K rows are fed to the S^T MFMA in a bit-swapped order (row r <- key r with bits 2 and 3 exchanged), so that the 8
//     accumulator values a lane contributes to one P^T k-slot are 8 CONSECUTIVE keys: every V^T fragment is one
//     ds_read_b128 and P goes from the accumulators to the B operand with no register shuffles;
//   * the global -> LDS source pointers are computed once and bumped per tile; the ragged last tile (key masking,
//     guarded loads) is peeled out of the loop, and the loop is unrolled by two so LDS offsets are immediates;
//   * row sums come from the MFMA pipe (a ones-row of V^T when the padded head dim leaves one, otherwise one extra MFMA
//     per 16 keys with an all-ones A operand);
//   * lazy rescaling: O / l are rescaled only when a row maximum grows by more than 2^8 since the last rescale (P then
//     stays <= 256, exact in fp16's relative precision); the check is one wave-uniform branch.
//   * NST-stage LDS ring (NST = 3 where it fits): tiles are requested two ahead and handed over with a counted vmcnt -
//     one tile of compute (~0.6 us) does not cover the global -> LDS latency (PMC: 35 % of wave cycles in s_waitcnt
//     with a 2-stage ring).
// Tried and dropped (same-box A/B): an intra-wave software pipeline (S^T MFMAs of tile t+1 issued before the softmax of
// tile t, second S accumulator set, 3-stage ring) is not faster, QT = 2 at d = 64 spills - at one query tile per wave
// every ds_read_b128 feeds exactly one MFMA, which is 125 B/clk of LDS traffic per CU at full MFMA rate: the loop is
// LDS-bandwidth bound, and only more MFMAs per fragment (QT = 2 where the registers allow it: d <= 48) lift that.
//
// MODE (round 3): the softmax numerator costs the VALU one FMA (scale, minus the running offset) and one v_exp_f32 per score, and the
// loop is VALU-issue bound.  With q PRE-SCALED by scale * log2(e) (ICD_ATTN_Q_PRESCALED: the executor folds the factor into the
// query projection's weights, so q carries it with no extra rounding) the scores leave the MFMA as base-2 exponents, and the
// running offset M can be subtracted by the MFMA as well:
//   MODE 1 (head dim = 16 KS - 8, e.g. 40): the first padded head-dim slot of the K tile is a column of ones and the same slot of
//           the Q fragment holds -M (fp16; any offset works as long as numerator and denominator use the same one - and they do,
//           the subtraction happens inside the fp32 accumulation), so S^T = K q^T - M for free;
//   MODE 2 (no free slot, e.g. 64 / 80): the S^T accumulators start from -M instead of zero (16 VGPRs per query tile).
// Either way p = exp2(s): the 32 FMAs per lane and tile are gone (-30 % VALU instructions in the steady state).
// MODE 0: scores scaled on the VALU as before (any q; also what a prescaled q takes where no MODE 1 / 2 instantiation exists).
// Measured (same-box round-robin, profiles/r03_attn_variants.txt): 4096 x 4096 d = 40: -5.3 %, d = 64: -2.6 %, 1024 x 1024: -1..2 %.
// The instruction count falls by 30 %, the time by 5 %: the loop is not VALU-throughput bound.  Also built and measured in round 3
// (profiles/r03_attn_pipe.txt): the two query tiles of a wave half a phase apart inside the instruction stream (S^T MFMAs of tile 1
// between the exponentials of tile 0, P.V of tile 0 between the exponentials of tile 1, pinned with sched_group_barrier and asm
// anchors against MachineSink) - bit-identical results, 934 us against 919 us: intra-wave MFMA / VALU overlap is not the limit
// either.  Removed again.  So was a ping-pong form (profiles/r03_attn_pingpong.txt): 8-wave blocks whose two waves per SIMD run
// the per-tile sequence exactly one phase apart (one issues MFMAs only - P.V of tile t, S^T of tile t + 1 - while the other runs
// the softmax VALU), locked by one s_barrier per phase, 4-stage ring, inline-asm fragment reads; bit-identical, and 15 - 40 % SLOWER.
// Its knock-outs say why: softmax phases alone 390 us, MFMA phases alone 450 us, loop / DMA / barrier skeleton 263 us, all three
// together 1058 us - the phases of the two groups do not overlap at all, with or without the barriers: ONE wave issues a 32x32x16
// MFMA every 64 cycles (28 MFMAs per 1800-cycle phase), the 32-cycle rate needs two waves issuing MFMAs on the SIMD at once, which
// is exactly what phase-locking forbids.  Two free-running waves per SIMD (this kernel) are the better schedule on this chip.
// CAUSAL is a template switch (round 3): as a run-time flag its 64 per-element key comparisons kept ~60 scalar registers alive
// through the whole loop, and the spill code (v_readlane / v_writelane around every use) sat on the non-causal hot path too.
//
// DR > 0 (round 3, head dim 40): O^T += V^T P^T on v_mfma_f32_16x16x32_f16 with DR 16-row tiles of the head dim (48 rows instead of
// 64: the 32-row tiles pad d = 40 by 60 %, a quarter of all MFMA cycles of the loop).  P leaves the 32x32 S^T accumulators as
// [32 queries] x [8 consecutive keys] per lane; the 16x16x32 B operand wants [16 queries] x [4 lane groups of 8 keys].  One
// v_permlane16_swap per packed register pair does that: with X = keys 8lh.. and Y = keys 16 + 8lh.. of a 32-key tile,
// swap(X, Y) leaves X' = queries 0-15 in all four 16-lane rows (keys 0-7, 16-23, 8-15, 24-31) and Y' = queries 16-31 likewise;
// the V^T fragment of lane group g is read from key chunk {0, 2, 1, 3}[g], so no data moves but those 16 swaps per tile.
// Measured (same-box round-robin, profiles/r03_attn_pv16.txt): 4096 x 4096 d = 40, B = 32: 877 -> 855 us (-2.6 %; 804 TFLOP/s
// algorithmic), bit-for-bit the same softmax.  14 % fewer MFMA cycles buy 2-3 % because the loop is bound by VALU issue, and the swap
// is not free: tools/valu_rate.hip (profiles/r03_valu_rate.txt) puts v_exp_f32 and v_permlane16_swap at 8 cycles per wave64
// instruction (v_fma / v_max3 / v_cvt_pk: 4), so a tile step carries ~1060 VALU cycles per wave against 768 of MFMA - and only about
// half of VALU time hides under MFMAs even in a register-only interleave of the two (same file).
template <int KS, int DT, int QT, int OCC, int NST, int MODE = 0, bool CAUSAL = false, int DR = 0>
__global__ __launch_bounds__(256, OCC) void attn_fused_kernel(AttnK p) {
    constexpr int NCH = 2 * KS;                       // 16-B chunks per K row
    constexpr bool P16 = DR > 0;
    constexpr int VROWS = P16 ? DR * 16 : DT * 32;    // rows of the V^T tile (head dim, padded)
    constexpr bool ONES = P16 || KS * 16 < DT * 32;   // a free padded V^T row exists: MFMA computes the denominator
    constexpr bool LS_MFMA = !ONES && QT == 1;        // otherwise: one extra MFMA per key step (QT = 1) or VALU sums
    constexpr int KT_BYTES = 64 * NCH * 16;           // K tile
    constexpr int VT_BYTES = VROWS * 128;             // V^T tile, 64 keys = 128 B per row
    constexpr int STAGE = KT_BYTES + VT_BYTES;
    constexpr int NKG = (2 * KS + 3) / 4;             // K-tile load groups per wave (64 chunks each)
    constexpr int NGV = VROWS / 8;                    // V^T load groups (64 chunks of 16 B each) per tile
    constexpr int NVG = (NGV + 3) / 4;                // ... per wave
    static_assert(!P16 || (QT == 2 && MODE == 1), "16-row P.V tiles: built for the wide d = 40 kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, l = tid & 63, lr = l & 31, lh = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int qt = bid % p.nqt, bh = bid / p.nqt;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * (128 * QT) + wv * (32 * QT);
    const half_t* zero = reinterpret_cast<const half_t*>(icd_zero_page);

    const half_t* Kb = p.k + (long long)b * p.Nk * p.ldk + h * p.d;
    const half_t* Vb = p.vt + (long long)b * p.vt_bs + (long long)h * p.d * p.ldvt;

    // Q fragments: lane = query row q0 + 32u + lr, head-dim offset ks*16 + lh*8
    f16x8 qf[QT][KS];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qrow = q0 + u * 32 + lr;
        const half_t* qp = p.q + ((long long)b * p.Nq + qrow) * p.ldq + h * p.d;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dd = ks * 16 + lh * 8;
            if (qrow < p.Nq && dd < p.d) qf[u][ks] = *reinterpret_cast<const f16x8*>(qp + dd);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[u][ks][e] = (half_t)0.f;
        }
    }

    // ---- global -> LDS: per-lane source pointers for full tiles (bumped by one tile per use) -----------------------
    const half_t* kp[NKG]; int kinc[NKG];
    const half_t* vp[NVG]; int vinc[NVG];
#pragma unroll
    for (int j = 0; j < NKG; ++j) {
        const int g = wv + 4 * j;
        const int cid = g * 64 + l;
        const int row = cid / NCH, pc = cid - row * NCH;
        const int dd = k_swz<NCH>(row, pc) * 8;
        const bool ok = dd < p.d;
        kp[j] = ok ? Kb + (long long)row * p.ldk + dd : zero;
        kinc[j] = ok ? 64 * p.ldk : 0;
        if (MODE == 1 && dd == p.d) kp[j] = icd_e0_page;     // ones column in head-dim slot d
    }
#pragma unroll
    for (int j = 0; j < NVG; ++j) {
        const int g = wv + 4 * j;
        const int cid = g * 64 + l;
        const int row = cid >> 3, pc = cid & 7;
        const int lc = pc ^ ((row >> 1) & 7);
        const bool ok = row < p.d;
        vp[j] = ok ? Vb + (long long)row * p.ldvt + lc * 8 : zero;
        vinc[j] = ok ? 64 : 0;
        if (ONES && row == VROWS - 1) { vp[j] = icd_ones_page; vinc[j] = 0; }
    }
    auto issue_fast = [&](int buf_off) {                  // tile is full: no key guards
        unsigned char* sk = smem + buf_off;
        unsigned char* sv = sk + KT_BYTES;
#pragma unroll
        for (int j = 0; j < NKG; ++j) {
            const int g = wv + 4 * j;
            if (g < 2 * KS) { glds16(kp[j], sk + g * 1024); kp[j] += kinc[j]; }
        }
#pragma unroll
        for (int j = 0; j < NVG; ++j) {
            const int g = wv + 4 * j;
            if (g < NGV) { glds16(vp[j], sv + g * 1024); vp[j] += vinc[j]; }
        }
    };
    auto issue_slow = [&](int t, int buf_off) {           // ragged last tile: rows / chunks past Nk park on the zero page
        unsigned char* sk = smem + buf_off;
        unsigned char* sv = sk + KT_BYTES;
        const int kv0 = t * 64;
#pragma unroll
        for (int j = 0; j < NKG; ++j) {
            const int g = wv + 4 * j;
            if (g < 2 * KS) {
                const int cid = g * 64 + l;
                const int row = cid / NCH, pc = cid - row * NCH;
                const int key = kv0 + row, dd = k_swz<NCH>(row, pc) * 8;
                const half_t* src = (key < p.Nk && dd < p.d) ? Kb + (long long)key * p.ldk + dd : zero;
                if (MODE == 1 && dd == p.d && key < p.Nk) src = icd_e0_page;
                glds16(src, sk + g * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < NVG; ++j) {
            const int g = wv + 4 * j;
            if (g < NGV) {
                const int cid = g * 64 + l;
                const int row = cid >> 3, pc = cid & 7;
                const int key = kv0 + (pc ^ ((row >> 1) & 7)) * 8;
                const half_t* src = (row < p.d && key < p.ldvt) ? Vb + (long long)row * p.ldvt + key : zero;
                if (ONES && row == VROWS - 1) src = icd_ones_page;
                glds16(src, sv + g * 1024);
            }
        }
    };

    // ---- LDS read offsets (bytes, buffer 0; buffer / sub-tile offsets are immediates) ------------------------------
    int kbase[KS], vbase[4];
    {
        const int prow = (lr & 0x13) | ((lr & 4) << 1) | ((lr & 8) >> 1);      // bits 2 and 3 exchanged
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kbase[ks] = (prow * NCH + k_swz<NCH>(prow, ks * 2 + lh)) * 16;
        const int x = (lr >> 1) & 7;
#pragma unroll
        for (int st = 0; st < 4; ++st) vbase[st] = KT_BYTES + lr * 128 + (((2 * st + lh) ^ x) << 4);
        if (P16) {
            // 16x16x32 A operand: lane = head-dim row l & 15 (+ 16 per tile: an immediate), lane group g = l >> 4 reads the 8 keys
            // the swapped P operand holds in k-slot g: chunk {0, 2, 1, 3}[g] of the 32-key half kt
            const int l15 = l & 15, g = l >> 4, pg = ((g & 1) << 1) | (g >> 1), x16 = (l15 >> 1) & 7;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) vbase[kt] = KT_BYTES + l15 * 128 + (((4 * kt + pg) ^ x16) << 4);
        }
    }

    constexpr int ODT = P16 ? 1 : DT, OQT = P16 ? 1 : QT, O16Q = P16 ? QT : 1, O16R = P16 ? DR : 1;
    f32x16 o[OQT][ODT];
    f32x4 o16[O16Q][2][O16R];       // [query tile][16-query half][16-row head-dim tile]
#pragma unroll
    for (int u = 0; u < OQT; ++u)
#pragma unroll
        for (int i = 0; i < ODT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[u][i][e] = 0.f;
#pragma unroll
    for (int u = 0; u < O16Q; ++u)
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int r = 0; r < O16R; ++r) o16[u][qh][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // O (and nothing else) shrinks by alpha at a rescale; alpha is per query = per lane of the S^T layout
    auto scale_o = [&](const int u, const float alpha) {
        if constexpr (P16) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
            const float a0 = __uint_as_float(sw[0]), a1 = __uint_as_float(sw[1]);      // queries l & 15 / 16 + (l & 15)
#pragma unroll
            for (int r = 0; r < O16R; ++r) { o16[u][0][r] *= a0; o16[u][1][r] *= a1; }
        } else {
#pragma unroll
            for (int i = 0; i < ODT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[u][i][e] *= alpha;
        }
    };
    // m_run: the row offset (raw score units) currently folded into O and l; l_run only when no ones-row exists
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) { m_run[u] = MODE ? 0.f : -INFINITY; l_run[u] = 0.f; }
    // MODE 2: the S^T accumulators start from -M (all 16 elements of a lane belong to its one query column)
    f32x16 minit[MODE == 2 ? QT : 1];
#pragma unroll
    for (int u = 0; u < (MODE == 2 ? QT : 1); ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) minit[u][e] = 0.f;
    const float c = p.scale_log2;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // folds into the MFMA's inline-constant srcC
    f16x8 ones8;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones8[e] = (half_t)1.f;
    asm volatile("" : "+v"(ones8));

    // BO = byte offset of the tile's LDS stage: a literal in the unrolled loop (folds into the ds_read immediates).
    // Fragment reads are issued in batches and fenced (sched_barrier): left alone, the register allocator reuses one
    // quad for every fragment and serialises ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, i.e. 16 exposed LDS latencies
    // per tile (measured: removing the LDS reads made the old loop 28 % faster, removing the exps did nothing).
    auto qk = [&](f32x16 (&s)[QT][2], const int BO, const bool rag, const int t) {
        // ---- K fragments in batches of 2 x KC, then S^T[key][q] for two 32-key tiles ----
        constexpr int KC = QT > 1 && KS > 3 ? 2 : KS <= 6 ? KS : (KS + 1) / 2;
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += KC) {
            f16x8 kf[2][KC];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int kk = 0; kk < KC; ++kk)
                    if (k0 + kk < KS)
                        kf[kt][kk] = *reinterpret_cast<const f16x8*>(smem + kbase[k0 + kk] + (BO + kt * 32 * NCH * 16));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int u = 0; u < QT; ++u)
                        if (k0 + kk < KS)
                            s[u][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][kk], qf[u][k0 + kk],
                                                                              k0 + kk == 0 ? (MODE == 2 ? minit[MODE == 2 ? u : 0] : zero16) : s[u][kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rag || CAUSAL) {                             // wave-uniform: ragged last tile (keys past Nk) / causal mask -> -inf
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                const int last = CAUSAL ? min(p.Nk - 1, q0 + u * 32 + lr) : p.Nk - 1;    // last key this query may see
                const int lim = last - t * 64 - 8 * lh;  // (constant) > (one per-lane limit): no scalar register per key slot
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kt * 32 + 16 * (e >> 3) + (e & 7) > lim) s[u][kt][e] = -INFINITY;
            }
        }
    };
    auto softmax_pv = [&](f32x16 (&s)[QT][2], const int BO, const int t) {
        // ---- V^T fragments of the first VPRE key steps: in flight while the softmax runs ----
        constexpr int VPRE = P16 ? 0 : QT * DT <= 2 ? 4 : QT * DT <= 4 ? 2 : 1;      // at most 8 fragments (32 VGPRs) ahead
        f16x8 vf[4][ODT];
        f16x8 vf16[2][O16R];                              // P16: all 2 x DR fragments of the tile (24 VGPRs) ahead of the softmax
#pragma unroll
        for (int st = 0; st < VPRE; ++st)
#pragma unroll
            for (int i = 0; i < ODT; ++i)
                vf[st][i] = *reinterpret_cast<const f16x8*>(smem + vbase[st] + (BO + i * 4096));
        if constexpr (P16) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < O16R; ++r)
                    vf16[kt][r] = *reinterpret_cast<const f16x8*>(smem + vbase[kt] + (BO + r * 2048));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- softmax numerators (lane owns query column lr; element e of s[kt] is key 32kt + 16(e>>3) + 8lh + (e&7)) ----
        f16x8 pf[QT][4];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            float mx0 = fmaxf(s[u][0][0], s[u][1][0]), mx1 = fmaxf(s[u][0][1], s[u][1][1]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) {
                mx0 = fmaxf(fmaxf(mx0, s[u][0][e]), s[u][1][e]);
                mx1 = fmaxf(fmaxf(mx1, s[u][0][e + 1]), s[u][1][e + 1]);
            }
            float mx = fmaxf(mx0, mx1);
            {   // other half-wave's maximum with v_permlane32_swap (VALU; a ds_bpermute would drain the LDS queue)
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            if (MODE != 0) {
                // s already is (base-2 exponent - M).  The first tile fixes M at its row maximum; afterwards M moves only when a
                // maximum outgrows it by more than 2^8 (lazy rescale, as MODE 0).  At a move by delta: O and l shrink by 2^-delta, the
                // scores of THIS tile (computed against the old M) drop by delta, and the -M carried by the MFMA is replaced.
                if (__builtin_amdgcn_ballot_w64(mx > 8.0f) || t == 0) {
                    float delta = t == 0 ? fmaxf(mx, -60000.f) : fmaxf(mx, 0.f);
                    float m_new = m_run[u] + delta;
                    if (MODE == 1) {                      // the offset travels as an fp16 operand: keep it fp16-exact
                        m_new = (float)(half_t)m_new;
                        delta = m_new - m_run[u];
                        if (lh == (KS * 16 - 8) % 16 / 8) qf[u][KS - 1][0] = (half_t)(-m_new);
                    } else if (MODE == 2) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) minit[MODE == 2 ? u : 0][e] = -m_new;
                    }
                    m_run[u] = m_new;
                    if (t != 0) {                         // (O and l are still zero on the first tile)
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        if (!ONES) l_run[u] *= alpha;
                        scale_o(u, alpha);
                    }
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) s[u][kt][e] -= delta;
                }
            } else if (__builtin_amdgcn_ballot_w64((mx - m_run[u]) * c > 8.0f)) {       // rare after the first tiles
                const float m_new = fmaxf(m_run[u], mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * c);
                m_run[u] = m_new;
                if (!ONES) l_run[u] *= alpha;
                scale_o(u, alpha);
            }
            const float nmc = -m_run[u] * c;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = MODE != 0 ? __builtin_amdgcn_exp2f(s[u][kt][e]) : __builtin_amdgcn_exp2f(fmaf(s[u][kt][e], c, nmc));
                    if (!ONES && !LS_MFMA) rs += pv;
                    pf[u][kt * 2 + (e >> 3)][e & 7] = (half_t)pv;
                }
            if (!ONES && !LS_MFMA) l_run[u] += rs;      // this lane's 32 keys; the other half-wave is added at the end
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T[dcol][q] += V^T[dcol][keys] . P^T[keys][q]  (each V^T fragment feeds the QT query tiles; the
        //      fragments of key step st+1 are requested before the MFMAs of step st) ----
        if constexpr (P16) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f16x8 pa[QT][2];                          // P^T operands of the two 16-query halves, 32 keys
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    const u32x4 X = __builtin_bit_cast(u32x4, pf[u][2 * kt]), Y = __builtin_bit_cast(u32x4, pf[u][2 * kt + 1]);
                    u32x4 a, b2;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(X[w], Y[w], false, false);
                        a[w] = sw[0]; b2[w] = sw[1];
                    }
                    pa[u][0] = __builtin_bit_cast(f16x8, a); pa[u][1] = __builtin_bit_cast(f16x8, b2);
                }
#pragma unroll
                for (int r = 0; r < O16R; ++r)
#pragma unroll
                    for (int u = 0; u < QT; ++u)
#pragma unroll
                        for (int qh = 0; qh < 2; ++qh)
                            o16[u][qh][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf16[kt][r], pa[u][qh], o16[u][qh][r], 0, 0, 0);
            }
        } else {
        f32x16 ls[QT];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st + 1 >= VPRE && st + 1 < 4) {
#pragma unroll
                for (int i = 0; i < ODT; ++i)
                    vf[st + 1][i] = *reinterpret_cast<const f16x8*>(smem + vbase[st + 1] + (BO + i * 4096));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < ODT; ++i)
#pragma unroll
                for (int u = 0; u < QT; ++u) o[u][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[st][i], pf[u][st], o[u][i], 0, 0, 0);
            if (LS_MFMA) {
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    ls[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones8, pf[u][st], st == 0 ? zero16 : ls[u], 0, 0, 0);
            }
        }
        if (LS_MFMA) {
#pragma unroll
            for (int u = 0; u < QT; ++u) l_run[u] += ls[u][0];          // every row of ones . P^T is the full 64-key sum
        }
        }
    };
    using F = std::false_type;
    using T = std::true_type;

    const int nfull = p.Nk >> 6;
    const bool ragged = (p.Nk & 63) != 0;
    const int nt = nfull + (ragged ? 1 : 0);
    constexpr int PD = NST - 1;                          // prefetch distance in tiles
    constexpr int KREM = (2 * KS) % 4;                   // waves >= KREM issue one K group less per tile
    auto issue = [&](int tt, int buf_off) {
        if (tt < nfull) issue_fast(buf_off);
        else issue_slow(tt, buf_off);
    };
    // tile t has landed (this wave's part) when at most the loads of the `keep` later tiles are outstanding
    constexpr int VREM = NGV % 4;                        // ... and one V^T group less
    auto wait_landed = [&](int keep) {
        const int fewer = (KREM != 0 && wv >= KREM ? 1 : 0) + (VREM != 0 && wv >= VREM ? 1 : 0);      // wave-uniform
        if (keep <= 0) __builtin_amdgcn_s_waitcnt(0x0f70);
        else if (fewer == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | (NKG + NVG >= 2 ? NKG + NVG - 2 : 0));
        else if (fewer == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (NKG + NVG - 1));
        else __builtin_amdgcn_s_waitcnt(0x0f70 | (NKG + NVG));
    };
    static_assert(NKG + NVG < 16 && NST <= 3, "vmcnt immediate / keep count");
    // INVARIANT the counted wait rests on (the bare s_barrier below no longer drains vmcnt): per tile, wave wv issues exactly one
    // LDS-DMA load for every j with wv + 4 j < 2 KS (K groups) and every j with wv + 4 j < NGV (V^T groups) - issue_fast and issue_slow
    // iterate the SAME two predicates and never skip a load inside them (rows / chunks past the tensor read the zero page instead), so
    // the per-wave count is (NKG - [KREM != 0 && wv >= KREM]) + (NVG - [VREM != 0 && wv >= VREM]) for full and ragged tiles alike.
    static_assert(NKG == (2 * KS + 3) / 4 && NVG == (NGV + 3) / 4 && KREM == (2 * KS) % 4 && VREM == NGV % 4,
                  "wait_landed counts the loads of issue_fast / issue_slow: keep the group arithmetic in one place");
    // -DICD_ATTN_DEBUG_SYNC restores the full fence (s_waitcnt vmcnt(0) lgkmcnt(0) + barrier) at every tile: a miscounted ring shows up
    // as a bitwise difference between the two builds (tools/attn_ring_check.py runs the ragged / causal / wide-head cases on both).
    {
        auto step = [&](const bool rag, const int buf, int t) {      // tile t sits in stage `buf`
            wait_landed(nt - 1 - t < PD - 1 ? nt - 1 - t : PD - 1);
            // a bare s_barrier: __syncthreads() is a release fence first (s_waitcnt vmcnt(0) lgkmcnt(0)), which drains the LDS-DMA of
            // the tiles still in flight and turns every ring into a 2-stage one.  The counted wait above is the whole contract:
            // this wave's part of tile t has landed, its fragment reads of tile t - 1 were consumed by MFMAs already.
#ifdef ICD_ATTN_DEBUG_SYNC
            __syncthreads();
#else
            __builtin_amdgcn_s_barrier();                 // tile t visible to all; everybody is done with tile t-1
#endif
            if (t + PD < nt) issue(t + PD, ((buf + PD) % NST) * STAGE);
            f32x16 s[QT][2];
            qk(s, buf * STAGE, rag, t);
            softmax_pv(s, buf * STAGE, t);
        };
#pragma unroll
        for (int i = 0; i < PD; ++i)
            if (i < nt) issue(i, i * STAGE);
        int t = 0;
        for (; t + NST <= nfull; t += NST) {
#pragma unroll
            for (int i = 0; i < NST; ++i) step(false, i, t + i);
        }
        for (int i = 0; t < nfull; ++t, ++i) step(false, i, t);      // < NST leftover full tiles, stages 0, 1, ..
        if (ragged) step(true, nfull % NST, nfull);
    }
    if constexpr (P16) {
        // ---- 16x16 accumulators: lane = query (l & 15) of half qh, head-dim rows 16r + 4(l >> 4) .. + 3; the denominator is row
        //      VROWS - 1 of O^T (the V^T ones-row): element 3 of the last tile in lane group 3 ----
        const int l15 = l & 15, g = l >> 4;
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                const float inv = 1.0f / __shfl(o16[u][qh][O16R - 1][3], 48 + l15);
                const int qrow = q0 + u * 32 + qh * 16 + l15;
                if (qrow < p.Nq) {
                    half_t* op = p.out + ((long long)b * p.Nq + qrow) * p.ldo + h * p.d;
#pragma unroll
                    for (int r = 0; r < O16R; ++r) {
                        const int dc = r * 16 + 4 * g;
                        if (dc < p.d) {
                            const f32x4 v = o16[u][qh][r];
                            *reinterpret_cast<f16x4*>(op + dc) = (f16x4){(half_t)(v[0] * inv), (half_t)(v[1] * inv), (half_t)(v[2] * inv), (half_t)(v[3] * inv)};
                        }
                    }
                }
            }
    } else {
    // ---- normalise and store: lane holds 4 consecutive head-dim columns of query row q0 + 32u + lr ----
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        float l_tot;
        if (ONES) l_tot = __shfl(o[u][ODT - 1][15], lr + 32);   // row DT*32-1 of O^T = sum_k P (the V^T ones-row), upper half-wave
        else if (LS_MFMA) l_tot = l_run[u];
        else {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[u]), __float_as_uint(l_run[u]), false, false);
            l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = 1.0f / l_tot;
        const int qrow = q0 + u * 32 + lr;
        if (qrow < p.Nq) {
            half_t* op = p.out + ((long long)b * p.Nq + qrow) * p.ldo + h * p.d;
#pragma unroll
            for (int i = 0; i < ODT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dc = i * 32 + 8 * g + 4 * lh;
                    if (dc < p.d) {
                        f16x4 v = {(half_t)(o[u][i][4 * g] * inv), (half_t)(o[u][i][4 * g + 1] * inv),
                                   (half_t)(o[u][i][4 * g + 2] * inv), (half_t)(o[u][i][4 * g + 3] * inv)};
                        *reinterpret_cast<f16x4*>(op + dc) = v;
                    }
                }
        }
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention (context of <= 96 tokens: 77 CLIP tokens everywhere in this path).  K and V^T of one (batch, head) are
// 2 x 77 x d halves: every wave loads ALL their MFMA fragments into registers once (straight from global memory in
// fragment layout) and then streams query tiles: Q fragments in, 12 + 12 MFMAs (d = 64), one exact softmax over the 96
// key slots, O out.  No LDS, no barriers, no online rescaling.  The kernel is HBM-bound by construction (Q in + O out,
// arithmetic intensity ~ 77 flop/B): what matters is bytes in flight, so each wave prefetches the
// next tile's Q fragments before it computes the current one.
// ---------------------------------------------------------------------------------------------------------------
// STL (round 5): live 16-key slots of the 96, ceil(keys / 16) - 5 for the 77 CLIP tokens: the dead slot's exponentials (8 of 48 per lane and
// tile), conversions and P.V MFMAs (DT of 6 DT) are dropped at compile time.
template <int KS, int DT, int STL = 6>
__global__ __launch_bounds__(256, 2) void attn_cross_kernel(AttnK p, int tiles_per_wave) {
    constexpr int KT = 3;                             // 32-key tiles (96 key slots)
    constexpr int ST = 6;                             // 16-key steps of the PV product
    const int tid = threadIdx.x, l = tid & 63, lr = l & 31, lh = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = (p.Nq + 32 * tiles_per_wave * 4 - 1) / (32 * tiles_per_wave * 4);      // blocks per (b, h)
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int qc = bid % nchunk, bh = bid / nchunk;
    const int b = bh / p.H, h = bh - b * p.H;
    const half_t* Kb = p.k + (long long)b * p.Nk * p.ldk + h * p.d;
    const half_t* Vb = p.vt + (long long)b * p.vt_bs + (long long)h * p.d * p.ldvt;
    f16x8 z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) z8[e] = (half_t)0.f;

    // K fragments (A operand of S^T = K.Q^T): lane row = key 32kt + perm(lr) (bits 2,3 exchanged, see attn_fused_kernel)
    const int prow = (lr & 0x13) | ((lr & 4) << 1) | ((lr & 8) >> 1);
    f16x8 kf[KT][KS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int key = kt * 32 + prow, dd = ks * 16 + lh * 8;
            kf[kt][ks] = (key < p.Nk && dd < p.d) ? *reinterpret_cast<const f16x8*>(Kb + (long long)key * p.ldk + dd) : z8;
        }
    // V^T fragments (A operand of O^T = V^T.P^T): lane row = head-dim 32i + lr, keys 16st + 8lh .. +7
    f16x8 vf[ST][DT];
#pragma unroll
    for (int st = 0; st < STL; ++st)
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const int row = i * 32 + lr, key = st * 16 + lh * 8;
            vf[st][i] = (row < p.d && key < p.ldvt) ? *reinterpret_cast<const f16x8*>(Vb + (long long)row * p.ldvt + key) : z8;
        }
    const float c = p.scale_log2;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int q_first = (qc * 4 + wv) * tiles_per_wave * 32;

    auto load_q = [&](f16x8 (&qf)[KS], int q0) {
        const int qrow = q0 + lr;
        const half_t* qp = p.q + ((long long)b * p.Nq + qrow) * p.ldq + h * p.d;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dd = ks * 16 + lh * 8;
            qf[ks] = (qrow < p.Nq && dd < p.d) ? *reinterpret_cast<const f16x8*>(qp + dd) : z8;
        }
    };
    // Key mask as an additive bias held in 16 VGPRs (round 3): as 48 per-element comparisons against Nk the loop-invariant predicates
    // lived in scalar-register pairs, 124 of them spilt, and every tile paid ~110 v_readlane_b32 to get them back (as many issue
    // slots as its 21 MFMAs + 48 exponentials).  The host sends only 64 < Nk <= 96 here, so tiles 0 and 1 need no mask at all.
    float mb[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        mb[e] = (KT - 1) * 32 + 16 * (e >> 3) + 8 * lh + (e & 7) >= p.Nk ? -INFINITY : 0.f;
        asm volatile("" : "+v"(mb[e]));                   // opaque: or the select is rematerialised inside the loop, predicates and all
    }
    // Q fragments of TWO tiles ahead are in flight under a tile's work (three named register sets, no copies): with one tile ahead the
    // kernel moved 3.1 TB/s - 2.5 KB in flight per wave against ~2 us of loaded HBM latency (round 3)
    f16x8 qa[KS], qb[KS], qd[KS];
    load_q(qa, q_first);
    if (tiles_per_wave > 1 && q_first + 32 < p.Nq) load_q(qb, q_first + 32);
    auto tile = [&](f16x8 (&qf)[KS], f16x8 (&qn)[KS], int q0, bool more) {
        if (more) load_q(qn, q0 + 64);                    // the tile after next
        f32x16 s[KT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][ks], qf[ks], ks == 0 ? zero16 : s[kt], 0, 0, 0);
        // element e of s[kt] is key 32kt + 16(e>>3) + 8lh + (e&7); the slots past Nk (all in the last tile: 64 < Nk <= 96) get -inf
        // from the per-lane bias `mb`, exact softmax over the rest
#pragma unroll
        for (int e = 0; e < 16; ++e) s[KT - 1][e] += mb[e];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (kt * 2 + (e >> 3) < STL) mx = fmaxf(mx, s[kt][e]);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float nmc = -mx * c;
        float rs = 0.f;
        f16x8 pf[ST];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (kt * 2 + (e >> 3) >= STL) continue;           // a slot no key lives in (compile time)
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[kt][e], c, nmc));
                rs += pv;
                pf[kt * 2 + (e >> 3)][e & 7] = (half_t)pv;
            }
        f32x16 o[DT];
#pragma unroll
        for (int st = 0; st < STL; ++st)
#pragma unroll
            for (int i = 0; i < DT; ++i)
                o[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[st][i], pf[st], st == 0 ? zero16 : o[i], 0, 0, 0);
        float l_tot;
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
            l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        const float inv = 1.0f / l_tot;
        const int qrow = q0 + lr;
        if (qrow < p.Nq) {
            half_t* op = p.out + ((long long)b * p.Nq + qrow) * p.ldo + h * p.d;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dc = i * 32 + 8 * g + 4 * lh;
                    if (dc < p.d) {
                        f16x4 v = {(half_t)(o[i][4 * g] * inv), (half_t)(o[i][4 * g + 1] * inv),
                                   (half_t)(o[i][4 * g + 2] * inv), (half_t)(o[i][4 * g + 3] * inv)};
                        *reinterpret_cast<f16x4*>(op + dc) = v;
                    }
                }
        }
    };
    auto live = [&](int t) { return t < tiles_per_wave && q_first + t * 32 < p.Nq; };
    for (int t = 0; live(t); t += 3) {
        const int q0 = q_first + t * 32;
        tile(qa, qd, q0, live(t + 2));
        if (live(t + 1)) tile(qb, qa, q0 + 32, live(t + 3));
        if (live(t + 2)) tile(qd, qb, q0 + 64, live(t + 4));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Materialised attention probabilities in ONE pass over the operands (layers whose controller reads or edits P,
// utils/p2p.py:335-338): P[b*H + h][q][key] = softmax_key(scale * q . k) as fp16, straight from the S^T accumulators -
// the fp32 score tensor of the old path (QK^T GEMM -> 4 B/element out, 4 B/element back in, softmax kernel) never exists.
// A wave owns one 32-query tile of one (batch, head): Q fragments in registers, K fragments streamed from global / L2
// (one k-tile ahead) as the MFMA A operand in the bit-swapped row order that makes a lane's 16 accumulators two runs of 8
// consecutive keys.  <= 96 keys (cross-attention): the three S^T tiles stay in registers, exact softmax.  More keys
// (self-attention of the <= 32^2 layers): two sweeps over K - running (max, sum) first, then S^T again and
// P = exp2(s - max) / sum; recomputing 4*Nq*Nk*d flops is far cheaper than 8 B/element of HBM traffic.  Probabilities
// leave through a per-wave LDS patch as 128-B row segments; pad columns [Nk, ldp) are written as zeros.
// ---------------------------------------------------------------------------------------------------------------
struct ProbsK {
    const half_t* q; const half_t* k; half_t* p;
    int B, H, Nq, Nk, d, ldq, ldk, ldp;
    float scale_log2;
    const unsigned char* qc; const unsigned char* kc;      // SPLIT: error carries of q / k (same indexing, one byte per element), or null
};

// 8 carry bytes -> 8 fp16 values bf8_e5m2(byte) (UNSCALED: a bf8 e5m2 number is the top byte of the fp16 with the same value)
__device__ __forceinline__ f16x8 carry8_as_f16(const u32x2 w) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 r = {__builtin_amdgcn_perm(w[0], 0u, 0x050c040cu), __builtin_amdgcn_perm(w[0], 0u, 0x070c060cu),
                     __builtin_amdgcn_perm(w[1], 0u, 0x050c040cu), __builtin_amdgcn_perm(w[1], 0u, 0x070c060cu)};
    return __builtin_bit_cast(f16x8, r);
}

// SPLIT (round 5, the accurate precision level on layers whose probabilities a controller keeps): q and k arrive with the error carry
// of their fp16 rounding (icd_gemm_desc.out_carry) and the scores are q.k = qh.kh + 2^-14 (ql.kh + qh.kl) - three MFMAs per k-step on
// two accumulators (the lo products in their own, unscaled).  A stored probability map is exp of these scores: its relative error IS the
// absolute error of the score, and the fp16 rounding of q and k is 12 % of the attention-store error budget (tests/error_budget_sim.py).
//
// EPI (round 5): what the reference's shipped controllers do to the probabilities of a layer (utils/p2p.py:138-221) in this kernel's
// epilogue instead of in passes of their own over P (icd_probs_epilogue):
//   * AttentionStore.between_steps: acc += P on the final probabilities (torch's fp16 add: fp16(float(acc) + float(P)));
//   * AttentionControlEdit.replace_self_attention inside its step window: the edited prompts take the base prompt's probabilities - their
//     blocks simply read the base sample's q and k (the same arithmetic, hence the same bits as the copy);
//   * the cross-attention edit of AttentionReplace / Refine / Reweight with its cross_replace_alpha blend, one linear operator per step,
//     new_row[e] = base_row . A_e + D_e (*) cur_row[e] (csrc/p2p.hip): the block of an edited prompt recomputes the base prompt's
//     probabilities of its query tile (<= 96 keys: a few MFMAs), uses them - rounded to fp16 as the two-pass form reads them from memory -
//     as the MFMA B operand straight from the accumulator layout, and combines with its own probabilities through the LDS patch that
//     already stages the rows for the store.  Same operands in the same order as icd_p2p_cross_edit: the same bits.
// The samples of the launch are [b0 unrelated samples (the unconditional half of a CFG-doubled batch) | base prompt | edited prompts ...].
struct ProbsEpi {
    half_t* acc;            // [ (B - b0) * H, Nq, ldp ] or null
    const half_t* At;       // [nedit][96][80] fp16 (ops.p2p_pack_operator) or null
    const float* D;         // [nedit][96]
    int b0, self_base;
};

template <int KS, bool SPLIT = false, int EPI = 0, bool FEW = false>     // EPI: 0 none, 1 store += P / self replacement, 2 also the cross edit
__global__ __launch_bounds__(256, (KS <= 5 && !FEW && EPI == 0) ? 3 : 2) void attn_probs_kernel(ProbsK a, ProbsEpi ep) {
    __shared__ __attribute__((aligned(16))) half_t patch_all[4][32 * 72];
    const int tid = threadIdx.x, l = tid & 63, lr = l & 31, lh = l >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
    const int q0 = (blockIdx.x * 4 + wv) * 32;
    const bool live = q0 < a.Nq;                           // (the many-keys path stages K per BLOCK: a wave without queries still loads and syncs)
    if (FEW && !live) return;
    half_t* patch = patch_all[wv];
    f16x8 z8;
#pragma unroll
    for (int e = 0; e < 8; ++e) z8[e] = (half_t)0.f;
    constexpr int KF = SPLIT ? 2 * KS : KS;            // fragments per operand: [hi (KS) | lo (KS)]
    const u32x2 zc = {0u, 0u};
    const int jp = EPI && b >= ep.b0 ? b - ep.b0 : 0;  // prompt index among the conditional samples (0 = the base prompt)
    const int bq = EPI && ep.self_base && jp > 0 ? ep.b0 : b;       // the sample whose q and k this block reads
    // q fragments: the carry stays packed (two registers per fragment, widened to fp16 where an MFMA takes it: two v_perm per use)
    struct QFrags { f16x8 hi[KS]; u32x2 lo[SPLIT ? KS : 1]; };
    auto load_q = [&](QFrags& qq, int bs) {
        const int qrow = q0 + lr;
        const long long qoff = ((long long)bs * a.Nq + qrow) * a.ldq + h * a.d;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dd = ks * 16 + lh * 8;
            const bool okq = qrow < a.Nq && dd < a.d;
            qq.hi[ks] = okq ? *reinterpret_cast<const f16x8*>(a.q + qoff + dd) : z8;
            if (SPLIT) qq.lo[ks] = (okq && a.qc) ? *reinterpret_cast<const u32x2*>(a.qc + qoff + dd) : zc;
        }
    };
    QFrags qf;
    if constexpr (!FEW) load_q(qf, bq);                  // (the few-keys path loads it after the base prompt's probabilities: registers)
    const int prow = (lr & 0x13) | ((lr & 4) << 1) | ((lr & 8) >> 1);
    auto load_k_of = [&](f16x8 (&kf)[KF], int kt, int bs) {
        const long long koff0 = (long long)bs * a.Nk * a.ldk + h * a.d;
        const int key = kt * 32 + prow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dd = ks * 16 + lh * 8;
            const bool okk = key < a.Nk && dd < a.d;
            const long long off = koff0 + (long long)key * a.ldk + dd;
            kf[ks] = okk ? *reinterpret_cast<const f16x8*>(a.k + off) : z8;
            if (SPLIT) kf[KS + ks] = carry8_as_f16((okk && a.kc) ? *reinterpret_cast<const u32x2*>(a.kc + off) : zc);
        }
    };
    auto load_k = [&](f16x8 (&kf)[KF], int kt) { load_k_of(kf, kt, bq); };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float c = a.scale_log2;
    auto scores_of = [&](f32x16& s, const f16x8 (&kf)[KF], const QFrags& qq, int kt) {     // s = log2(e) * scale * q.k, keys past Nk -> -inf
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qq.hi[ks], ks == 0 ? zero16 : s, 0, 0, 0);
        if (SPLIT) {                                                           // + 2^-14 (kl.qh + kh.ql)
            f32x16 t;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[KS + ks], qq.hi[ks], ks == 0 ? zero16 : t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], carry8_as_f16(qq.lo[ks]), t, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = __builtin_fmaf(t[e], 1.f / 16384.f, s[e]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = 16 * (e >> 3) + (e & 7) < a.Nk - kt * 32 - 8 * lh ? s[e] * c : -INFINITY;      // key < Nk
        }
    };
    auto scores = [&](f32x16& s, const f16x8 (&kf)[KF], int kt) { scores_of(s, kf, qf, kt); };
    auto half_swap_max = [&](float v) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    auto half_swap_sum = [&](float v) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    };
    half_t* Pb = a.p + ((long long)bh * a.Nq + q0) * a.ldp;
    half_t* Ab = EPI && ep.acc && b >= ep.b0 ? ep.acc + ((long long)(jp * a.H + h) * a.Nq + q0) * a.ldp : nullptr;
    // one k-tile of probabilities (this lane: query lr, keys 32kt + 16j + 8lh .. +7, j = 0, 1) -> patch -> global rows
    auto emit = [&](const f32x16& pv, int kt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)pv[8 * j + e];
            *reinterpret_cast<f16x8*>(patch + lr * 72 + (kt & 1) * 32 + 16 * j + 8 * lh) = o;
        }
    };
    // store += P rides in the flush; its read of the accumulator is requested a k-tile pair ahead (acc_fetch), so the round trip to HBM
    // sits under the scores / exponentials of the pair and not between the patch read and the two stores
    struct AccRows { f16x8 t[4]; };
    auto acc_fetch = [&](AccRows& ar, int kt_first, int ncols) {
        if (!(EPI && Ab)) return;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + (l >> 3), c8 = (l & 7) * 8;
            const int col = kt_first * 32 + c8;
            if (c8 < ncols && col < a.ldp && q0 + r < a.Nq) ar.t[pass] = *reinterpret_cast<const f16x8*>(Ab + (long long)r * a.ldp + col);
        }
    };
    auto flush = [&](int kt_first, int ncols, const AccRows& ar) {             // patch columns [0, ncols) are keys 32*kt_first ...
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + (l >> 3), c8 = (l & 7) * 8;
            const int col = kt_first * 32 + c8;
            if (c8 < ncols && col < a.ldp && q0 + r < a.Nq) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(patch + r * 72 + c8);
                *reinterpret_cast<f16x8*>(Pb + (long long)r * a.ldp + col) = v;
                if (EPI && Ab) {                                               // store += P (torch's fp16 in-place add)
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)ar.t[pass][e] + (float)v[e]);
                    *reinterpret_cast<f16x8*>(Ab + (long long)r * a.ldp + col) = o;
                }
            }
        }
    };
    const int nt = (a.ldp + 31) >> 5;                      // k-tiles incl. the pad columns (they are written as zeros)
    if constexpr (FEW) {
        // ---- few keys (cross-attention, <= 96 key slots; the host picks the instantiation): all S^T tiles in registers, exact softmax ----
        auto probs3 = [&](f32x16 (&s)[3], const QFrags& qq, int bs) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                if (kt < nt) {
                    f16x8 kf[KF];
                    load_k_of(kf, kt, bs);
                    scores_of(s[kt], kf, qq, kt);
#pragma unroll
                    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kt][e]);
                }
            }
            mx = half_swap_max(mx);
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
                if (kt < nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { s[kt][e] = __builtin_amdgcn_exp2f(s[kt][e] - mx); rs += s[kt][e]; }
            const float inv = 1.0f / half_swap_sum(rs);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
                if (kt < nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s[kt][e] *= inv;
        };
        const bool edit = EPI == 2 && ep.At != nullptr && jp > 0;
        f16x8 pb[5];                                       // the base prompt's probabilities of this query tile as the MFMA B operand
        if (EPI == 2 && edit) {                            // (before this block's own q is loaded: one set of q fragments live at a time)
            QFrags qb;
            load_q(qb, ep.b0);
            f32x16 sb[3];
            probs3(sb, qb, ep.b0);
#pragma unroll
            for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[ks][e] = (ks >> 1) < nt ? (half_t)sb[ks >> 1][8 * (ks & 1) + e] : (half_t)0.f;
        }
        load_q(qf, bq);
        f32x16 s[3];
        probs3(s, qf, bq);
        AccRows ar;
        acc_fetch(ar, 0, nt >= 2 ? 64 : 32);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            if (kt < nt) {
                emit(s[kt], kt);
                if (EPI == 2 && edit) {
                    // new[n] = sum_w At[n][w] base[w] + D[n] cur[n] for the 32 tokens of this tile; cur comes back from the patch in
                    // the accumulator layout (row lr, tokens 8g + 4lh .. +3)
                    const half_t* Ae = ep.At + (long long)(jp - 1) * (96 * 80);
                    f32x16 acc;
#pragma unroll
                    for (int ks = 0; ks < 5; ++ks) {
                        const f16x8 af = *reinterpret_cast<const f16x8*>(Ae + (kt * 32 + lr) * 80 + ks * 16 + lh * 8);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, pb[ks], ks == 0 ? zero16 : acc, 0, 0, 0);
                    }
                    const float* De = ep.D + (long long)(jp - 1) * 96 + kt * 32;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = 8 * g + 4 * lh;
                        if (kt * 32 + n + 4 > a.ldp) continue;                 // token slots past the row (pad columns stay zero)
                        half_t* pp = patch + lr * 72 + (kt & 1) * 32 + n;
                        const f16x4 cur = *reinterpret_cast<const f16x4*>(pp);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(De + n);
                        f16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (half_t)fmaf(d[i], (float)cur[i], acc[4 * g + i]);
                        *reinterpret_cast<f16x4*>(pp) = o;
                    }
                }
                if ((kt & 1) || kt == nt - 1) {
                    flush(kt & ~1, (kt & 1) ? 64 : 32, ar);
                    if (kt == 1 && nt == 3) acc_fetch(ar, 2, 32);
                }
            }
        }
    } else {
    // ---- many keys: sweep 1 = running row maximum and sum, sweep 2 = probabilities ----
    // K is staged through LDS once per BLOCK and sweep (round 5): the four waves of a block score the same keys, and with every wave
    // fetching its own fragments from global memory one tile ahead the 1024 x 1024 layers of SD1.5 sat at 99 us (fp16 operands) / 178 us
    // (split operands, two waves per SIMD) for 134 MB of P - one exposed L2 round trip per key tile.  Now 256 threads copy SK keys (hi
    // rows, and the raw carry bytes when SPLIT) global -> registers -> LDS one stage ahead of the MFMAs, one barrier per stage; fragments
    // come from LDS (row pitches of an odd number of 16-B / 8-B units: conflict-free b128 / b64 reads).
    constexpr int SK = KS <= 5 ? 64 : 32, TPS = SK / 32;          // keys / key tiles per stage
    constexpr int HS = KS * 32 + 16, LS = KS * 16 + 8;            // row pitch of the hi / carry image (bytes)
    constexpr int HI_ITEMS = SK * KS * 2, LO_ITEMS = SPLIT ? SK * KS * 2 : 0;      // 16-B / 8-B items per stage
    constexpr int NHI = (HI_ITEMS + 255) / 256, NLO = SPLIT ? (LO_ITEMS + 255) / 256 : 1;
    constexpr int LO_OFF = SK * HS, STAGE = SK * HS + (SPLIT ? SK * LS : 0);
    __shared__ __attribute__((aligned(16))) unsigned char kst[2][STAGE];
    const long long kbase = (long long)bq * a.Nk * a.ldk + h * a.d;
    f16x8 ph[NHI];
    u32x2 pl[NLO];
    auto fetch = [&](int stage) {                                  // global -> registers
        const int key0 = stage * SK;
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
            const int item = j * 256 + tid, key = item / (2 * KS), c = item - key * (2 * KS);
            const bool okk = item < HI_ITEMS && key0 + key < a.Nk && c * 8 < a.d;
            ph[j] = okk ? *reinterpret_cast<const f16x8*>(a.k + kbase + (long long)(key0 + key) * a.ldk + c * 8) : z8;
        }
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < NLO; ++j) {
                const int item = j * 256 + tid, key = item / (2 * KS), c = item - key * (2 * KS);
                const bool okk = a.kc && item < LO_ITEMS && key0 + key < a.Nk && c * 8 < a.d;
                pl[j] = okk ? *reinterpret_cast<const u32x2*>(a.kc + kbase + (long long)(key0 + key) * a.ldk + c * 8) : zc;
            }
        }
    };
    auto stash = [&](int buf) {                                    // registers -> LDS
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
            const int item = j * 256 + tid, key = item / (2 * KS), c = item - key * (2 * KS);
            if (item < HI_ITEMS) *reinterpret_cast<f16x8*>(kst[buf] + key * HS + c * 16) = ph[j];
        }
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < NLO; ++j) {
                const int item = j * 256 + tid, key = item / (2 * KS), c = item - key * (2 * KS);
                if (item < LO_ITEMS) *reinterpret_cast<u32x2*>(kst[buf] + LO_OFF + key * LS + c * 8) = pl[j];
            }
        }
    };
    auto frags = [&](f16x8 (&kf)[KF], int buf, int t2) {           // this lane's A-operand fragments of key tile t2 of the stage
        const int key = t2 * 32 + prow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *reinterpret_cast<const f16x8*>(kst[buf] + key * HS + ks * 32 + lh * 16);
            if (SPLIT) kf[KS + ks] = carry8_as_f16(*reinterpret_cast<const u32x2*>(kst[buf] + LO_OFF + key * LS + (ks * 2 + lh) * 8));
        }
    };
    const int nst = (nt + TPS - 1) / TPS;
    float m_run = -INFINITY, l_run = 0.f;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int sg = 0; sg < nst; ++sg) {
        if (sg + 1 < nst) fetch(sg + 1);
#pragma unroll 1
        for (int t2 = 0; t2 < TPS; ++t2) {       // (not unrolled: one tile's fragments live at a time)
            const int kt = sg * TPS + t2;
            if (live && kt < nt) {
                f16x8 kf[KF];
                frags(kf, sg & 1, t2);
                f32x16 s;
                scores(s, kf, kt);
                float tm = -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) tm = fmaxf(tm, s[e]);
                const float mn = fmaxf(m_run, tm);               // this half-wave's view; the halves are merged after the sweep
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc += __builtin_amdgcn_exp2f(s[e] - mn);
                l_run = l_run * __builtin_amdgcn_exp2f(m_run - mn) + acc;
                m_run = mn;
            }
        }
        if (sg + 1 < nst) stash((sg + 1) & 1);
        __syncthreads();
    }
    {   // merge the two half-waves (each saw 16 of every 32 keys); a half that saw only masked keys has m = -inf, l = 0
        const float m_all = half_swap_max(m_run);
        const float mine = m_run == -INFINITY ? 0.f : l_run * __builtin_amdgcn_exp2f(m_run - m_all);
        l_run = half_swap_sum(mine);
        m_run = m_all;
    }
    const float inv = 1.0f / l_run;
    AccRows ar;
    fetch(0);
    stash(0);                                                    // (every wave is past the last barrier of sweep 1: both buffers are free)
    __syncthreads();
    for (int sg = 0; sg < nst; ++sg) {
        if (sg + 1 < nst) fetch(sg + 1);
#pragma unroll 1
        for (int t2 = 0; t2 < TPS; ++t2) {       // (not unrolled: one tile's fragments live at a time)
            const int kt = sg * TPS + t2;
            if (live && kt < nt) {
                if (!(kt & 1)) acc_fetch(ar, kt, kt + 1 < nt ? 64 : 32);
                f16x8 kf[KF];
                frags(kf, sg & 1, t2);
                f32x16 s;
                scores(s, kf, kt);
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(s[e] - m_run) * inv;
                emit(s, kt);
                if ((kt & 1) || kt == nt - 1) flush(kt & ~1, (kt & 1) ? 64 : 32, ar);
            }
        }
        if (sg + 1 < nst) stash((sg + 1) & 1);
        __syncthreads();
    }
    }
}

template <int KS, int DT>
int launch_attn_cross(AttnK k, hipStream_t st) {
    // tiles per wave: amortise the K / V^T fragment loads (24 x 1 KiB per wave from L2) but keep >= ~4 blocks per CU
    // (round-3 sweep of 4 - 32 tiles per wave and 512 - 1024 blocks: within the noise of each other)
    int tpw = 8;
    while (tpw > 1 && (long long)((k.Nq + 128 * tpw - 1) / (128 * tpw)) * k.B * k.H < 1024) tpw >>= 1;
    const int nchunk = (k.Nq + 128 * tpw - 1) / (128 * tpw);
    if (k.Nk <= 80) hipLaunchKernelGGL((attn_cross_kernel<KS, DT, 5>), dim3(nchunk * k.B * k.H), dim3(256), 0, st, k, tpw);
    else hipLaunchKernelGGL((attn_cross_kernel<KS, DT, 6>), dim3(nchunk * k.B * k.H), dim3(256), 0, st, k, tpw);
    ICD_CHECK_LAUNCH("icd_attention_fused(cross)");
    return ICD_OK;
}

template <int KS, int DT, int QT, int OCC, int NST, int MODE, bool CAUSAL, int DR = 0>
int launch_attn_c(AttnK k, hipStream_t st) {
    constexpr int smem = NST * (64 * 2 * KS * 16 + (DR ? DR * 16 : DT * 32) * 128);
    static_assert(smem * OCC <= 160 * 1024, "LDS ring x occupancy exceeds the CU's 160 KiB");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fused_kernel<KS, DT, QT, OCC, NST, MODE, CAUSAL, DR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    k.nqt = (k.Nq + 128 * QT - 1) / (128 * QT);
    hipLaunchKernelGGL((attn_fused_kernel<KS, DT, QT, OCC, NST, MODE, CAUSAL, DR>), dim3(k.nqt * k.B * k.H), dim3(256), smem, st, k);
    ICD_CHECK_LAUNCH("icd_attention_fused");
    return ICD_OK;
}
template <int KS, int DT, int QT, int OCC = 2, int NST = 2, int MODE = 0, int DR = 0>
int launch_attn(AttnK k, hipStream_t st) {
    // causal instantiations: every one-query-tile-per-wave kernel of MODE 0, and the head-dim-64 MODE 2 kernel (the CLIP text
    // encoders); icd_attention_fused_ex routes causal calls to those
    if constexpr (QT == 1 && (MODE == 0 || (MODE == 2 && KS == 4))) { if (k.causal) return launch_attn_c<KS, DT, QT, OCC, NST, MODE, true>(k, st); }
    if (k.causal) { icd_set_error("icd_attention_fused: internal: no causal instantiation for this tile"); return ICD_ERR_UNSUPPORTED; }
    return launch_attn_c<KS, DT, QT, OCC, NST, MODE, false, DR>(k, st);
}

}  // namespace

extern "C" int icd_attention_fused_ex(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t H,
                                      int32_t Nq, int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldvt, int32_t ldo,
                                      int64_t vt_batch_stride, float scale, int32_t flags, void* stream);

static int attention_probs_run(const void* q, const void* qc, const void* k, const void* kc, void* probs, int32_t B, int32_t H, int32_t Nq,
                               int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldp, float scale, void* stream,
                               const icd_probs_epilogue* epi = nullptr) {
    ICD_CHECK_ARG(q && k && probs, "icd_attention_probs: null pointer");
    ICD_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0, "icd_attention_probs: empty shape");
    ICD_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 160, "icd_attention_probs: head dim must be a multiple of 8, <= 160 (got %d)", d);
    ICD_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldp % 8 == 0 && ldp >= Nk, "icd_attention_probs: leading dims must be 16-byte aligned, ldp >= Nk");
    ICD_CHECK_ARG(scale > 0.f, "icd_attention_probs: scale must be positive");
    ICD_CHECK_ARG((long long)B * H <= 65535, "icd_attention_probs: B * H exceeds the grid limit");
    ProbsK a;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.p = (half_t*)probs;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d; a.ldq = ldq; a.ldk = ldk; a.ldp = ldp;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.qc = (const unsigned char*)qc; a.kc = (const unsigned char*)kc;
    ProbsEpi ep{nullptr, nullptr, nullptr, 0, 0};
    if (epi) {
        ICD_CHECK_ARG(epi->first_cond_sample >= 0 && epi->first_cond_sample < B, "icd_attention_probs_ex: first_cond_sample out of range");
        ICD_CHECK_ARG(!epi->edit_At || (epi->edit_D && Nk <= 80 && ldp <= 96),
                      "icd_attention_probs_ex: the cross-attention edit needs edit_D, <= 80 keys and a base + >= 1 edited prompt");
        ep.acc = (half_t*)epi->acc; ep.At = (const half_t*)epi->edit_At; ep.D = epi->edit_D;
        ep.b0 = epi->first_cond_row > 0 ? (int)(epi->first_cond_row / H) : epi->first_cond_sample; ep.self_base = epi->self_from_base != 0;
        ICD_CHECK_ARG(epi->first_cond_row % H == 0 && ep.b0 < B, "icd_attention_probs_ex: first_cond_row must be a multiple of H inside the batch");
        ICD_CHECK_ARG(!(ep.At || ep.self_base) || B - ep.b0 >= 2, "icd_attention_probs_ex: an edit needs a base prompt and at least one edited prompt");
        ICD_CHECK_ARG(epi->edit_count <= 0 || epi->edit_count == B - ep.b0 - 1,
                      "icd_attention_probs_ex: edit_count %d does not match the %d edited samples of the launch", epi->edit_count, B - ep.b0 - 1);
    }
    const bool use_epi = ep.acc || ep.At || ep.self_base;
    const dim3 grid((unsigned)((Nq + 127) / 128), (unsigned)(B * H));
    hipStream_t st = (hipStream_t)stream;
#define ICD_PROBS(SP, EP, FW)                                                                                  \
    do {                                                                                                      \
        if (d <= 48) hipLaunchKernelGGL((attn_probs_kernel<3, SP, EP, FW>), grid, dim3(256), 0, st, a, ep);       \
        else if (d <= 80) hipLaunchKernelGGL((attn_probs_kernel<5, SP, EP, FW>), grid, dim3(256), 0, st, a, ep);  \
        else if (d <= 128) hipLaunchKernelGGL((attn_probs_kernel<8, SP, EP, FW>), grid, dim3(256), 0, st, a, ep); \
        else hipLaunchKernelGGL((attn_probs_kernel<10, SP, EP, FW>), grid, dim3(256), 0, st, a, ep);              \
    } while (0)
#define ICD_PROBS2(SP, EP) do { if (ldp <= 96) ICD_PROBS(SP, EP, true); else ICD_PROBS(SP, EP, false); } while (0)
    if (qc || kc) { if (ep.At) ICD_PROBS(true, 2, true); else if (use_epi) ICD_PROBS2(true, 1); else ICD_PROBS2(true, 0); }
    else { if (ep.At) ICD_PROBS(false, 2, true); else if (use_epi) ICD_PROBS2(false, 1); else ICD_PROBS2(false, 0); }
#undef ICD_PROBS2
#undef ICD_PROBS
    ICD_CHECK_LAUNCH("icd_attention_probs");
    return ICD_OK;
}

extern "C" int icd_attention_probs(const void* q, const void* k, void* probs, int32_t B, int32_t H, int32_t Nq, int32_t Nk,
                                   int32_t d, int32_t ldq, int32_t ldk, int32_t ldp, float scale, void* stream) {
    return attention_probs_run(q, nullptr, k, nullptr, probs, B, H, Nq, Nk, d, ldq, ldk, ldp, scale, stream);
}

extern "C" int icd_attention_probs_split(const void* q, const void* q_carry, const void* k, const void* k_carry, void* probs, int32_t B,
                                         int32_t H, int32_t Nq, int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldp, float scale,
                                         void* stream) {
    return attention_probs_run(q, q_carry, k, k_carry, probs, B, H, Nq, Nk, d, ldq, ldk, ldp, scale, stream);
}

extern "C" int icd_attention_probs_ex(const void* q, const void* q_carry, const void* k, const void* k_carry, void* probs, int32_t B,
                                      int32_t H, int32_t Nq, int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldp, float scale,
                                      const icd_probs_epilogue* epilogue, void* stream) {
    return attention_probs_run(q, q_carry, k, k_carry, probs, B, H, Nq, Nk, d, ldq, ldk, ldp, scale, stream, epilogue);
}

extern "C" int icd_attention_fused(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t H,
                                   int32_t Nq, int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldvt,
                                   int32_t ldo, int64_t vt_batch_stride, float scale, void* stream) {
    return icd_attention_fused_ex(q, k, vt, out, B, H, Nq, Nk, d, ldq, ldk, ldvt, ldo, vt_batch_stride, scale, 0, stream);
}

extern "C" int icd_attention_fused_ex(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t H,
                                      int32_t Nq, int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldvt, int32_t ldo,
                                      int64_t vt_batch_stride, float scale, int32_t flags, void* stream) {
    ICD_CHECK_ARG((flags & ~(ICD_ATTN_CAUSAL | ICD_ATTN_Q_PRESCALED | ICD_ATTN_TUNE_MODE0)) == 0, "icd_attention_fused: unknown flags 0x%x", flags);
    ICD_CHECK_ARG(!(flags & ICD_ATTN_CAUSAL) || Nq == Nk, "icd_attention_fused: the causal mask needs Nq == Nk");
    ICD_CHECK_ARG(scale > 0.f, "icd_attention_fused: scale must be positive");
    ICD_CHECK_ARG(q && k && vt && out, "icd_attention_fused: null pointer");
    ICD_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nk > 0, "icd_attention_fused: empty shape");
    ICD_CHECK_ARG(d > 0 && d % 8 == 0 && d <= 160, "icd_attention_fused: head dim must be a multiple of 8, <= 160 (got %d)", d);
    ICD_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0 && ldvt >= Nk,
                  "icd_attention_fused: leading dims must be 16-byte aligned and ldvt >= Nk");
    AttnK a;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.vt = (const half_t*)vt; a.out = (half_t*)out;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d; a.ldq = ldq; a.ldk = ldk; a.ldvt = ldvt; a.ldo = ldo;
    a.vt_bs = vt_batch_stride > 0 ? vt_batch_stride : (long long)H * d * ldvt;
    const bool presc = (flags & ICD_ATTN_Q_PRESCALED) != 0;
    a.scale_log2 = presc ? 1.0f : scale * 1.4426950408889634f;
    // MODE 1 / 2 kernels (exponent offset subtracted by the MFMA); a causal mask has the fast form at head dim 64 only
    const bool fast = presc && !(flags & ICD_ATTN_TUNE_MODE0) && (!(flags & ICD_ATTN_CAUSAL) || d == 64);
    a.nqt = 0;
    a.causal = (flags & ICD_ATTN_CAUSAL) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    // two query tiles per wave when the sequence is long enough to still fill the chip with 256-row workgroups
    const bool wide = (long long)((Nq + 255) / 256) * B * H >= 512 && Nk >= 256 && !a.causal;
    // cross-attention with enough query tiles to give every wave >= 2 of them (so the 24 KiB of K / V^T fragments per
    // wave amortise and the next Q tile prefetches): K / V^T fragments live in registers.  Shorter problems (SDXL's
    // 1024-query layers at batch 8) are launch + latency bound either way (~20 us for 42 MB) and stay on the tiled kernel.
    if (Nk > 64 && Nk <= 96 && d <= 64 && !a.causal && (long long)((Nq + 127) / 128) * B * H >= 2048) {
        if (d <= 32) return launch_attn_cross<2, 1>(a, st);
        if (d <= 48) return launch_attn_cross<3, 2>(a, st);
        return launch_attn_cross<4, 2>(a, st);
    }
    if (d <= 16) return launch_attn<1, 1, 1, 2, 2>(a, st);
    if (d <= 32) return launch_attn<2, 1, 1, 2, 2>(a, st);
    // (round 3, same-box round-robin A/B at 4096 x 4096: one query tile per wave, three blocks per CU and a 3-stage ring are all
    //  1 - 12 % slower than the configurations below with MODE 1 / 2; profiles/r03_attn_variants.txt)
    if (fast && d == 40 && wide) return launch_attn<3, 2, 2, 2, 2, 1, 3>(a, st);       // P.V on 16-row tiles (48 rows, not 64)
    if (fast && d == 40) return launch_attn<3, 2, 1, 2, 2, 1>(a, st);
    if (fast && d == 64) return launch_attn<4, 2, 1, 2, 2, 2>(a, st);
    if (fast && d == 80) return launch_attn<5, 3, 1, 2, 2, 2>(a, st);
    if (d <= 48) return wide ? launch_attn<3, 2, 2, 2, 2>(a, st) : launch_attn<3, 2, 1, 2, 2>(a, st);
    if (d <= 64) return launch_attn<4, 2, 1, 2, 2>(a, st);      // QT = 2 spills at d = 64 (measured slower)
    if (d <= 80) return launch_attn<5, 3, 1, 2, 2>(a, st);
    if (d <= 96) return launch_attn<6, 3, 1, 2, 2>(a, st);
    if (d <= 128) return launch_attn<8, 4, 1, 2, 2>(a, st);
    // head dim 160 (SD1.5's 16 x 16 / 8 x 8 levels): its accumulators and fragments do not fit the 256 registers of two waves per SIMD
    // (187 spilt); one wave per SIMD with the whole register file is 10 - 35 % faster (profiles/r03_attn_small_kernels.txt)
    return launch_attn<10, 5, 1, 1, 2>(a, st);
}
