// Shared declarations of the GEMM kernels (gemm.hip: 128x128 / 256x128 tiles; gemm_big.hip: 256x320 / 256x256 tiles).
#pragma once
#include "common.h"

namespace icd_gemm_detail {

constexpr int BK = 64;
constexpr int EPI_LD = 132;                      // fp32 staging row stride (floats) for a 64x128 slab
constexpr int EPI_LD_T = 68;                     // transposed staging: 128 rows (n) x 64 (m)

struct GemmK {
    const half_t* a0; const half_t* a1; const half_t* w;
    const float* bias; const half_t* rowbias; const half_t* resid; void* out;
    float* partial;                              // split-K: fp32 [S][M][N]
    int M, N, K, Nw;
    int lda, ldw, ldo, ldr, ld_rowbias, rps;
    int C0, C1, Hin, Win, Hout, Wout, ksize, stride, upsample;
    int zdiv; long long a_bs0, a_bs1, w_bs0, w_bs1, o_bs0, o_bs1;
    float alpha; int flags;
    int nbm, nbn, ksplit, kt_per_split;
    int gm;                                      // m-tiles per L2 group of the block -> tile map (tile_of_block)
    const float* ln_stats;                       // fused LayerNorm: fp32 [M][2] = (mean, rstd) of the A rows, or null
    const float* ln_s;                           //                  fp32 [N] = row sums of the gamma-scaled weights
    float* ln_stats_w;                           // ICD_GEMM_LN_COMPUTE taken in-kernel: (mean, rstd) of the A rows are computed from the
    float ln_eps;                                //   fragments in the main loop (gemm_big.hip), used from LDS and stored here by n-tile 0
    // cross-attention fused behind the query projection (icd_gemm_desc.xattn_*; gemm.hip xattn_epilogue)
    const half_t* xk; const half_t* xvt;
    int x_nk, x_ldk, x_ldvt; long long x_vt_bs; float x_scale_log2;
    float* out32;                                // second output (icd_gemm_desc.out_f32): the values before the fp16 rounding, or null
    const unsigned char* resid_c;                // error carry of `resid` (icd_gemm_desc.resid_carry): bf8 e5m2 of what fp16 lost, x 2^14
    unsigned char* out_c;                        // error carry of `out` (icd_gemm_desc.out_carry), or null
    unsigned long long* timeline;                // diagnostics (icd_debug_gemm_timeline): 4 s_memrealtime stamps per block, or null
    // conv over a 2 x 2 subset of the 3 x 3 taps (icd_gemm_desc.conv_tap_base / conv_ktaps: the phase form of the upsampling conv):
    // ktaps taps are iterated (4, or ksize^2), iterated tap u is tap tap_base + (u & 1) + 3 (u >> 1) of the 3 x 3 geometry
    // (few fields on purpose: the conv kernels sit at the scalar-register limit, every kernel argument is one more live SGPR)
    unsigned long long tapmap;                   // bits 4u..4u+3: index of iterated tap u in the 3 x 3 geometry (identity 0x876543210); bits 60..63: taps iterated
    // output row of GEMM row m = f * m + s * floor(m / W) + c, floor(m / W) = umulhi(m, orm_magic); orm_pack = f | s << 2 | c << 17 (identity: 1, magic 0)
    unsigned orm_pack, orm_magic;
};

// output row of GEMM row m (icd_gemm_desc.out_remap_w: pixel (2y + py, 2x + px) of the upsampled map; identity otherwise)
__device__ __forceinline__ long long out_row(const GemmK& p, int m) {
    const unsigned f = p.orm_pack & 3u, s = (p.orm_pack >> 2) & 0x7fffu, c = p.orm_pack >> 17;
    return (long long)(f * (unsigned)m + s * __umulhi((unsigned)m, p.orm_magic) + c);
}

// block id -> (m-tile, n-tile).  Block b runs on XCD b % 8 (observed, speed only): every XCD gets a contiguous range of
// the tile sequence, and inside it tiles are ordered in groups of `gm` m-tiles with the n-tile index outermost, so the
// ~32 blocks an XCD runs concurrently cover gm x (32 / gm) tiles: they share gm activation tiles and 32 / gm weight tiles
// through that XCD's L2 (12 operand tiles instead of 33 for gm = 8), and most global -> LDS loads become L2 hits.
__device__ __forceinline__ void tile_of_block(int bid, int nbm, int nbn, int gm, int& mt, int& nt) {
    const int nblk = nbm * nbn;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = gm * nbn;
    const int g = bid / per_group, rem = bid - g * per_group;
    const int rows = min(gm, nbm - g * gm);
    nt = rem / rows;
    mt = g * gm + (rem - nt * rows);
}

__device__ __forceinline__ int swz_off(int row, int chunk) {        // byte offset inside a [rows][64] half tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// gelu(x) = x Phi(x) = max(x, 0) - |x| erfc(|x| / sqrt 2) / 2, with log2(erfc(a / sqrt 2) / 2) as a degree-5 polynomial in
// a = min(|x|, 6) (weighted minimax fit; beyond 6 the term is < |x| 1e-9): |error| < 6.4e-7 in absolute terms over the fp16 range,
// evaluated in fp32 (tests/test_gelu_poly.py) - below half an fp16 ulp of any result of magnitude >= 2^-9, a few ulps of the 1e-4
// values of the negative tail; the Abramowitz-Stegun form it replaces had 2e-7.  ONE transcendental and 9 plain VALU operations per element; the
// Abramowitz-Stegun erf of rounds 1-2 (v_rcp + v_exp + 15 others) cost 52 issue cycles per element against ~30 (v_exp_f32 and
// v_rcp_f32 are 8-cycle instructions, tools/valu_rate.hip), and the GEGLU epilogue is VALU-bound: 64 evaluations per lane and tile.
// Two elements at a time so the Horner steps are v_pk_fma_f32 (hipcc keeps literal-constant FMAs scalar otherwise): per element
// 2.5 packed + 3 plain VALU issues and one v_exp_f32.
__device__ __forceinline__ f32x2 gelu_fast2(const f32x2 x) {
    const float x0 = x[0], x1 = x[1];
    const f32x2 a = {__builtin_amdgcn_fmed3f(__builtin_fabsf(x0), 0.f, 6.0f), __builtin_amdgcn_fmed3f(__builtin_fabsf(x1), 0.f, 6.0f)};
    const f32x2 c5 = {-0.00047329376684501767f, -0.00047329376684501767f}, c4 = {0.007084457669407129f, 0.007084457669407129f};
    const f32x2 c3 = {-0.05182715505361557f, -0.05182715505361557f}, c2 = {-0.4599926769733429f, -0.4599926769733429f};
    const f32x2 c1 = {-1.1507877111434937f, -1.1507877111434937f}, c0 = {-1.000037670135498f, -1.000037670135498f};
    f32x2 p = __builtin_elementwise_fma(c5, a, c4);
    p = __builtin_elementwise_fma(p, a, c3);
    p = __builtin_elementwise_fma(p, a, c2);
    p = __builtin_elementwise_fma(p, a, c1);
    p = __builtin_elementwise_fma(p, a, c0);
    return (f32x2){__builtin_fmaf(-__builtin_fabsf(x0), __builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_fmed3f(x0, 0.f, __builtin_inff())),
                   __builtin_fmaf(-__builtin_fabsf(x1), __builtin_amdgcn_exp2f(p[1]), __builtin_amdgcn_fmed3f(x1, 0.f, __builtin_inff()))};
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast2((f32x2){x, x})[0]; }

// LayerNorm folded into the GEMM that consumes it (icd_gemm_desc.ln_stats): with W' = W * gamma (folded at load time),
// LN(x) @ W^T = rstd_m * (x @ W'^T - mean_m * rowsum(W')_n) + (W @ beta)_n - the last term travels in `bias`.
// v[0..8) are 8 consecutive output columns n.. of row m (already scaled by alpha).
__device__ __forceinline__ void ln_correct8(float (&v)[8], const float* ln_stats, const float* ln_s, int m, int n) {
    const f32x2 st = *reinterpret_cast<const f32x2*>(ln_stats + 2 * (long long)m);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(ln_s + n), s1 = *reinterpret_cast<const f32x4*>(ln_s + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = st[1] * (v[e] - st[0] * s0[e]);
        v[4 + e] = st[1] * (v[4 + e] - st[0] * s1[e]);
    }
}


// Error carry of the residual stream (icd_gemm_desc.resid_carry / out_carry): value = fp16 + 2^-14 * bf8_e5m2.  |v - fp16(v)| is at most
// half an ulp of the fp16 value, so with the fixed scale 2^14 the carry of any |v| < 2^13 sits inside the bf8 range (e5m2: 32 binades)
// without a per-element exponent, and v_cvt_pk_bf8_f32 / v_cvt_pk_f32_bf8 convert two elements per instruction: 3.5 + 1.5 VALU
// issues per element for the write and the read side.  What is left of the rounding error is <= 2^-3 of it (2 mantissa bits,
// round to nearest even): the stream behaves like a 14..15-bit-mantissa tensor at 3 bytes per element.
constexpr float CARRY_SCALE = 16384.f, CARRY_INV = 1.f / 16384.f;
// v[0..8) += the 8 carries packed in w (element e in byte e)
__device__ __forceinline__ void carry_add8(float (&v)[8], const u32x2 w) {
    const f32x2 c0 = __builtin_amdgcn_cvt_pk_f32_bf8(w[0], false), c1 = __builtin_amdgcn_cvt_pk_f32_bf8(w[0], true);
    const f32x2 c2 = __builtin_amdgcn_cvt_pk_f32_bf8(w[1], false), c3 = __builtin_amdgcn_cvt_pk_f32_bf8(w[1], true);
    v[0] = __builtin_fmaf(c0[0], CARRY_INV, v[0]); v[1] = __builtin_fmaf(c0[1], CARRY_INV, v[1]);
    v[2] = __builtin_fmaf(c1[0], CARRY_INV, v[2]); v[3] = __builtin_fmaf(c1[1], CARRY_INV, v[3]);
    v[4] = __builtin_fmaf(c2[0], CARRY_INV, v[4]); v[5] = __builtin_fmaf(c2[1], CARRY_INV, v[5]);
    v[6] = __builtin_fmaf(c3[0], CARRY_INV, v[6]); v[7] = __builtin_fmaf(c3[1], CARRY_INV, v[7]);
}
// ... of the 8 consecutive elements at c (8-byte aligned)
__device__ __forceinline__ void carry_add8(float (&v)[8], const unsigned char* c) { carry_add8(v, *reinterpret_cast<const u32x2*>(c)); }
// the carry bytes of 8 values v whose fp16 roundings are o
__device__ __forceinline__ u32x2 carry_of8(const float (&v)[8], const f16x8& o) {
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)              // clamped to the largest finite e5m2 value: beyond |v| = 2^13 the carry saturates instead of
        d[e] = __builtin_amdgcn_fmed3f((v[e] - (float)o[e]) * CARRY_SCALE, -57344.f, 57344.f);      // turning into inf / NaN in the stream
    unsigned w0 = __builtin_amdgcn_cvt_pk_bf8_f32(d[0], d[1], 0, false);
    w0 = __builtin_amdgcn_cvt_pk_bf8_f32(d[2], d[3], w0, true);
    unsigned w1 = __builtin_amdgcn_cvt_pk_bf8_f32(d[4], d[5], 0, false);
    w1 = __builtin_amdgcn_cvt_pk_bf8_f32(d[6], d[7], w1, true);
    return (u32x2){w0, w1};
}

// gemm_big.hip tile configurations and their measured cost (tools/gemm_bench.py with forced configurations, one box):
// one launch costs rounds x (k-tiles x tk + fixed) where a block owns its CU (1 block / CU), tk = one k-tile of one
// block and fixed = prologue + exposed epilogue, both in units of one k-tile of a 256 x 256 block on a partly filled
// chip (about 1.3 - 1.7 us depending on the box's clocks).  tk_full applies when >= 200 CUs are busy (chip-level
// ceiling: the same block runs about 10 - 20 % slower), tk_part when <= 160.
struct BigTile { int bm, bn; bool geglu_ok; double tk_part, tk_full, fixed; };
constexpr int NUM_BIG_TILES = 7;
constexpr int PP_TILE = 4;                       // index of the ping-pong 256 x 256 tile (gemm_pp.hip)
constexpr int PP320_TILE = 5;                    //              ... 256 x 320 tile (gemm_pp320.hip)
constexpr int PP192_TILE = 6;                    //              ... 192 x 256 tile (gemm_pp.hip, TM = 3)
constexpr BigTile BIG_TILES[NUM_BIG_TILES] = {
    {256, 256, true, 1.00, 1.08, 9.5},
    {256, 320, false, 1.07, 1.30, 13.5},
    {192, 256, true, 0.70, 0.76, 9.6},
    {128, 320, false, 0.72, 0.80, 9.0},
    {256, 256, true, 0.90, 0.97, 9.0},           // gemm_pp.hip: the ping-pong main loop (1.585 vs 1.755 us per k-tile at 8192^3, prologue -0.5 us)
    {256, 320, false, 1.00, 1.21, 14.2},         // gemm_pp320.hip: k-tiles 3 - 7 % cheaper than the lockstep 256 x 320 tile, one more phase of fill and
                                                 // drain: slower below ~8 k-tiles (131072 x 320 x 320: 60.7 vs 55.3 us), faster from K = 640 on
                                                 // (32768 x 640 x 640: 37.6 vs 40.6; conv 131072 x 320 x 8640: 604 vs 640) - profiles/r06_tune_*.txt
    {192, 256, true, 0.66, 0.72, 9.3},           // gemm_pp.hip TM = 3: the lockstep 192 x 256 tile x 0.95 (8192 x 1280 x 5120: 103.5 vs 108.1 us, x 1280: 33.5 vs 35.0)
};
// Round 6: 128 x 320 / 128 x 256 tiles with four waves and TWO co-resident blocks per CU on a 32-wide k-step (gemm_duo.hip) were built for the
// short-K dense layers, measured and removed: 131072 x 320 x 320 58.1 us against 55.8 on the 256 x 320 tile, the K = 320 GEGLU projection
// 386.8 against 330.8 (profiles/r06_duo_tiles.txt) - two blocks sharing a CU do not hide each other's epilogue any better than the chip already does.
// Further configurations were built, measured and removed in round 2 (tools/gemm_timeline.py): a
// generated hand-scheduled 4-wave 128 x 128 main loop, a 256 x 128 x 32 tile with two co-resident blocks per CU, and a 256 x 160
// tile (exactly 256 blocks for M = 8192, N = 1280) both as 4 waves of 64 x 160 and as 8 waves sharing each wave tile between
// two k-halves.  None is faster: under a full-chip launch every tile family delivers the same ~4 TFLOP/s per CU because the chip is at its
// power limit (shader clock 1.3 - 1.7 GHz measured inside the main loop, 2.3 GHz when few CUs are busy).
int launch_big(const GemmK& k, int cfg, hipStream_t st);     // cfg = index into BIG_TILES
int launch_pp(const GemmK& k, hipStream_t st, int rows = 256);   // gemm_pp.hip: the ping-pong 256 x 256 / 192 x 256 tiles (BIG_TILES[PP_TILE / PP192_TILE])
int launch_pp320(const GemmK& k, hipStream_t st);            // gemm_pp320.hip: the ping-pong 256 x 320 tile (BIG_TILES[PP320_TILE])
bool pp_operands_ok(const GemmK& k, bool conv);              // operands addressable by its 31-bit buffer offsets

}  // namespace icd_gemm_detail
