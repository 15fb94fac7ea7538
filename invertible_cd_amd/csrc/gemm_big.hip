// gemm_big.hip - high-arithmetic-intensity GEMM / implicit-GEMM conv tiles for the large layers of the UNets.
//
// Why a second tile family: with 128-wide tiles the main loop is bound by the LDS ports (fragment reads + the
// global_load_lds DMA writes) and by global->LDS bandwidth (1/64..1/85 B per flop), see DESIGN.md section 4.  Here:
//   * block tile 256 x 256 x 64, 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_f16 (128 accumulator
//     registers): 6 ds_read_b128 per 8 MFMAs (0.75 vs 1.0 for the 64 x 64 wave tile), 1/128 B of DMA per flop (vs 1/64).
//     (A one-wave-per-SIMD variant with 128 x 160 wave tiles in the 512-entry register file was tried: hipcc spills
//     hundreds of registers once the accumulators exceed the 256 AGPRs - that form needs hand-written asm.)
//   * serves every layer with N % 256 == 0: the GEGLU projections (N = 8C) and the 1280-channel convs / Linears.
//   * 2 LDS stages of 64 KiB.  The barrier of k-tile t+1 is taken in the shadow of the last sub-step's 8 queued MFMAs of
//     k-tile t: wait vmcnt -> s_barrier -> issue the loads of t+2 -> prefetch the first fragments of t+1.
//   * Loader, swizzle, epilogue and split-K are those of gemm.hip (same LDS image, same epilogue arithmetic).
#include <type_traits>
#include "gemm_common.h"
#include "gemm_epilogue.h"

using namespace icd_gemm_detail;

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Cross-attention as the epilogue of the query projection on the 256 x 256 tile (the north-star kernel, see gemm.hip's
// xattn_epilogue for the arithmetic: S^T = K q^T on the accumulators used in place as the MFMA B operand, exact softmax over
// <= 96 key slots, O^T = V^T P^T).  A block = 256 queries of ONE sample x 4 heads of 64; wave (wm, wn) owns head wn for the
// 128 queries of tile row wm (TM = 4 query tiles of 32).  Unlike the 128-wide host, K and V^T of the block's 4 heads are staged
// ONCE in the (now idle) operand stages - 96 keys x 256 dims (row stride 520 B: conflict-free 8-B fragment reads) and 256 dims x
// 96 keys (stride 208 B).  The four query tiles are first rounded to fp16 q (128 accumulator registers -> 64), then processed in
// pairs on one register-resident set of K fragments and one of V^T fragments (reading each fragment from LDS right before
// its MFMA left ~36 exposed LDS latencies per tile: 17 us of epilogue per block).  O leaves through a private LDS patch per
// wave as 128-B row segments.
constexpr int XA_K_OFF = 0, XA_K_LD = 520, XA_V_OFF = 96 * XA_K_LD, XA_V_LD = 208, XA_TABLE_OFF = XA_V_OFF + 256 * XA_V_LD;
constexpr int XA_PATCH_OFF = XA_TABLE_OFF + 5 * 256 * 8, XA_SMEM = XA_PATCH_OFF + 8 * 32 * 72 * 2;     // 49920 + 53248 + 10240 + 36864
static_assert(XA_SMEM <= 160 * 1024, "fused cross-attention: LDS budget");

// STL (round 5): live 16-key slots, ceil(keys / 16) - 5 for the 77 CLIP tokens.  The sixth slot (keys 80..95) is masked for every lane, so
// its 8 exponentials per lane and query tile (of 48), its conversions and its two P.V MFMAs (of 12) are dropped at compile time; as a
// wave-uniform run-time branch the same skip made hipcc spill 54 registers (round 2).
template <int TM, int STL = 6>
__device__ __forceinline__ void xattn_epilogue_big(const GemmK& p, f32x16 (&acc)[TM][2], unsigned char* smem, int wv, int wm, int wn,
                                                   int l, int m0, int m_lim, int n0, const float* ln_lds) {
    constexpr int KT = 3, KS = 4, ST = 6, DT = 2;
    static_assert(STL == 5 || STL == 6, "live key slots");
    const int lr = l & 31, lh = l >> 5, tid = threadIdx.x;
    const int key_lim = p.x_nk - 8 * lh;                   // key slot constant c of this lane is masked when c >= key_lim
    const int b = m0 / p.rps;                              // the rows [m0, m_lim) of a block lie inside one sample
    // ---- stage K [96 keys][256 dims] and V^T [256 dims][96 keys] of the block's 4 heads (the caller's barrier freed the stages)
    {
        const half_t* Kb = p.xk + (long long)b * p.x_nk * p.x_ldk + n0;
        const half_t* Vb = p.xvt + (long long)b * p.x_vt_bs + (long long)n0 * p.x_ldvt;
        const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 6; ++j) {                      // 96 rows x 32 chunks of 16 B, 512 threads
            const int item = j * 512 + tid, key = item >> 5, ch = item & 31;
            const f16x8 v = key < p.x_nk ? *reinterpret_cast<const f16x8*>(Kb + (long long)key * p.x_ldk + ch * 8) : z8;
            // the 520-B row pitch (conflict-free 8-B fragment reads) is only 8-byte aligned: two 8-byte stores
            half_t* dst = reinterpret_cast<half_t*>(smem + XA_K_OFF + key * XA_K_LD + ch * 16);
            *reinterpret_cast<f16x4*>(dst) = (f16x4){v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f16x4*>(dst + 4) = (f16x4){v[4], v[5], v[6], v[7]};
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {                      // 256 rows x 12 chunks of 16 B (keys 0..95)
            const int item = j * 512 + tid, row = item / 12, ch = item - row * 12;
            const f16x8 v = ch * 8 < p.x_ldvt ? *reinterpret_cast<const f16x8*>(Vb + (long long)row * p.x_ldvt + ch * 8) : z8;
            *reinterpret_cast<f16x8*>(smem + XA_V_OFF + row * XA_V_LD + ch * 16) = v;
        }
    }
    __syncthreads();
    const unsigned char* Kl = smem + XA_K_OFF + wn * 128;  // this wave's head: 64 dims = 128 B into every key row
    const unsigned char* Vl = smem + XA_V_OFF + wn * 64 * XA_V_LD;
    const int prow = (lr & 0x13) | ((lr & 4) << 1) | ((lr & 8) >> 1);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half_t* patch = reinterpret_cast<half_t*>(smem + XA_PATCH_OFF) + wv * (32 * 72);
    const int ncol = n0 + wn * 64;
    // ---- 1. every query tile -> fp16 q (LayerNorm correction, bias, softmax scale * log2 e folded in, one rounding): the 128
    //         accumulator registers become 64, which is what lets the K and V^T fragments stay in registers across two tiles ----
    f16x8 qf[TM][KS];
    {
        f32x2 lst[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mrow = m0 + (wm * TM + i) * 32;
            lst[i] = (f32x2){0.f, 1.f};
            if (p.ln_stats) {
                if (ln_lds) lst[i] = *reinterpret_cast<const f32x2*>(ln_lds + 2 * (mrow + lr - m0));
                else if (TM == 4 || mrow + lr < m_lim) lst[i] = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (long long)(mrow + lr));
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {                  // column group outermost: its bias / column sums are loaded once for 4 tiles
                const int n = ncol + j * 32 + 8 * g + 4 * lh;
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, t4 = {0.f, 0.f, 0.f, 0.f};
                if (p.ln_stats) s4 = *reinterpret_cast<const f32x4*>(p.ln_s + n);
                if (p.bias) t4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * g + e] * p.alpha;
                        v = lst[i][1] * (v - lst[i][0] * s4[e]) + t4[e];
                        qf[i][2 * j + (g >> 1)][4 * (g & 1) + e] = (half_t)(v * p.x_scale_log2);
                    }
            }
    }
    // NI query tiles (two, or the odd one left of a 96-query wave) from tile I0 on
    auto tiles = [&](auto ni_tag, auto i0_tag) {
        constexpr int NI = decltype(ni_tag)::value, i0 = decltype(i0_tag)::value;
        // ---- 2. S^T = K q^T for two query tiles on ONE set of K fragments (24 reads in flight instead of 24 exposed latencies
        //         per tile), exact softmax over the <= 96 key slots ----
        f16x8 pf[NI][ST];
        float inv[NI];
        {
            f16x8 kf[KT][KS];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const unsigned char* src = Kl + (kt * 32 + prow) * XA_K_LD + (ks * 16 + lh * 4) * 2;
                    const f16x4 lo = *reinterpret_cast<const f16x4*>(src), hi = *reinterpret_cast<const f16x4*>(src + 16);
                    kf[kt][ks] = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) {
                f32x16 sc[KT];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
                        sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][ks], qf[i0 + ii][ks], ks == 0 ? zero16 : sc[kt], 0, 0, 0);
                float mx = -INFINITY;
                // the limit is made opaque per query tile: as a loop invariant the 48 predicates (constant >= limit) were hoisted into
                // scalar-register pairs, 181 of them spilt, and every tile read them back with v_readlane_b32 (round 3)
                int kl = key_lim;
                asm volatile("" : "+v"(kl));
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (kt * 2 + (e >> 3) >= STL) continue;          // a slot no key lives in (compile time)
                        // key = kt*32 + 16*(e>>3) + 8*lh + (e&7) >= x_nk, written as (compile-time constant) >= (one per-lane limit)
                        if (kt * 32 + 16 * (e >> 3) + (e & 7) >= kl) sc[kt][e] = -INFINITY;
                        mx = fmaxf(mx, sc[kt][e]);
                    }
                {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                }
                float rs = 0.f;                          // (skipping the mask / the exponentials of key slots past the last key with
#pragma unroll                                           //  wave-uniform branches made hipcc spill 54 registers: slower)
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if (kt * 2 + (e >> 3) >= STL) continue;
                        const float pv = __builtin_amdgcn_exp2f(sc[kt][e] - mx);
                        rs += pv;
                        pf[ii][kt * 2 + (e >> 3)][e & 7] = (half_t)pv;
                    }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rs), __float_as_uint(rs), false, false);
                inv[ii] = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
            }
        }
        // ---- 3. O^T = V^T P^T (V^T fragments from LDS as they are used: holding them too makes hipcc spill); O leaves through the
        //         wave's LDS patch ----
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const int mrow = m0 + (wm * TM + i0 + ii) * 32;
            f32x16 o[DT];
#pragma unroll
            for (int st = 0; st < STL; ++st)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                {
                    const f16x8 vf = *reinterpret_cast<const f16x8*>(Vl + (dt * 32 + lr) * XA_V_LD + (st * 16 + lh * 8) * 2);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[ii][st], st == 0 ? zero16 : o[dt], 0, 0, 0);
                }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 v = {(half_t)(o[dt][4 * g] * inv[ii]), (half_t)(o[dt][4 * g + 1] * inv[ii]), (half_t)(o[dt][4 * g + 2] * inv[ii]),
                               (half_t)(o[dt][4 * g + 3] * inv[ii])};
                    *reinterpret_cast<f16x4*>(patch + lr * 72 + dt * 32 + 8 * g + 4 * lh) = v;
                }
            half_t* out = reinterpret_cast<half_t*>(p.out);
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int r = pass * 8 + (l >> 3), c8 = (l & 7) * 8;
                const f16x8 v = *reinterpret_cast<const f16x8*>(patch + r * 72 + c8);
                if (TM == 4 || mrow + r < m_lim) *reinterpret_cast<f16x8*>(out + (long long)(mrow + r) * p.ldo + ncol + c8) = v;
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    tiles(I2{}, I0{});
    if constexpr (TM == 4) tiles(I2{}, I2{});
    if constexpr (TM == 3) tiles(I1{}, I2{});
}

#ifndef ICD_CONV_CHUNK_MAJOR
#define ICD_CONV_CHUNK_MAJOR 1
#endif
constexpr bool CONV_CHUNK_MAJOR = ICD_CONV_CHUNK_MAJOR != 0;       // K order of the conv tiles (see the loader)

constexpr int enc_vmcnt(int n) { return ((n >> 4) << 14) | 0x0F70 | (n & 15); }

// LNS (round 5): the instantiation that may compute LayerNorm statistics in its main loop (p.ln_stats_w).  As a run-time flag in ONE kernel
// the statistics code put a branch and a join into every k sub-step of every dense launch, and at each join the waitcnt pass drained ALL
// outstanding ds_reads (s_waitcnt lgkmcnt(0)) - including the fragments just requested for the NEXT sub-step, i.e. the register double
// buffering was dead in every dense GEMM (8 full drains per k-tile pair against 4 in the conv kernels, which never had the branch).
template <int MODE, int WM, int WN, int TM, int TN, bool XATTN = false, bool CARRY = false, bool LNS = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_big_kernel(GemmK p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BNt = WN * TN * 32;
    constexpr int A_BYTES = BM * 128, W_BYTES = BNt * 128, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int NAJ = BM * 8 / NT, NWJ = BNt * 8 / NT;        // 16-B chunks per thread per stage
    constexpr int LOADS = NAJ + NWJ;
    // past the epilogue's staging patches (8 x 9 KiB), inside the stage buffers; the fused cross-attention epilogue has its own map
    constexpr int LN_TABLE_OFF = XATTN ? XA_TABLE_OFF : 96 * 1024;
    static_assert(XATTN || LN_TABLE_OFF + (1 + WN) * BM * 8 <= 2 * STAGE_BYTES, "LayerNorm table does not fit");
    static_assert(!XATTN || (MODE == 0 && WM == 2 && WN == 4 && TN == 2 && (TM == 4 || TM == 3)), "fused cross-attention: 256 / 192 x 256 tile, a wave per head");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int wm = wv / WN, wn = wv - wm * WN;
    unsigned long long* tl = p.timeline ? p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
    if (tl && tid == 0) tl[0] = __builtin_amdgcn_s_memrealtime();

    int mt, nt;
    tile_of_block(blockIdx.x, p.nbm, p.nbn, p.gm, mt, nt);
    int m0 = mt * BM, m_lim = p.M;
    const int n0 = nt * BNt;
    if constexpr (XATTN) {                       // m-tiles are laid out per sample: ceil(rps / BM) of them, the last one possibly partial
        const int bps = (p.rps + BM - 1) / BM, smp = mt / bps;
        m0 = smp * p.rps + (mt - smp * bps) * BM;
        m_lim = (smp + 1) * p.rps;
    }
    const int split = blockIdx.y;
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.kt_per_split;
    const int nk = min(nk_total, kt_begin + p.kt_per_split) - kt_begin;
    const half_t* zero = reinterpret_cast<const half_t*>(icd_zero_page);

    // ---- loader state (see gemm.hip: biased pointers, zero-page parking, one M0 per group of 4 chunks) ----------
    const int lrow = l >> 3, pchunk = l & 7;
    const int Cin = p.C0 + p.C1;
    const int ntaps = p.ksize * p.ksize, pad = (p.flags & ICD_GEMM_PAD_HI) ? 0 : p.ksize >> 1;
    const int ktaps = (int)(p.tapmap >> 60);     // taps iterated: ntaps, or 4 of the 9 (phase form of the upsampling conv: tap u -> tap_base + (u & 1) + 3 (u >> 1))
    const int Hu = p.Hin << p.upsample, Wu = p.Win << p.upsample;
    const int k_begin = kt_begin * BK;

    // conv A operand (round 4): LDS-DMA through a BUFFER descriptor (buffer_load_dwordx4 ... offen lds) instead of 64-bit global pointers.
    // Per 16-B chunk the loader keeps a 32-bit byte offset of the row's tap-(0,0) pixel in the source being read and the complement of a
    // 9-bit tap-validity mask (zero padding, rows >= M); per k-tile the offset to issue is (that offset + a wave-uniform tap / channel term)
    // with bit 31 set where the tap is invalid - out of the descriptor's range (sources < 2 GiB on this path, checked on the host), so the
    // DMA writes zeros: 3 VALU issues per chunk and k-tile (add, bfe, lshl_or), no zero-page pointer select, no per-tap recompute of
    // (y, x, pixel), two registers per chunk less than round 3's loader.  That economy is what makes the CHUNK-major K order affordable
    // (k-tile = chunk * taps + tap: the nine taps of a 64-channel chunk back to back, the nine uses of a [rows + halo] x 64-channel slab inside
    // nine consecutive k-tiles: L2 hits instead of 9 fabric reads per line; weights stay tap-major in memory), see profiles/r04_conv_korder.txt.
    const half_t* a_ptr[NAJ]; int a_inc[NAJ];    // dense (MODE 0)
    unsigned a_off[NAJ], a_nmsk[NAJ], a_voff[NAJ];   // conv: see above; a_nmsk bits 0..8 = tap INVALID, bits 9 / 10 = row / column parity (upsample)
    int a_pix[NAJ];                              // conv: source pixel index of tap (0, 0) (a_off for the second concat source derives from it)
    const half_t* w_ptr[NWJ]; int w_inc[NWJ];
    const int cpt = MODE == 1 ? Cin / BK : 1;    // k-tiles per tap
    int u_tap, u_c;
    if (CONV_CHUNK_MAJOR && MODE == 1) { const int ch = kt_begin / ktaps; u_tap = kt_begin - ch * ktaps; u_c = ch * BK; }
    else { u_tap = MODE == 1 ? kt_begin / cpt : 0; u_c = MODE == 1 ? (kt_begin - u_tap * cpt) * BK : 0; }
    const int w_k0 = (CONV_CHUNK_MAJOR && MODE == 1) ? u_tap * Cin + u_c : k_begin;
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
        const int r = (wv * NAJ + j) * 8 + lrow;
        const int lc = pchunk ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const int boff = (j & 3) * 512;          // bias for the instruction's immediate offset (halves)
        a_ptr[j] = zero - boff; a_inc[j] = 0; a_pix[j] = 0; a_nmsk[j] = 0x1ff; a_off[j] = 0; a_voff[j] = 0x80000000u;
        if (m < m_lim) {
            if (MODE == 0) {
                a_ptr[j] = p.a0 + (long long)m * p.lda + k_begin + lc * 8 - boff; a_inc[j] = BK;
            } else {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / p.Wout, x = rem - y * p.Wout;
                const int yu0 = y * p.stride - pad, xu0 = x * p.stride - pad;
                unsigned nm = 0x1ff;
                for (int t = 0; t < ntaps; ++t) {
                    const int dy = ntaps == 9 ? t / 3 : 0, dx = ntaps == 9 ? t - dy * 3 : 0;
                    if ((unsigned)(yu0 + dy) < (unsigned)Hu && (unsigned)(xu0 + dx) < (unsigned)Wu) nm &= ~(1u << t);
                }
                if (p.upsample) nm |= ((unsigned)(yu0 & 1) << 9) | ((unsigned)(xu0 & 1) << 10);
                a_nmsk[j] = nm;
                a_pix[j] = b * p.Hin * p.Win + (yu0 >> p.upsample) * p.Win + (xu0 >> p.upsample);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
        const int r = (wv * NWJ + j) * 8 + lrow;
        const int lc = pchunk ^ ((r >> 1) & 7);
        const int n = n0 + r;
        const bool ok = n < p.Nw;
        const int boff = (j & 3) * 512;
        w_ptr[j] = ok ? p.w + (long long)n * p.ldw + w_k0 + lc * 8 - boff : zero - boff;
        w_inc[j] = ok ? ((CONV_CHUNK_MAJOR && MODE == 1) ? -1 : BK) : 0;       // chunk-major: a mask for the per-k-tile step
    }
    // buffer descriptors of the (up to two) conv sources: base, bytes, raw 32-bit offsets, bounds-checked
    const int conv_nb = MODE == 1 ? (p.M + p.Hout * p.Wout - 1) / (p.Hout * p.Wout) : 0;
    const unsigned src_px = (unsigned)conv_nb * (unsigned)(p.Hin * p.Win);
    (void)src_px;
    // (the descriptor type exists in the device pass only: the host pass of hipcc, which needs nothing but the kernel's stub, would drop
    //  the whole template silently - stubs undefined at load time - if it met it)
#if defined(__HIP_DEVICE_COMPILE__)
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.a0), 0, MODE == 1 ? src_px * (unsigned)p.C0 * 2u : 0u, 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.a1 ? p.a1 : p.a0), 0, MODE == 1 ? src_px * (unsigned)p.C1 * 2u : 0u, 0x00020000);
#endif
    bool u_first = true;                         // which concat source a_off[] was computed for
    auto set_source = [&](bool first) {
        const int Cs = first ? p.C0 : p.C1;
#pragma unroll
        for (int j = 0; j < NAJ; ++j) {
            const int r = (wv * NAJ + j) * 8 + lrow;
            const int lc = pchunk ^ ((r >> 1) & 7);
            a_off[j] = ((unsigned)a_pix[j] * (unsigned)Cs + (unsigned)(lc * 8)) * 2u;
        }
        u_first = first;
    };
    int w_step = BK;
    // offsets of the NEXT k-tile to be issued, computed right after the loads of the current one are in flight
    auto conv_next = [&]() {
        const int t3 = (int)((p.tapmap >> (4 * u_tap)) & 15u);       // iterated tap -> tap of the 3 x 3 geometry (wave-uniform: one 64-bit scalar shift)
        const int dy = (t3 * 11) >> 5, dx = t3 - dy * 3;             // (0, 0) for a 1 x 1 conv
        const bool first = u_c < p.C0;
        if (first != u_first) set_source(first);
        const int Cs = first ? p.C0 : p.C1, cc = first ? u_c : u_c - p.C0;
        const unsigned s_tap = (unsigned)(((dy * p.Win + dx) * Cs + cc) * 2);        // wave-uniform (no upsample)
#pragma unroll
        for (int j = 0; j < NAJ; ++j) {
            unsigned off = a_off[j] + s_tap;
            if (p.upsample) {
                const int doff = (int)((((a_nmsk[j] >> 9) & 1) + dy) >> 1) * p.Win + (int)((((a_nmsk[j] >> 10) & 1) + dx) >> 1);
                off = a_off[j] + (unsigned)((doff * Cs + cc) * 2);
            }
            a_voff[j] = off | (__builtin_amdgcn_ubfe(a_nmsk[j], (unsigned)t3, 1u) << 31);
        }
        if (CONV_CHUNK_MAJOR) {                                      // k-tile -> (chunk, tap)
            w_step = Cin;
            if (++u_tap == ktaps) { u_tap = 0; u_c += BK; w_step = BK - (ktaps - 1) * Cin; }
        } else {                                                     // k-tile -> (tap, chunk)
            u_c += BK;
            if (u_c == Cin) { u_c = 0; ++u_tap; }
        }
    };
    if (MODE == 1) { set_source(u_c < p.C0); conv_next(); }          // offsets of the first k-tile

    const int wave_a = __builtin_amdgcn_readfirstlane(wv * NAJ * 1024);
    const int wave_w = __builtin_amdgcn_readfirstlane(A_BYTES + wv * NWJ * 1024);

#define GLDS(PTR, BASE, IMM)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTR),                \
                                     (__attribute__((address_space(3))) void*)(BASE), 16, IMM, 0)

    auto issue_stage = [&](int stage_off) {
        unsigned char* sa = smem + stage_off + wave_a;
        unsigned char* sw = smem + stage_off + wave_w;
        if (MODE == 1) {
            const bool first = u_first;
#pragma unroll
            for (int j = 0; j < NAJ; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (__attribute__((address_space(3))) void*)(sa + j * 1024), 16, a_voff[j], 0, 0, 0);
#endif
            }
        } else {
#pragma unroll
            for (int j = 0; j < NAJ; ++j) {
                unsigned char* base = sa + (j >> 2) * 4096;
                if ((j & 3) == 0) GLDS(a_ptr[j], base, 0);
                else if ((j & 3) == 1) GLDS(a_ptr[j], base, 1024);
                else if ((j & 3) == 2) GLDS(a_ptr[j], base, 2048);
                else GLDS(a_ptr[j], base, 3072);
                a_ptr[j] += a_inc[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NWJ; ++j) {
            unsigned char* base = sw + (j >> 2) * 4096;
            if ((j & 3) == 0) GLDS(w_ptr[j], base, 0);
            else if ((j & 3) == 1) GLDS(w_ptr[j], base, 1024);
            else if ((j & 3) == 2) GLDS(w_ptr[j], base, 2048);
            else GLDS(w_ptr[j], base, 3072);
            w_ptr[j] += (CONV_CHUNK_MAJOR && MODE == 1) ? (w_step & w_inc[j]) : w_inc[j];
        }
        if (MODE == 1) conv_next();
    };
#undef GLDS

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- LDS read addresses per stage -----------------------------------------------------------------------
    const int lr = l & 31, lh = l >> 5;
    int rd_a[2][4], rd_w[2][4];
    {
        const int x = (lr >> 1) & 7;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int off = ((s4 * 2 + lh) ^ x) << 4;
                rd_a[st][s4] = st * STAGE_BYTES + (wm * TM * 32 + lr) * 128 + off;
                rd_w[st][s4] = st * STAGE_BYTES + A_BYTES + (wn * TN * 32 + lr) * 128 + off;
            }
    }
    f16x8 af[2][TM], wf[2][TN];               // set 1 is unused (and dead-code eliminated) without DBUF
    // LayerNorm statistics of the A rows, in-kernel (p.ln_stats_w): the WN waves of a tile row see every element of their rows go
    // by as MFMA operands (K = the LayerNorm width) and share the work - wave wn sums the k sub-steps s4 with s4 % WN == wn;
    // lane (lr, lh) covers the k-columns it holds, v_dot2_f32_f16 with fp32 accumulate.  (All of it on the waves of column 0 made
    // them the block's critical path: +10 % on a 40-n-tile GEGLU launch.)
    const bool stat_on = LNS && MODE == 0 && p.ln_stats_w != nullptr;
    float st_s[TM], st_q[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    auto load_frags = [&](auto st_tag, auto s_tag, auto set_tag) {
        constexpr int ST = decltype(st_tag)::value, S4 = decltype(s_tag)::value, SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[SET][i] = *reinterpret_cast<const f16x8*>(smem + rd_a[ST][S4] + i * 4096);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[SET][j] = *reinterpret_cast<const f16x8*>(smem + rd_w[ST][S4] + j * 4096);
    };
    auto mfmas = [&](auto set_tag, auto sub_tag) {
        constexpr int SET = decltype(set_tag)::value, S4 = decltype(sub_tag)::value;
        if (stat_on && (S4 % WN) == wn) {        // (slipping the dots in between the MFMAs makes hipcc spill hundreds of registers)
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const h2 v = {af[SET][i][2 * e], af[SET][i][2 * e + 1]};
                    st_s[i] = __builtin_amdgcn_fdot2(v, one, st_s[i], false);
                    st_q[i] = __builtin_amdgcn_fdot2(v, v, st_q[i], false);
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[SET][j], af[SET][i], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // ---- prologue: tiles 0 and 1 in flight, wait for tile 0 ------------------------------------------------------
    issue_stage(0);
    if (nk > 1) { issue_stage(STAGE_BYTES); __builtin_amdgcn_s_waitcnt(enc_vmcnt(LOADS)); }
    else __builtin_amdgcn_s_waitcnt(enc_vmcnt(0));
    __builtin_amdgcn_s_barrier();
    if (tl && tid == 0) { tl[1] = __builtin_amdgcn_s_memrealtime(); tl[4] = __builtin_amdgcn_s_memtime(); }
    load_frags(I0{}, I0{}, I0{});

    // register double-buffering of the fragments only where the accumulators leave room (128 x 64 wave tile)
    constexpr bool DBUF = TM * TN <= 8;
    auto k_tile = [&](auto st_tag, int t) {
        constexpr int ST = decltype(st_tag)::value;
        using STt = std::integral_constant<int, ST>;
        using SNt = std::integral_constant<int, ST ^ 1>;
        if (DBUF) {
            load_frags(STt{}, I1{}, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{}, I0{});
            load_frags(STt{}, I2{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{}, I1{});
            load_frags(STt{}, I3{}, I1{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I0{}, I2{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(I1{}, I3{});                   // last sub-step: its MFMAs are queued on the matrix pipe ...
        } else {
            mfmas(I0{}, I0{});
            load_frags(STt{}, I1{}, I0{});
            mfmas(I0{}, I1{});
            load_frags(STt{}, I2{}, I0{});
            mfmas(I0{}, I2{});
            load_frags(STt{}, I3{}, I0{});
            mfmas(I0{}, I3{});
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) {                        // ... and in their shadow: hand over to the next k-tile
            __builtin_amdgcn_s_waitcnt(enc_vmcnt(0));          // tile t+1 has landed (this wave's part)
            __builtin_amdgcn_s_barrier();                      // everybody's part landed; everybody done reading stage ST
            if (t + 2 < nk) issue_stage(ST * STAGE_BYTES);     // tile t+2 -> the stage just released
            load_frags(SNt{}, I0{}, I0{});
        }
    };
    for (int t = 0; t < nk; t += 2) {
        k_tile(I0{}, t);
        if (t + 1 < nk) k_tile(I1{}, t + 1);
    }

    if (tl && tid == 0) tl[5] = __builtin_amdgcn_s_memtime();        // shader-clock ticks of the main loop (clock = ticks / time)
    const float* ln_lds = nullptr;
    if (stat_on) {
        // per-wave partial sums -> LDS (behind the epilogue's staging patches), summed over the WN waves of the tile row in a fixed
        // order; (mean, rstd) of the block's rows -> an LDS table for the epilogue, and to memory by n-tile 0 (a later GEMM
        // normalised by the same LayerNorm reads them there)
        float* table = reinterpret_cast<float*>(smem + LN_TABLE_OFF);
        float* parts = table + 2 * BM;                                     // [WN][BM][2]
        __syncthreads();                         // every wave is done with the stage buffers
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s_ = st_s[i] + __shfl_xor(st_s[i], 32), q_ = st_q[i] + __shfl_xor(st_q[i], 32);
            if (lh == 0) *reinterpret_cast<f32x2*>(parts + 2 * (wn * BM + (wm * TM + i) * 32 + lr)) = (f32x2){s_, q_};
        }
        __syncthreads();
        if (wn == 0 && lh == 0) {
            const float inv_k = 1.f / (float)p.K;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + lr;
                float s_ = 0.f, q_ = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(parts + 2 * (w * BM + row));
                    s_ += v[0]; q_ += v[1];
                }
                const float mean = s_ * inv_k;
                float var = fmaxf(q_ * inv_k - mean * mean, 0.f);
                // One-pass variance: E[x^2] - mean^2 loses ~ (1 + mean^2 / var) x 1e-6 of relative accuracy.  Rows whose offset
                // dominates their spread (|mean| > 4 sigma: not seen on zero-centred transformer activations, but a row is a row)
                // take the exact second pass instead - sum (x - mean)^2 in fp32 over the row, which the tile just streamed through L2.
                if (mean * mean > 16.f * var && m0 + row < m_lim) {
                    const half_t* ar = p.a0 + (long long)(m0 + row) * p.lda;
                    float acc2 = 0.f;
                    for (int kk = 0; kk < p.K; kk += 8) {
                        const f16x8 v = *reinterpret_cast<const f16x8*>(ar + kk);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float dlt = (float)v[e] - mean; acc2 = __builtin_fmaf(dlt, dlt, acc2); }
                    }
                    var = acc2 * inv_k;
                }
                const float rstd = rsqrtf(var + p.ln_eps);
                *reinterpret_cast<f32x2*>(table + 2 * row) = (f32x2){mean, rstd};
                if (nt == 0 && m0 + row < m_lim) *reinterpret_cast<f32x2*>(p.ln_stats_w + 2 * (long long)(m0 + row)) = (f32x2){mean, rstd};
            }
        }
        ln_lds = table;                          // indexed by row - m0 (the epilogue's own barrier publishes it)
    }
    // ---- epilogue (gemm_epilogue.h): per-wave LDS patches, no block-wide slabs -----------------------------------------
    // (since the fast path is specialised by operand mix, the 256 x 320 conv tile - 160 accumulators + the im2col loader state -
    //  fits it too: FAST_OK stays a template switch for experiments)
    if constexpr (XATTN) {
        if (!stat_on) __syncthreads();           // (the statistics block above already synchronised) every wave is done with the stages
        if (tl && tid == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
        // (XATTN kernels know no error carry: their CARRY flag selects the five-slot epilogue - one instantiation each, because both
        //  epilogues in ONE kernel made hipcc spill 83 registers)
        xattn_epilogue_big<TM, CARRY ? 5 : 6>(p, acc, smem, wv, wm, wn, l, m0, m_lim, n0, ln_lds);
    } else {
        wave_epilogue<TM, TN, true, CARRY>(p, acc, smem, wv, wm, wn, l, m0, n0, split, tl, ln_lds);
    }
    if (tl) {                                    // last wave out writes the end stamp (stores of this wave are issued, not drained)
        __syncthreads();
        if (tid == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// LNS (round 5): the instantiation that may compute LayerNorm statistics in its main loop (p.ln_stats_w).  As a run-time flag in ONE kernel
// the statistics code put a branch and a join into every k sub-step of every dense launch, and at each join the waitcnt pass drained ALL
// outstanding ds_reads (s_waitcnt lgkmcnt(0)) - including the fragments just requested for the NEXT sub-step, i.e. the register double
// buffering was dead in every dense GEMM (8 full drains per k-tile pair against 4 in the conv kernels, which never had the branch).
template <int MODE, int WM, int WN, int TM, int TN, bool XATTN = false, bool CARRY = false, bool LNS = false>
int launch_one(const GemmK& k, hipStream_t st) {
    constexpr int smem0 = 2 * (WM * TM * 32 + WN * TN * 32) * 128;
    constexpr int smem = XATTN && XA_SMEM > smem0 ? XA_SMEM : smem0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<MODE, WM, WN, TM, TN, XATTN, CARRY, LNS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_big_kernel<MODE, WM, WN, TM, TN, XATTN, CARRY, LNS>), dim3(k.nbm * k.nbn, k.ksplit, 1), dim3(WM * WN * 64), smem, st, k);
    ICD_CHECK_LAUNCH("icd_gemm(big tile)");
    return ICD_OK;
}

}  // namespace

namespace icd_gemm_detail {

// k.nbm / k.nbn / k.ksplit / k.kt_per_split are set by the caller for the tile of configuration `cfg` (BIG_TILES[])
int launch_big(const GemmK& k, int cfg, hipStream_t st) {
    const bool conv = k.ksize > 0 && k.Hout > 0;
    // launches with an error carry (the residual adds of the executor) have their own instantiation; with split-K the carry is the
    // reduce kernel's business and the tile kernel is the plain one
    const bool carry = (k.out_c || k.resid_c) && k.ksplit == 1;
#define ICD_BIG(WM, WN, TM, TN)                                                                                                    \
    (carry ? (conv ? launch_one<1, WM, WN, TM, TN, false, true>(k, st) : launch_one<0, WM, WN, TM, TN, false, true>(k, st))        \
           : (conv ? launch_one<1, WM, WN, TM, TN>(k, st)                                                                          \
                   : (k.ln_stats_w ? launch_one<0, WM, WN, TM, TN, false, false, true>(k, st) : launch_one<0, WM, WN, TM, TN>(k, st))))
    switch (cfg) {
    case 0: return ICD_BIG(2, 4, 4, 2);   // 256 x 256: 2 x 4 waves of 128 x 64 (GEGLU capable)
    case 1: return ICD_BIG(4, 2, 2, 5);   // 256 x 320: 4 x 2 waves of 64 x 160
    case 2: return ICD_BIG(2, 4, 3, 2);   // 192 x 256: 2 x 4 waves of 96 x 64 (GEGLU capable) - chip fill for M = 8192-class layers
    case 3: return ICD_BIG(4, 2, 1, 5);   // 128 x 320: 4 x 2 waves of 32 x 160
    case PP_TILE: return launch_pp(k, st);   // 256 x 256 with the ping-pong main loop (gemm_pp.hip)
    case PP320_TILE: return launch_pp320(k, st);   // 256 x 320 with it (gemm_pp320.hip)
    case PP192_TILE: return launch_pp(k, st, 192); // 192 x 256 with it
    }
#undef ICD_BIG
    if (cfg == 100)   // query projection + cross-attention in one launch (icd_gemm_desc.xattn_*): 256 x 256 = 256 queries x 4 heads
        return k.x_nk <= 80 ? launch_one<0, 2, 4, 4, 2, true, true, true>(k, st) : launch_one<0, 2, 4, 4, 2, true, false, true>(k, st);   // 5 / 6 live key slots
    if (cfg == 101)   // the same on 192 x 256: six m-tiles per 1024-query sample (the last one 64 rows) - 240 blocks where 160 leave 96 CUs idle
        return k.x_nk <= 80 ? launch_one<0, 2, 4, 3, 2, true, true, true>(k, st) : launch_one<0, 2, 4, 3, 2, true, false, true>(k, st);
    icd_set_error("icd_gemm: unknown big-tile configuration %d", cfg);
    return ICD_ERR_INVALID_ARG;
}

}  // namespace icd_gemm_detail
