// gemm_pp.hip - 256 x 256 x 64 GEMM tile with a ping-pong ("8-phase") main loop (round 6).
//
// gemm_big.hip runs its 8 waves in lockstep: both waves of a SIMD read their fragments at the same time and then queue their MFMAs at
// the same time, one barrier per k-tile, the DMA queue drained (vmcnt 0) before it.  Here the two wave rows of the 2 x 4 wave grid are
// two GROUPS that run one barrier apart (group 1 takes one extra s_barrier before its first phase, group 0 one after its last):
//   * a k-tile is four PHASES, one 64 x 32 quadrant of the wave's 128 x 64 tile over the whole BK = 64 each (8 MFMAs 32x32x16 on two
//     accumulators); a phase = [ds_reads of the quadrant's new fragments | one half-tile of a later k-tile sent to the DMA queue] ->
//     lgkmcnt(0) -> s_barrier -> 8 MFMAs at raised priority -> s_barrier.  While group 0 is between its two barriers (matrix pipe),
//     group 1 - its SIMD partners, wave w + 4 sits on the SIMD of wave w - is in its load section, and vice versa: the matrix pipe
//     of every SIMD always has one wave feeding it and the other wave's LDS / DMA issue costs it nothing.
//   * an operand tile (256 rows x 64 k) is staged as two HALF-tiles of 128 rows: half h of A holds rows wm * 128 + h * 64 + [0, 64) of
//     both wave rows, half h of W the columns wn * 64 + h * 32 + [0, 32) of the four wave columns - i.e. exactly what quadrant h of
//     every wave reads.  Staging order per k-tile: W0, A0 (read in phase 0), W1 (phase 1), A1 (phase 2); phase P sends half-tile
//     number P + 7 (two buffer_load ... lds of 1 KiB per wave), i.e. the slot whose last read was one (W0) or two phases ago.
//   * the DMA queue is never drained inside the loop.  Shipped schedule (V = 2): phase P reads half-tile P + 1 (8 / 4 / 8 / 4 ds_read_b128: phase 3
//     fetches W0 of the NEXT k-tile into the register set W1 has left, the two sets swap roles every k-tile), sends half-tile P + 7 and takes a
//     rolling COUNTED wait - vmcnt(10): half-tile P + 2 has landed, five stay in flight across the barriers - before the phase's first
//     barrier; what a wait covers is read one phase later.  (V = 1, ICD_GEMM_TUNE_PP_V1: reads 12 / 4 / 8 / 0, one vmcnt(6) per k-tile; within 0.5 %.)
//   * reads of a slot are retired (lgkmcnt(0)) BEFORE the phase's first barrier: the other group restages that slot right after it.
//   * 2180 shader cycles per k-tile at 8192^3 against a floor of 2048 (64 MFMAs of 32 cycles per SIMD); the lockstep tile: 2873
//     (profiles/r06_pp_timeline.txt).
// Both operands come through buffer descriptors (32-bit per-lane offsets computed once, the k offset in a scalar register): no
// per-k-tile pointer arithmetic on the VALU.  LDS image, swizzle, accumulator layout and epilogue are those of gemm_big.hip.
#include <type_traits>
#include "gemm_common.h"
#include "gemm_epilogue.h"

using namespace icd_gemm_detail;

namespace {

constexpr int PP_BN = 256;
constexpr int PP_HALF = 128 * 128;               // bytes of one half-tile (128 rows x 64 halfs)
constexpr int PP_BUF = 4 * PP_HALF;              // one k-tile: W0 | A0 | W1 | A1
constexpr int PP_SMEM = 2 * PP_BUF;

#define PP_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PP_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)

// TM = 4: 256 x 256 (wave tile 128 x 64).  TM = 3 (round 6): 192 x 256, wave tile 96 x 64 - chip fill for the M = 8192-class layers (43 x 5 = 215
// tiles where 256-row tiles make 160): A half 0 = the first 64 rows of both wave rows (two 32-row MFMA tiles), A half 1 = their last 32 rows
// (8 KiB: ONE DMA instruction per wave), phases of 8 / 8 / 4 / 4 MFMAs; rolling wait vmcnt(8) in phase 0, vmcnt(9) otherwise.
template <int MODE, bool CARRY, int V = 2, bool LNS = false, int TM = 4>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmK p) {
    static_assert(TM == 4 || (TM == 3 && V == 2), "256 or 192 rows");
    constexpr int PP_BM = TM * 64;
    constexpr int NA1 = TM == 4 ? 2 : 1;          // DMA instructions per wave for A half 1 / MFMA row tiles of A half 1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int wm = wv >> 2, wn = wv & 3;
    const int lr = l & 31, lh = l >> 5;
    unsigned long long* tl = p.timeline ? p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
    if (tl && tid == 0) tl[0] = __builtin_amdgcn_s_memrealtime();

    int mt, nt;
    tile_of_block(blockIdx.x, p.nbm, p.nbn, p.gm, mt, nt);
    const int m0 = mt * PP_BM, n0 = nt * PP_BN;
    const int split = blockIdx.y;
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.kt_per_split;
    const int nk = min(nk_total, kt_begin + p.kt_per_split) - kt_begin;
    const int nseq = 4 * nk;                     // half-tiles of this block
    const int k_begin = kt_begin * BK;

    // ---- loader: per-lane byte offsets of the 2 x 4 half-tile chunks (slot order W0, A0, W1, A1), bit 31 = outside the operand ----
    // MODE 1 (implicit-GEMM conv, gemm_big.hip's loader in this kernel's staging order): the A offsets change with every k-tile - tap and
    // 64-channel chunk of the (up to two) NHWC sources, K order chunk-major (k-tile = chunk * taps + tap), weights tap-major in memory;
    // per chunk the lane keeps the byte offset of its row's tap-(0,0) pixel and the complement of a 9-bit tap-validity mask.
    unsigned voff[4][2];
    unsigned a_off[2][2], a_nmsk[2][2];
    int a_pix[2][2], a_lc[2][2];             // (a_lc: the lane's logical 16-B chunk of that row: conv only)
    const int Cin = p.C0 + p.C1;
    const int ntaps = p.ksize * p.ksize, pad = (p.flags & ICD_GEMM_PAD_HI) ? 0 : p.ksize >> 1;
    const int ktaps = (int)(p.tapmap >> 60);
    {
        const int lrow = l >> 3, pchunk = l & 7;
        const int Hu = p.Hin << p.upsample, Wu = p.Win << p.upsample;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = (wv * 2 + j) * 8 + lrow;                   // row of the half-tile this lane fills
            const int lc = pchunk ^ ((r >> 1) & 7);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n0 + (r >> 5) * 64 + h * 32 + (r & 31);
                voff[2 * h][j] = n < p.Nw ? (unsigned)(n * p.ldw + lc * 8) * 2u : 0x80000000u;
                // A half 0: 64 rows of each wave row; half 1: the remaining 64 (TM = 4) or 32 (TM = 3: rows wv * 8 + lrow, instruction j = 0 only)
                int m, lca = lc;
                bool live = true;
                if (h == 0 || TM == 4) m = m0 + (r >> 6) * (TM * 32) + h * 64 + (r & 63);
                else {
                    const int r1 = wv * 8 + lrow;
                    lca = pchunk ^ ((r1 >> 1) & 7);
                    m = m0 + (r1 >> 5) * (TM * 32) + 64 + (r1 & 31);
                    live = j == 0;
                }
                if constexpr (MODE == 0) {
                    voff[2 * h + 1][j] = (live && m < p.M) ? (unsigned)(m * p.lda + lca * 8) * 2u : 0x80000000u;
                } else {
                    voff[2 * h + 1][j] = 0x80000000u; a_pix[h][j] = 0; a_nmsk[h][j] = 0x1ff; a_off[h][j] = 0; a_lc[h][j] = lca;
                    if (live && m < p.M) {
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
                        a_nmsk[h][j] = nm;
                        a_pix[h][j] = b * p.Hin * p.Win + (yu0 >> p.upsample) * p.Win + (xu0 >> p.upsample);
                    }
                }
            }
        }
    }
    const int conv_nb = MODE == 1 ? (p.M + p.Hout * p.Wout - 1) / (p.Hout * p.Wout) : 0;
    const unsigned src_px = (unsigned)conv_nb * (unsigned)(p.Hin * p.Win);
    (void)src_px;
#if defined(__HIP_DEVICE_COMPILE__)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.a0), 0,
                                                       MODE == 1 ? src_px * (unsigned)p.C0 * 2u : (unsigned)(((long long)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.w), 0, (unsigned)(((long long)(p.Nw - 1) * p.ldw + p.K) * 2), 0x00020000);
#endif
    // conv: state of the k-tile whose half-tiles are being sent (u_tap, u_c), advanced once per k-tile - right before its first half-tile
    int u_tap = 0, u_c = 0, w_soff = 0, src_lo = 0, src_hi = 0, src_bytes = 0;
    bool u_first = true;
    if constexpr (MODE == 1) { const int ch = kt_begin / ktaps; u_tap = kt_begin - ch * ktaps; u_c = ch * BK; }
    auto set_source = [&](bool first) {
        const int Cs = first ? p.C0 : p.C1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) a_off[h][j] = ((unsigned)a_pix[h][j] * (unsigned)Cs + (unsigned)(a_lc[h][j] * 8)) * 2u;
        u_first = first;
    };
    auto conv_advance = [&]() {                  // offsets of k-tile (u_tap, u_c) -> voff[A slots], w_soff; then step to the next k-tile
        const int t3 = (int)((p.tapmap >> (4 * u_tap)) & 15u);
        const int dy = (t3 * 11) >> 5, dx = t3 - dy * 3;
        const bool first = u_c < p.C0;
        if (first != u_first) set_source(first);
        const int Cs = first ? p.C0 : p.C1, cc = first ? u_c : u_c - p.C0;
        const unsigned s_tap = (unsigned)(((dy * p.Win + dx) * Cs + cc) * 2);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned off = a_off[h][j] + s_tap;
                if (p.upsample) {
                    const int doff = (int)((((a_nmsk[h][j] >> 9) & 1) + dy) >> 1) * p.Win + (int)((((a_nmsk[h][j] >> 10) & 1) + dx) >> 1);
                    off = a_off[h][j] + (unsigned)((doff * Cs + cc) * 2);
                }
                voff[2 * h + 1][j] = off | (__builtin_amdgcn_ubfe(a_nmsk[h][j], (unsigned)t3, 1u) << 31);
            }
        w_soff = __builtin_amdgcn_readfirstlane((u_tap * Cin + u_c) * 2);
        {   // the descriptor of this k-tile's source from PROVABLY wave-uniform words: a select between two descriptors that hipcc keeps in
            // VGPRs (the conv kernels sit at the scalar-register limit) turns every buffer_load into a waterfall loop (4 per k-tile)
            const unsigned long long b = reinterpret_cast<unsigned long long>(first ? p.a0 : p.a1);
            src_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
            src_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
            src_bytes = __builtin_amdgcn_readfirstlane((int)(src_px * (unsigned)Cs * 2u));
        }
        if (++u_tap == ktaps) { u_tap = 0; u_c += BK; }
    };
    if constexpr (MODE == 1) set_source(u_c < p.C0);
    const int wave_base = __builtin_amdgcn_readfirstlane(wv * 2048);
    // half-tile `slot` of k-tile kt -> buffer BUF (conv: slot 0, W0, is the first half-tile of a k-tile: the state steps there)
    auto stage = [&](auto buf_tag, auto slot_tag, int kt) {
        constexpr int BUF = decltype(buf_tag)::value, SLOT = decltype(slot_tag)::value;
        if constexpr (MODE == 1 && SLOT == 0) conv_advance();
        const int soff = MODE == 1 ? ((SLOT & 1) ? 0 : w_soff) : (k_begin + kt * BK) * 2;
        unsigned char* dst = smem + BUF * PP_BUF + SLOT * PP_HALF + wave_base;
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr bool TWO = SLOT != 3 || NA1 == 2;       // (A half 1 of the 192-row tile is 64 rows: one instruction per wave)
        unsigned char* dst1 = SLOT == 3 && NA1 == 1 ? smem + BUF * PP_BUF + SLOT * PP_HALF + (wave_base >> 1) : dst;
        if constexpr (MODE == 1 && (SLOT & 1)) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<half_t*>(((unsigned long long)(unsigned)src_hi << 32) | (unsigned long long)(unsigned)src_lo), 0, (unsigned)src_bytes, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst1, 16, voff[SLOT][0], 0, 0, 0);
            if constexpr (TWO) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voff[SLOT][1], 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds((SLOT & 1) ? rsA : rsW, (__attribute__((address_space(3))) void*)dst1, 16, voff[SLOT][0], soff, 0, 0);
            if constexpr (TWO) __builtin_amdgcn_raw_ptr_buffer_load_lds((SLOT & 1) ? rsA : rsW, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voff[SLOT][1], soff, 0, 0);
        }
#endif
    };

    // ---- fragment addresses (byte offset inside a half-tile, per k sub-step) ----
    int rd_a[4], rd_a1[4], rd_w[4];
    {
        const int x = (lr >> 1) & 7;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int off = ((s4 * 2 + lh) ^ x) << 4;
            rd_a[s4] = (wm * 64 + lr) * 128 + off;
            rd_a1[s4] = TM == 4 ? rd_a[s4] : (wm * 32 + lr) * 128 + off;          // A half 1 of the 192-row tile: 32 rows per wave row
            rd_w[s4] = (wn * 32 + lr) * 128 + off;
        }
    }
    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f16x8 af[2][4], wf[2][4];
    // LayerNorm statistics of the A rows from the operand fragments (p.ln_stats_w, see gemm_big.hip): wave wn sums k sub-step s4 == wn of
    // its rows; here the sums are taken at the head of the load section that follows the A half's matrix sections (the fragments are
    // still in registers, the wave has nothing else to issue while its partner holds the matrix pipe)
    const bool stat_on = LNS && MODE == 0 && p.ln_stats_w != nullptr;
    float st_s[TM], st_q[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    auto stats_of = [&](auto ha_tag) {
        constexpr int HA = decltype(ha_tag)::value;
        if constexpr (LNS) {
            if (stat_on) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    if (s4 != wn) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int ii = 0; ii < (HA == 0 ? 2 : NA1); ++ii) {
                            const h2 v = {af[ii][s4][2 * e], af[ii][s4][2 * e + 1]};
                            st_s[2 * HA + ii] = __builtin_amdgcn_fdot2(v, one, st_s[2 * HA + ii], false);
                            st_q[2 * HA + ii] = __builtin_amdgcn_fdot2(v, v, st_q[2 * HA + ii], false);
                        }
                }
            }
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    auto read_a = [&](auto buf_tag, auto h_tag) {
        constexpr int BUF = decltype(buf_tag)::value, H = decltype(h_tag)::value;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int ii = 0; ii < (H == 0 ? 2 : NA1); ++ii)
                af[ii][s4] = *reinterpret_cast<const f16x8*>(smem + BUF * PP_BUF + (2 * H + 1) * PP_HALF + (H == 0 ? rd_a[s4] : rd_a1[s4]) + ii * 4096);
    };
    // W half H of buffer BUF -> register set R
    auto read_w = [&](auto buf_tag, auto h_tag, auto r_tag) {
        constexpr int BUF = decltype(buf_tag)::value, H = decltype(h_tag)::value, R = decltype(r_tag)::value;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            wf[R][s4] = *reinterpret_cast<const f16x8*>(smem + BUF * PP_BUF + (2 * H) * PP_HALF + rd_w[s4]);
    };
    // quadrant (A half HA, W half HB) with the W fragments of register set R
    auto quadrant = [&](auto ha_tag, auto hb_tag, auto r_tag) {
        constexpr int HA = decltype(ha_tag)::value, HB = decltype(hb_tag)::value, R = decltype(r_tag)::value;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int ii = 0; ii < (HA == 0 ? 2 : NA1); ++ii)
                acc[2 * HA + ii][HB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[R][s4], af[ii][s4], acc[2 * HA + ii][HB], 0, 0, 0);
    };
    auto matrix_section = [&](auto ha_tag, auto hb_tag, auto r_tag) {
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
        __builtin_amdgcn_s_setprio(1);
        quadrant(ha_tag, hb_tag, r_tag);
        __builtin_amdgcn_s_setprio(0);
        PP_FENCE();
        __builtin_amdgcn_s_barrier();
        PP_FENCE();
    };

    if constexpr (V == 1) {
    // ---- V1: reads 12 / 4 / 8 / 0 per phase, one counted wait per k-tile (vmcnt(6) in phase 3) ---------------------------------
    stage(I0{}, I0{}, 0); stage(I0{}, I1{}, 0); stage(I0{}, I2{}, 0); stage(I0{}, I3{}, 0);
    if (nk > 1) {
        stage(I1{}, I0{}, 1); stage(I1{}, I1{}, 1); stage(I1{}, I2{}, 1);
        PP_WAIT_VM(6);
    } else {
        PP_WAIT_VM(0);
    }
    PP_FENCE();
    __builtin_amdgcn_s_barrier();
    PP_FENCE();
    if (tl && tid == 0) { tl[1] = __builtin_amdgcn_s_memrealtime(); tl[4] = __builtin_amdgcn_s_memtime(); }
    if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind
    PP_FENCE();
    auto phase = [&](auto buf_tag, auto p_tag, int t) {
        constexpr int BUF = decltype(buf_tag)::value, P = decltype(p_tag)::value;
        using Bt = std::integral_constant<int, BUF>;
        using Bo = std::integral_constant<int, BUF ^ 1>;
        if constexpr (P == 0) { read_w(Bt{}, I0{}, I0{}); PP_FENCE(); read_a(Bt{}, I0{}); }
        if constexpr (P == 1) read_w(Bt{}, I1{}, I1{});
        if constexpr (P == 2) read_a(Bt{}, I1{});
        PP_FENCE();
        const int seq = 4 * t + P + 7;
        if (seq < nseq) {
            if constexpr (P == 0) stage(Bo{}, I3{}, t + 1);
            if constexpr (P == 1) stage(Bt{}, I0{}, t + 2);
            if constexpr (P == 2) stage(Bt{}, I1{}, t + 2);
            if constexpr (P == 3) stage(Bt{}, I2{}, t + 2);
        }
        if constexpr (P == 3) {                  // k-tile t + 1 has landed (this wave's part); <= 3 half-tiles stay in flight
            if (t + 2 < nk) PP_WAIT_VM(6);
            else PP_WAIT_VM(0);
        }
        PP_WAIT_LGKM0();
        if constexpr (P == 0) matrix_section(I0{}, I0{}, I0{});
        if constexpr (P == 1) matrix_section(I0{}, I1{}, I1{});
        if constexpr (P == 2) matrix_section(I1{}, I1{}, I1{});
        if constexpr (P == 3) matrix_section(I1{}, I0{}, I0{});
    };
    auto k_tile = [&](auto buf_tag, int t) {
        phase(buf_tag, I0{}, t);
        phase(buf_tag, I1{}, t);
        phase(buf_tag, I2{}, t);
        phase(buf_tag, I3{}, t);
    };
    for (int t = 0; t < nk; t += 2) {
        k_tile(I0{}, t);
        if (t + 1 < nk) k_tile(I1{}, t + 1);
    }
    } else {
    // ---- V2: reads 8 / 4 / 8 / 4 per phase (phase 3 fetches W0 of the NEXT k-tile into the register set W1 has left: the two sets swap
    //      roles every k-tile), a rolling counted wait: phase P reads half-tile P + 1, sends P + 7 and waits until P + 2 has landed
    //      (vmcnt(10): five half-tiles in flight across the barriers) ------------------------------------------------------------
    stage(I0{}, I0{}, 0); stage(I0{}, I1{}, 0); stage(I0{}, I2{}, 0); stage(I0{}, I3{}, 0);
    if (nk > 1) {
        stage(I1{}, I0{}, 1); stage(I1{}, I1{}, 1); stage(I1{}, I2{}, 1);
        if constexpr (TM == 4) PP_WAIT_VM(10);   // W0, A0 of k-tile 0 have landed (14 instructions sent)
        else PP_WAIT_VM(9);                      // (13 sent: A half 1 is one instruction)
    } else {
        if constexpr (TM == 4) PP_WAIT_VM(4);
        else PP_WAIT_VM(3);
    }
    PP_FENCE();
    __builtin_amdgcn_s_barrier();
    PP_FENCE();
    if (tl && tid == 0) { tl[1] = __builtin_amdgcn_s_memrealtime(); tl[4] = __builtin_amdgcn_s_memtime(); }
    read_w(I0{}, I0{}, I0{});
    if (wm == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind
    PP_FENCE();
    auto tail_wait = [&](int rem) {              // rem half-tiles may stay in flight (the last phases of the block)
        if constexpr (TM == 4) {
            if (rem >= 4) PP_WAIT_VM(8);
            else if (rem == 3) PP_WAIT_VM(6);
            else if (rem == 2) PP_WAIT_VM(4);
            else if (rem == 1) PP_WAIT_VM(2);
            else PP_WAIT_VM(0);
        } else {                                 // the newest half-tiles of a block are A1 (1 instruction), W1, A0, W0 (2 each)
            if (rem >= 4) PP_WAIT_VM(7);
            else if (rem == 3) PP_WAIT_VM(5);
            else if (rem == 2) PP_WAIT_VM(3);
            else if (rem == 1) PP_WAIT_VM(1);
            else PP_WAIT_VM(0);
        }
    };
    auto phase = [&](auto buf_tag, auto p_tag, int t) {
        constexpr int BUF = decltype(buf_tag)::value, P = decltype(p_tag)::value;
        using Bt = std::integral_constant<int, BUF>;
        using Bo = std::integral_constant<int, BUF ^ 1>;
        using R0 = std::integral_constant<int, BUF>;          // register set of this k-tile's W0
        using R1 = std::integral_constant<int, BUF ^ 1>;      //                               W1 (and of the next k-tile's W0)
        if constexpr (P == 0) read_a(Bt{}, I0{});
        if constexpr (P == 1) { read_w(Bt{}, I1{}, R1{}); stats_of(I0{}); }
        if constexpr (P == 2) read_a(Bt{}, I1{});
        if constexpr (P == 3) { if (t + 1 < nk) read_w(Bo{}, I0{}, R1{}); stats_of(I1{}); }
        PP_FENCE();
        const int gp = 4 * t + P;                // phase number of the block
        if (gp + 7 < nseq) {
            if constexpr (P == 0) stage(Bo{}, I3{}, t + 1);
            if constexpr (P == 1) stage(Bt{}, I0{}, t + 2);
            if constexpr (P == 2) stage(Bt{}, I1{}, t + 2);
            if constexpr (P == 3) stage(Bt{}, I2{}, t + 2);
            // half-tiles P + 3 .. P + 7 may stay in flight: 10 instructions (TM = 4); TM = 3: 8 when that window starts at an A half 1, else 9
            if constexpr (TM == 4) PP_WAIT_VM(10);
            else if constexpr (P == 0) PP_WAIT_VM(8);
            else PP_WAIT_VM(9);
        } else {
            tail_wait(nseq - 3 - gp);
        }
        PP_WAIT_LGKM0();
        if constexpr (P == 0) matrix_section(I0{}, I0{}, R0{});
        if constexpr (P == 1) matrix_section(I0{}, I1{}, R1{});
        if constexpr (P == 2) matrix_section(I1{}, I1{}, R1{});
        if constexpr (P == 3) matrix_section(I1{}, I0{}, R0{});
    };
    auto k_tile = [&](auto buf_tag, int t) {
        phase(buf_tag, I0{}, t);
        phase(buf_tag, I1{}, t);
        phase(buf_tag, I2{}, t);
        phase(buf_tag, I3{}, t);
    };
    for (int t = 0; t < nk; t += 2) {
        k_tile(I0{}, t);
        if (t + 1 < nk) k_tile(I1{}, t + 1);
    }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();   // group 0 waits for group 1's last phase
    PP_FENCE();
    if (tl && tid == 0) tl[5] = __builtin_amdgcn_s_memtime();

    const float* ln_lds = nullptr;
    if constexpr (LNS) {
        if (stat_on) {                               // as gemm_big.hip: per-wave sums -> LDS, (mean, rstd) table for the epilogue, stored by n-tile 0
            constexpr int LN_TABLE_OFF = 96 * 1024;
            static_assert(LN_TABLE_OFF + 5 * PP_BM * 8 <= PP_SMEM, "LayerNorm table does not fit");
            float* table = reinterpret_cast<float*>(smem + LN_TABLE_OFF);
            float* parts = table + 2 * PP_BM;        // [4][BM][2]
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float s_ = st_s[i] + __shfl_xor(st_s[i], 32), q_ = st_q[i] + __shfl_xor(st_q[i], 32);
                if (lh == 0) *reinterpret_cast<f32x2*>(parts + 2 * (wn * PP_BM + (wm * TM + i) * 32 + lr)) = (f32x2){s_, q_};
            }
            __syncthreads();
            if (wn == 0 && lh == 0) {
                const float inv_k = 1.f / (float)p.K;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = (wm * TM + i) * 32 + lr;
                    float s_ = 0.f, q_ = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const f32x2 v = *reinterpret_cast<const f32x2*>(parts + 2 * (w * PP_BM + row));
                        s_ += v[0]; q_ += v[1];
                    }
                    const float mean = s_ * inv_k;
                    float var = fmaxf(q_ * inv_k - mean * mean, 0.f);
                    if (mean * mean > 16.f * var && m0 + row < p.M) {      // offset-dominated row: exact second pass (gemm_big.hip)
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
                    if (nt == 0 && m0 + row < p.M) *reinterpret_cast<f32x2*>(p.ln_stats_w + 2 * (long long)(m0 + row)) = (f32x2){mean, rstd};
                }
            }
            ln_lds = table;
        }
    }
    wave_epilogue<TM, 2, true, CARRY>(p, acc, smem, wv, wm, wn, l, m0, n0, split, tl, ln_lds);
    if (tl) {
        __syncthreads();
        if (tid == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int MODE, bool CARRY, int V = 2, bool LNS = false, int TM = 4>
int launch_pp_one(const GemmK& k, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<MODE, CARRY, V, LNS, TM>), hipFuncAttributeMaxDynamicSharedMemorySize, PP_SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pp_kernel<MODE, CARRY, V, LNS, TM>), dim3(k.nbm * k.nbn, k.ksplit, 1), dim3(512), PP_SMEM, st, k);
    ICD_CHECK_LAUNCH("icd_gemm(ping-pong tile)");
    return ICD_OK;
}

}  // namespace

namespace icd_gemm_detail {

// the operands must be addressable with 31-bit byte offsets (bit 31 marks rows outside the operand)
bool pp_operands_ok(const GemmK& k, bool conv) {
    if (conv) {              // conv: the sources are checked by the planner (conv_fast); the weights here
        const long long w_bytes = ((long long)(k.Nw - 1) * k.ldw + k.K) * 2;
        return w_bytes < (1LL << 31) - 4096;
    }
    const long long a_bytes = ((long long)(k.M - 1) * k.lda + k.K) * 2, w_bytes = ((long long)(k.Nw - 1) * k.ldw + k.K) * 2;
    return a_bytes < (1LL << 31) - 4096 && w_bytes < (1LL << 31) - 4096 && k.lda >= 0 && k.ldw >= 0;
}

int launch_pp(const GemmK& k, hipStream_t st, int rows) {
    const bool carry = (k.out_c || k.resid_c) && k.ksplit == 1;
    const bool conv = k.ksize > 0 && k.Hout > 0;
    if (rows == 192) {
        if (conv) return carry ? launch_pp_one<1, true, 2, false, 3>(k, st) : launch_pp_one<1, false, 2, false, 3>(k, st);
        if (k.ln_stats_w) return launch_pp_one<0, false, 2, true, 3>(k, st);
        return carry ? launch_pp_one<0, true, 2, false, 3>(k, st) : launch_pp_one<0, false, 2, false, 3>(k, st);
    }
    if (conv) return carry ? launch_pp_one<1, true>(k, st) : launch_pp_one<1, false>(k, st);
    if ((k.flags & ICD_GEMM_TUNE_PP_V1) && !carry && !k.ln_stats_w) return launch_pp_one<0, false, 1>(k, st);      // A/B: the first schedule
    if (k.ln_stats_w) return launch_pp_one<0, false, 2, true>(k, st);        // (the planner never combines inline statistics with a carry)
    return carry ? launch_pp_one<0, true>(k, st) : launch_pp_one<0, false>(k, st);
}

}  // namespace icd_gemm_detail
