// gemm_pp.hip - two co-resident workgroups per CU: GEMM / implicit-GEMM conv tile whose prologue, epilogue and memory
// stalls hide behind the OTHER workgroup's MFMAs.
//
// Why (tools/gemm_timeline.py, per-block s_memrealtime stamps of the 1-block-per-CU tiles of gemm_big.hip on a full
// chip): every block of a round runs in lockstep, so the MFMA pipes idle through each round's epilogue (8 us with GEGLU,
// 23 us with a residual: that phase moves 66 MB and is HBM-bound while the matrix cores wait) and prologue (2 - 3 us), and
// the main loop itself runs at 2.1 us per 256 x 256 x 64 k-tile instead of the 1.0 us of its MFMAs because a 2-stage ring
// keeps only one k-tile in flight against ~2 us of loaded-fabric latency.  K <= 1280 layers (20 k-tiles) lose 40 % of the
// launch to this.
//
// Design:
//   * block tile 256 x 128 x 32, FOUR waves (2 x 2), wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_f16 (128 accumulator
//     registers, the fragment economy of the 256 x 256 tile: 6 ds_read_b128 per 8 MFMAs), one wave per SIMD;
//   * 3 LDS stages of 24 KiB (BK = 32: 64-B rows) = 72 KiB, so TWO blocks are resident per CU (2 waves per SIMD, from
//     different blocks, with independent barriers): while one block stores its tile or waits for operands, the other one's
//     MFMAs own the matrix pipe.  Counted vmcnt: two k-tiles stay in flight across each barrier;
//   * 64-B LDS rows: 16-B chunk c of row r sits at physical chunk c ^ ((r >> 2) & 3) - the 16 lanes of every ds_read_b128
//     service group then touch 16 distinct (row % 4, chunk) slots of the 256-B bank row (conflict free); the swizzle is
//     applied on the global SOURCE address, the global_load_lds destination stays lane-linear;
//   * loader state, im2col gather (tap / upsample / stride / concat / zero page), XCD + L2-group tile order, epilogue and
//     split-K are those of gemm_big.hip (gemm_epilogue.h).
#include <type_traits>
#include "gemm_common.h"
#include "gemm_epilogue.h"

using namespace icd_gemm_detail;

namespace {

constexpr int enc_vmcnt(int n) { return ((n >> 4) << 14) | 0x0F70 | (n & 15); }

constexpr int BKP = 32;                          // k-tile depth (halves)
constexpr int PP_BM = 256, PP_BN = 128;
constexpr int PP_NST = 3;

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(GemmK p) {
    constexpr int TM = 4, TN = 2, WN = 2;
    constexpr int A_BYTES = PP_BM * 64, W_BYTES = PP_BN * 64, STAGE_BYTES = A_BYTES + W_BYTES;     // 16 + 8 KiB
    constexpr int NAJ = PP_BM / 64, NWJ = PP_BN / 64;            // global_load_lds per thread per stage: 4 + 2 (16 rows each)
    constexpr int LOADS = NAJ + NWJ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int wm = wv / WN, wn = wv - wm * WN;
    unsigned long long* tl = p.timeline ? p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 : nullptr;
    if (tl && tid == 0) tl[0] = __builtin_amdgcn_s_memrealtime();

    int mt, nt;
    tile_of_block(blockIdx.x, p.nbm, p.nbn, p.gm, mt, nt);
    const int m0 = mt * PP_BM, n0 = nt * PP_BN;
    const int split = blockIdx.y;
    const int nk_total = p.K / BKP;
    const int kt_begin = split * p.kt_per_split;                 // in BKP units
    const int nk = min(nk_total, kt_begin + p.kt_per_split) - kt_begin;
    const half_t* zero = reinterpret_cast<const half_t*>(icd_zero_page);

    // ---- loader state: lane -> (row within a 16-row group, physical 16-B chunk of the 64-B row) -----------------------
    const int lrow = l >> 2, pchunk = l & 3;
    const int Cin = p.C0 + p.C1;
    const int ntaps = p.ksize * p.ksize, pad = (p.flags & ICD_GEMM_PAD_HI) ? 0 : p.ksize >> 1;
    const int Hu = p.Hin << p.upsample, Wu = p.Win << p.upsample;
    const int k_begin = kt_begin * BKP;

    const half_t* a_ptr[NAJ]; int a_inc[NAJ];
    int a_pix[NAJ], a_yx[NAJ];
    const half_t* w_ptr[NWJ]; int w_inc[NWJ];
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
        const int r = (wv * NAJ + j) * 16 + lrow;
        const int lc = pchunk ^ ((r >> 2) & 3);
        const int m = m0 + r;
        const int boff = j * 512;                // the instruction's immediate offset (j KiB) also moves the source: bias it back
        a_ptr[j] = zero - boff; a_inc[j] = 0; a_pix[j] = -1; a_yx[j] = 0;
        if (m < p.M) {
            if (MODE == 0) {
                a_ptr[j] = p.a0 + (long long)m * p.lda + k_begin + lc * 8 - boff; a_inc[j] = BKP;
            } else {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / p.Wout;
                a_yx[j] = (y << 16) | (rem - y * p.Wout);
                a_pix[j] = b * p.Hin * p.Win;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
        const int r = (wv * NWJ + j) * 16 + lrow;
        const int lc = pchunk ^ ((r >> 2) & 3);
        const int n = n0 + r;
        const bool ok = n < p.Nw;
        const int boff = j * 512;
        w_ptr[j] = ok ? p.w + (long long)n * p.ldw + k_begin + lc * 8 - boff : zero - boff;
        w_inc[j] = ok ? BKP : 0;
    }
    int u_tap = MODE == 1 ? k_begin / Cin : 0;
    int u_c = MODE == 1 ? k_begin - u_tap * Cin : 0;
    bool u_recompute = true;

    const int wave_a = __builtin_amdgcn_readfirstlane(wv * NAJ * 1024);
    const int wave_w = __builtin_amdgcn_readfirstlane(A_BYTES + wv * NWJ * 1024);

#define GLDS(PTR, BASE, IMM)                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(PTR),                \
                                     (__attribute__((address_space(3))) void*)(BASE), 16, IMM, 0)

    auto issue_stage = [&](int stage_off) {
        unsigned char* sa = smem + stage_off + wave_a;
        unsigned char* sw = smem + stage_off + wave_w;
        if (MODE == 1) {
            if (u_recompute) {
                const int dy = (u_tap * 11) >> 5, dx = u_tap - dy * 3;
                const int oy = (ntaps == 9 ? dy : 0) - pad, ox = (ntaps == 9 ? dx : 0) - pad;
                const bool first = u_c < p.C0;
#pragma unroll
                for (int j = 0; j < NAJ; ++j) {
                    const int r = (wv * NAJ + j) * 16 + lrow;
                    const int lc = pchunk ^ ((r >> 2) & 3);
                    const int yu = (a_yx[j] >> 16) * p.stride + oy, xu = (a_yx[j] & 0xffff) * p.stride + ox;
                    const bool ok = a_pix[j] >= 0 && (unsigned)yu < (unsigned)Hu && (unsigned)xu < (unsigned)Wu;
                    const long long pix = a_pix[j] + (yu >> p.upsample) * p.Win + (xu >> p.upsample);
                    const half_t* s0 = first ? p.a0 + pix * p.C0 + u_c : p.a1 + pix * p.C1 + (u_c - p.C0);
                    a_ptr[j] = (ok ? s0 + lc * 8 : zero) - j * 512;
                    a_inc[j] = ok ? BKP : 0;
                }
            }
            u_c += BKP;
            u_recompute = false;
            if (u_c == Cin) { u_c = 0; ++u_tap; u_recompute = true; }
            else if (u_c == p.C0) u_recompute = true;
        }
        GLDS(a_ptr[0], sa, 0); GLDS(a_ptr[1], sa, 1024); GLDS(a_ptr[2], sa, 2048); GLDS(a_ptr[3], sa, 3072);
        GLDS(w_ptr[0], sw, 0); GLDS(w_ptr[1], sw, 1024);
#pragma unroll
        for (int j = 0; j < NAJ; ++j) a_ptr[j] += a_inc[j];
#pragma unroll
        for (int j = 0; j < NWJ; ++j) w_ptr[j] += w_inc[j];
    };
#undef GLDS

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- LDS fragment addresses (stage 0; stage offsets are added as immediates) ------------------------------------------
    const int lr = l & 31, lh = l >> 5;
    int rd_a[2], rd_w[2];
    {
        const int x = (lr >> 2) & 3;             // rows of one fragment are base + lr with base % 32 == 0: the key depends on lr only
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int off = ((s2 * 2 + lh) ^ x) << 4;
            rd_a[s2] = (wm * TM * 32 + lr) * 64 + off;
            rd_w[s2] = A_BYTES + (wn * TN * 32 + lr) * 64 + off;
        }
    }
    f16x8 af[2][TM], wf[2][TN];
    auto load_frags = [&](int stage_off, auto s_tag, auto set_tag) {
        constexpr int S2 = decltype(s_tag)::value, SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[SET][i] = *reinterpret_cast<const f16x8*>(smem + stage_off + rd_a[S2] + i * 2048);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[SET][j] = *reinterpret_cast<const f16x8*>(smem + stage_off + rd_w[S2] + j * 2048);
    };
    auto mfmas = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[SET][j], af[SET][i], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- main loop: 3-stage ring, tiles t+1 and t+2 in flight while tile t is multiplied ------------------------------------
    issue_stage(0);
    if (nk > 1) issue_stage(STAGE_BYTES);
    auto k_tile = [&](auto st_tag, int t) {
        constexpr int ST = decltype(st_tag)::value;
        // tile t has landed (this wave's part) once at most the loads of tile t+1 are outstanding
        if (t + 1 < nk) __builtin_amdgcn_s_waitcnt(enc_vmcnt(LOADS));
        else __builtin_amdgcn_s_waitcnt(enc_vmcnt(0));
        __builtin_amdgcn_s_barrier();                // everybody's part landed; everybody finished reading stage (ST + 2) % 3
        if (t + 2 < nk) issue_stage(((ST + 2) % PP_NST) * STAGE_BYTES);
        if (tl && t == 0 && tid == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
        load_frags(ST * STAGE_BYTES, I0{}, I0{});
        load_frags(ST * STAGE_BYTES, I1{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I0{});
        __builtin_amdgcn_sched_barrier(0);
        mfmas(I1{});
    };
    for (int t = 0; t < nk; t += PP_NST) {
        k_tile(I0{}, t);
        if (t + 1 < nk) k_tile(I1{}, t + 1);
        if (t + 2 < nk) k_tile(std::integral_constant<int, 2>{}, t + 2);
    }

    wave_epilogue<TM, TN>(p, acc, smem, wv, wm, wn, l, m0, n0, split, tl);
    if (tl) {
        __syncthreads();
        if (tid == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int MODE>
int launch_one(const GemmK& k, hipStream_t st) {
    constexpr int smem = PP_NST * (PP_BM + PP_BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pp_kernel<MODE>), dim3(k.nbm * k.nbn, k.ksplit, 1), dim3(256), smem, st, k);
    ICD_CHECK_LAUNCH("icd_gemm(256x128 two-per-CU tile)");
    return ICD_OK;
}

}  // namespace

namespace icd_gemm_detail {

// k.nbm / k.nbn / k.ksplit / k.kt_per_split (in 32-deep k-tiles) are set by the caller
int launch_pp(const GemmK& k, hipStream_t st) {
    const bool conv = k.ksize > 0 && k.Hout > 0;
    return conv ? launch_one<1>(k, st) : launch_one<0>(k, st);
}

}  // namespace icd_gemm_detail
