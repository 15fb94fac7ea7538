// gemm_pp320.hip - the 256 x 320 x 64 tile (4 x 2 waves of 64 x 160) with the ping-pong main loop of gemm_pp.hip (round 6).
//
// The 320-wide tile is what the planner takes for every layer with N = 320 / 640 (and several with N = 1280 / 2560): the SD1.5 convs at
// 64^2 / 32^2, the C = 320 / 640 projections - about a third of a step.  Same structure as gemm_pp.hip - two wave groups (wave rows 0-1 and
// 2-3; wave w + 4 shares the SIMD of wave w) one barrier apart, a phase = [fragment reads | DMA issue] -> lgkmcnt(0) -> s_barrier -> 8
// MFMAs at raised priority -> s_barrier - with this tile's geometry:
//   * a k-tile is FIVE phases, phase j = column tile j (32 columns) of the wave's 64 x 160 tile over the whole BK: 2 x 4 MFMAs on the
//     accumulators acc[0..1][j]; the wave's A fragments (2 x 4 ds_read_b128) are read once per k-tile in phase 0, the W fragments of
//     column tile j (4 reads) in phase j;
//   * staging units: A-g (the 128 rows of wave group g, 16 KiB = 2 DMA instructions per wave) and W-j (column tile j of both wave columns,
//     64 rows = 8 KiB = 1 per wave); a k-tile is A-0, A-1, W-0 .. W-4 = 9 instructions per wave, 72 KiB; two buffers (144 KiB);
//   * the phases of k-tile t send the units of k-tile t + 1 (2, 2, 2, 2, 1 instructions), every phase takes a COUNTED wait sized so that
//     what the NEXT phase reads has landed (vmcnt 5, 6, 7, 8, 4: about one k-tile stays in flight across the barriers; never 0 before the tail).
#include <type_traits>
#include "gemm_common.h"
#include "gemm_epilogue.h"

using namespace icd_gemm_detail;

namespace {

constexpr int P3_BM = 256, P3_BN = 320;
constexpr int P3_A = 128 * 128;                  // bytes of one A unit (128 rows x 64 halfs)
constexpr int P3_W = 64 * 128;                   //              W unit (64 rows)
constexpr int P3_BUF = 2 * P3_A + 5 * P3_W;      // 73728
constexpr int P3_SMEM = 2 * P3_BUF;              // 147456

#define P3_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define P3_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define P3_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE, bool CARRY, bool LNS = false>
__global__ __launch_bounds__(512, 2) void gemm_pp320_kernel(GemmK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int wm = wv >> 1, wn = wv & 1;
    const int grp = wv >> 2;                     // wave group = wave rows {0, 1} / {2, 3}
    const int lr = l & 31, lh = l >> 5;
    unsigned long long* tl = p.timeline ? p.timeline + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
    if (tl && tid == 0) tl[0] = __builtin_amdgcn_s_memrealtime();

    int mt, nt;
    tile_of_block(blockIdx.x, p.nbm, p.nbn, p.gm, mt, nt);
    const int m0 = mt * P3_BM, n0 = nt * P3_BN;
    const int split = blockIdx.y;
    const int nk_total = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.kt_per_split;
    const int nk = min(nk_total, kt_begin + p.kt_per_split) - kt_begin;
    const int k_begin = kt_begin * BK;

    // ---- loader: per-lane byte offsets (bit 31 = outside the operand); MODE 1: gemm_pp.hip's conv loader ----
    unsigned voff_a[2][2], voff_w[5];
    unsigned a_off[2][2], a_nmsk[2][2];
    int a_pix[2][2];
    const int Cin = p.C0 + p.C1;
    const int ntaps = p.ksize * p.ksize, pad = (p.flags & ICD_GEMM_PAD_HI) ? 0 : p.ksize >> 1;
    const int ktaps = (int)(p.tapmap >> 60);
    {
        const int lrow = l >> 3, pchunk = l & 7;
        const int Hu = p.Hin << p.upsample, Wu = p.Win << p.upsample;
        {
            const int r = wv * 8 + lrow;                             // row of a W unit this lane fills (0 .. 63)
            const int lc = pchunk ^ ((r >> 1) & 7);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = n0 + (r >> 5) * 160 + j * 32 + (r & 31);
                voff_w[j] = n < p.Nw ? (unsigned)(n * p.ldw + lc * 8) * 2u : 0x80000000u;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int r = (wv * 2 + jj) * 8 + lrow;                  // row of an A unit (0 .. 127)
            const int lc = pchunk ^ ((r >> 1) & 7);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int m = m0 + g * 128 + r;
                if constexpr (MODE == 0) {
                    voff_a[g][jj] = m < p.M ? (unsigned)(m * p.lda + lc * 8) * 2u : 0x80000000u;
                } else {
                    voff_a[g][jj] = 0x80000000u; a_pix[g][jj] = 0; a_nmsk[g][jj] = 0x1ff; a_off[g][jj] = 0;
                    if (m < p.M) {
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
                        a_nmsk[g][jj] = nm;
                        a_pix[g][jj] = b * p.Hin * p.Win + (yu0 >> p.upsample) * p.Win + (xu0 >> p.upsample);
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
    int u_tap = 0, u_c = 0, w_soff = 0, src_lo = 0, src_hi = 0, src_bytes = 0;
    bool u_first = true;
    if constexpr (MODE == 1) { const int ch = kt_begin / ktaps; u_tap = kt_begin - ch * ktaps; u_c = ch * BK; }
    auto set_source = [&](bool first) {
        const int Cs = first ? p.C0 : p.C1;
        const int lrow = l >> 3, pchunk = l & 7;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int r = (wv * 2 + jj) * 8 + lrow;
            const int lc = pchunk ^ ((r >> 1) & 7);
#pragma unroll
            for (int g = 0; g < 2; ++g) a_off[g][jj] = ((unsigned)a_pix[g][jj] * (unsigned)Cs + (unsigned)(lc * 8)) * 2u;
        }
        u_first = first;
    };
    auto conv_advance = [&]() {                  // offsets of k-tile (u_tap, u_c) -> voff_a, w_soff, the source's descriptor words; then step
        const int t3 = (int)((p.tapmap >> (4 * u_tap)) & 15u);
        const int dy = (t3 * 11) >> 5, dx = t3 - dy * 3;
        const bool first = u_c < p.C0;
        if (first != u_first) set_source(first);
        const int Cs = first ? p.C0 : p.C1, cc = first ? u_c : u_c - p.C0;
        const unsigned s_tap = (unsigned)(((dy * p.Win + dx) * Cs + cc) * 2);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                unsigned off = a_off[g][jj] + s_tap;
                if (p.upsample) {
                    const int doff = (int)((((a_nmsk[g][jj] >> 9) & 1) + dy) >> 1) * p.Win + (int)((((a_nmsk[g][jj] >> 10) & 1) + dx) >> 1);
                    off = a_off[g][jj] + (unsigned)((doff * Cs + cc) * 2);
                }
                voff_a[g][jj] = off | (__builtin_amdgcn_ubfe(a_nmsk[g][jj], (unsigned)t3, 1u) << 31);
            }
        w_soff = __builtin_amdgcn_readfirstlane((u_tap * Cin + u_c) * 2);
        {   // (provably wave-uniform descriptor words: see gemm_pp.hip)
            const unsigned long long b = reinterpret_cast<unsigned long long>(first ? p.a0 : p.a1);
            src_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
            src_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
            src_bytes = __builtin_amdgcn_readfirstlane((int)(src_px * (unsigned)Cs * 2u));
        }
        if (++u_tap == ktaps) { u_tap = 0; u_c += BK; }
    };
    if constexpr (MODE == 1) set_source(u_c < p.C0);
    const int wave_a = __builtin_amdgcn_readfirstlane(wv * 2048);
    const int wave_w = __builtin_amdgcn_readfirstlane(wv * 1024);
    // A unit G of k-tile kt -> buffer BUF (conv: A-0 is the first unit of a k-tile: the state steps there)
    auto stage_a = [&](auto buf_tag, auto g_tag, int kt) {
        constexpr int BUF = decltype(buf_tag)::value, G = decltype(g_tag)::value;
        if constexpr (MODE == 1 && G == 0) conv_advance();
        unsigned char* dst = smem + BUF * P3_BUF + G * P3_A + wave_a;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MODE == 1) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<half_t*>(((unsigned long long)(unsigned)src_hi << 32) | (unsigned long long)(unsigned)src_lo), 0, (unsigned)src_bytes, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff_a[G][0], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voff_a[G][1], 0, 0, 0);
        } else {
            const int soff = (k_begin + kt * BK) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)dst, 16, voff_a[G][0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + 1024), 16, voff_a[G][1], soff, 0, 0);
        }
#endif
    };
    auto stage_w = [&](auto buf_tag, auto j_tag, int kt) {
        constexpr int BUF = decltype(buf_tag)::value, J = decltype(j_tag)::value;
        const int soff = MODE == 1 ? w_soff : (k_begin + kt * BK) * 2;
        unsigned char* dst = smem + BUF * P3_BUF + 2 * P3_A + J * P3_W + wave_w;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)dst, 16, voff_w[J], soff, 0, 0);
#endif
    };

    // ---- fragment addresses ----
    int rd_a[4], rd_w[4];
    {
        const int x = (lr >> 1) & 7;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int off = ((s4 * 2 + lh) ^ x) << 4;
            rd_a[s4] = grp * P3_A + ((wm & 1) * 64 + lr) * 128 + off;
            rd_w[s4] = 2 * P3_A + (wn * 32 + lr) * 128 + off;
        }
    }
    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f16x8 af[2][4], wf[4];
    const bool stat_on = LNS && MODE == 0 && p.ln_stats_w != nullptr;
    float st_s[2], st_q[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }
    auto stats = [&]() {                         // wave wn sums the k sub-steps s4 with s4 % 2 == wn of its rows (gemm_big.hip)
        if constexpr (LNS) {
            if (stat_on) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    if ((s4 & 1) != wn) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const h2 v = {af[i][s4][2 * e], af[i][s4][2 * e + 1]};
                            st_s[i] = __builtin_amdgcn_fdot2(v, one, st_s[i], false);
                            st_q[i] = __builtin_amdgcn_fdot2(v, v, st_q[i], false);
                        }
                }
            }
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;

    auto read_a = [&](auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i][s4] = *reinterpret_cast<const f16x8*>(smem + BUF * P3_BUF + rd_a[s4] + i * 4096);
    };
    auto read_w = [&](auto buf_tag, auto j_tag) {
        constexpr int BUF = decltype(buf_tag)::value, J = decltype(j_tag)::value;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            wf[s4] = *reinterpret_cast<const f16x8*>(smem + BUF * P3_BUF + J * P3_W + rd_w[s4]);
    };
    auto matrix_section = [&](auto j_tag) {
        constexpr int J = decltype(j_tag)::value;
        P3_FENCE();
        __builtin_amdgcn_s_barrier();
        P3_FENCE();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s4], af[i][s4], acc[i][J], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P3_FENCE();
        __builtin_amdgcn_s_barrier();
        P3_FENCE();
    };

    // ---- prologue: k-tile 0 in flight, A + W-0 landed --------------------------------------------------------------------------
    stage_a(I0{}, I0{}, 0); stage_a(I0{}, I1{}, 0);
    stage_w(I0{}, I0{}, 0); stage_w(I0{}, I1{}, 0); stage_w(I0{}, I2{}, 0); stage_w(I0{}, I3{}, 0); stage_w(I0{}, I4{}, 0);
    P3_WAIT_VM(4);
    P3_FENCE();
    __builtin_amdgcn_s_barrier();
    P3_FENCE();
    if (tl && tid == 0) { tl[1] = __builtin_amdgcn_s_memrealtime(); tl[4] = __builtin_amdgcn_s_memtime(); }
    if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind
    P3_FENCE();

    // phase J of k-tile t (buffer BUF); `more`: k-tile t + 1 exists (its units are sent by these phases)
    auto phase = [&](auto buf_tag, auto j_tag, int t, bool more) {
        constexpr int BUF = decltype(buf_tag)::value, J = decltype(j_tag)::value;
        using Bt = std::integral_constant<int, BUF>;
        using Bo = std::integral_constant<int, BUF ^ 1>;
        if constexpr (J == 0) { read_w(Bt{}, I0{}); P3_FENCE(); read_a(Bt{}); }
        else read_w(Bt{}, j_tag);
        if constexpr (J == 1) stats();
        P3_FENCE();
        if (more) {
            if constexpr (J == 0) { stage_a(Bo{}, I0{}, t + 1); P3_WAIT_VM(5); }
            if constexpr (J == 1) { stage_a(Bo{}, I1{}, t + 1); P3_WAIT_VM(6); }
            if constexpr (J == 2) { stage_w(Bo{}, I0{}, t + 1); stage_w(Bo{}, I1{}, t + 1); P3_WAIT_VM(7); }
            if constexpr (J == 3) { stage_w(Bo{}, I2{}, t + 1); stage_w(Bo{}, I3{}, t + 1); P3_WAIT_VM(8); }
            if constexpr (J == 4) { stage_w(Bo{}, I4{}, t + 1); P3_WAIT_VM(4); }
        } else {                                 // last k-tile: W-(J+1) has landed
            if constexpr (J == 0) P3_WAIT_VM(3);
            if constexpr (J == 1) P3_WAIT_VM(2);
            if constexpr (J == 2) P3_WAIT_VM(1);
            if constexpr (J == 3) P3_WAIT_VM(0);
        }
        P3_WAIT_LGKM0();
        matrix_section(j_tag);
    };
    auto k_tile = [&](auto buf_tag, int t) {
        const bool more = t + 1 < nk;
        phase(buf_tag, I0{}, t, more);
        phase(buf_tag, I1{}, t, more);
        phase(buf_tag, I2{}, t, more);
        phase(buf_tag, I3{}, t, more);
        phase(buf_tag, I4{}, t, more);
    };
    for (int t = 0; t < nk; t += 2) {
        k_tile(I0{}, t);
        if (t + 1 < nk) k_tile(I1{}, t + 1);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();  // group 0 waits for group 1's last phase
    P3_FENCE();
    if (tl && tid == 0) tl[5] = __builtin_amdgcn_s_memtime();

    const float* ln_lds = nullptr;
    if constexpr (LNS) {
        if (stat_on) {                               // as gemm_big.hip: per-wave sums -> LDS, (mean, rstd) table for the epilogue, stored by n-tile 0
            constexpr int LN_TABLE_OFF = 96 * 1024;
            static_assert(LN_TABLE_OFF + 3 * P3_BM * 8 <= P3_SMEM, "LayerNorm table does not fit");
            float* table = reinterpret_cast<float*>(smem + LN_TABLE_OFF);
            float* parts = table + 2 * P3_BM;        // [2][BM][2]
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float s_ = st_s[i] + __shfl_xor(st_s[i], 32), q_ = st_q[i] + __shfl_xor(st_q[i], 32);
                if (lh == 0) *reinterpret_cast<f32x2*>(parts + 2 * (wn * P3_BM + (wm * 2 + i) * 32 + lr)) = (f32x2){s_, q_};
            }
            __syncthreads();
            if (wn == 0 && lh == 0) {
                const float inv_k = 1.f / (float)p.K;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = (wm * 2 + i) * 32 + lr;
                    const f32x2 v0 = *reinterpret_cast<const f32x2*>(parts + 2 * row), v1 = *reinterpret_cast<const f32x2*>(parts + 2 * (P3_BM + row));
                    const float s_ = v0[0] + v1[0], q_ = v0[1] + v1[1];
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
    wave_epilogue<2, 5, true, CARRY>(p, acc, smem, wv, wm, wn, l, m0, n0, split, tl, ln_lds);
    if (tl) {
        __syncthreads();
        if (tid == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int MODE, bool CARRY, bool LNS = false>
int launch_pp320_one(const GemmK& k, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp320_kernel<MODE, CARRY, LNS>), hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_pp320_kernel<MODE, CARRY, LNS>), dim3(k.nbm * k.nbn, k.ksplit, 1), dim3(512), P3_SMEM, st, k);
    ICD_CHECK_LAUNCH("icd_gemm(ping-pong 256 x 320 tile)");
    return ICD_OK;
}

}  // namespace

namespace icd_gemm_detail {

int launch_pp320(const GemmK& k, hipStream_t st) {
    const bool carry = (k.out_c || k.resid_c) && k.ksplit == 1;
    const bool conv = k.ksize > 0 && k.Hout > 0;
    if (conv) return carry ? launch_pp320_one<1, true>(k, st) : launch_pp320_one<1, false>(k, st);
    if (k.ln_stats_w) return launch_pp320_one<0, false, true>(k, st);
    return carry ? launch_pp320_one<0, true>(k, st) : launch_pp320_one<0, false>(k, st);
}

}  // namespace icd_gemm_detail
