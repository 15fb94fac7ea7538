// gemm.hip - the MFMA workhorse of the iCD U-Net path on gfx950.
//
// One kernel family covers every dense contraction diffusers issues for the UNet:
//   conv3x3 (pad 1, stride 1|2, optional nearest-2x upsample and channel concat folded into the loader) as implicit GEMM,
//   1x1 conv / Linear (proj_in/out, to_q/k/v, to_out, FF, time-embedding MLPs), and the batched attention matmuls of
//   the materialised-P path (utils/p2p.py:335-338: baddbmm -> softmax -> controller -> bmm).
//
// Design (CDNA4):
//   * 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_f16 tiles, fp32 accum.
//   * Both operands are streamed global -> LDS with global_load_lds (16 B / lane, no VGPR round trip).  The im2col
//     gather is done on the per-lane SOURCE address (tap / upsample / stride / concat / zero padding via a zero page),
//     the LDS image stays lane-linear as the DMA requires.
//   * LDS tile rows are 128 B (64 halves); 16-B chunks are XOR-swizzled with ((row>>1)&7) so the ds_read_b128
//     fragment reads of a 16-lane group hit 16 distinct (bank-row half, chunk) slots - conflict free.  The swizzle is
//     applied on the source address (which chunk a lane fetches) and on the read address (same involution).
//   * 2-stage pipeline: the loads of k-tile t+1 are issued before the MFMAs of k-tile t; one barrier per k-tile.
//   * MFMA operands are swapped (weights as A, activations as B) so each lane ends up with 4 consecutive output
//     channels of one output row; the tile is staged through LDS in fp32 and written with fully coalesced 16-B
//     stores, with bias / time-bias / residual / GEGLU applied once, in fp32, before the single fp16 rounding.
//   * blockIdx -> tile mapping is XCD-aware (block b runs on XCD b%8): every XCD gets a contiguous range of tiles
//     so the n-tiles that share an activation tile hit the same private L2.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;      // 64 KiB
constexpr int EPI_LD = 132;                      // fp32 staging row stride (floats) for a 64x128 half tile
constexpr int EPI_LD_T = 68;                     // transposed staging: 128 rows (n) x 64 (m)

struct GemmK {
    const half_t* a0; const half_t* a1; const half_t* w;
    const float* bias; const half_t* rowbias; const half_t* resid; void* out;
    int M, N, K, Nw;
    int lda, ldw, ldo, ldr, ld_rowbias, rps;
    int C0, C1, Hin, Win, Hout, Wout, ksize, stride, upsample;
    int zdiv; long long a_bs0, a_bs1, w_bs0, w_bs1, o_bs0, o_bs1;
    float alpha; int flags;
    int nbm, nbn;
};

__device__ __forceinline__ int swz_off(int row, int chunk) {        // byte offset inside a [128][64] half tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int MODE, bool TRANS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmK p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int wm = wv >> 1, wn = wv & 1;

    // ---- XCD-aware tile id ------------------------------------------------------------------------------
    const int nblk = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mt = bid / p.nbn, nt = bid - mt * p.nbn;
    const int m0 = mt * BM, n0 = nt * BN;

    const int z = blockIdx.z;
    const int z0 = z / p.zdiv, z1 = z - z0 * p.zdiv;
    const half_t* A0 = p.a0 + z0 * p.a_bs0 + z1 * p.a_bs1;
    const half_t* Wp = p.w + z0 * p.w_bs0 + z1 * p.w_bs1;
    const half_t* zero = reinterpret_cast<const half_t*>(icd_zero_page);

    // ---- loader state: 4 A chunks + 4 W chunks per thread per k-tile -------------------------------------
    const int lrow = l >> 3, pchunk = l & 7;
    int a_k[4];              // dense: k offset of this lane's chunk; conv: channel offset within Cin
    int a_tap[4];            // conv: current tap
    int a_pix[4];            // conv: b*Hin*Win ; dense: unused
    int a_y[4], a_x[4];      // conv: output pixel coords
    bool a_ok[4];
    long long a_rowoff[4];   // dense: m*lda
    int w_k[4]; bool w_ok[4]; long long w_rowoff[4];
    const int Cin = p.C0 + p.C1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wv * 32 + j * 8 + lrow;
        const int lc = pchunk ^ ((r >> 1) & 7);
        const int m = m0 + r;
        a_ok[j] = m < p.M;
        if (MODE == 0) {
            a_k[j] = lc * 8;
            a_rowoff[j] = (long long)m * p.lda;
            a_tap[j] = 0; a_pix[j] = 0; a_y[j] = 0; a_x[j] = 0;
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            a_y[j] = rem / p.Wout;
            a_x[j] = rem - a_y[j] * p.Wout;
            a_pix[j] = b * p.Hin * p.Win;
            int c = lc * 8, tap = 0;
            while (c >= Cin) { c -= Cin; ++tap; }
            a_k[j] = c; a_tap[j] = tap;
            a_rowoff[j] = 0;
        }
        const int n = n0 + r;
        w_ok[j] = n < p.Nw;
        w_k[j] = lc * 8;
        w_rowoff[j] = (long long)n * p.ldw;
    }
    const int ntaps = p.ksize * p.ksize;
    const int pad = p.ksize >> 1;
    const int Hu = p.Hin << p.upsample, Wu = p.Win << p.upsample;

    auto issue_stage = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE_BYTES + wv * 4096;
        unsigned char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const half_t* src;
            if (MODE == 0) {
                src = (a_ok[j] && a_k[j] < p.K) ? A0 + a_rowoff[j] + a_k[j] : zero;
                a_k[j] += BK;
            } else {
                const int tap = a_tap[j];
                const int dy = (tap * 11) >> 5, dx = tap - dy * 3;      // tap / 3, tap % 3 for tap < 9
                const int yu = a_y[j] * p.stride + (ntaps == 9 ? dy : 0) - pad;
                const int xu = a_x[j] * p.stride + (ntaps == 9 ? dx : 0) - pad;
                const bool ok = a_ok[j] && tap < ntaps && (unsigned)yu < (unsigned)Hu && (unsigned)xu < (unsigned)Wu;
                const long long pix = a_pix[j] + (yu >> p.upsample) * p.Win + (xu >> p.upsample);
                const int c = a_k[j];
                const half_t* s0 = (c < p.C0) ? p.a0 + pix * p.C0 + c : p.a1 + pix * p.C1 + (c - p.C0);
                src = ok ? s0 : zero;
                int cn = c + BK, tp = tap;
                while (cn >= Cin) { cn -= Cin; ++tp; }
                a_k[j] = cn; a_tap[j] = tp;
            }
            glds16(src, sa + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const half_t* src = (w_ok[j] && w_k[j] < p.K) ? Wp + w_rowoff[j] + w_k[j] : zero;
            w_k[j] += BK;
            glds16(src, sw + j * 1024);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    issue_stage(0);
    const int lr = l & 31, lh = l >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0f70 | 0);     // vmcnt(0) (expcnt/lgkmcnt untouched)
        __syncthreads();
        if (kt + 1 < nk) issue_stage((kt + 1) & 1);
        const unsigned char* sa = smem + (kt & 1) * STAGE_BYTES;
        const unsigned char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = s * 2 + lh;
            f16x8 af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f16x8*>(sa + swz_off(wm * 64 + i * 32 + lr, c));
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = *reinterpret_cast<const f16x8*>(sw + swz_off(wn * 64 + j * 32 + lr, c));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (TRANS) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], wf[j], acc[i][j], 0, 0, 0);
                    else       acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: fp32 staging through LDS, two 64-row halves -------------------------------------------
    float* stage = reinterpret_cast<float*>(smem);
    const bool geglu = p.flags & ICD_GEMM_GEGLU;
    const bool out_f32 = p.flags & ICD_GEMM_OUT_F32;
    const long long o_off = z0 * p.o_bs0 + z1 * p.o_bs1;
    for (int h = 0; h < 2; ++h) {
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
        if (TRANS) {
            half_t* out = reinterpret_cast<half_t*>(p.out) + o_off;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int item = pass * 256 + tid;
                const int nl = item >> 3, mc = (item & 7) * 8;
                const int n = n0 + nl, m = m0 + h * 64 + mc;
                if (n >= p.N || m >= p.M) continue;
                const float* sp = stage + nl * EPI_LD_T + mc;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const int b = m / p.rps, key = m - b * p.rps;
                if (key + 8 <= p.rps && m + 8 <= p.M && (key & 7) == 0 && (p.ldo & 7) == 0) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[e] * p.alpha);
                    *reinterpret_cast<f16x8*>(out + ((long long)b * p.N + n) * p.ldo + key) = o;
                    if (key + 8 == p.rps)
                        for (int kk = p.rps; kk < p.ldo; ++kk) out[((long long)b * p.N + n) * p.ldo + kk] = (half_t)0.f;
                } else {
                    for (int e = 0; e < 8; ++e) {
                        const int mm = m + e;
                        if (mm >= p.M) break;
                        const int bb = mm / p.rps, kk = mm - bb * p.rps;
                        half_t* row = out + ((long long)bb * p.N + n) * p.ldo;
                        row[kk] = (half_t)(v[e] * p.alpha);
                        if (kk == p.rps - 1)
                            for (int k2 = p.rps; k2 < p.ldo; ++k2) row[k2] = (half_t)0.f;
                    }
                }
            }
        } else if (geglu) {
            half_t* out = reinterpret_cast<half_t*>(p.out) + o_off;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int r = pass * 32 + (tid >> 3), oc = (tid & 7) * 8;
                const int m = m0 + h * 64 + r;
                const int hcol = (oc >> 5) * 64 + (oc & 31);
                if (m >= p.M || n0 + hcol >= p.N) continue;
                const float* sp = stage + r * EPI_LD + hcol;
                f32x4 h0 = *reinterpret_cast<const f32x4*>(sp), h1 = *reinterpret_cast<const f32x4*>(sp + 4);
                f32x4 g0 = *reinterpret_cast<const f32x4*>(sp + 32), g1 = *reinterpret_cast<const f32x4*>(sp + 36);
                float hv[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                float gv[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float hh = hv[e] * p.alpha, gg = gv[e] * p.alpha;
                    if (p.bias) { hh += p.bias[n0 + hcol + e]; gg += p.bias[n0 + hcol + 32 + e]; }
                    o[e] = (half_t)(hh * gelu_erf_f(gg));
                }
                *reinterpret_cast<f16x8*>(out + (long long)m * p.ldo + (n0 >> 1) + oc) = o;
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int r = pass * 16 + (tid >> 4), c8 = (tid & 15) * 8;
                const int m = m0 + h * 64 + r, n = n0 + c8;
                if (m >= p.M || n >= p.N) continue;
                const float* sp = stage + r * EPI_LD + c8;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
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
                    f16x8 rs = *reinterpret_cast<const f16x8*>(p.resid + (long long)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rs[e];
                }
                if (out_f32) {
                    float* out = reinterpret_cast<float*>(p.out) + o_off + (long long)m * p.ldo + n;
                    *reinterpret_cast<f32x4*>(out) = (f32x4){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(out + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<f16x8*>(reinterpret_cast<half_t*>(p.out) + o_off + (long long)m * p.ldo + n) = o;
                }
            }
        }
    }
}

template <int MODE, bool TRANS>
int launch(const GemmK& k, int batch, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, TRANS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        attr_set = true;
    }
    dim3 grid(k.nbm * k.nbn, 1, batch);
    hipLaunchKernelGGL((gemm_kernel<MODE, TRANS>), grid, dim3(256), SMEM_BYTES, st, k);
    ICD_CHECK_LAUNCH("icd_gemm");
    return ICD_OK;
}

}  // namespace

extern "C" int icd_gemm(const icd_gemm_desc* d, void* stream) {
    ICD_CHECK_ARG(d != nullptr, "icd_gemm: null descriptor");
    ICD_CHECK_ARG(d->a0 && d->w && d->out, "icd_gemm: a0/w/out must be non-null");
    ICD_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "icd_gemm: M,N,K must be positive (got %d,%d,%d)", d->M, d->N, d->K);
    ICD_CHECK_ARG(d->K % 8 == 0 && d->ldw % 8 == 0, "icd_gemm: K and ldw must be multiples of 8 (K=%d ldw=%d)", d->K, d->ldw);
    ICD_CHECK_ARG(d->N % 8 == 0, "icd_gemm: N must be a multiple of 8 (got %d)", d->N);
    ICD_CHECK_ARG(d->mode == 0 || d->mode == 1, "icd_gemm: mode must be 0 (dense) or 1 (conv)");
    const bool trans = d->flags & ICD_GEMM_OUT_TRANS;
    const bool geglu = d->flags & ICD_GEMM_GEGLU;
    ICD_CHECK_ARG(!(trans && (geglu || (d->flags & ICD_GEMM_OUT_F32) || d->bias || d->resid || d->rowbias)),
                  "icd_gemm: transposed output supports alpha only");
    ICD_CHECK_ARG(!(geglu && (d->resid || d->rowbias || (d->flags & ICD_GEMM_OUT_F32) || d->N % 64 != 0)),
                  "icd_gemm: GEGLU needs N %% 64 == 0 and no resid/rowbias/f32 output");
    if (d->rowbias || trans) ICD_CHECK_ARG(d->rows_per_sample > 0, "icd_gemm: rows_per_sample required");
    GemmK k;
    k.a0 = (const half_t*)d->a0; k.a1 = (const half_t*)d->a1; k.w = (const half_t*)d->w;
    k.bias = d->bias; k.rowbias = (const half_t*)d->rowbias; k.resid = (const half_t*)d->resid; k.out = d->out;
    k.M = d->M; k.N = d->N; k.K = d->K; k.Nw = d->Nw > 0 ? d->Nw : d->N;
    k.lda = d->lda; k.ldw = d->ldw; k.ldo = d->ldo; k.ldr = d->ldr; k.ld_rowbias = d->ld_rowbias;
    k.rps = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
    k.C0 = d->C0; k.C1 = d->C1; k.Hin = d->Hin; k.Win = d->Win; k.Hout = d->Hout; k.Wout = d->Wout;
    k.ksize = d->ksize; k.stride = d->stride; k.upsample = d->upsample;
    k.zdiv = d->zdiv > 0 ? d->zdiv : 1;
    k.a_bs0 = d->a_bs0; k.a_bs1 = d->a_bs1; k.w_bs0 = d->w_bs0; k.w_bs1 = d->w_bs1; k.o_bs0 = d->o_bs0; k.o_bs1 = d->o_bs1;
    k.alpha = d->alpha; k.flags = d->flags;
    k.nbm = (d->M + BM - 1) / BM; k.nbn = (d->N + BN - 1) / BN;
    const int batch = d->batch > 0 ? d->batch : 1;
    if (d->mode == 1) {
        ICD_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "icd_gemm: conv ksize must be 1 or 3");
        ICD_CHECK_ARG(d->stride == 1 || d->stride == 2, "icd_gemm: conv stride must be 1 or 2");
        ICD_CHECK_ARG(d->upsample == 0 || d->upsample == 1, "icd_gemm: upsample must be 0 or 1");
        ICD_CHECK_ARG(d->C0 > 0 && d->C0 % 8 == 0 && d->C1 % 8 == 0 && d->C1 >= 0, "icd_gemm: conv channels must be multiples of 8");
        ICD_CHECK_ARG((d->C1 == 0) == (d->a1 == nullptr), "icd_gemm: a1/C1 mismatch");
        ICD_CHECK_ARG(d->K == d->ksize * d->ksize * (d->C0 + d->C1), "icd_gemm: K != taps*Cin");
        ICD_CHECK_ARG(d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0,
                      "icd_gemm: bad conv geometry");
        ICD_CHECK_ARG(batch == 1 && !trans, "icd_gemm: conv mode is not batched / transposed");
        return launch<1, false>(k, 1, (hipStream_t)stream);
    }
    ICD_CHECK_ARG(d->lda % 8 == 0, "icd_gemm: lda must be a multiple of 8");
    return trans ? launch<0, true>(k, batch, (hipStream_t)stream) : launch<0, false>(k, batch, (hipStream_t)stream);
}
