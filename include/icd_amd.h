/*
 * icd_amd.h - C ABI of the MI355X-native iCD U-Net hot path (libicd_amd.so).
 *
 * This is the drop-in boundary for the ONE path of yandex-research/invertible-cd that this project accelerates:
 * the few-step consistency U-Net loop.  In the reference the operator boundary is a Python callable
 *     model.unet(latents, t, timestep_cond=w_emb, encoder_hidden_states=ctx)["sample"]     utils/generation.py:241-244
 *     pipe.unet(latents, t, encoder_hidden_states=..., timestep_cond=..., added_cond_kwargs=...)[0]
 *                                                                                           utils/generation_sdxl.py:445-453,288-295
 * plus the per-Attention-module plugin callback controller(attention_probs, is_cross, place)  utils/p2p.py:336
 * and the boundary step predicted_origin(...)                                                utils/generation.py:136-155.
 * All arithmetic below that callable is third-party (diffusers/torch/cuDNN) in the reference; here it is
 * hand-written HIP for gfx950.  Python (ctypes) binds exactly these entry points - see INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - no allocation inside: outputs and workspace are caller-owned;
 *   - return 0 on success, negative icd_status on failure; icd_last_error() gives the message
 *     (the Python host raises RuntimeError, keeping the reference's exception convention);
 *   - activations are fp16 "NHWC" ([B, H*W, C] tokens) inside the path; latents cross the boundary as the
 *     reference's contiguous NCHW [B,4,H,W];  accumulation is fp32 everywhere.
 */
#ifndef ICD_AMD_H
#define ICD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ICD_OK = 0,
    ICD_ERR_INVALID_ARG = -1,
    ICD_ERR_UNSUPPORTED = -2,
    ICD_ERR_HIP = -3,
    ICD_ERR_MISSING_TENSOR = -4,
    ICD_ERR_WORKSPACE = -5,
    ICD_ERR_HOOK = -6
} icd_status;

const char* icd_last_error(void);
int icd_version(void);
/* 12 hex digits of the sha1 over the kernel sources (csrc/ *.hip, *.h, *.cpp) this binary was built from; "unstamped" for a build that
 * did not go through invertible_cd_amd/build.py.  The Python binding compares it with the sources next to it and refuses a stale library;
 * bench.py reports it as roofline.kernels_sha. */
const char* icd_build_sha(void);

/* ------------------------------------------------------------------------------------------------------------
 * Block-level operators (diffusers semantics, SURVEY.md section 8a row "a12/13-ops").
 * ---------------------------------------------------------------------------------------------------------- */

/* flags of icd_gemm_desc */
#define ICD_GEMM_GEGLU      1   /* out[m, j] = h * gelu_erf(g); weights/bias pre-interleaved in 32-column groups  */
#define ICD_GEMM_OUT_F32    2   /* out is float (attention scores before softmax)                                   */
#define ICD_GEMM_OUT_TRANS  4   /* out[(b*N + n)*ldo + (m % rows_per_sample)], b = m / rows_per_sample  (V^T)      */
#define ICD_GEMM_RESID_F32 16   /* resid is float [M, ldr] (fp32 residual stream of the fp32-fidelity VAE path)             */
#define ICD_GEMM_LN_COMPUTE 32  /* ln_stats is an OUTPUT: this launch also computes the (mean, rstd) of A's rows (eps = ln_eps) - */
                                /* from the MFMA operand fragments of its own main loop where the tile kernel allows (no pass  */
                                /* over A, no extra launch), else with an icd_layernorm_stats launch first (needs lda == K)     */
#define ICD_GEMM_PAD_HI     8   /* conv: zero padding on the bottom / right edge only (AutoencoderKL Downsample2D:  */
                                /* F.pad(x, (0,1,0,1)) + conv3x3 stride 2 pad 0), instead of ksize/2 on every side   */

/* Planner overrides for A/B tuning (tools/gemm_bench.py) and for tests that must exercise one tile family; they never
 * change results beyond fp32 summation order.  Not needed by callers. */
#define ICD_GEMM_TUNE_WM2        0x00040000   /* 128x128 tile, no split-K                                            */
#define ICD_GEMM_TUNE_WM4        0x00080000   /* 256x128 tile, no split-K                                            */
#define ICD_GEMM_TUNE_FORCE_BIG  0x00100000   /* take a 256-wide tile (gemm_big.hip) whatever the chip fill           */
#define ICD_GEMM_TUNE_NO_BIG     0x00200000   /* never take one                                                      */
#define ICD_GEMM_TUNE_NO_LN_INLINE 0x00800000 /* ICD_GEMM_LN_COMPUTE: always take the separate statistics launch (A/B)        */
#define ICD_GEMM_TUNE_BN256      0x00400000   /* big tiles: only the BN = 256 shapes                                 */
#define ICD_GEMM_TUNE_PP_V1      0x10000000   /* ping-pong 256 x 256 tile, dense, no carry: its first schedule (reads 12 / 4 / 8 / 0 per phase, one vmcnt(6) per k-tile): A/B */
#define ICD_GEMM_TUNE_NO_PP      0x20000000   /* never take the ping-pong 256 x 256 tile (gemm_pp.hip): A/B against the lockstep tiles */
#define ICD_GEMM_TUNE_BIG_CFG(i) (((i) + 1) << 24)   /* force big-tile configuration i (0..6, see gemm_common.h)      */

/* out = alpha * (A (*) W^T) + bias[n] + rowbias[m / rows_per_sample][n] + resid[m][n]
 * A is either a dense row-major [M, K] matrix (mode 0; Linear, 1x1 conv, attention bmm) or the implicit im2col
 * view of one/two NHWC tensors (mode 1; conv3x3 pad 1, stride 1|2, optional nearest-2x upsample in the loader,
 * optional channel concat cat([a0, a1], C) in the loader).  W is [N, K] row-major, K = taps * Cin (tap-major).
 * Replaces: torch conv2d / linear / baddbmm / bmm issued by diffusers ResnetBlock2D, Transformer2DModel,
 * Attention (utils/p2p.py:321-342), FeedForward(GEGLU), Down/Upsample2D, TimestepEmbedding. */
typedef struct {
    const void* a0;          /* fp16 */
    const void* a1;          /* fp16, second concat source or NULL */
    const void* w;           /* fp16 [Nw, K] */
    const float* bias;       /* fp32 [N] or NULL */
    const void* rowbias;     /* fp16 [B, ld_rowbias] or NULL (time-embedding projection) */
    const void* resid;       /* fp16 [M, ldr] or NULL */
    void* out;               /* fp16 (or fp32 with ICD_GEMM_OUT_F32) */
    int32_t M, N, K, Nw;     /* Nw = rows of W that exist (rows >= Nw read as zero), usually == N */
    int32_t lda, ldw, ldo, ldr, ld_rowbias;
    int32_t rows_per_sample; /* Hout*Wout (tokens per sample) */
    int32_t mode;            /* 0 dense, 1 conv */
    int32_t C0, C1;          /* channels of a0 / a1 (conv mode) */
    int32_t Hin, Win, Hout, Wout, ksize, stride, upsample;
    int32_t batch;           /* grid.z; z -> (z / zdiv, z % zdiv) */
    int32_t zdiv;
    int64_t a_bs0, a_bs1, w_bs0, w_bs1, o_bs0, o_bs1;  /* element strides */
    float alpha;
    int32_t flags;
    void* splitk_ws;          /* optional fp32 scratch for split-K partial tiles (small-M / deep-K shapes); NULL: never split */
    int64_t splitk_ws_bytes;
    /* LayerNorm fused into the GEMM that consumes it (BasicTransformerBlock: norm1 -> to_q/k/v, norm2 -> attn2.to_q,
     * norm3 -> ff.net.0.proj): A is the UN-normalised row-major activation, W has the LayerNorm gamma folded into its
     * columns, ln_stats is fp32 [M][2] = (mean, rstd) per row (icd_layernorm_stats) and ln_colsum fp32 [N] the row sums of
     * the gamma-scaled W; the epilogue computes  rstd_m * (alpha * (A W^T)[m][n] - mean_m * ln_colsum[n]) + bias[n] ...
     * with (W beta + original bias) passed as `bias`.  Exactly LN(A) W^T + bias in real arithmetic; the normalised
     * activation is never written.  Both NULL: plain GEMM.  Dense (mode 0), batch 1, fp16 output only. */
    const float* ln_stats;    /* written, not read, under ICD_GEMM_LN_COMPUTE (valid for later launches on the same stream) */
    const float* ln_colsum;
    /* Cross-attention fused behind the query projection (the north-star kernel; replaces to_q -> baddbmm -> softmax -> bmm
     * of utils/p2p.py:321-342 on layers whose controller does not need the probabilities).  When xattn_k is set, A W^T
     * (+ fused LayerNorm, + bias) is the query q of heads of 64 columns, and out[m, h*64 : (h+1)*64] =
     * softmax(xattn_scale * q_h[m] . K_h^T) V_h with K = xattn_k [B * xattn_nk rows, xattn_ldk] (head h at column h*64,
     * sample b at row b * xattn_nk) and V^T = xattn_vt [B][N][xattn_ldvt] (sample stride xattn_vt_bs elements, pad keys
     * zero).  q never reaches memory.  Needs N %% 128 == 0, rows_per_sample %% 256 == 0, xattn_nk <= 96, dense mode, no
     * residual. */
    const void* xattn_k;
    const void* xattn_vt;
    int32_t xattn_nk, xattn_ldk, xattn_ldvt;
    int64_t xattn_vt_bs;
    float xattn_scale;
    float ln_eps;             /* ICD_GEMM_LN_COMPUTE: epsilon of the LayerNorm (0 -> 1e-5) */
    /* Tuning and diagnostics travel with the call - the library keeps no process-wide GEMM state.  Zero / NULL: defaults.
     * None of them changes a result beyond fp32 summation order. */
    int32_t tune_group_m;     /* m-tiles per L2 group of the block -> tile map (0: the planner's choice) */
    int32_t tune_xattn_tile;  /* host tile of the fused query-projection + cross-attention launch: 0 planner, 2 = 128 x 128, 4 = 256 x 128, 5 = 256 x 256, 6 = 192 x 256 */
    void* debug_timeline;     /* device buffer of 8 x uint64 per block: every block of this launch stamps s_memrealtime (100 MHz) at */
                              /* start, first k-tile landed, main loop done, epilogue done (+ s_memtime ticks of the main loop)     */
    /* Second output for the fp32 twin of the residual stream (icd_unet option ICD_UNET_OPT_RESIDUAL_MODE = ICD_RESIDUAL_F32): the SAME values as `out` before the
     * fp16 rounding, fp32 [M, ldo] (row stride ldo).  With an fp32 `resid` (ICD_GEMM_RESID_F32) a chain h <- h + f(h) accumulates
     * in fp32 while every consumer still reads the fp16 copy.  NULL: off.  fp16 `out` only, no GEGLU / transposed / batched. */
    float* out_f32;
    /* Error carry of the residual stream (icd_unet option ICD_UNET_OPT_RESIDUAL_MODE = ICD_RESIDUAL_CARRY, the default): a tensor of a chain
     * h <- h + f(h) lives as the fp16 value every consumer reads PLUS one byte per element holding what its rounding lost,
     *     carry = bf8_e5m2((v - fp16(v)) * 2^14)          value = fp16 + carry * 2^-14,
     * i.e. about four more mantissa bits on the stream (the rounding of the stream per add is the dominant error term of an
     * fp16-storage UNet) for 1 + 1 bytes per element and add instead of the 2 + 4 of an fp32 twin.
     * resid_carry: uint8 [M, ldr] beside an fp16 `resid`; out_carry: uint8 [M, ldo] beside the fp16 `out` (same leading dimensions, in
     * elements).  Either may be NULL.  Plain fp16 output only (no GEGLU / transposed / fp32 / batched / fused cross-attention). */
    const void* resid_carry;
    void* out_carry;
    /* Phase form of Upsample2D (nearest 2x + conv3x3, diffusers): output pixel (2y + py, 2x + px) only sees the 2 x 2 input pixels
     * (y + py - 1 .. y + py, x + px - 1 .. x + px), so the conv is four 2 x 2 convs on the INPUT grid with tap-summed weights - 16 tap
     * GEMMs instead of 36, 4/9 of the flops.  One launch per phase: a stride-1 3 x 3 geometry on the input grid (upsample = 0, Hout = Hin)
     * restricted to the conv_ktaps = 4 taps {t0, t0 + 1, t0 + 3, t0 + 4}, t0 = conv_tap_base = 3 py + px, with K = 4 * Cin and W the
     * phase's [N, 4 * Cin] tap sums, and the output row of GEMM row m scattered to pixel (2y + py, 2x + px) of the [B, 2H, 2W] output:
     * out_remap_w = Win (> 0 switches it on: row(m) = 2 m + 2 Win (m / Win) + out_remap_c), out_remap_c = 2 py Win + px... times ldo
     * elements as usual; out_carry follows the same rows.  conv_ktaps = 0: all ksize^2 taps, no remap (everything above off). */
    int32_t conv_tap_base, conv_ktaps, out_remap_w, out_remap_c;
} icd_gemm_desc;

int icd_gemm(const icd_gemm_desc* d, void* stream);
/* What icd_gemm would run for this descriptor (pure function of the descriptor; nothing is enqueued): kernel 0 = the 128-wide
 * tiles of gemm.hip, 1 = the 256-wide tiles of gemm_big.hip; tile_m x tile_n the block tile; ksplit > 1: split-K + reduce launch;
 * ln_inline: ICD_GEMM_LN_COMPUTE statistics come from the main loop's operand fragments (no statistics launch); xattn: the
 * cross-attention epilogue runs in this launch.  The executor records it per launch (icd_profile_dump) so that tests can assert
 * which code path a UNet forward took. */
typedef struct {
    int32_t kernel, tile_m, tile_n, ksplit, ln_inline, xattn;
} icd_gemm_plan_info;
int icd_gemm_plan(const icd_gemm_desc* d, icd_gemm_plan_info* out);
/* Bytes of split-K scratch icd_gemm would like for this shape (0: the shape is not split). */
int64_t icd_gemm_workspace_bytes(int32_t M, int32_t N, int32_t K);

/* GroupNorm(32 groups, affine, eps) [+ SiLU] over NHWC fp16, input optionally the channel concat of two tensors
 * (up-block skip connections), output a single NHWC tensor.  stats_ws: >= B*split*groups*2 floats.
 * Replaces: torch group_norm + silu in ResnetBlock2D / Transformer2DModel.norm / conv_norm_out. */
int icd_groupnorm(const void* x0, int32_t C0, const void* x1, int32_t C1, int32_t B, int32_t HW, int32_t groups,
                  const float* gamma, const float* beta, float eps, int32_t silu, void* out, float* stats_ws,
                  void* stream);
int64_t icd_groupnorm_ws_floats(int32_t B, int32_t HW, int32_t groups);
/* GroupNorm of tensors of the residual stream that travel with their error carry (icd_gemm_desc.out_carry: value = fp16 + 2^-14 * bf8):
 * the normalisation reads fp16 + carry (the statistics pass reads the fp16 part only).  carry0 / carry1: uint8 [B*HW, C0 / C1], either may
 * be NULL.  aux (optional): fp16 [B*HW, ld_aux] written beside `out` for the split shortcut conv of a ResnetBlock2D
 * (ICD_RESIDUAL_SPLIT) - with two sources [x1 (C1) | lo0 (C0) | lo1 (C1)], with one source [lo0 (C0)], lo = fp16(2^-14 * carry) - so that
 * a two-source 1x1 conv over (x0, aux) with weights [W0 | W1 | W0 | W1] sees x0 + lo0 and x1 + lo1.  Everything else as icd_groupnorm. */
int icd_groupnorm_carry(const void* x0, int32_t C0, const void* carry0, const void* x1, int32_t C1, const void* carry1, int32_t B,
                        int32_t HW, int32_t groups, const float* gamma, const float* beta, float eps, int32_t silu, void* out, void* aux,
                        int32_t ld_aux, float* stats_ws, void* stream);
/* lo[i] = fp16(2^-14 * bf8(carry[i])), i < n (n %% 8 == 0): the second K segment of a split-operand GEMM, A = [hi | lo] against
 * W = [W | W], for consumers of a carried tensor that no GroupNorm reads first (Transformer2DModel.proj_out, the downsampler conv). */
int icd_carry_expand(const void* carry, int64_t n, void* lo, void* stream);
/* out[r] = [lo (C) | hi (C)] for rows r < rows of a carried tensor hi [rows, C] + carry: the second source of a conv over [x | lo | x] against
 * per-tap weights [W_hi | W_hi | W_lo] - split activations and split weights (the tap sums of the phase-form upsampling conv are not fp16
 * numbers; with W_lo = fp16(W - W_hi) the product is that of the exact sums). */
int icd_carry_expand2(const void* carry, const void* hi, int64_t rows, int32_t C, void* out, void* stream);

/* fp32-fidelity path of the VAE (reference: vae.to(torch.float32), utils/generation_sdxl.py:465-466).  Activations that can
 * exceed the fp16 range (conv outputs, residual stream) are fp32 [rows, C]; GEMM operands are fp16 "split3" tensors
 * [rows, 3C] = [hi | lo | hi] with hi = fp16(v), lo = fp16(v - hi), multiplied against weights packed [w_hi | w_hi | w_lo]
 * by the unchanged fp16 MFMA kernels (icd_gemm with K = 3 * K0, ICD_GEMM_OUT_F32, ICD_GEMM_RESID_F32).
 * icd_groupnorm_f32_split: GroupNorm (+SiLU) of an fp32 tensor -> split3.  stats_ws as icd_groupnorm_ws_floats.
 * icd_split_cast: (x * scale) -> split3 for tensors that reach a conv without a GroupNorm (scale = a power of two keeps the
 * values inside the fp16 range; the consumer's alpha undoes it exactly). */
int icd_groupnorm_f32_split(const float* x, int32_t C, int32_t B, int32_t HW, int32_t groups, const float* gamma, const float* beta,
                            float eps, int32_t silu, void* out_split3, float* stats_ws, void* stream);
int icd_split_cast(const float* x, int64_t rows, int32_t C, float scale, void* out_split3, void* stream);
/* *out (device fp32) = max |x[i]|, i < n: the host picks the power-of-two scale of icd_split_cast from it. */
int icd_absmax(const float* x, int64_t n, float* out, void* stream);

/* LayerNorm over the last dim of [rows, C] fp16 (eps 1e-5, affine).  Replaces torch layer_norm in BasicTransformerBlock. */
int icd_layernorm(const void* x, int64_t rows, int32_t C, const float* gamma, const float* beta, float eps, void* out,
                  void* stream);
/* Statistics half of LayerNorm: stats[r] = (mean, 1/sqrt(var + eps)) of row r (fp32 [rows][2], exact two-pass variance);
 * the normalisation itself is applied by the consuming GEMM (icd_gemm_desc.ln_stats). */
int icd_layernorm_stats(const void* x, int64_t rows, int32_t C, float eps, float* stats, void* stream);

/* Row softmax: P[r, 0:cols] = softmax(scale * S[r, 0:cols]) (fp32 in, fp16 out, pad columns [cols, ld) zeroed).
 * Replaces Attention.get_attention_scores' softmax (utils/p2p.py:335). */
int icd_softmax_rows(const float* s, int64_t rows, int32_t cols, int32_t ld_s, float scale, void* p, int32_t ld_p,
                     void* stream);

/* Fused attention without materialised probabilities (flash style):
 *   out[b, n, h*d:(h+1)*d] = softmax(scale * q[b,n,h,:] . k[b,:,h,:]) @ v
 * q: [B, Nq, ldq] (head h at column h*d), k: [B, Nk, ldk], vt: V TRANSPOSED [B, H*d, ldvt] (keys contiguous),
 * out: [B, Nq, ldo].  d in {40, 64, 80, 160} (any multiple of 8 up to 160).  vt_batch_stride: elements between the V^T
 * blocks of consecutive samples (0 = H*d*ldvt; larger when vt is a row slice of a wider [B, sum(C), ldvt] buffer that
 * holds the V^T of every cross-attention layer).
 * Used on layers whose controller does not need P (utils/p2p.py:147,184-188: N > 32^2, or no controller). */
int icd_attention_fused(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t H, int32_t Nq,
                        int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldvt, int32_t ldo,
                        int64_t vt_batch_stride, float scale, void* stream);

/* Same with flags: ICD_ATTN_CAUSAL masks keys after the query (needs Nq == Nk) - the causal self-attention of the CLIP
 * text encoders behind `text_encoder(ids)[0]` (utils/generation.py:293,301) and `encode_prompt`
 * (utils/generation_sdxl.py:31-44). */
#define ICD_ATTN_CAUSAL 1
/* ICD_ATTN_Q_PRESCALED: q already carries scale * log2(e) (the executor folds the factor into the query projection's weights at
 * load time, so no extra rounding happens): the scores are base-2 exponents as they leave the matrix cores, `scale` is ignored,
 * and for head dims 40 / 64 / 80 the running softmax offset is subtracted by the MFMA itself (no scale FMA per score on the
 * VALU, which bounds this kernel).  ICD_ATTN_TUNE_MODE0 (A/B, tests): keep the VALU form for a prescaled q. */
#define ICD_ATTN_Q_PRESCALED 2
#define ICD_ATTN_TUNE_MODE0  4
int icd_attention_fused_ex(const void* q, const void* k, const void* vt, void* out, int32_t B, int32_t H, int32_t Nq,
                           int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldvt, int32_t ldo,
                           int64_t vt_batch_stride, float scale, int32_t flags, void* stream);

/* Materialised attention probabilities in one pass (the layers whose controller reads or edits P, utils/p2p.py:335-338):
 *   probs[(b*H + h), n, 0:Nk] = softmax_key(scale * q[b,n,h,:] . k[b,:,h,:])   fp16, row stride ldp, columns [Nk, ldp) zero.
 * Replaces baddbmm -> softmax of Attention.get_attention_scores without the fp32 score tensor in between; the controller
 * callback and P.V (icd_gemm, batched) follow as before.  q / k layouts as in icd_attention_fused. */
int icd_attention_probs(const void* q, const void* k, void* probs, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t d,
                        int32_t ldq, int32_t ldk, int32_t ldp, float scale, void* stream);
/* The same with split operands: q_carry / k_carry (either may be NULL) are the error carries of q / k as icd_gemm_desc.out_carry writes
 * them (uint8, the indexing of q / k); scores = qh.kh + 2^-14 (ql.kh + qh.kl).  A stored probability is the exponential of its score:
 * the fp16 rounding of q and k is an absolute error of the exponent, i.e. a relative error of every map a controller keeps. */
int icd_attention_probs_split(const void* q, const void* q_carry, const void* k, const void* k_carry, void* probs, int32_t B, int32_t H,
                              int32_t Nq, int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldp, float scale, void* stream);
/* What the reference's shipped controllers do to the probabilities of a layer (utils/p2p.py:138-221), in the probability kernel's epilogue
 * instead of in passes of their own over P.  The samples of the launch are [first_cond_sample unrelated samples (the unconditional half of a
 * CFG-doubled batch) | base prompt | edited prompts ...]; everything below applies to the samples from first_cond_sample on.
 *   acc             fp16 [(B - first_cond_sample) * H, Nq, ldp] or NULL: acc += P_final with torch's fp16 in-place-add rounding
 *                   (AttentionStore.between_steps, utils/p2p.py:164-170);
 *   self_from_base  != 0: the edited prompts take the base prompt's probabilities (AttentionControlEdit.replace_self_attention inside its
 *                   step window, utils/p2p.py:183-188) - their blocks read the base sample's q and k: the same bits as the copy;
 *   edit_At, edit_D the cross-attention edit of AttentionReplace / Refine / Reweight with the cross_replace_alpha blend as one linear
 *                   operator per step, new_row[e] = base_row . A_e + D_e (*) cur_row[e]: fp16 [nedit][96][80] / fp32 [nedit][96] exactly as
 *                   icd_p2p_cross_edit takes them (nedit = B - first_cond_sample - 1; <= 80 keys), or NULL.  Bit-identical to
 *                   icd_attention_probs followed by icd_p2p_cross_edit. */
typedef struct {
    void* acc;
    const void* edit_At;
    const float* edit_D;
    int32_t first_cond_sample, self_from_base;
    int64_t first_cond_row;       /* alternative to first_cond_sample for callers that know rows, not heads: first_cond_sample = first_cond_row / H */
    int32_t edit_count;           /* number of edited prompts the caller built edit_At / edit_D (or planned self_from_base) for; when > 0 the launch
                                   * is rejected unless it equals B - first_cond_sample - 1 (the kernel indexes the operators by sample: a caller
                                   * whose conditional samples are not exactly [base | edit_count edited prompts] must not use this epilogue) */
    int32_t reserved0;
} icd_probs_epilogue;
int icd_attention_probs_ex(const void* q, const void* q_carry, const void* k, const void* k_carry, void* probs, int32_t B, int32_t H,
                           int32_t Nq, int32_t Nk, int32_t d, int32_t ldq, int32_t ldk, int32_t ldp, float scale,
                           const icd_probs_epilogue* epilogue, void* stream);

/* Sinusoidal embeddings.  kind 0: diffusers Timesteps(dim, flip_sin_to_cos=True, shift=0) -> [cos || sin];
 * kind 1: guidance_scale_embedding (utils/generation.py:96-122) -> [sin || cos] of 1000*w, denominator half-1.
 * vals: fp32 [n] on device; out fp16 [n, dim]. */
int icd_sinusoid(const float* vals, int32_t n, int32_t dim, int32_t kind, void* out, void* stream);
/* The same with fp32 output and the correctly rounded fp32 frequency table (the precise time-embedding path, ICD_SPLIT_TEMB). */
int icd_sinusoid_f32(const float* vals, int32_t n, int32_t dim, int32_t kind, float* out, void* stream);
/* out[r] = [hi (C) | lo (C)] fp16 of act(x[r]) (act 0: identity, 1: SiLU) for fp32 x [rows, C]: hi = fp16(v), lo = fp16(v - hi); a GEMM
 * over it against [W | W] sees v to ~2^-22.  The time-embedding MLPs of the precise path run on it (fp32 in, fp32 out). */
int icd_split2_act(const float* x, int64_t rows, int32_t C, int32_t act, void* out, void* stream);

/* y = silu(x) over n fp16 elements (time-embedding activation shared by all ResnetBlock2D.time_emb_proj). */
int icd_silu(const void* x, int64_t n, void* out, void* stream);
/* kind 0 silu, 1 quick_gelu x*sigmoid(1.702x) (CLIP ViT-L text MLP), 2 gelu erf form (OpenCLIP bigG text MLP). */
int icd_activation(const void* x, int64_t n, int32_t kind, void* out, void* stream);
/* CLIPTextEmbeddings: out[r, :] = tok_emb[ids[r], :] + pos_emb[r % T, :]; ids int64 [rows] on device, tables fp16. */
int icd_embed_tokens(const int64_t* ids, const void* tok_emb, const void* pos_emb, int64_t rows, int32_t T, int32_t C,
                     int32_t vocab, void* out, void* stream);

/* conv_in: 3x3 pad 1, Cin=4 NCHW latents (fp16 or fp32) -> NHWC fp16 [B, H*W, Cout].  w: fp16 [Cout, 3,3,4]. */
int icd_conv_in(const void* x_nchw, int32_t x_is_f32, int32_t B, int32_t H, int32_t W, const void* w,
                const float* bias, int32_t Cout, void* out, void* stream);

/* NCHW [B,4,HW] latents (fp16 or fp32) -> token-major [B*HW, 8] fp16 with channels 4..7 zero: lets conv_in run as an
 * implicit GEMM (icd_gemm mode 1, C0 = 8, K = 72, weights [Cout, 3,3,8]) on the matrix cores - what the executor does. */
int icd_pack_latent(const void* x_nchw, int32_t x_is_f32, int32_t B, int32_t HW, void* out, void* stream);

/* General form of icd_pack_latent: NCHW [B,C,HW], C <= 8 -> [B*HW, 8] fp16; padding channels are zero except
 * `ones_channel` (>= C, or -1 for none), which is 1.0 inside the image.  The VAE decoder uses it to fold
 * post_quant_conv (1x1, with bias) into conv_in exactly: the bias rides on the ones channel, so the zero padding of
 * the 3x3 conv still sees zeros outside the image (utils/generation.py:258 vae.decode). */
int icd_pack_nchw(const void* x_nchw, int32_t x_is_f32, int32_t B, int32_t C, int32_t HW, int32_t ones_channel, void* out,
                  void* stream);

/* conv_out: 3x3 pad 1 over NHWC fp16 [B,H*W,Cin] -> NCHW eps [B,4,H,W] (fp16 or fp32).  w: fp16 [4, 3,3,Cin]. */
int icd_conv_out(const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w, const float* bias,
                 void* eps_nchw, int32_t out_is_f32, void* stream);
/* Same with Cout = 1..4 output channels, result NCHW [B,Cout,H,W]; w stays [4, 3,3,Cin] (rows >= Cout ignored).
 * AutoencoderKL decoder.conv_out (128 -> 3) and encoder.conv_out + quant_conv folded to the 4 mean channels. */
int icd_conv_out_n(const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w, const float* bias,
                   int32_t Cout, void* out_nchw, int32_t out_is_f32, void* stream);

/* Prompt-to-prompt cross-attention edit, fused and in place (utils/p2p.py:190-207 with the replace_cross_attention of
 * AttentionReplace :227, AttentionRefine :238-241, AttentionReweight :254-258 and the cross_replace_alpha blend):
 *   probs[(g*heads + h), p, :] for g = 1 .. n_prompts-1  <-  probs[(0*heads + h), p, :] . A[g-1] + D[g-1] (*) itself
 * probs: fp16 [n_prompts*heads, nq, ld] (the conditional rows of one Attention module), tokens nk <= 80 <= ld, ld % 8 == 0,
 * pad columns zero (the executor's 80-column buffers for the 77 CLIP tokens);
 * At: fp16 [n_prompts-1, 96, 80] = A transposed and zero padded (At[e][n][w] = A[e][w][n]), D: fp32 [n_prompts-1, 96] zero
 * padded - built per step on the device from the controller's mapper / alphas / equalizer / cross_replace_alpha tensors
 * (invertible_cd_amd/p2p.py).  The product runs on the matrix cores with fp32 accumulation. */
int icd_p2p_cross_edit(void* probs, int32_t n_prompts, int32_t heads, int64_t nq, int32_t nk, int32_t ld, const void* At,
                       const float* D, void* stream);

/* LocalBlend (utils/p2p.py:18-44) in one launch: word-weighted mean over layers x heads of the res x res cross-attention maps
 * of each prompt -> 3x3 max pool -> (nearest resize to the latent) -> normalise by the maximum -> threshold th_pool, OR-ed with
 * the base prompt's mask; optional substruct words (alpha_sub: no pooling, threshold th_sub) are cleared from it;
 * out[p] = x[0] + float(mask_p) * (x[p] - x[0]).  maps[l]: fp16 [n_prompts * heads[l], res*res, ld] (the accumulated
 * AttentionStore tensors, row stride ld >= n_words), alpha / alpha_sub: fp32 [n_prompts][n_words] word masks, x: [n_prompts, C,
 * H, W] fp16 or fp32, out: fp32 (torch's promotion of `base + mask.float() * (x_t - base)`).  maps / heads are HOST arrays of
 * n_layers <= 8 entries. */
int icd_local_blend(const void* const* maps, const int32_t* heads, int32_t n_layers, int32_t n_prompts, int32_t res,
                    int32_t n_words, int32_t ld, const float* alpha, const float* alpha_sub, float th_pool, float th_sub,
                    const void* x, int32_t x_is_f32, int32_t C, int32_t H, int32_t W, float* out, void* stream);
/* dst[t][i] += src[t][i] (fp16, rounded like torch's in-place add) for up to 32 tensors in one launch: the per-step accumulation
 * of AttentionStore.between_steps (utils/p2p.py:164-170).  dst / src / counts are HOST arrays; tensors 16-byte aligned. */
int icd_accumulate_multi(void* const* dst, const void* const* src, const int64_t* counts, int32_t n_tensors, void* stream);

/* Consistency boundary step, eps-prediction (utils/generation.py:136-155 == utils/generation_sdxl.py:112-132):
 *   x0 = (x - sigma_t*eps)/alpha_t ; out = alpha_s*x0 + sigma_s*eps, (alpha_s, sigma_s) := (1, 0) where s == 0.
 * coef: fp32 [B,4] = (alpha_t, sigma_t, alpha_s, sigma_s) per sample (host gathers them from the 1000-entry
 * tables, as the reference does with extract_into_tensor).  x/eps/out: [B, per_sample]; dtype_flags bit0: x is fp32,
 * bit1: eps is fp32, bit2: out is fp32 (otherwise fp16).  Arithmetic is fp32 without FMA contraction, i.e. the
 * reference's torch expression evaluated in fp32 (tables are fp32, so torch promotes). */
int icd_x0_step(const void* x, const void* eps, const float* coef, int32_t B, int64_t per_sample, int32_t dtype_flags,
                void* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Whole-U-Net executor (native runtime: plan + arena + launches; one call per UNet evaluation).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t in_channels, out_channels;
    int32_t num_levels;                 /* 4 (SD1.5) or 3 (SDXL) */
    int32_t block_out_channels[4];
    int32_t down_has_attn[4], up_has_attn[4];
    int32_t transformer_layers[4];      /* per down level */
    int32_t num_heads[4];               /* per down level */
    int32_t layers_per_block;
    int32_t cross_dim;
    int32_t use_linear_projection;      /* informational: proj_in/out are GEMMs either way in NHWC */
    int32_t time_cond_proj_dim;         /* 512 or 0 */
    int32_t addition_time_embed_dim;    /* 256 (SDXL) or 0 */
    int32_t add_in_dim;                 /* 2816 (SDXL) or 0 */
    int32_t norm_groups;
} icd_unet_config;

typedef struct icd_unet icd_unet;

/* Attention plugin callback - the C form of `controller(attention_probs, is_cross, place_in_unet)` (utils/p2p.py:336).
 * Called on the host, in module-execution order, once per Attention module per forward with phase 0:
 *   phase 0 (ICD_HOOK_QUERY): return 0 -> run this layer fused (no P); return 1 -> materialise P: the hook must have
 *            stored in *probs a device buffer of bh*nq*ld fp16 elements (a FRESH allocation if it intends to keep
 *            it, cf. AttentionStore's views utils/p2p.py:148).  Return 2 (round 5; even batch) -> materialise P for the SECOND HALF of
 *            the batch only - the conditional samples of a [uncond; cond] batch, all the reference's controllers touch
 *            (utils/p2p.py:153-155): the buffer holds bh/2 rows, phase 1 is called with bh/2, an icd_probs_epilogue counts its rows
 *            from the first materialised one, and the first half of the batch runs the fused kernel.
 *   phase 1 (ICD_HOOK_PROBS): P has been enqueued on `stream`; the hook may enqueue in-place edits on the same
 *            stream.  Return 0; negative aborts the forward with ICD_ERR_HOOK.
 * place: 0 down, 1 mid, 2 up.  P layout: [bh = B*heads (row b*heads+h), nq, ld] with nk valid columns. */
#define ICD_HOOK_QUERY 0
#define ICD_HOOK_PROBS 1
/*   In phase 0 `probs` points at TWO slots (round 5): [0] the P buffer as above, [1] optionally a pointer to an icd_probs_epilogue for this
 *            layer call (copied before the call returns; first_cond_row in rows of P, first_cond_sample ignored) - the controller's work
 *            on P that the probability kernel performs in its epilogue: store accumulation, self-attention replacement, the
 *            cross-attention edit.  The hook must not repeat that work in phase 1.  Hooks that leave slot 1 alone get the plain kernel;
 *            in phase 1 `probs` points at one slot. */
typedef int (*icd_attn_hook)(void* user, int32_t phase, int32_t layer, int32_t is_cross, int32_t place, int64_t bh,
                             int64_t nq, int64_t nk, int64_t ld, void** probs);

int icd_unet_create(const icd_unet_config* cfg, icd_unet** out);
void icd_unet_destroy(icd_unet* u);
/* Bind one packed tensor (see invertible_cd_amd/unet.py for the packing).  dtype: 0 fp16, 1 fp32. */
int icd_unet_set_tensor(icd_unet* u, const char* name, const void* ptr, int32_t dtype, int64_t numel);
/* Validate that every tensor the plan needs is bound. */
int icd_unet_finalize(icd_unet* u);
int32_t icd_unet_num_attention_layers(const icd_unet* u);
/* Per-handle execution options (two handles on two streams may hold different settings; nothing is process-wide).
 *   ICD_UNET_OPT_XATTN_FUSION   LayerNorm -> to_q -> cross-attention of a layer (head dim 64, tokens %% 256 == 0, <= 96 keys, no
 *       controller asking for its probabilities) as ONE launch (icd_gemm_desc.xattn_*): 0 never, 1 every eligible layer,
 *       2 (default) where it measured faster - the 256 x 256 host tile (C %% 256 == 0) in one round of 128..256 blocks (SDXL's
 *       1024-token layers at 8 images per GPU).  Results differ only by the fp16 rounding order of q.
 *   ICD_UNET_OPT_LN_INLINE_STATS  1 (default): the first GEMM behind every LayerNorm computes the statistics itself
 *       (ICD_GEMM_LN_COMPUTE); 0: a separate icd_layernorm_stats pass over the residual stream.  Statistics agree to ~1e-6.
 *   ICD_UNET_OPT_XATTN_TILE     A/B tuning of the fused launch's host tile (icd_gemm_desc.tune_xattn_tile): 0 (default) planner,
 *       2 = 128 x 128, 4 = 256 x 128, 5 = 256 x 256, 6 = 192 x 256 (six m-tiles per 1024-query sample).
 *   ICD_UNET_OPT_ATTN_VALU_SCALE  A/B: 1 = the flash attention kernels apply the softmax offset with one FMA per score on the VALU
 *       (ICD_ATTN_TUNE_MODE0), 0 (default) = the MFMA subtracts it (head dims 40 / 64 / 80).  Same arithmetic up to fp32 rounding.
 *   ICD_UNET_OPT_RESIDUAL_MODE  precision of the residual stream.  Every chain x <- x + f(x) of the UNet (ResnetBlock2D conv2 + input /
 *       shortcut, the three branch adds of a BasicTransformerBlock, Transformer2DModel proj_out + input) rounds the whole stream to
 *       fp16 once per add, 100 - 300 adds deep - the dominant error term of fp16 storage (eps vs an fp32 evaluation 1.1e-3 rel-L2).
 *         ICD_RESIDUAL_FP16  (0)  plain fp16 stream (rounds 1 - 3's default);
 *         ICD_RESIDUAL_F32   (1)  fp32 twin: the chain accumulates in fp32 beside the fp16 copy the next operator reads
 *                                 (icd_gemm_desc.out_f32 + ICD_GEMM_RESID_F32): 0.7e-3, 6 more bytes per element and add;
 *         ICD_RESIDUAL_CARRY (2)  error carry: one bf8 byte per element keeps what the rounding lost
 *                                 (icd_gemm_desc.resid_carry / out_carry): the same 0.7e-3 for 2 more bytes per element and add;
 *         ICD_RESIDUAL_SPLIT (3, default)  the carry, and the consumers that dominate what is left read it too: every GroupNorm
 *                                 normalises fp16 + carry (icd_groupnorm_carry; conv1 and the sampler convs pass a carry on for it), and
 *                                 the shortcut conv of the channel-changing resnets, proj_out and the downsampler conv take hi + lo as a
 *                                 two-source GEMM over [x | lo] against [W | W] (tensors `<name>.weight2`; lo = fp16(2^-14 carry) from
 *                                 icd_groupnorm_carry's aux output / icd_carry_expand): 0.40 - 0.45e-3, so that the 3 - 4 step forward
 *                                 and edit loops of the reference stay inside 1e-3 (tests/error_budget_sim.py has the budget).
 *       load_models(dtype='fp32') (the reference's default, utils/loading.py:34,38) adds fp32 latents / eps at the boundary to it.
 *       Set before sizing the workspace.
 * Returns ICD_ERR_INVALID_ARG for an unknown option or value. */
#define ICD_UNET_OPT_RESIDUAL_MODE   5
#define ICD_UNET_OPT_RESIDUAL_F32    5   /* round-3 name of the same option */
#define ICD_RESIDUAL_FP16  0
#define ICD_RESIDUAL_F32   1
#define ICD_RESIDUAL_CARRY 2
#define ICD_RESIDUAL_SPLIT 3
/* ICD_UNET_OPT_SPLIT_MASK: which consumers ICD_RESIDUAL_SPLIT covers (bit mask; default ICD_SPLIT_DEFAULT).  Diagnostic / A-B: the
 * error budget (profiles/r05_error_budget.txt) switches them one at a time.  Set before sizing the workspace. */
#define ICD_UNET_OPT_SPLIT_MASK      6
/* ICD_UNET_OPT_UPSAMPLE_PHASES: 1 (default) = Upsample2D's nearest-2x + conv3x3 runs as four 2 x 2 convs on the input grid, one per
 * output pixel phase, with tap-summed weights `<name>.phase.<2 py + px>` (icd_gemm_desc.conv_ktaps): 4/9 of the flops; 0 = the 3 x 3
 * conv with the upsampling folded into its loader (rounds 1 - 4). */
#define ICD_UNET_OPT_UPSAMPLE_PHASES 7
/* ICD_UNET_OPT_GEMM_TUNE: ICD_GEMM_TUNE_* planner bits (NO_PP, NO_BIG, BN256) OR-ed into every icd_gemm launch of this handle: same-process, same-box
 * A/B of a tile family over a whole forward (bench.py --gemm-tune); 0 (default) = the planner's own choice.  Results differ by fp32 summation
 * order where a different split-K plan is taken, not otherwise. */
#define ICD_UNET_OPT_GEMM_TUNE       8
#define ICD_SPLIT_GN          1    /* every GroupNorm normalises fp16 + carry (skip tensors keep their carry for it) */
#define ICD_SPLIT_CONV1       2    /* conv1 of a ResnetBlock2D hands its output to GroupNorm 2 with a carry */
#define ICD_SPLIT_SHORTCUT    4    /* conv_shortcut over [x | lo] (needs ICD_SPLIT_GN: that GroupNorm writes lo) */
#define ICD_SPLIT_PROJ_OUT    8    /* Transformer2DModel.proj_out over [h | lo] */
#define ICD_SPLIT_DOWN       16    /* the downsampler conv over [h | lo] */
#define ICD_SPLIT_SAMPLER_OUT 32   /* down / up sampler outputs carry their rounding error on */
#define ICD_SPLIT_UP         64    /* the LAST upsampler conv (the one at the output resolution) over [h | lo | h] against split tap sums: an error  */
                                   /* made there is damped by nothing downstream - 70 % of what splitting all of them buys for 45 % of the cost     */
#define ICD_SPLIT_TEMB      128    /* the time-embedding path (Timesteps -> MLPs -> add_embedding -> SiLU -> time_emb_proj) in fp32 precision: */
                                   /* fp32 sinusoids, split [hi | lo] operands, fp32 outputs; only the per-resnet time biases are fp16.   */
                                   /* An error there is the same perturbation in every ResnetBlock2D (20 % of SDXL's remaining variance). */
#define ICD_SPLIT_QK        256    /* layers whose probabilities a controller keeps: q and k leave their projections with an error carry and the  */
                                   /* probability kernel computes qh.kh + 2^-14 (ql.kh + qh.kl) (icd_attention_probs_split)                        */
#define ICD_SPLIT_UP_ALL    512    /* ... every upsampler conv (with ICD_SPLIT_UP) */
#define ICD_SPLIT_ALL      1023
#define ICD_SPLIT_ACCURATE 1023    /* the "accurate" level of the Python precision policy: everything.  (Without ICD_SPLIT_UP_ALL: -3 % / -1 % time, */
                                   /* eps 0.390 -> 0.399e-3 / 0.403 -> 0.418e-3, the worst edited store tensor 0.92e-3 -> 0.96e-3: kept for the margin.) */
#define ICD_SPLIT_DEFAULT   447
#define ICD_UNET_OPT_XATTN_FUSION    1
#define ICD_UNET_OPT_LN_INLINE_STATS 2
#define ICD_UNET_OPT_XATTN_TILE      3
#define ICD_UNET_OPT_ATTN_VALU_SCALE 4
int icd_unet_set_option(icd_unet* u, int32_t option, int32_t value);
int64_t icd_unet_workspace_bytes(const icd_unet* u, int32_t batch, int32_t H, int32_t W, int32_t n_ctx);
/* Same for a known materialisation rule of the attention hook: probs_mode 0 = no layer is ever materialised (no hook), 1 =
 * the rule of the reference's shipped controllers (every cross-attention layer and every layer with <= 32^2 queries,
 * utils/p2p.py:147,184-188), 2 = any layer (what icd_unet_workspace_bytes assumes).  Since the one-pass probability kernel
 * (icd_attention_probs) removed the fp32 score tensor the rule no longer moves the arena's peak - all three modes return the
 * same size today; the entry point stays for ABI stability.  Probabilities are the hook's allocations, never arena memory. */
int64_t icd_unet_workspace_bytes_ex(const icd_unet* u, int32_t batch, int32_t H, int32_t W, int32_t n_ctx,
                                    int32_t probs_mode);

typedef struct {
    const void* sample;        /* NCHW [B, in_ch, H, W], fp16 or fp32 (sample_is_f32) */
    const float* timesteps;    /* fp32 [B] (device) */
    const void* context;       /* fp16 [B, n_ctx, cross_dim] */
    const void* timestep_cond; /* fp16 [B, time_cond_proj_dim] or NULL */
    const void* text_embeds;   /* fp16 [B, add_in_dim - 6*addition_time_embed_dim] (SDXL) or NULL */
    const float* time_ids;     /* fp32 [B, 6] (SDXL) or NULL */
    void* eps;                 /* NCHW [B, out_ch, H, W] (same dtype as sample) */
    void* workspace;
    int64_t workspace_bytes;
    int32_t batch, H, W, n_ctx;
    int32_t sample_is_f32;
    icd_attn_hook hook;        /* NULL: every attention layer fused */
    void* hook_user;
    /* Cross-attention K / V^T of every layer depend on the context only - the same tensor at every step of a sampling loop
     * (utils/generation.py:241-244 passes self.context, utils/generation_sdxl.py:445-453 prompt_embeds).  With a caller-owned
     * kv_cache of icd_unet_kv_cache_bytes() the two context projections of a forward are written there, and a later forward with
     * kv_cache_valid != 0 (the caller vouches that `context`, batch and n_ctx are those of the forward that filled it) skips
     * them and reads the cache.  NULL: projections live in the workspace and are recomputed every forward.  Results are
     * bit-identical either way.
     * INVALIDATION: the cache's layout belongs to the options it was filled under.  With ICD_RESIDUAL_SPLIT + ICD_SPLIT_QK the K rows of the
     * layers a controller keeps carry an error byte that the plain levels never write: after icd_unet_set_option(ICD_UNET_OPT_RESIDUAL_MODE
     * / ICD_UNET_OPT_SPLIT_MASK) - and after a change of batch, n_ctx or context - the next forward must run with kv_cache_valid = 0
     * (icd_unet_kv_cache_bytes may also differ: size it again).  The handle cannot check this: a stale `valid` flag reads bytes the
     * filling forward did not write.  (unet.UNet2DConditionModel drops its cache on every precision change.) */
    void* kv_cache;
    int64_t kv_cache_bytes;
    int32_t kv_cache_valid;
} icd_unet_io;

int icd_unet_forward(icd_unet* u, const icd_unet_io* io, void* stream);
int64_t icd_unet_kv_cache_bytes(const icd_unet* u, int32_t batch, int32_t n_ctx);

/* ------------------------------------------------------------------------------------------------------------
 * Per-kernel-family timing of the executor's launches with HIP events recorded on the launch stream (bench.py's
 * roofline leg).  No reference counterpart (the reference has no timing code at all).
 * ---------------------------------------------------------------------------------------------------------- */
#define ICD_PROF_GEMM_CONV    0   /* implicit-GEMM conv3x3 / 1x1 (gemm_kernel<1,*>)            */
#define ICD_PROF_GEMM_DENSE   1   /* Linear layers (gemm_kernel<0,*>, batch 1)                 */
#define ICD_PROF_GEMM_BATCHED 2   /* attention bmm of the materialised-P path                  */
#define ICD_PROF_ATTN_FUSED   3
#define ICD_PROF_GROUPNORM    4
#define ICD_PROF_LAYERNORM    5
#define ICD_PROF_SOFTMAX      6
#define ICD_PROF_MISC         7
#define ICD_PROF_XATTN        8   /* query projection + cross-attention in one launch (gemm.hip xattn_epilogue) */
#define ICD_PROF_KINDS        9
typedef struct {
    int32_t kind;
    int32_t launches;
    double ms;        /* sum of event-to-event durations */
    double flops;     /* ALGORITHMIC flops (2*M*N*K incl. zero-padded taps; 4*B*H*Nq*Nk*d for attention) */
    double bytes;     /* ALGORITHMIC HBM bytes for the bandwidth-bound families */
    double flops_executed;  /* flops issued to the matrix cores: = flops except for the split-operand launches (K doubled by the lo */
                            /* segment: more) and the phase form of the upsampling conv (4 of the 9 taps: fewer)                   */
} icd_profile_row;
/* enable > 0: start a fresh recording of every family; enable < 0: record only the families whose bit is set in
 * (-enable) (bit k = family k; keeps the event overhead out of a timed region); enable == 0: stop.  While enabled every
 * selected executor launch is bracketed by two hipEventRecord calls on its stream. */
int icd_profile_enable(int32_t enable);
/* After the stream has been synchronised: fills rows[0..ICD_PROF_KINDS) and returns the number of rows. */
int icd_profile_read(icd_profile_row* rows, int32_t max_rows);
/* Per-launch records (GEMM: M,N,K, aux = flags or ksize*100+stride*10+upsample; attention: Nq,Nk,d, aux = heads).
 * recs == NULL: returns the number of records available. */
typedef struct {
    int32_t kind, M, N, K, aux;
    float ms;
    double flops;
    int32_t tile_m, tile_n;   /* GEMM launches: the planner's block tile (icd_gemm_plan) */
    int32_t plan_flags;       /* bit 0: gemm_big.hip tile, bit 1: LayerNorm statistics in the main loop, bit 2: fused cross-attention */
    int32_t ksplit;
} icd_profile_record;
int icd_profile_dump(icd_profile_record* recs, int32_t max_recs);
#ifdef __cplusplus
}
#endif
#endif /* ICD_AMD_H */
