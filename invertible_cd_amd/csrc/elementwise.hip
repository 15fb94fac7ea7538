// elementwise.hip - small HBM/latency-bound pieces of the UNet path on gfx950:
//   sinusoidal embeddings, SiLU, conv_in (NCHW latents -> NHWC), conv_out (NHWC -> NCHW eps) and the consistency
//   boundary step predicted_origin (utils/generation.py:136-155).
#include "common.h"

namespace {

// kind 0: diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos(t f_i) || sin(t f_i)],
//         f_i = exp(-ln(1e4) i / half)
// kind 1: guidance_scale_embedding (utils/generation.py:96-122): [sin(1000 w f_i) || cos(..)], f_i = exp(-ln(1e4) i/(half-1))
// OUT: half_t, or float for the precise time-embedding path of the UNet (ICD_SPLIT_TEMB): there the angle is vals * f with f the
// CORRECTLY ROUNDED fp32 frequency (exp in double), so that an fp32 host evaluation of the same expression sees the same angles - at
// t = 999 / time_ids = 1024 one fp32 ulp of f moves the angle by 1e-4, as much as the fp16 rounding this path removes
template <typename OUT>
__global__ void sinusoid_kernel(const float* __restrict__ vals, int n, int dim, int kind, OUT* __restrict__ out) {
    const int half = dim >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * half) return;
    const int r = idx / half, i = idx - r * half;
    const float lg = 9.210340371976184f;   // ln(10000)
    if (kind == 0) {
        const float f = sizeof(OUT) == 4 ? (float)exp((double)(-lg * (float)i / (float)half)) : expf(-lg * (float)i / (float)half);
        const float a = vals[r] * f;
        out[(long long)r * dim + i] = (OUT)cosf(a);
        out[(long long)r * dim + half + i] = (OUT)sinf(a);
    } else {
        const float step = lg / (float)(half - 1);
        const float f = expf((float)i * -step);
        const float a = (vals[r] * 1000.0f) * f;
        out[(long long)r * dim + i] = (OUT)sinf(a);
        out[(long long)r * dim + half + i] = (OUT)cosf(a);
        if ((dim & 1) && i == 0) out[(long long)r * dim + dim - 1] = (OUT)0.f;
    }
}

// out[r] = [hi (C) | lo (C)] of act(x[r]) for an fp32 tensor: hi = fp16(v), lo = fp16(v - hi) - the operand of a GEMM over [hi | lo]
// against [W | W] that sees v to ~2^-22 (the time-embedding MLPs of the precise path: a handful of rows, every resnet downstream)
__global__ void split2_act_kernel(const float* __restrict__ x, long long rows, int C, int act, half_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    float v = x[i];
    if (act == 1) v = v / (1.0f + expf(-v));
    const half_t hi = (half_t)v;
    out[r * 2 * C + c] = hi;
    out[r * 2 * C + C + c] = (half_t)(v - (float)hi);
}

// KIND 0: silu   1: quick_gelu x*sigmoid(1.702x) (CLIP ViT-L text MLP)   2: gelu (erf form; OpenCLIP bigG text MLP)
template <int KIND>
__global__ void activation_kernel(const half_t* __restrict__ x, long long n8, half_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    f16x8 v = *reinterpret_cast<const f16x8*>(x + i * 8), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = (float)v[e];
        o[e] = (half_t)(KIND == 0 ? silu_f(f) : KIND == 1 ? f / (1.0f + __expf(-1.702f * f)) : gelu_erf_f(f));
    }
    *reinterpret_cast<f16x8*>(out + i * 8) = o;
}

// out[b*T + t, :] = tok[ids[b*T + t], :] + pos[t, :]   (CLIPTextEmbeddings); thread = one 8-channel chunk of one token
__global__ void embed_tokens_kernel(const long long* __restrict__ ids, const half_t* __restrict__ tok,
                                    const half_t* __restrict__ pos, long long rows, int T, int C, int vocab,
                                    half_t* __restrict__ out) {
    const int nch = C >> 3;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * nch) return;
    const long long r = i / nch;
    const int c = (int)(i - r * nch) * 8;
    long long id = ids[r];
    id = id < 0 ? 0 : id >= vocab ? vocab - 1 : id;               // the host validates; never read out of bounds
    f16x8 a = *reinterpret_cast<const f16x8*>(tok + id * C + c);
    f16x8 b = *reinterpret_cast<const f16x8*>(pos + (r % T) * C + c), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] + (float)b[e]);
    *reinterpret_cast<f16x8*>(out + r * C + c) = o;
}

// conv_in: thread = (pixel, 8 output channels).  Input NCHW with 4 channels; weights [Cout][3][3][4] fp16.
template <typename TIn>
__global__ __launch_bounds__(256) void conv_in_kernel(const TIn* __restrict__ x, int B, int H, int W,
                                                       const half_t* __restrict__ w, const float* __restrict__ bias,
                                                       int Cout, half_t* __restrict__ out) {
    const int nch = Cout >> 3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * nch;
    if (idx >= total) return;
    const int oc = (int)(idx % nch) * 8;
    const long long pix = idx / nch;
    const int HW = H * W;
    const int b = (int)(pix / HW), rem = (int)(pix - (long long)b * HW);
    const int y = rem / W, xx = rem - y * W;
    float in[36];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xc = xx + t % 3 - 1;
        const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xc < (unsigned)W;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            in[t * 4 + c] = ok ? (float)x[((long long)(b * 4 + c) * H + yy) * W + xc] : 0.f;
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const half_t* wr = w + (long long)(oc + e) * 36;
        float acc = bias ? bias[oc + e] : 0.f;
#pragma unroll
        for (int k = 0; k < 36; k += 4) {
            f16x4 wv = *reinterpret_cast<const f16x4*>(wr + k);
            acc += in[k] * (float)wv[0] + in[k + 1] * (float)wv[1] + in[k + 2] * (float)wv[2] + in[k + 3] * (float)wv[3];
        }
        o[e] = (half_t)acc;
    }
    *reinterpret_cast<f16x8*>(out + pix * Cout + oc) = o;
}

// NCHW [B,4,H,W] latents -> token-major [B*H*W, 8] fp16 (channels 4..7 zero) so that conv_in can run on the MFMA
// implicit-GEMM path (K = 9 * 8 = 72).
template <typename TIn>
__global__ void pack_latent_kernel(const TIn* __restrict__ x, int B, int C, int HW, int ones_ch, half_t* __restrict__ out) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (long long)B * HW) return;
    const int b = (int)(pix / HW), r = (int)(pix - (long long)b * HW);
    f16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        o[c] = c < C ? (half_t)(float)x[((long long)b * C + c) * HW + r] : (half_t)(c == ones_ch ? 1.f : 0.f);
    *reinterpret_cast<f16x8*>(out + pix * 8) = o;
}

// conv_out: LPP lanes per output pixel (64 / LPP pixels per wave) stride over 8-channel chunks; up to 4 output
// channels, NCHW result.  w: [4][3][3][Cin] (rows >= Cout are ignored).  LPP = 64 for the UNet's 320 input channels,
// 16 for the VAE decoder's 128 (a full wave per pixel would leave 48 lanes idle on 512 x 512 images).
template <typename TOut, int LPP>
__global__ __launch_bounds__(256) void conv_out_kernel(const half_t* __restrict__ x, int B, int H, int W, int Cin,
                                                        const half_t* __restrict__ w, const float* __restrict__ bias,
                                                        int Cout, TOut* __restrict__ eps) {
    constexpr int PPW = 64 / LPP;
    const int l = threadIdx.x & 63, sub = l % LPP;
    const long long pix = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + l / LPP;
    const int HW = H * W;
    const bool live = pix < (long long)B * HW;
    const long long pc = live ? pix : 0;
    const int b = (int)(pc / HW), rem = (int)(pc - (long long)b * HW);
    const int y = rem / W, xx = rem - y * W;
    const int nch = Cin >> 3;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xc = xx + t % 3 - 1;
        if (!live || (unsigned)yy >= (unsigned)H || (unsigned)xc >= (unsigned)W) continue;
        const half_t* xp = x + ((long long)b * HW + yy * W + xc) * Cin;
        for (int c = sub; c < nch; c += LPP) {
            f16x8 v = *reinterpret_cast<const f16x8*>(xp + c * 8);
            // v_dot2_f32_f16 (two fp16 products, fp32 accumulate) instead of 8 converts + 8 FMAs per weight chunk: 220 -> 169 us per
            // 131072 pixels x 320 channels.  Still latency-bound on its 45 loads per pixel: a form with the weights in registers and 16
            // pixels per wave was no faster (217 us, serial per pixel); the real fix is an MFMA tile over a halo patch
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                f16x8 wv = *reinterpret_cast<const f16x8*>(w + ((long long)(o * 9 + t)) * Cin + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[o] = __builtin_amdgcn_fdot2((h2){v[2 * e], v[2 * e + 1]}, (h2){wv[2 * e], wv[2 * e + 1]}, acc[o], false);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = group_sum<LPP>(acc[o]);
    if (live && sub < Cout) {
        const float v = (sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3]) + (bias ? bias[sub] : 0.f);
        eps[((long long)(b * Cout + sub) * H + y) * W + xx] = (TOut)v;
    }
}

// conv_out on the matrix cores (Cin % 32 == 0): D[16 output channels (4 real)][16 pixels] += W[16 x 32 k] . X[32 k x 16 pixels] with
// v_mfma_f32_16x16x32_f16.  The B operand is read straight from the NHWC activations - lane (pixel n = l & 15, k-slot g = l >> 4) loads the
// 8 channels c0 + 8g.. of its pixel's tap (16 B, the zero page outside the image); the A operand comes from an LDS copy of the 4 x 9 x Cin
// weights (lanes of rows 4..15 read a zero chunk).  A wave owns Q 16-pixel tiles of consecutive pixels that share every weight
// fragment (Q tiles per wave, UN k-chunks of loads in flight); 9 taps x Cin / 32 MFMAs per tile.  The VALU form above spends 169 us on
// 131072 pixels x 320 channels (45 loads and 144 v_dot2 per pixel and wave), this one 103 us (28 against 44 us at 32768 pixels).
template <typename TOut, int Q, int UN>
__global__ __launch_bounds__(256) void conv_out_mfma_kernel(const half_t* __restrict__ x, int B, int H, int W, int Cin,
                                                             const half_t* __restrict__ w, const float* __restrict__ bias,
                                                             int Cout, TOut* __restrict__ eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* wl = reinterpret_cast<half_t*>(smem_raw);          // [9 * ncc][4 rows][32 halves], then one zero chunk of 8 halves
    const int tid = threadIdx.x, l = tid & 63, n = l & 15, g = l >> 4;
    const int ncc = Cin >> 5, HW = H * W;
    const long long total = (long long)B * HW;
    // stage the weights: w is [4][9][Cin]; 16-B chunks, chunk index = ((t * ncc + cc) * 4 + o) * 4 + j (j: 8-half slice of the 32)
    for (int i = tid; i < 9 * ncc * 16; i += 256) {
        const int j = i & 3, o = (i >> 2) & 3, kc = i >> 4, t = kc / ncc, cc = kc - t * ncc;
        // (w holds 4 rows by contract, icd_conv_out_n; rows >= Cout are not trusted to exist or to be zero: staged as zeros)
        f16x8 wv;
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = (half_t)0.f;
        if (o < Cout) wv = *reinterpret_cast<const f16x8*>(w + ((long long)(o * 9 + t)) * Cin + cc * 32 + j * 8);
        *reinterpret_cast<f16x8*>(wl + (long long)i * 8) = wv;
    }
    if (tid == 0) {
        f16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (half_t)0.f;
        *reinterpret_cast<f16x8*>(wl + (long long)9 * ncc * 128) = z;
    }
    __syncthreads();
    const long long p0 = ((long long)blockIdx.x * 4 + (tid >> 6)) * (16 * Q);
    if (p0 >= total) return;
    const half_t* zero = reinterpret_cast<const half_t*>(icd_zero_page);
    int pb[Q], py[Q], px[Q];
    bool live[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const long long pix = p0 + q * 16 + n;
        live[q] = pix < total;
        const long long pc = live[q] ? pix : 0;
        pb[q] = (int)(pc / HW);
        const int rem = (int)(pc - (long long)pb[q] * HW);
        py[q] = rem / W; px[q] = rem - py[q] * W;
    }
    f32x4 acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A-fragment address of this lane inside one k-chunk: rows >= 4 read the zero chunk
    const int a_off = n < 4 ? (n * 4 + g) * 8 : -1;
    const half_t* a_zero = wl + (long long)9 * ncc * 128;
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const half_t* src[Q];
        int inc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int yy = py[q] + dy, xc = px[q] + dx;
            const bool ok = live[q] && (unsigned)yy < (unsigned)H && (unsigned)xc < (unsigned)W;
            src[q] = ok ? x + ((long long)pb[q] * HW + yy * W + xc) * Cin + g * 8 : zero;
            inc[q] = ok ? 32 : 0;
        }
        const half_t* ap = a_off >= 0 ? wl + (long long)t * ncc * 128 + a_off : a_zero;
        const int a_inc = a_off >= 0 ? 128 : 0;
        for (int c0 = 0; c0 < ncc; c0 += UN) {                 // UN k-chunks = UN * Q activation loads in flight per lane
            f16x8 a[UN], bq[UN][Q];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool on = c0 + u < ncc;
                a[u] = *reinterpret_cast<const f16x8*>(on ? ap + u * a_inc : a_zero);
#pragma unroll
                for (int q = 0; q < Q; ++q) bq[u][q] = *reinterpret_cast<const f16x8*>(on ? src[q] + u * inc[q] : zero);
            }
            ap += UN * a_inc;
#pragma unroll
            for (int q = 0; q < Q; ++q) src[q] += UN * inc[q];
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], bq[u][q], acc[q], 0, 0, 0);
        }
    }
    if (g == 0) {                                               // rows 0..3 of D = the output channels of pixel n
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (!live[q]) continue;
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < Cout) eps[((long long)(pb[q] * Cout + o) * H + py[q]) * W + px[q]] = (TOut)(acc[q][o] + (bias ? bias[o] : 0.f));
        }
    }
}

// predicted_origin, eps-prediction.  Evaluation order and roundings mirror the reference's fp32 torch expression
// (no FMA contraction): x0 = (x - sigma_t*eps) / alpha_t ; out = alpha_s*x0 + sigma_s*eps.
template <typename TX, typename TE, typename TO>
__global__ void x0_step_kernel(const TX* __restrict__ x, const TE* __restrict__ eps, const float* __restrict__ coef,
                               long long per_sample, long long total, TO* __restrict__ out) {
#pragma clang fp contract(off)      // hipcc contracts a*b+c into FMA by default; the reference rounds every op
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / per_sample);
    const float a_t = coef[b * 4 + 0], s_t = coef[b * 4 + 1], a_s = coef[b * 4 + 2], s_s = coef[b * 4 + 3];
    const float xv = (float)x[i], ev = (float)eps[i];
    const float prod = s_t * ev;
    const float diff = xv - prod;
    const float x0 = diff / a_t;            // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt is the default)
    const float p1 = a_s * x0;
    const float p2 = s_s * ev;
    out[i] = (TO)(p1 + p2);
}

}  // namespace

extern "C" int icd_sinusoid(const float* vals, int32_t n, int32_t dim, int32_t kind, void* out, void* stream) {
    ICD_CHECK_ARG(vals && out && n > 0 && dim >= 2, "icd_sinusoid: bad arguments");
    ICD_CHECK_ARG(kind == 0 || kind == 1, "icd_sinusoid: kind must be 0 or 1");
    ICD_CHECK_ARG(kind == 1 || dim % 2 == 0, "icd_sinusoid: Timesteps dim must be even");
    const int total = n * (dim / 2);
    hipLaunchKernelGGL(sinusoid_kernel<half_t>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, vals, n, dim, kind,
                       (half_t*)out);
    ICD_CHECK_LAUNCH("icd_sinusoid");
    return ICD_OK;
}

extern "C" int icd_sinusoid_f32(const float* vals, int32_t n, int32_t dim, int32_t kind, float* out, void* stream) {
    ICD_CHECK_ARG(vals && out && n > 0 && dim >= 2, "icd_sinusoid_f32: bad arguments");
    ICD_CHECK_ARG(kind == 0 || kind == 1, "icd_sinusoid_f32: kind must be 0 or 1");
    ICD_CHECK_ARG(kind == 1 || dim % 2 == 0, "icd_sinusoid_f32: Timesteps dim must be even");
    const int total = n * (dim / 2);
    hipLaunchKernelGGL(sinusoid_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, vals, n, dim, kind, out);
    ICD_CHECK_LAUNCH("icd_sinusoid_f32");
    return ICD_OK;
}

extern "C" int icd_split2_act(const float* x, int64_t rows, int32_t C, int32_t act, void* out, void* stream) {
    ICD_CHECK_ARG(x && out && rows > 0 && C > 0 && (act == 0 || act == 1), "icd_split2_act: bad arguments (act: 0 none, 1 silu)");
    const long long total = (long long)rows * C;
    hipLaunchKernelGGL(split2_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, C, act,
                       (half_t*)out);
    ICD_CHECK_LAUNCH("icd_split2_act");
    return ICD_OK;
}

extern "C" int icd_silu(const void* x, int64_t n, void* out, void* stream) {
    ICD_CHECK_ARG(x && out && n > 0 && n % 8 == 0, "icd_silu: n must be a positive multiple of 8");
    const long long n8 = n / 8;
    hipLaunchKernelGGL(activation_kernel<0>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, n8, (half_t*)out);
    ICD_CHECK_LAUNCH("icd_silu");
    return ICD_OK;
}

extern "C" int icd_activation(const void* x, int64_t n, int32_t kind, void* out, void* stream) {
    ICD_CHECK_ARG(x && out && n > 0 && n % 8 == 0, "icd_activation: n must be a positive multiple of 8");
    ICD_CHECK_ARG(kind >= 0 && kind <= 2, "icd_activation: kind must be 0 (silu), 1 (quick_gelu) or 2 (gelu)");
    const long long n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(activation_kernel<0>, grid, dim3(256), 0, st, (const half_t*)x, n8, (half_t*)out);
    else if (kind == 1) hipLaunchKernelGGL(activation_kernel<1>, grid, dim3(256), 0, st, (const half_t*)x, n8, (half_t*)out);
    else hipLaunchKernelGGL(activation_kernel<2>, grid, dim3(256), 0, st, (const half_t*)x, n8, (half_t*)out);
    ICD_CHECK_LAUNCH("icd_activation");
    return ICD_OK;
}

extern "C" int icd_embed_tokens(const int64_t* ids, const void* tok_emb, const void* pos_emb, int64_t rows, int32_t T,
                                int32_t C, int32_t vocab, void* out, void* stream) {
    ICD_CHECK_ARG(ids && tok_emb && pos_emb && out, "icd_embed_tokens: null pointer");
    ICD_CHECK_ARG(rows > 0 && T > 0 && rows % T == 0 && C > 0 && C % 8 == 0 && vocab > 0, "icd_embed_tokens: bad shape");
    const long long total = (long long)rows * (C / 8);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const long long*)ids, (const half_t*)tok_emb, (const half_t*)pos_emb, (long long)rows, T, C, vocab,
                       (half_t*)out);
    ICD_CHECK_LAUNCH("icd_embed_tokens");
    return ICD_OK;
}

extern "C" int icd_conv_in(const void* x_nchw, int32_t x_is_f32, int32_t B, int32_t H, int32_t W, const void* w,
                           const float* bias, int32_t Cout, void* out, void* stream) {
    ICD_CHECK_ARG(x_nchw && w && out, "icd_conv_in: null pointer");
    ICD_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0, "icd_conv_in: bad shape");
    const long long total = (long long)B * H * W * (Cout / 8);
    dim3 grid((unsigned)((total + 255) / 256));
    if (x_is_f32)
        hipLaunchKernelGGL(conv_in_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x_nchw, B, H, W,
                           (const half_t*)w, bias, Cout, (half_t*)out);
    else
        hipLaunchKernelGGL(conv_in_kernel<half_t>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)x_nchw, B, H,
                           W, (const half_t*)w, bias, Cout, (half_t*)out);
    ICD_CHECK_LAUNCH("icd_conv_in");
    return ICD_OK;
}

extern "C" int icd_pack_nchw(const void* x_nchw, int32_t x_is_f32, int32_t B, int32_t C, int32_t HW, int32_t ones_channel,
                             void* out, void* stream) {
    ICD_CHECK_ARG(x_nchw && out && B > 0 && HW > 0, "icd_pack_nchw: bad arguments");
    ICD_CHECK_ARG(C > 0 && C <= 8 && ones_channel < 8 && (ones_channel < 0 || ones_channel >= C),
                  "icd_pack_nchw: C must be 1..8 and the ones channel one of the padding channels (C=%d, ones=%d)", C, ones_channel);
    const long long pixels = (long long)B * HW;
    dim3 grid((unsigned)((pixels + 255) / 256));
    if (x_is_f32)
        hipLaunchKernelGGL(pack_latent_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x_nchw, B, C, HW,
                           ones_channel, (half_t*)out);
    else
        hipLaunchKernelGGL(pack_latent_kernel<half_t>, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)x_nchw, B, C, HW,
                           ones_channel, (half_t*)out);
    ICD_CHECK_LAUNCH("icd_pack_nchw");
    return ICD_OK;
}

extern "C" int icd_pack_latent(const void* x_nchw, int32_t x_is_f32, int32_t B, int32_t HW, void* out, void* stream) {
    return icd_pack_nchw(x_nchw, x_is_f32, B, 4, HW, -1, out, stream);
}

extern "C" int icd_conv_out_n(const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w, const float* bias,
                              int32_t Cout, void* out_nchw, int32_t out_is_f32, void* stream) {
    ICD_CHECK_ARG(x && w && out_nchw, "icd_conv_out: null pointer");
    ICD_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0, "icd_conv_out: bad shape");
    ICD_CHECK_ARG(Cout >= 1 && Cout <= 4, "icd_conv_out: Cout must be 1..4 (got %d)", Cout);
    const long long pixels = (long long)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    const int nch = Cin / 8;
    if (Cin % 32 == 0 && Cin >= 256 && (size_t)Cin * 72 + 16 <= 64 * 1024) {       // matrix-core form (narrower inputs: the VALU form is faster)
        const size_t smem = (size_t)Cin * 72 + 16;                     // 9 * Cin / 32 k-chunks x 4 rows x 64 B, + the zero chunk
        // two 16-pixel tiles per wave, five k-chunks (ten 16-B activation loads per lane) in flight: of the forms tried the best at 32768
        // and 131072 pixels (profiles/r03_conv_out.txt).  All of them stop at ~104 us for 131072 pixels x 320 channels: the nine taps
        // re-read every activation row through the L2 (755 MB); a halo patch in LDS would be the next step
        const dim3 grid((unsigned)((pixels + 127) / 128));
        if (out_is_f32) hipLaunchKernelGGL((conv_out_mfma_kernel<float, 2, 5>), grid, dim3(256), smem, st, (const half_t*)x, B, H, W, Cin,
                                           (const half_t*)w, bias, Cout, (float*)out_nchw);
        else hipLaunchKernelGGL((conv_out_mfma_kernel<half_t, 2, 5>), grid, dim3(256), smem, st, (const half_t*)x, B, H, W, Cin,
                                (const half_t*)w, bias, Cout, (half_t*)out_nchw);
        ICD_CHECK_LAUNCH("icd_conv_out(mfma)");
        return ICD_OK;
    }
#define CO_LAUNCH(T, LPP)                                                                                              \
    hipLaunchKernelGGL((conv_out_kernel<T, LPP>), dim3((unsigned)((pixels + 4 * (64 / LPP) - 1) / (4 * (64 / LPP)))),  \
                       dim3(256), 0, st, (const half_t*)x, B, H, W, Cin, (const half_t*)w, bias, Cout, (T*)out_nchw)
    if (nch <= 8) { if (out_is_f32) CO_LAUNCH(float, 8); else CO_LAUNCH(half_t, 8); }
    else if (nch <= 16) { if (out_is_f32) CO_LAUNCH(float, 16); else CO_LAUNCH(half_t, 16); }
    else if (nch <= 32) { if (out_is_f32) CO_LAUNCH(float, 32); else CO_LAUNCH(half_t, 32); }
    else { if (out_is_f32) CO_LAUNCH(float, 64); else CO_LAUNCH(half_t, 64); }
#undef CO_LAUNCH
    ICD_CHECK_LAUNCH("icd_conv_out");
    return ICD_OK;
}

extern "C" int icd_conv_out(const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w,
                            const float* bias, void* eps_nchw, int32_t out_is_f32, void* stream) {
    return icd_conv_out_n(x, B, H, W, Cin, w, bias, 4, eps_nchw, out_is_f32, stream);
}

extern "C" int icd_x0_step(const void* x, const void* eps, const float* coef, int32_t B, int64_t per_sample,
                           int32_t dtype_flags, void* out, void* stream) {
    ICD_CHECK_ARG(x && eps && coef && out && B > 0 && per_sample > 0, "icd_x0_step: bad arguments");
    ICD_CHECK_ARG(dtype_flags >= 0 && dtype_flags < 8, "icd_x0_step: bad dtype flags");
    const long long total = (long long)B * per_sample;
    dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define X0_CASE(F, TX, TE, TO)                                                                                       \
    case F:                                                                                                          \
        hipLaunchKernelGGL((x0_step_kernel<TX, TE, TO>), grid, blk, 0, st, (const TX*)x, (const TE*)eps, coef,      \
                           (long long)per_sample, total, (TO*)out);                                                  \
        break;
    switch (dtype_flags) {
        X0_CASE(0, half_t, half_t, half_t)
        X0_CASE(1, float, half_t, half_t)
        X0_CASE(2, half_t, float, half_t)
        X0_CASE(3, float, float, half_t)
        X0_CASE(4, half_t, half_t, float)
        X0_CASE(5, float, half_t, float)
        X0_CASE(6, half_t, float, float)
        X0_CASE(7, float, float, float)
    }
#undef X0_CASE
    ICD_CHECK_LAUNCH("icd_x0_step");
    return ICD_OK;
}
