"""Block-level operators of the UNet path: thin torch-tensor front ends over the C ABI (include/icd_amd.h).

torch is used here for device memory and streams only; every arithmetic op is a HIP kernel in libicd_amd.so.
Activations are fp16, token-major ("NHWC"): a feature map is a [B, H*W, C] tensor.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import GemmDesc, ICD_GEMM_GEGLU, ICD_GEMM_OUT_F32, ICD_GEMM_OUT_TRANS, ICD_GEMM_PAD_HI, ICD_GEMM_RESID_F32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk16(t, name):
    assert t.is_cuda and t.dtype == torch.float16 and t.is_contiguous(), f"{name}: need contiguous cuda fp16"


# ------------------------------------------------------------------------------------------------ packing helpers
def _chk_rows(t, name):
    """2-D fp16 operand that may be a column slice of a wider matrix (q / k halves of a fused projection)."""
    assert t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 \
        and t.data_ptr() % 16 == 0, f"{name}: need a row-major cuda fp16 matrix with 16-byte aligned rows"


def pack_conv_weight(w_oihw):
    """[O, I, kh, kw] -> [O, kh*kw*I] fp16 (tap-major, channel-minor: the K order of the implicit GEMM)."""
    o = w_oihw.shape[0]
    return w_oihw.permute(0, 2, 3, 1).reshape(o, -1).to(torch.float16).contiguous()


def geglu_perm(n_out):
    """Row permutation of ff.net.0.proj ([2*n_out, C]) so that packed rows alternate 32 'value' / 32 'gate' rows."""
    assert n_out % 32 == 0
    idx = torch.arange(2 * n_out)
    blk, within = idx // 32, idx % 32
    return torch.where(blk % 2 == 0, (blk // 2) * 32 + within, n_out + (blk // 2) * 32 + within)


# ------------------------------------------------------------------------------------------------ operators
def _splitk_ws(d, device):
    """Attach split-K scratch when the shape wants it (small M, deep K); returns the tensor to keep it alive."""
    n = _lib.load().icd_gemm_workspace_bytes(d.M, d.N, d.K)
    if n <= 0:
        return None
    ws = torch.empty((n,), dtype=torch.uint8, device=device)
    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), n
    return ws


FORCE_BIG_TILE = 0x100000      # tuning / test override understood by icd_gemm: take the 256x256 tile path regardless of size
FORBID_BIG_TILE = 0x200000


def gemm(a, w, bias=None, resid=None, rowbias=None, rows_per_sample=0, geglu=False, out=None, alpha=1.0,
         out_f32=False, debug_flags=0, ln_stats=None, ln_colsum=None, ln_compute=False, ln_eps=1e-5, group_m=0, timeline=None,
         out32=None, resid_carry=None, out_carry=None):
    """out[M, N] = alpha * a[M, K] @ w[N, K]^T (+bias[N] fp32) (+rowbias[m // rps]) (+resid) ; GEGLU halves N.
    ln_stats [M, 2] fp32 + ln_colsum [N] fp32: LayerNorm of `a` fused into the epilogue (w carries gamma, bias carries W beta).
    resid_carry / out_carry: uint8 [M, N] error carries of `resid` / `out` (icd_gemm_desc.resid_carry / out_carry; carry_decode)."""
    _chk16(a, "a"); _chk16(w, "w")
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=torch.float32 if out_f32 else torch.float16)
    d = GemmDesc()
    d.a0, d.w, d.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.rowbias = rowbias.data_ptr() if rowbias is not None else None
    d.M, d.N, d.K, d.Nw = M, N, K, N
    d.lda, d.ldw, d.ldo = a.stride(0), w.stride(0), out.stride(0)
    d.ldr = resid.stride(0) if resid is not None else 0
    d.ld_rowbias = rowbias.stride(0) if rowbias is not None else 0
    d.rows_per_sample = rows_per_sample
    d.mode, d.batch, d.zdiv, d.alpha = 0, 1, 1, alpha
    d.flags = (ICD_GEMM_GEGLU if geglu else 0) | (ICD_GEMM_OUT_F32 if out_f32 else 0) | debug_flags
    if resid is not None and resid.dtype == torch.float32:
        d.flags |= ICD_GEMM_RESID_F32
    if ln_stats is not None:
        assert ln_stats.dtype == torch.float32 and ln_stats.is_contiguous() and tuple(ln_stats.shape) == (M, 2)
        assert ln_colsum.dtype == torch.float32 and ln_colsum.is_contiguous() and ln_colsum.numel() == N
        d.ln_stats, d.ln_colsum = ln_stats.data_ptr(), ln_colsum.data_ptr()
        if ln_compute:                        # ln_stats is filled by this launch (ICD_GEMM_LN_COMPUTE), not read
            d.flags |= _lib.ICD_GEMM_LN_COMPUTE
            d.ln_eps = ln_eps
    if out32 is not None:                     # second output: the same values before the fp16 rounding (fp32 residual stream)
        assert out32.dtype == torch.float32 and out32.is_cuda and tuple(out32.shape) == (M, n_out) and out32.stride(0) == out.stride(0)
        d.out_f32 = out32.data_ptr()
    _set_carry(d, resid, resid_carry, out, out_carry)
    d.tune_group_m = group_m                  # tuning / diagnostics travel in the descriptor (no process-wide state)
    d.debug_timeline = timeline.data_ptr() if timeline is not None else None
    ws = _splitk_ws(d, a.device)
    _lib.check(_lib.load().icd_gemm(C.byref(d), _stream()), "icd_gemm")
    return out


def _set_carry(d, resid, resid_carry, out, out_carry):
    if resid_carry is not None:
        assert resid is not None and resid_carry.dtype == torch.uint8 and resid_carry.is_cuda and resid_carry.shape == resid.shape
        assert resid_carry.stride(0) == resid.stride(0)
        d.resid_carry = resid_carry.data_ptr()
    if out_carry is not None:
        assert out_carry.dtype == torch.uint8 and out_carry.is_cuda and out_carry.shape == out.shape and out_carry.stride(0) == out.stride(0)
        d.out_carry = out_carry.data_ptr()


CARRY_SCALE = 16384.0


def carry_decode(hi, carry):
    """fp32 value of a carried tensor: fp16 `hi` + 2^-14 * bf8_e5m2 `carry` (uint8), see icd_gemm_desc.resid_carry."""
    return hi.float() + carry.view(torch.float8_e5m2).float() / CARRY_SCALE


def carry_encode(v):
    """(hi, carry) of an fp32 tensor, as the GEMM epilogues produce them (round to nearest even twice)."""
    hi = v.half()
    return hi, ((v.float() - hi.float()) * CARRY_SCALE).clamp(-57344.0, 57344.0).to(torch.float8_e5m2).view(torch.uint8)


def conv3x3(x, B, H, W, w_packed, bias=None, x2=None, stride=1, upsample=False, resid=None, rowbias=None, ksize=3,
            debug_flags=0, pad_hi=False, out_f32=False, alpha=1.0, resid_carry=None, out_carry=None, phase=None, out=None):
    """Implicit-GEMM conv over NHWC x [B*H*W, C0] (optionally cat with x2 [.., C1]); returns [B*Ho*Wo, Cout].
    pad_hi: zero padding on the bottom/right edge only (AutoencoderKL Downsample2D).  out_f32 / an fp32 `resid` / alpha: the
    fp32-fidelity VAE path (fp32 conv outputs and residual stream, power-of-two input scaling undone by alpha)."""
    _chk16(x, "x"); _chk16(w_packed, "w")
    C0 = x.shape[-1]
    C1 = x2.shape[-1] if x2 is not None else 0
    Hu, Wu = (H * 2, W * 2) if upsample else (H, W)
    Ho, Wo = (Hu + stride - 1) // stride, (Wu + stride - 1) // stride
    N = w_packed.shape[0]
    if phase is None:
        out = torch.empty((B * Ho * Wo, N), device=x.device, dtype=torch.float32 if out_f32 else torch.float16)
    else:       # one pixel phase (2 py + px) of the upsampling conv: writes its quarter of `out` [B * 2H * 2W, N] (icd_gemm_desc.conv_ktaps)
        assert ksize == 3 and stride == 1 and not upsample and out is not None and out.shape == (B * 4 * H * W, N) and out.is_contiguous()
    d = GemmDesc()
    d.a0, d.a1, d.w, d.out = x.data_ptr(), (x2.data_ptr() if x2 is not None else None), w_packed.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.rowbias = rowbias.data_ptr() if rowbias is not None else None
    d.M, d.N, d.K, d.Nw = B * Ho * Wo, N, (4 if phase is not None else ksize * ksize) * (C0 + C1), N
    if phase is not None:
        d.conv_tap_base, d.conv_ktaps, d.out_remap_w, d.out_remap_c = 3 * (phase >> 1) + (phase & 1), 4, W, 2 * (phase >> 1) * W + (phase & 1)
    d.lda, d.ldw, d.ldo = 0, w_packed.stride(0), N
    d.ldr = resid.stride(0) if resid is not None else 0
    d.ld_rowbias = rowbias.stride(0) if rowbias is not None else 0
    d.rows_per_sample = Ho * Wo
    d.mode, d.C0, d.C1 = 1, C0, C1
    d.Hin, d.Win, d.Hout, d.Wout, d.ksize, d.stride, d.upsample = H, W, Ho, Wo, ksize, stride, int(upsample)
    d.batch, d.zdiv, d.alpha, d.flags = 1, 1, alpha, debug_flags | (ICD_GEMM_PAD_HI if pad_hi else 0) | (ICD_GEMM_OUT_F32 if out_f32 else 0)
    if resid is not None and resid.dtype == torch.float32:
        d.flags |= ICD_GEMM_RESID_F32
    _set_carry(d, resid, resid_carry, out, out_carry)
    ws = _splitk_ws(d, x.device)
    _lib.check(_lib.load().icd_gemm(C.byref(d), _stream()), "icd_gemm(conv)")
    return out


def groupnorm(x, B, HW, gamma, beta, eps, silu, x2=None, groups=32):
    _chk16(x, "x")
    C0 = x.shape[-1]
    C1 = x2.shape[-1] if x2 is not None else 0
    lib = _lib.load()
    ws = torch.empty((lib.icd_groupnorm_ws_floats(B, HW, groups),), device=x.device, dtype=torch.float32)
    out = torch.empty((B * HW, C0 + C1), device=x.device, dtype=torch.float16)
    _lib.check(lib.icd_groupnorm(_p(x), C0, _p(x2), C1, B, HW, groups, _p(gamma), _p(beta), eps, int(silu), _p(out),
                                 _p(ws), _stream()), "icd_groupnorm")
    return out


def groupnorm_carry(x, B, HW, gamma, beta, eps, silu, carry=None, x2=None, carry2=None, groups=32, with_aux=False):
    """GroupNorm (+SiLU) of carried tensors (icd_groupnorm_carry): normalises fp16 + 2^-14 * bf8 carry.  with_aux: also returns the
    second source of a split shortcut conv, fp16 [B*HW, C0 + 2 C1] = [x2 | lo | lo2] ([lo] without x2)."""
    _chk16(x, "x")
    C0 = x.shape[-1]
    C1 = x2.shape[-1] if x2 is not None else 0
    lib = _lib.load()
    ws = torch.empty((lib.icd_groupnorm_ws_floats(B, HW, groups),), device=x.device, dtype=torch.float32)
    out = torch.empty((B * HW, C0 + C1), device=x.device, dtype=torch.float16)
    aux = torch.empty((B * HW, C0 + 2 * C1), device=x.device, dtype=torch.float16) if with_aux else None
    for c, t in ((carry, x), (carry2, x2)):
        assert c is None or (c.dtype == torch.uint8 and c.is_cuda and c.is_contiguous() and c.shape == t.shape)
    _lib.check(lib.icd_groupnorm_carry(_p(x), C0, _p(carry), _p(x2), C1, _p(carry2), B, HW, groups, _p(gamma), _p(beta), eps, int(silu),
                                       _p(out), _p(aux), C0 + 2 * C1 if with_aux else 0, _p(ws), _stream()), "icd_groupnorm_carry")
    return (out, aux) if with_aux else out


def carry_expand(carry):
    """lo = fp16(2^-14 * bf8 carry) (icd_carry_expand): the second K segment of a split-operand GEMM."""
    assert carry.dtype == torch.uint8 and carry.is_cuda and carry.is_contiguous() and carry.numel() % 8 == 0
    lo = torch.empty(carry.shape, device=carry.device, dtype=torch.float16)
    _lib.check(_lib.load().icd_carry_expand(_p(carry), carry.numel(), _p(lo), _stream()), "icd_carry_expand")
    return lo


def groupnorm_f32_split(x32, B, HW, gamma, beta, eps, silu, groups=32):
    """GroupNorm (+SiLU) of an fp32 [B*HW, C] tensor -> split3 fp16 [B*HW, 3C] = [hi | lo | hi] (fp32-fidelity VAE path)."""
    assert x32.is_cuda and x32.dtype == torch.float32 and x32.is_contiguous()
    Cc = x32.shape[-1]
    lib = _lib.load()
    ws = torch.empty((lib.icd_groupnorm_ws_floats(B, HW, groups),), device=x32.device, dtype=torch.float32)
    out = torch.empty((B * HW, 3 * Cc), device=x32.device, dtype=torch.float16)
    _lib.check(lib.icd_groupnorm_f32_split(_p(x32), Cc, B, HW, groups, _p(gamma), _p(beta), eps, int(silu), _p(out), _p(ws), _stream()),
               "icd_groupnorm_f32_split")
    return out


def split_cast(x32, scale=1.0):
    """fp32 [rows, C] * scale -> split3 fp16 [rows, 3C] = [hi | lo | hi]."""
    assert x32.is_cuda and x32.dtype == torch.float32 and x32.is_contiguous() and x32.dim() == 2
    rows, Cc = x32.shape
    out = torch.empty((rows, 3 * Cc), device=x32.device, dtype=torch.float16)
    _lib.check(_lib.load().icd_split_cast(_p(x32), rows, Cc, scale, _p(out), _stream()), "icd_split_cast")
    return out


def absmax(x32):
    """max |x| of an fp32 tensor as a Python float (one small kernel + a host sync: only the fp32-fidelity VAE path uses it)."""
    assert x32.is_cuda and x32.dtype == torch.float32 and x32.is_contiguous()
    out = torch.empty((1,), device=x32.device, dtype=torch.float32)
    _lib.check(_lib.load().icd_absmax(_p(x32), x32.numel(), _p(out), _stream()), "icd_absmax")
    return float(out.item())


def split_cast_guarded(x32, headroom=8192.0):
    """split3 of a raw (un-normalised) fp32 tensor with a power-of-two scale chosen so that max |x| * scale <= headroom (well
    inside the fp16 range, and never below 1 for tensors that are already small): returns (split3 tensor, 1 / scale)."""
    import math
    m = absmax(x32)
    if not math.isfinite(m):
        raise FloatingPointError("non-finite activation in the fp32-fidelity path")
    scale = 1.0 if m <= headroom else 2.0 ** -math.ceil(math.log2(m / headroom))
    return split_cast(x32, scale), 1.0 / scale


def split_weight(w2d):
    """fp32 [N, K] (or packed conv [N, taps*Cin] viewed as [N, taps, Cin]) -> fp16 [.., 3*Cin] = [w_hi | w_hi | w_lo] per tap."""
    w32 = w2d.float()
    hi = w32.to(torch.float16)
    lo = (w32 - hi.float()).to(torch.float16)
    return torch.cat([hi, hi, lo], dim=-1).contiguous()


def layernorm(x, gamma, beta, eps=1e-5):
    _chk16(x, "x")
    rows, Cc = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().icd_layernorm(_p(x), rows, Cc, _p(gamma), _p(beta), eps, _p(out), _stream()), "icd_layernorm")
    return out


def layernorm_stats(x, eps=1e-5):
    """(mean, rstd) per row of x [rows, C] -> fp32 [rows, 2]; the normalisation is applied by gemm(..., ln_stats=)."""
    _chk16(x, "x")
    rows, Cc = x.shape
    out = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().icd_layernorm_stats(_p(x), rows, Cc, eps, _p(out), _stream()), "icd_layernorm_stats")
    return out


def fold_layernorm(w, gamma, beta, bias=None):
    """Weights of a Linear that consumes LayerNorm(x; gamma, beta): (fp16 W * gamma, fp32 row sums of it, fp32 W beta + bias)."""
    w32 = w.float()
    w16 = (w32 * gamma.float()[None, :]).to(torch.float16).contiguous()
    t = w32 @ beta.float() + (0 if bias is None else bias.float())
    return w16, w16.float().sum(1).contiguous(), t.contiguous()


def softmax_rows(s, cols, ld_p, scale=1.0):
    assert s.dtype == torch.float32 and s.is_contiguous()
    rows, ld_s = s.shape
    p = torch.empty((rows, ld_p), device=s.device, dtype=torch.float16)
    _lib.check(_lib.load().icd_softmax_rows(_p(s), rows, cols, ld_s, scale, _p(p), ld_p, _stream()), "icd_softmax_rows")
    return p


def project_vt(x, w, B, n_tokens, ld_keys, ln_stats=None, ln_colsum=None, debug_flags=0):
    """V^T[b, c, key] = (x[b*n_tokens + key] @ w^T)[c]  -> [B, N, ld_keys] (pad columns zero)."""
    _chk16(x, "x"); _chk16(w, "w")
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((B, N, ld_keys), device=x.device, dtype=torch.float16)
    d = GemmDesc()
    d.a0, d.w, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K, d.Nw = M, N, K, N
    d.lda, d.ldw, d.ldo = x.stride(0), w.stride(0), ld_keys
    d.rows_per_sample = n_tokens
    d.mode, d.batch, d.zdiv, d.alpha, d.flags = 0, 1, 1, 1.0, ICD_GEMM_OUT_TRANS | debug_flags
    if ln_stats is not None:
        d.ln_stats, d.ln_colsum = ln_stats.data_ptr(), ln_colsum.data_ptr()
    _lib.check(_lib.load().icd_gemm(C.byref(d), _stream()), "icd_gemm(V^T)")
    return out


def query_cross_attention(a, w, k, vt, B, n_tokens, nk, scale, bias=None, ln_stats=None, ln_colsum=None, ln_compute=False,
                          debug_flags=0, xattn_tile=0, timeline=None):
    """Query projection + cross-attention in ONE launch (icd_gemm_desc.xattn_*): q = a[M, K] @ w[C, K]^T (+ fused LayerNorm,
    + bias), heads of 64 columns; out[m, h*64:(h+1)*64] = softmax(scale * q_h[m] . K_h^T) V_h.  k: [B*nk, ldk] rows (a column
    slice of a wider matrix is fine), vt: [B, C, ldvt] (V transposed, pad keys zero).  Needs C % 128 == 0, n_tokens % 256 == 0,
    nk <= 96."""
    _chk16(a, "a"); _chk16(w, "w"); _chk_rows(k, "k")
    assert vt.is_cuda and vt.dtype == torch.float16 and vt.dim() == 3 and vt.stride(2) == 1
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=torch.float16)
    d = GemmDesc()
    d.a0, d.w, d.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.M, d.N, d.K, d.Nw = M, N, K, N
    d.lda, d.ldw, d.ldo = a.stride(0), w.stride(0), out.stride(0)
    d.rows_per_sample = n_tokens
    d.mode, d.batch, d.zdiv, d.alpha, d.flags = 0, 1, 1, 1.0, debug_flags
    if ln_stats is not None:
        d.ln_stats, d.ln_colsum = ln_stats.data_ptr(), ln_colsum.data_ptr()
        if ln_compute:
            d.flags |= _lib.ICD_GEMM_LN_COMPUTE
    d.xattn_k, d.xattn_vt = k.data_ptr(), vt.data_ptr()
    d.xattn_nk, d.xattn_ldk, d.xattn_ldvt, d.xattn_vt_bs, d.xattn_scale = nk, k.stride(0), vt.stride(1), vt.stride(0), scale
    d.tune_xattn_tile = xattn_tile            # 0 planner; 2 / 4: force the 128 x 128 / 256 x 128 host tile, 5 / 6: 256 x 256 / 192 x 256 (A/B)
    d.debug_timeline = timeline.data_ptr() if timeline is not None else None
    _lib.check(_lib.load().icd_gemm(C.byref(d), _stream()), "icd_gemm(xattn)")
    return out


def attention_fused(q, k, vt, B, H, Nq, Nk, d, scale, causal=False, prescaled=False, valu_scale=False):
    """q [B*Nq, H*d], k [B*Nk, H*d], vt [B, H*d, ld] -> out [B*Nq, H*d]; causal: keys after the query are masked.
    prescaled: q already carries scale * log2(e) (ICD_ATTN_Q_PRESCALED; `scale` is ignored); valu_scale: A/B switch
    ICD_ATTN_TUNE_MODE0."""
    _chk_rows(q, "q"); _chk_rows(k, "k"); _chk16(vt, "vt")
    out = torch.empty((q.shape[0], q.shape[1]), device=q.device, dtype=torch.float16)
    flags = (_lib.ICD_ATTN_CAUSAL if causal else 0) | (_lib.ICD_ATTN_Q_PRESCALED if prescaled else 0) | \
            (_lib.ICD_ATTN_TUNE_MODE0 if valu_scale else 0)
    _lib.check(_lib.load().icd_attention_fused_ex(_p(q), _p(k), _p(vt), _p(out), B, H, Nq, Nk, d, q.stride(0), k.stride(0),
                                                  vt.stride(1), out.stride(0), vt.stride(0), scale, flags, _stream()),
               "icd_attention_fused")
    return out


def attention_probs(q, k, B, H, Nq, Nk, d, scale, ld=None, out=None, q_carry=None, k_carry=None, acc=None, edit=None, self_from_base=False,
                    first_cond_sample=0):
    """P[b*H+h, n, :Nk] = softmax(scale * q.k) as fp16 [B*H, Nq, ld] in one pass (no fp32 score tensor); pad columns zero.
    q_carry / k_carry: uint8 error carries of q / k (icd_attention_probs_split: scores from hi + lo operands)."""
    _chk_rows(q, "q"); _chk_rows(k, "k")
    ld = ld or (Nk + 7) // 8 * 8
    if out is None:
        out = torch.empty((B * H, Nq, ld), device=q.device, dtype=torch.float16)
    if acc is not None or edit is not None or self_from_base:
        # icd_attention_probs_ex: the shipped controllers' work on P in the kernel's epilogue - acc [(B - first) * H, Nq, ld] += P,
        # edit = (At, Dp) of p2p_pack_operator, self_from_base: the edited prompts take the base prompt's rows
        epi = _lib.ProbsEpilogue()
        epi.first_cond_sample, epi.self_from_base = first_cond_sample, int(self_from_base)
        if acc is not None:
            assert acc.dtype == torch.float16 and acc.is_cuda and acc.stride() == (Nq * ld, ld, 1) and acc.shape[0] == (B - first_cond_sample) * H
            epi.acc = acc.data_ptr()
        if edit is not None:
            epi.edit_At, epi.edit_D = edit[0].data_ptr(), edit[1].data_ptr()
        _lib.check(_lib.load().icd_attention_probs_ex(_p(q), _p(q_carry), _p(k), _p(k_carry), _p(out), B, H, Nq, Nk, d, q.stride(0),
                                                      k.stride(0), ld, scale, C.byref(epi), _stream()), "icd_attention_probs_ex")
        return out
    if q_carry is None and k_carry is None:
        _lib.check(_lib.load().icd_attention_probs(_p(q), _p(k), _p(out), B, H, Nq, Nk, d, q.stride(0), k.stride(0), ld, scale, _stream()),
                   "icd_attention_probs")
    else:
        for c, t in ((q_carry, q), (k_carry, k)):
            assert c is None or (c.dtype == torch.uint8 and c.is_cuda and c.shape == t.shape and c.stride() == t.stride())
        _lib.check(_lib.load().icd_attention_probs_split(_p(q), _p(q_carry), _p(k), _p(k_carry), _p(out), B, H, Nq, Nk, d, q.stride(0),
                                                         k.stride(0), ld, scale, _stream()), "icd_attention_probs_split")
    return out


def attention_scores(q, k, B, H, Nq, Nk, d, scale, ld):
    """S[b*H+h, n, m] = scale * q[b,n,h,:].k[b,m,h,:]  (fp32, [B*H, Nq, ld], columns >= Nk are zero)."""
    s = torch.empty((B * H, Nq, ld), device=q.device, dtype=torch.float32)
    dsc = GemmDesc()
    dsc.a0, dsc.w, dsc.out = q.data_ptr(), k.data_ptr(), s.data_ptr()
    dsc.M, dsc.N, dsc.K, dsc.Nw = Nq, ld, d, Nk
    dsc.lda, dsc.ldw, dsc.ldo = q.stride(0), k.stride(0), ld
    dsc.mode, dsc.batch, dsc.zdiv = 0, B * H, H
    dsc.a_bs0, dsc.a_bs1 = Nq * q.stride(0), d
    dsc.w_bs0, dsc.w_bs1 = Nk * k.stride(0), d
    dsc.o_bs0, dsc.o_bs1 = H * Nq * ld, Nq * ld
    dsc.alpha, dsc.flags = scale, ICD_GEMM_OUT_F32
    _lib.check(_lib.load().icd_gemm(C.byref(dsc), _stream()), "icd_gemm(QK^T)")
    return s


def attention_apply(p, vt, B, H, Nq, d, out=None):
    """out[b, n, h*d:(h+1)*d] = P[b*H+h, n, :] @ V[b, :, h, :]   with P [B*H, Nq, ld], vt [B, H*d, ld]."""
    ld = p.stride(1)
    if out is None:
        out = torch.empty((B * Nq, H * d), device=p.device, dtype=torch.float16)
    dsc = GemmDesc()
    dsc.a0, dsc.w, dsc.out = p.data_ptr(), vt.data_ptr(), out.data_ptr()
    dsc.M, dsc.N, dsc.K, dsc.Nw = Nq, ((d + 7) // 8) * 8, ld, d
    dsc.lda, dsc.ldw, dsc.ldo = ld, vt.stride(1), out.stride(0)
    dsc.mode, dsc.batch, dsc.zdiv = 0, B * H, H
    dsc.a_bs0, dsc.a_bs1 = H * p.stride(0), p.stride(0)
    dsc.w_bs0, dsc.w_bs1 = vt.stride(0), d * vt.stride(1)
    dsc.o_bs0, dsc.o_bs1 = Nq * out.stride(0), d
    dsc.alpha, dsc.flags = 1.0, 0
    _lib.check(_lib.load().icd_gemm(C.byref(dsc), _stream()), "icd_gemm(PV)")
    return out


def sinusoid(vals, dim, kind):
    vals = vals.to(torch.float32).contiguous()
    out = torch.empty((vals.numel(), dim), device=vals.device, dtype=torch.float16)
    _lib.check(_lib.load().icd_sinusoid(_p(vals), vals.numel(), dim, kind, _p(out), _stream()), "icd_sinusoid")
    return out


def silu(x):
    _chk16(x, "x")
    out = torch.empty_like(x)
    _lib.check(_lib.load().icd_silu(_p(x), x.numel(), _p(out), _stream()), "icd_silu")
    return out


def p2p_cross_edit_supported(probs):
    return (probs.is_cuda and probs.dtype == torch.float16 and probs.dim() == 3 and probs.stride(2) == 1 and probs.shape[2] <= 80
            and probs.stride(1) >= 80 and probs.stride(1) % 8 == 0 and probs.stride(0) == probs.shape[1] * probs.stride(1)
            and probs.data_ptr() % 16 == 0)


def p2p_pack_operator(A, D):
    """A fp32 [E, T, T] (row w, column n), D fp32 [E, T]  ->  (At fp16 [E, 96, 80], Dp fp32 [E, 96]) for icd_p2p_cross_edit."""
    E, T, _ = A.shape
    At = torch.zeros((E, 96, 80), device=A.device, dtype=torch.float16)
    At[:, :T, :T] = A.transpose(1, 2)
    Dp = torch.zeros((E, 96), device=A.device, dtype=torch.float32)
    Dp[:, :T] = D
    return At, Dp


def p2p_cross_edit(probs, n_prompts, At, Dp):
    """In place on the conditional rows of one cross-attention module: probs fp16 [n_prompts*heads, nq, nk <= 80], a view
    with row stride ld >= 80 (pad columns zero) of a contiguous buffer; (At, Dp) from p2p_pack_operator."""
    assert p2p_cross_edit_supported(probs)
    bh, nq, nk = probs.shape
    assert bh % n_prompts == 0 and tuple(At.shape) == (n_prompts - 1, 96, 80) and tuple(Dp.shape) == (n_prompts - 1, 96)
    assert At.dtype == torch.float16 and At.is_contiguous() and Dp.dtype == torch.float32 and Dp.is_contiguous()
    _lib.check(_lib.load().icd_p2p_cross_edit(_p(probs), n_prompts, bh // n_prompts, nq, nk, probs.stride(1), _p(At), _p(Dp),
                                              _stream()), "icd_p2p_cross_edit")
    return probs


ACT_SILU, ACT_QUICK_GELU, ACT_GELU = 0, 1, 2


def activation(x, kind):
    _chk16(x, "x")
    out = torch.empty_like(x)
    _lib.check(_lib.load().icd_activation(_p(x), x.numel(), kind, _p(out), _stream()), "icd_activation")
    return out


def embed_tokens(ids, tok_emb, pos_emb):
    """ids int64 [B, T] -> [B*T, C] fp16 = tok_emb[ids] + pos_emb[t]."""
    assert ids.is_cuda and ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2
    _chk16(tok_emb, "tok_emb"); _chk16(pos_emb, "pos_emb")
    B, T = ids.shape
    out = torch.empty((B * T, tok_emb.shape[1]), device=ids.device, dtype=torch.float16)
    _lib.check(_lib.load().icd_embed_tokens(_p(ids), _p(tok_emb), _p(pos_emb), B * T, T, tok_emb.shape[1], tok_emb.shape[0],
                                            _p(out), _stream()), "icd_embed_tokens")
    return out


def conv_in(x_nchw, w_packed, bias):
    B, Cin, H, W = x_nchw.shape
    assert Cin == 4 and x_nchw.is_contiguous() and x_nchw.dtype in (torch.float16, torch.float32)
    Cout = w_packed.shape[0]
    out = torch.empty((B * H * W, Cout), device=x_nchw.device, dtype=torch.float16)
    _lib.check(_lib.load().icd_conv_in(_p(x_nchw), int(x_nchw.dtype == torch.float32), B, H, W, _p(w_packed), _p(bias),
                                       Cout, _p(out), _stream()), "icd_conv_in")
    return out


def conv_out(x, B, H, W, w_packed, bias, out_dtype=torch.float16, cout=4):
    """3x3 pad-1 conv NHWC [B*H*W, Cin] -> NCHW [B, cout, H, W], cout <= 4; w_packed [4, 9*Cin]."""
    _chk16(x, "x")
    eps = torch.empty((B, cout, H, W), device=x.device, dtype=out_dtype)
    _lib.check(_lib.load().icd_conv_out_n(_p(x), B, H, W, x.shape[-1], _p(w_packed), _p(bias), cout, _p(eps),
                                          int(out_dtype == torch.float32), _stream()), "icd_conv_out")
    return eps


def pack_nchw(x_nchw, ones_channel=-1):
    """NCHW [B, C<=8, H, W] (fp16/fp32) -> token-major [B*H*W, 8] fp16; padding channels zero, `ones_channel` = 1."""
    B, Cc, H, W = x_nchw.shape
    assert x_nchw.is_contiguous() and x_nchw.dtype in (torch.float16, torch.float32)
    out = torch.empty((B * H * W, 8), device=x_nchw.device, dtype=torch.float16)
    _lib.check(_lib.load().icd_pack_nchw(_p(x_nchw), int(x_nchw.dtype == torch.float32), B, Cc, H * W, ones_channel, _p(out),
                                         _stream()), "icd_pack_nchw")
    return out


def x0_step(x, eps, coef, out_dtype=None):
    """predicted_origin (eps-prediction); coef fp32 [B,4] = (alpha_t, sigma_t, alpha_s, sigma_s)."""
    assert x.is_contiguous() and eps.is_contiguous() and x.shape == eps.shape
    out_dtype = out_dtype or x.dtype
    out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    B = x.shape[0]
    flags = int(x.dtype == torch.float32) | (int(eps.dtype == torch.float32) << 1) | (int(out_dtype == torch.float32) << 2)
    coef = coef.to(device=x.device, dtype=torch.float32).contiguous()
    _lib.check(_lib.load().icd_x0_step(_p(x), _p(eps), _p(coef), B, x.numel() // B, flags, _p(out), _stream()), "icd_x0_step")
    return out


def local_blend(maps, alpha, alpha_sub, th_pool, th_sub, x_t, res=16):
    """LocalBlend in one launch (icd_local_blend).  maps: list of <= 8 fp16 cuda tensors [P*heads_l, res*res, n_words] (last dim
    contiguous, rows evenly strided), alpha / alpha_sub: fp32 [P, n_words] word masks, x_t: [P, C, H, W] fp16 / fp32 -> fp32."""
    P = alpha.shape[0]
    n_words = alpha.shape[1]
    assert x_t.is_cuda and x_t.is_contiguous() and x_t.dtype in (torch.float16, torch.float32) and x_t.shape[0] == P
    ld = maps[0].stride(1)
    ptrs, heads = (C.c_void_p * len(maps))(), (C.c_int32 * len(maps))()
    for i, m in enumerate(maps):
        assert m.is_cuda and m.dtype == torch.float16 and m.dim() == 3 and m.shape[1] == res * res and m.shape[2] == n_words
        assert m.stride(2) == 1 and m.stride(1) == ld and m.stride(0) == res * res * ld and m.shape[0] % P == 0
        ptrs[i], heads[i] = m.data_ptr(), m.shape[0] // P
    al = alpha.to(device=x_t.device, dtype=torch.float32).contiguous()
    als = None if alpha_sub is None else alpha_sub.to(device=x_t.device, dtype=torch.float32).contiguous()
    out = torch.empty(x_t.shape, device=x_t.device, dtype=torch.float32)
    _lib.check(_lib.load().icd_local_blend(ptrs, heads, len(maps), P, res, n_words, ld, _p(al), None if als is None else _p(als),
                                           float(th_pool), float(th_sub), _p(x_t), int(x_t.dtype == torch.float32), x_t.shape[1],
                                           x_t.shape[2], x_t.shape[3], _p(out), _stream()), "icd_local_blend")
    return out


def accumulate_multi(dst, src):
    """dst[t] += src[t] for lists of fp16 cuda tensors (<= 32 per launch), rounded like torch's in-place add."""
    lib = _lib.load()
    for i in range(0, len(dst), 32):
        d, s = dst[i:i + 32], src[i:i + 32]
        n = len(d)
        dp, sp, cn = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
        for t, (a, b) in enumerate(zip(d, s)):
            assert a.is_cuda and b.is_cuda and a.dtype == b.dtype == torch.float16 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
            dp[t], sp[t], cn[t] = a.data_ptr(), b.data_ptr(), a.numel()
        _lib.check(lib.icd_accumulate_multi(dp, sp, cn, n, _stream()), "icd_accumulate_multi")
