"""AutoencoderKL (the VAE either side of the iCD path) on the HIP kernels of this package - SURVEY.md section 8f rank 1.

Drop-in for the attributes the reference touches on `model.vae` / `pipe.vae`:
    vae.decode(z)['sample']                      utils/generation.py:258   (z already divided by 0.18215 by the caller)
    vae.decode(z, return_dict=False)[0]          utils/generation_sdxl.py:466
    vae.encode(x)['latent_dist'].mean            utils/generation.py:277,282
    vae.dtype, vae.config.scaling_factor, vae.to(dtype)

Architecture: diffusers 0.25.1 AutoencoderKL (oracle/vae_ref.py restates it and is what the GPU tests compare against).
Everything runs on the same operators as the UNet: GroupNorm(+SiLU) `icd_groupnorm`, 3x3 / 1x1 convs as implicit GEMM
`icd_gemm` (nearest-2x upsample folded into the loader; the encoder's bottom/right-only padding is ICD_GEMM_PAD_HI), the
mid-block attention (one head of 512 channels) as QK^T -> row softmax -> PV (`icd_gemm` batched + `icd_softmax_rows`; the
fused flash kernel when the head dim is <= 160), first / last convs through `icd_pack_nchw` + implicit GEMM and
`icd_conv_out_n`.  Host-side algebra done once at load, all exact in real arithmetic:
  * post_quant_conv (1x1 + bias) is folded into decoder.conv_in; its bias rides on a ones-channel of the packed latent
    so the 3x3 conv's zero padding still sees zeros outside the image;
  * quant_conv is folded into encoder.conv_out (mean rows and log-variance rows as two 4-channel output convs);
  * the V bias of the attention is folded into the output projection's bias (softmax rows sum to one).
Two arithmetic modes, selected like the reference does it, through `vae.to(dtype)`:
  * fp16 (default, `AutoencoderKL(..., dtype=torch.float16)` / `.to(torch.float16)`): activations are fp16 token-major
    [B*H*W, C] with fp32 accumulation, like the UNet.
  * fp32 fidelity (`.to(torch.float32)` - what utils/generation_sdxl.py:465-466 and diffusers' force_upcast ask for,
    because real SDXL-VAE activations overflow the fp16 range): every tensor that can grow without bound (conv outputs, the
    residual stream) is stored in fp32; the operands of the matrix cores are "split3" fp16 tensors [hi | lo | hi] against
    weights packed [w_hi | w_hi | w_lo] (ops.split_weight), i.e. (a_hi + a_lo)(w_hi + w_lo) minus the lo*lo term, ~2^-21
    relative operand error on the unchanged fp16 MFMA kernels at 3x their work; GroupNorm reads fp32 and writes split3
    (icd_groupnorm_f32_split); tensors that reach a conv without a GroupNorm (Up/Downsample, conv_shortcut) are scaled by a
    power of two chosen from their max |x| (icd_absmax) on the way to fp16 and the conv's alpha restores the scale exactly
    (icd_split_cast).  Only the
    mid-block attention keeps fp16 q / k / v / P (bounded by construction: they follow a GroupNorm and a softmax).
"""
from dataclasses import dataclass, replace
from types import SimpleNamespace
from typing import Tuple

import torch

from . import ops

EPS = 1e-6


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    def scaled(self, block_out_channels):
        """Reduced-width copy (tests): channel counts must stay multiples of 32 (GroupNorm groups) ."""
        return replace(self, block_out_channels=tuple(block_out_channels))

    def state_dict_shapes(self):
        """diffusers AutoencoderKL keys -> shapes (same layout the oracle pins by the exact SD VAE parameter count)."""
        ch, L, zc = list(self.block_out_channels), self.layers_per_block, self.latent_channels
        out = {}

        def resnet(p, cin, cout):
            out[p + "norm1.weight"] = (cin,); out[p + "norm1.bias"] = (cin,)
            out[p + "conv1.weight"] = (cout, cin, 3, 3); out[p + "conv1.bias"] = (cout,)
            out[p + "norm2.weight"] = (cout,); out[p + "norm2.bias"] = (cout,)
            out[p + "conv2.weight"] = (cout, cout, 3, 3); out[p + "conv2.bias"] = (cout,)
            if cin != cout:
                out[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); out[p + "conv_shortcut.bias"] = (cout,)

        def mid(p, c):
            resnet(p + "resnets.0.", c, c)
            a = p + "attentions.0."
            out[a + "group_norm.weight"] = (c,); out[a + "group_norm.bias"] = (c,)
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                out[a + n + ".weight"] = (c, c); out[a + n + ".bias"] = (c,)
            resnet(p + "resnets.1.", c, c)

        out["encoder.conv_in.weight"] = (ch[0], self.in_channels, 3, 3); out["encoder.conv_in.bias"] = (ch[0],)
        cin = ch[0]
        for i, c in enumerate(ch):
            for j in range(L):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}.", cin, c)
                cin = c
            if i < len(ch) - 1:
                out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
                out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
        mid("encoder.mid_block.", ch[-1])
        out["encoder.conv_norm_out.weight"] = (ch[-1],); out["encoder.conv_norm_out.bias"] = (ch[-1],)
        out["encoder.conv_out.weight"] = (2 * zc, ch[-1], 3, 3); out["encoder.conv_out.bias"] = (2 * zc,)
        out["quant_conv.weight"] = (2 * zc, 2 * zc, 1, 1); out["quant_conv.bias"] = (2 * zc,)
        out["post_quant_conv.weight"] = (zc, zc, 1, 1); out["post_quant_conv.bias"] = (zc,)
        rev = ch[::-1]
        out["decoder.conv_in.weight"] = (rev[0], zc, 3, 3); out["decoder.conv_in.bias"] = (rev[0],)
        mid("decoder.mid_block.", rev[0])
        cin = rev[0]
        for i, c in enumerate(rev):
            for j in range(L + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}.", cin, c)
                cin = c
            if i < len(rev) - 1:
                out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
                out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
        out["decoder.conv_norm_out.weight"] = (rev[-1],); out["decoder.conv_norm_out.bias"] = (rev[-1],)
        out["decoder.conv_out.weight"] = (self.out_channels, rev[-1], 3, 3); out["decoder.conv_out.bias"] = (self.out_channels,)
        return out


SD_VAE = VAEConfig()
SDXL_VAE = VAEConfig(scaling_factor=0.13025)


class _Out(dict):
    """dict with attribute access: the reference indexes `['sample']` / `['latent_dist']`, diffusers pipelines use `.x`."""
    __getattr__ = dict.__getitem__


class LatentDist:
    """diffusers' `DiagonalGaussianDistribution` over the 8 moment channels of quant_conv(encoder(x)).

    The SD1.5 path reads `.mean` (utils/generation.py:277,282); the SDXL img2img `prepare_latents` the reference calls at
    utils/generation_sdxl.py:273-276 draws `.sample(generator)`: mean + exp(0.5 * clamp(logvar, -30, 20)) * randn, the
    noise drawn like diffusers' randn_tensor (a CPU generator draws on the CPU in the parameters' dtype, then moves)."""

    def __init__(self, mean, logvar=None):
        self.mean = mean
        self.logvar = None if logvar is None else torch.clamp(logvar, -30.0, 20.0)
        self.deterministic = logvar is None
        if self.logvar is not None:
            self.std = torch.exp(0.5 * self.logvar)
            self.var = torch.exp(self.logvar)
        else:
            self.std = self.var = torch.zeros_like(mean)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        gen_dev = generator.device.type if generator is not None else self.mean.device.type
        if gen_dev == "cpu":
            noise = torch.randn(self.mean.shape, generator=generator, device="cpu", dtype=self.mean.dtype).to(self.mean.device)
        else:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise


def pack_vae_state_dict(cfg: VAEConfig, sd, device="cuda"):
    """diffusers-layout AutoencoderKL state dict -> kernel layout (fp16 weights, fp32 biases / affine)."""
    missing = [k for k in cfg.state_dict_shapes() if k not in sd]
    if missing:
        raise KeyError(f"AutoencoderKL state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
    for k, shp in cfg.state_dict_shapes().items():
        if tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: expected shape {tuple(shp)}, got {tuple(sd[k].shape)}")
    f32 = lambda k: sd[k].detach().to("cpu", torch.float32)
    P = {}
    half = lambda t: t.to(device=device, dtype=torch.float16).contiguous()
    full = lambda t: t.to(device=device, dtype=torch.float32).contiguous()

    def resnet(p):
        for n in ("norm1", "norm2"):
            P[p + n + ".weight"], P[p + n + ".bias"] = full(f32(p + n + ".weight")), full(f32(p + n + ".bias"))
        for n in ("conv1", "conv2"):
            P[p + n + ".weight"], P[p + n + ".bias"] = half(ops.pack_conv_weight(f32(p + n + ".weight"))), full(f32(p + n + ".bias"))
        if p + "conv_shortcut.weight" in sd:
            w = f32(p + "conv_shortcut.weight")
            P[p + "conv_shortcut.weight"], P[p + "conv_shortcut.bias"] = half(w.reshape(w.shape[0], -1)), full(f32(p + "conv_shortcut.bias"))

    def mid(p):
        resnet(p + "resnets.0."); resnet(p + "resnets.1.")
        a = p + "attentions.0."
        P[a + "group_norm.weight"], P[a + "group_norm.bias"] = full(f32(a + "group_norm.weight")), full(f32(a + "group_norm.bias"))
        P[a + "to_qk.weight"] = half(torch.cat([f32(a + "to_q.weight"), f32(a + "to_k.weight")]))
        P[a + "to_qk.bias"] = full(torch.cat([f32(a + "to_q.bias"), f32(a + "to_k.bias")]))
        P[a + "to_v.weight"] = half(f32(a + "to_v.weight"))
        wo = f32(a + "to_out.0.weight")
        P[a + "to_out.weight"] = half(wo)
        P[a + "to_out.bias"] = full(wo @ f32(a + "to_v.bias") + f32(a + "to_out.0.bias"))      # softmax rows sum to one

    ch, L, zc = list(cfg.block_out_channels), cfg.layers_per_block, cfg.latent_channels
    # ---- encoder
    w = f32("encoder.conv_in.weight")
    w8 = torch.zeros(w.shape[0], 8, 3, 3); w8[:, :cfg.in_channels] = w
    P["encoder.conv_in.weight"], P["encoder.conv_in.bias"] = half(ops.pack_conv_weight(w8)), full(f32("encoder.conv_in.bias"))
    for i in range(len(ch)):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.")
        if i < len(ch) - 1:
            k = f"encoder.down_blocks.{i}.downsamplers.0.conv."
            P[k + "weight"], P[k + "bias"] = half(ops.pack_conv_weight(f32(k + "weight"))), full(f32(k + "bias"))
    mid("encoder.mid_block.")
    P["encoder.conv_norm_out.weight"], P["encoder.conv_norm_out.bias"] = full(f32("encoder.conv_norm_out.weight")), full(f32("encoder.conv_norm_out.bias"))
    wq, bq = f32("quant_conv.weight").reshape(2 * zc, 2 * zc), f32("quant_conv.bias")
    wo, bo = f32("encoder.conv_out.weight"), f32("encoder.conv_out.bias")
    wm = torch.einsum("om,mchw->ochw", wq[:zc], wo)                          # mean rows of quant_conv(conv_out(.))
    w4 = torch.zeros(4, wo.shape[1], 3, 3); w4[:zc] = wm
    b4 = torch.zeros(4); b4[:zc] = wq[:zc] @ bo + bq[:zc]
    P["encoder.conv_out_mean.weight"], P["encoder.conv_out_mean.bias"] = half(ops.pack_conv_weight(w4)), full(b4)
    wl = torch.einsum("om,mchw->ochw", wq[zc:], wo)                          # log-variance rows (latent_dist.sample / .std)
    w4 = torch.zeros(4, wo.shape[1], 3, 3); w4[:zc] = wl
    b4 = torch.zeros(4); b4[:zc] = wq[zc:] @ bo + bq[zc:]
    P["encoder.conv_out_logvar.weight"], P["encoder.conv_out_logvar.bias"] = half(ops.pack_conv_weight(w4)), full(b4)
    # ---- decoder
    wpq, bpq = f32("post_quant_conv.weight").reshape(zc, zc), f32("post_quant_conv.bias")
    wi = f32("decoder.conv_in.weight")
    w8 = torch.zeros(wi.shape[0], 8, 3, 3)
    w8[:, :zc] = torch.einsum("ochw,cd->odhw", wi, wpq)
    w8[:, zc] = torch.einsum("ochw,c->ohw", wi, bpq)                          # rides on the ones channel
    P["decoder.conv_in.weight"], P["decoder.conv_in.bias"] = half(ops.pack_conv_weight(w8)), full(f32("decoder.conv_in.bias"))
    mid("decoder.mid_block.")
    for i in range(len(ch)):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.")
        if i < len(ch) - 1:
            k = f"decoder.up_blocks.{i}.upsamplers.0.conv."
            P[k + "weight"], P[k + "bias"] = half(ops.pack_conv_weight(f32(k + "weight"))), full(f32(k + "bias"))
            # the phase form of Upsample2D (four 2 x 2 convs on the input grid with tap-summed weights: 4/9 of the flops, unet.py)
            from .unet import upsample_phase_weights
            for ph, wp in enumerate(upsample_phase_weights(f32(k + "weight"))):
                P[k + f"phase.{ph}"] = half(wp.reshape(wp.shape[0], -1).contiguous())
    P["decoder.conv_norm_out.weight"], P["decoder.conv_norm_out.bias"] = full(f32("decoder.conv_norm_out.weight")), full(f32("decoder.conv_norm_out.bias"))
    wo, bo = f32("decoder.conv_out.weight"), f32("decoder.conv_out.bias")
    w4 = torch.zeros(4, wo.shape[1], 3, 3); w4[:cfg.out_channels] = wo
    b4 = torch.zeros(4); b4[:cfg.out_channels] = bo
    P["decoder.conv_out.weight"], P["decoder.conv_out.bias"] = half(ops.pack_conv_weight(w4)), full(b4)
    return P


class AutoencoderKL:
    def __init__(self, cfg: VAEConfig, state_dict, device="cuda", dtype=torch.float16, max_chunk=8, fused_attention=None):
        if cfg.latent_channels != 4 or cfg.in_channels > 4 or cfg.out_channels > 4:
            raise ValueError("AutoencoderKL: this build supports 4 latent channels and <= 4 image channels")
        if any(c % 32 for c in cfg.block_out_channels):
            raise ValueError("AutoencoderKL: block_out_channels must be multiples of 32")
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.config = SimpleNamespace(scaling_factor=cfg.scaling_factor, latent_channels=cfg.latent_channels,
                                      block_out_channels=list(cfg.block_out_channels), in_channels=cfg.in_channels,
                                      out_channels=cfg.out_channels)
        self.w = pack_vae_state_dict(cfg, state_dict, device)
        self._sd_ref = {k: state_dict[k] for k in cfg.state_dict_shapes()}     # masters of the fp32-fidelity weights (converted lazily)
        self._ws = None                        # split3 weights of the fp32-fidelity path, packed on first use
        self.max_chunk = max_chunk            # samples per pass (bounds the 512x512x128-channel activations and the scores)
        self.fused_attention = fused_attention    # None: fused flash kernel when the head dim allows it (<= 160)
        self.carry = True                         # fp16 path: error-carried residual stream + GroupNorm over fp16 + carry (round 6); False: plain fp16 stream
        self.upsample_phases = True               # fp16 decoder: Upsample2D as four 2 x 2 convs on the input grid (4/9 of its flops); False: 3 x 3 form

    # -- diffusers plumbing the reference uses
    def to(self, *args, **kw):
        for a in list(args) + [kw.get("dtype")]:
            if isinstance(a, torch.dtype):
                self.dtype = a                 # float32 selects the fp32-fidelity arithmetic (module docstring), not just the I/O dtype
        return self

    # ------------------------------------------------------------------ fp32-fidelity path (dtype == torch.float32)
    def _split_weights(self):
        """[w_hi | w_hi | w_lo] per tap for every conv / projection whose input is a split3 tensor, from the fp32 masters."""
        if self._ws is not None:
            return self._ws
        sd = {k: v.detach().to("cpu", torch.float32) for k, v in self._sd_ref.items()}
        dev, cfg = self.device, self.cfg
        S = {}

        def conv(key, w=None, cin_pad=None):
            w = sd[key] if w is None else w                                   # [O, I, kh, kw]
            if cin_pad:
                wp = torch.zeros(w.shape[0], cin_pad, *w.shape[2:]); wp[:, :w.shape[1]] = w; w = wp
            o = w.shape[0]
            S[key] = ops.split_weight(w.permute(0, 2, 3, 1)).reshape(o, -1).to(dev)          # [O, taps * 3I]

        def dense(key, w):
            S[key] = ops.split_weight(w.reshape(w.shape[0], -1)).to(dev)

        def resnet(p):
            conv(p + "conv1.weight"); conv(p + "conv2.weight")
            if p + "conv_shortcut.weight" in sd:
                conv(p + "conv_shortcut.weight")

        def mid(p):
            resnet(p + "resnets.0."); resnet(p + "resnets.1.")
            a = p + "attentions.0."
            dense(a + "to_qk.weight", torch.cat([sd[a + "to_q.weight"], sd[a + "to_k.weight"]]))
            dense(a + "to_v.weight", sd[a + "to_v.weight"])

        ch, L, zc = list(cfg.block_out_channels), cfg.layers_per_block, cfg.latent_channels
        conv("encoder.conv_in.weight", cin_pad=8)
        for i in range(len(ch)):
            for j in range(L):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}.")
            if i < len(ch) - 1:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv.weight")
        mid("encoder.mid_block.")
        wq, wo = sd["quant_conv.weight"].reshape(2 * zc, 2 * zc), sd["encoder.conv_out.weight"]
        for name, rows in (("mean", wq[:zc]), ("logvar", wq[zc:])):
            w4 = torch.zeros(4, wo.shape[1], 3, 3); w4[:zc] = torch.einsum("om,mchw->ochw", rows, wo)
            conv(f"encoder.conv_out_{name}.weight", w=w4)
        wpq, bpq, wi = sd["post_quant_conv.weight"].reshape(zc, zc), sd["post_quant_conv.bias"], sd["decoder.conv_in.weight"]
        w8 = torch.zeros(wi.shape[0], 8, 3, 3)
        w8[:, :zc] = torch.einsum("ochw,cd->odhw", wi, wpq)
        w8[:, zc] = torch.einsum("ochw,c->ohw", wi, bpq)
        conv("decoder.conv_in.weight", w=w8)
        mid("decoder.mid_block.")
        for i in range(len(ch)):
            for j in range(L + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}.")
            if i < len(ch) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv.weight")
        wo = sd["decoder.conv_out.weight"]
        w4 = torch.zeros(4, wo.shape[1], 3, 3); w4[:cfg.out_channels] = wo
        conv("decoder.conv_out.weight", w=w4)
        self._ws = S
        return S

    def _pack32(self, x_nchw, ones_channel=-1):
        """NCHW fp32 (<= 8 channels) -> split3 [B*H*W, 24] of the 8-channel token-major packing (data movement + split only)."""
        B, Cc, H, W = x_nchw.shape
        t = torch.zeros((B, H, W, 8), device=self.device, dtype=torch.float32)
        t[..., :Cc] = x_nchw.to(self.device, torch.float32).permute(0, 2, 3, 1)
        if ones_channel >= 0:
            t[..., ones_channel] = 1.0
        return ops.split_cast(t.reshape(B * H * W, 8), 1.0)

    def _resnet32(self, p, x, B, H, W):
        w, ws, g = self.w, self._ws, self.cfg.norm_num_groups
        h = ops.groupnorm_f32_split(x, B, H * W, w[p + "norm1.weight"], w[p + "norm1.bias"], EPS, True, groups=g)
        h = ops.conv3x3(h, B, H, W, ws[p + "conv1.weight"], w[p + "conv1.bias"], out_f32=True)
        h = ops.groupnorm_f32_split(h, B, H * W, w[p + "norm2.weight"], w[p + "norm2.bias"], EPS, True, groups=g)
        sc = x
        if p + "conv_shortcut.weight" in ws:
            xs, inv = ops.split_cast_guarded(x)
            sc = ops.gemm(xs, ws[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"], out_f32=True, alpha=inv)
        return ops.conv3x3(h, B, H, W, ws[p + "conv2.weight"], w[p + "conv2.bias"], resid=sc, out_f32=True)

    def _attention32(self, p, x, B, H, W):
        w, ws, g = self.w, self._ws, self.cfg.norm_num_groups
        C, N = x.shape[1], H * W
        h = ops.groupnorm_f32_split(x, B, N, w[p + "group_norm.weight"], w[p + "group_norm.bias"], EPS, False, groups=g)
        qk = ops.gemm(h, ws[p + "to_qk.weight"], w[p + "to_qk.bias"])                  # fp16: bounded (GroupNorm output x weights)
        q, k = qk[:, :C], qk[:, C:]
        ld = (N + 7) // 8 * 8
        vt = ops.project_vt(h, ws[p + "to_v.weight"], B, N, ld)
        scale = C ** -0.5
        fused = self.fused_attention if self.fused_attention is not None else C <= 160
        if fused:
            o = ops.attention_fused(q, k, vt, B, 1, N, N, C, scale)
        else:
            o = torch.empty((B * N, C), device=x.device, dtype=torch.float16)
            step = max(1, min(B, (1 << 30) // (N * ld * 4)))
            for b0 in range(0, B, step):
                bc = min(step, B - b0)
                s_ = ops.attention_scores(q[b0 * N:(b0 + bc) * N], k[b0 * N:(b0 + bc) * N], bc, 1, N, N, C, scale, ld)
                pr = ops.softmax_rows(s_.reshape(bc * N, ld), N, ld).reshape(bc, N, ld)
                ops.attention_apply(pr, vt[b0:b0 + bc], bc, 1, N, C, out=o[b0 * N:(b0 + bc) * N])
        return ops.gemm(o, w[p + "to_out.weight"], w[p + "to_out.bias"], resid=x, out_f32=True)

    def _mid32(self, p, x, B, H, W):
        x = self._resnet32(p + "resnets.0.", x, B, H, W)
        x = self._attention32(p + "attentions.0.", x, B, H, W)
        return self._resnet32(p + "resnets.1.", x, B, H, W)

    def _decode_chunk32(self, z):
        w, ws, cfg = self.w, self._split_weights(), self.cfg
        B, _, H, W = z.shape
        x = ops.conv3x3(self._pack32(z, ones_channel=cfg.latent_channels), B, H, W, ws["decoder.conv_in.weight"],
                        w["decoder.conv_in.bias"], out_f32=True)
        x = self._mid32("decoder.mid_block.", x, B, H, W)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet32(f"decoder.up_blocks.{i}.resnets.{j}.", x, B, H, W)
            if i < nb - 1:
                k = f"decoder.up_blocks.{i}.upsamplers.0.conv."
                xs, inv = ops.split_cast_guarded(x)
                x = ops.conv3x3(xs, B, H, W, ws[k + "weight"], w[k + "bias"], upsample=True, out_f32=True, alpha=inv)
                H, W = 2 * H, 2 * W
        h = ops.groupnorm_f32_split(x, B, H * W, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], EPS, True,
                                    groups=cfg.norm_num_groups)
        return ops.conv_out(h, B, H, W, ws["decoder.conv_out.weight"], w["decoder.conv_out.bias"], out_dtype=torch.float32,
                            cout=cfg.out_channels)

    def _encode_chunk32(self, img):
        w, ws, cfg = self.w, self._split_weights(), self.cfg
        B, _, H, W = img.shape
        x = ops.conv3x3(self._pack32(img), B, H, W, ws["encoder.conv_in.weight"], w["encoder.conv_in.bias"], out_f32=True)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                x = self._resnet32(f"encoder.down_blocks.{i}.resnets.{j}.", x, B, H, W)
            if i < nb - 1:
                k = f"encoder.down_blocks.{i}.downsamplers.0.conv."
                xs, inv = ops.split_cast_guarded(x)
                x = ops.conv3x3(xs, B, H, W, ws[k + "weight"], w[k + "bias"], stride=2, pad_hi=True, out_f32=True, alpha=inv)
                H, W = H // 2, W // 2
        x = self._mid32("encoder.mid_block.", x, B, H, W)
        h = ops.groupnorm_f32_split(x, B, H * W, w["encoder.conv_norm_out.weight"], w["encoder.conv_norm_out.bias"], EPS, True,
                                    groups=cfg.norm_num_groups)
        mean = ops.conv_out(h, B, H, W, ws["encoder.conv_out_mean.weight"], w["encoder.conv_out_mean.bias"], out_dtype=torch.float32, cout=4)
        logvar = ops.conv_out(h, B, H, W, ws["encoder.conv_out_logvar.weight"], w["encoder.conv_out_logvar.bias"], out_dtype=torch.float32,
                              cout=4)
        return mean, logvar

    def eval(self):
        return self

    # -- blocks (token-major fp16 [B*H*W, C]).  Round 6: the residual stream carries its rounding error (`self.carry`, default on): every
    # x <- x + f(x) of a ResnetBlock2D / the mid-block attention keeps what the fp16 rounding of the sum lost in one bf8 byte per element
    # (icd_gemm_desc.resid_carry / out_carry, the UNet's residual mode 2) and every GroupNorm normalises fp16 + carry (icd_groupnorm_carry):
    # the stream behaves like a 14-bit-mantissa tensor.  A stream is the pair (x, xc); xc None = no carry yet.
    def _gn(self, x, xc, B, HW, wk, bk, silu):
        g = self.cfg.norm_num_groups
        if xc is None:
            return ops.groupnorm(x, B, HW, self.w[wk], self.w[bk], EPS, silu, groups=g)
        return ops.groupnorm_carry(x, B, HW, self.w[wk], self.w[bk], EPS, silu, carry=xc, groups=g)

    def _oc(self, rows, cols, like):
        return torch.empty((rows, cols), device=like.device, dtype=torch.uint8) if self.carry else None

    def _resnet(self, p, x, xc, B, H, W):
        w = self.w
        h = self._gn(x, xc, B, H * W, p + "norm1.weight", p + "norm1.bias", True)
        c1 = self._oc(h.shape[0], w[p + "conv1.weight"].shape[0], h)
        h = ops.conv3x3(h, B, H, W, w[p + "conv1.weight"], w[p + "conv1.bias"], out_carry=c1)
        h = self._gn(h, c1, B, H * W, p + "norm2.weight", p + "norm2.bias", True)
        sc, scc = x, xc
        if p + "conv_shortcut.weight" in w:
            scc = self._oc(x.shape[0], w[p + "conv_shortcut.weight"].shape[0], x)
            sc = ops.gemm(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"], out_carry=scc)
        oc = self._oc(sc.shape[0], sc.shape[1], sc)
        return ops.conv3x3(h, B, H, W, w[p + "conv2.weight"], w[p + "conv2.bias"], resid=sc, resid_carry=scc, out_carry=oc), oc

    def _attention(self, p, x, xc, B, H, W):
        w = self.w
        C, N = x.shape[1], H * W
        h = self._gn(x, xc, B, N, p + "group_norm.weight", p + "group_norm.bias", False)
        qk = ops.gemm(h, w[p + "to_qk.weight"], w[p + "to_qk.bias"])
        q, k = qk[:, :C], qk[:, C:]
        ld = (N + 7) // 8 * 8
        vt = ops.project_vt(h, w[p + "to_v.weight"], B, N, ld)
        scale = C ** -0.5
        fused = self.fused_attention if self.fused_attention is not None else C <= 160
        if fused:
            o = ops.attention_fused(q, k, vt, B, 1, N, N, C, scale)
        else:
            o = torch.empty((B * N, C), device=x.device, dtype=torch.float16)
            step = max(1, min(B, (1 << 30) // (N * ld * 4)))                    # <= 1 GiB of fp32 scores at a time
            for b0 in range(0, B, step):
                bc = min(step, B - b0)
                s = ops.attention_scores(q[b0 * N:(b0 + bc) * N], k[b0 * N:(b0 + bc) * N], bc, 1, N, N, C, scale, ld)
                pr = ops.softmax_rows(s.reshape(bc * N, ld), N, ld).reshape(bc, N, ld)
                ops.attention_apply(pr, vt[b0:b0 + bc], bc, 1, N, C, out=o[b0 * N:(b0 + bc) * N])
        oc = self._oc(x.shape[0], x.shape[1], x)
        return ops.gemm(o, w[p + "to_out.weight"], w[p + "to_out.bias"], resid=x, resid_carry=xc, out_carry=oc), oc

    def _mid(self, p, x, xc, B, H, W):
        x, xc = self._resnet(p + "resnets.0.", x, xc, B, H, W)
        x, xc = self._attention(p + "attentions.0.", x, xc, B, H, W)
        return self._resnet(p + "resnets.1.", x, xc, B, H, W)

    # -- public
    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """z [B,4,h,w] (already / scaling_factor) -> image [B,3,8h,8w] in `self.dtype`."""
        if z.dim() != 4 or z.shape[1] != self.cfg.latent_channels:
            raise ValueError(f"AutoencoderKL.decode: expected [B,{self.cfg.latent_channels},h,w], got {tuple(z.shape)}")
        z = z.to(self.device)
        if z.dtype not in (torch.float16, torch.float32):
            z = z.float()
        chunk = self._decode_chunk32 if self.dtype == torch.float32 else self._decode_chunk
        outs = [chunk(z[i:i + self.max_chunk].contiguous()) for i in range(0, z.shape[0], self.max_chunk)]
        img = (outs[0] if len(outs) == 1 else torch.cat(outs)).to(self.dtype)
        return _Out(sample=img) if return_dict else (img,)

    def _decode_chunk(self, z):
        w, cfg = self.w, self.cfg
        B, _, H, W = z.shape
        x = ops.pack_nchw(z, ones_channel=cfg.latent_channels)
        xc = self._oc(B * H * W, w["decoder.conv_in.weight"].shape[0], x)
        x = ops.conv3x3(x, B, H, W, w["decoder.conv_in.weight"], w["decoder.conv_in.bias"], out_carry=xc)
        x, xc = self._mid("decoder.mid_block.", x, xc, B, H, W)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                x, xc = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}.", x, xc, B, H, W)
            if i < nb - 1:
                k = f"decoder.up_blocks.{i}.upsamplers.0.conv."
                uc = self._oc(B * 4 * H * W, x.shape[-1], x)
                if self.upsample_phases and W >= 2:
                    up = torch.empty((B * 4 * H * W, x.shape[-1]), device=x.device, dtype=torch.float16)
                    for ph in range(4):
                        ops.conv3x3(x, B, H, W, w[k + f"phase.{ph}"], w[k + "bias"], phase=ph, out=up, out_carry=uc)
                    x = up
                else:
                    x = ops.conv3x3(x, B, H, W, w[k + "weight"], w[k + "bias"], upsample=True, out_carry=uc)
                xc = uc
                H, W = 2 * H, 2 * W
        x = self._gn(x, xc, B, H * W, "decoder.conv_norm_out.weight", "decoder.conv_norm_out.bias", True)
        return ops.conv_out(x, B, H, W, w["decoder.conv_out.weight"], w["decoder.conv_out.bias"],
                            out_dtype=torch.float32 if self.dtype == torch.float32 else torch.float16, cout=cfg.out_channels)

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """image [B,3,H,W] in [-1,1] (H, W multiples of 8) -> latent_dist with `.mean` [B,4,H/8,W/8]."""
        if x.dim() != 4 or x.shape[1] != self.cfg.in_channels or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"AutoencoderKL.encode: expected [B,{self.cfg.in_channels},8h,8w], got {tuple(x.shape)}")
        x = x.to(self.device)
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        chunk = self._encode_chunk32 if self.dtype == torch.float32 else self._encode_chunk
        outs = [chunk(x[i:i + self.max_chunk].contiguous()) for i in range(0, x.shape[0], self.max_chunk)]
        mean = torch.cat([o[0] for o in outs]).to(self.dtype)
        logvar = torch.cat([o[1] for o in outs]).to(self.dtype)
        dist = LatentDist(mean, logvar)
        return _Out(latent_dist=dist) if return_dict else (dist,)

    def _encode_chunk(self, img):
        w, cfg = self.w, self.cfg
        B, _, H, W = img.shape
        x = ops.pack_nchw(img)
        xc = self._oc(B * H * W, w["encoder.conv_in.weight"].shape[0], x)
        x = ops.conv3x3(x, B, H, W, w["encoder.conv_in.weight"], w["encoder.conv_in.bias"], out_carry=xc)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                x, xc = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}.", x, xc, B, H, W)
            if i < nb - 1:
                k = f"encoder.down_blocks.{i}.downsamplers.0.conv."
                dc = self._oc(B * (H // 2) * (W // 2), x.shape[-1], x)
                x = ops.conv3x3(x, B, H, W, w[k + "weight"], w[k + "bias"], stride=2, pad_hi=True, out_carry=dc)
                xc = dc
                H, W = H // 2, W // 2
        x, xc = self._mid("encoder.mid_block.", x, xc, B, H, W)
        x = self._gn(x, xc, B, H * W, "encoder.conv_norm_out.weight", "encoder.conv_norm_out.bias", True)
        od = torch.float32 if self.dtype == torch.float32 else torch.float16
        mean = ops.conv_out(x, B, H, W, w["encoder.conv_out_mean.weight"], w["encoder.conv_out_mean.bias"], out_dtype=od, cout=4)
        logvar = ops.conv_out(x, B, H, W, w["encoder.conv_out_logvar.weight"], w["encoder.conv_out_logvar.bias"], out_dtype=od, cout=4)
        return mean, logvar
