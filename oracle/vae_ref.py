"""CPU oracle for the VAE either side of the iCD path - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(invertible_cd_amd/vae.py) never does.

What it restates: diffusers==0.25.1 `AutoencoderKL` as the reference calls it -
  decode:  utils/generation.py:255-262   (`latents = 1 / 0.18215 * latents; vae.decode(latents)['sample']`),
           utils/generation_sdxl.py:465-468 (`vae.decode(latents / vae.config.scaling_factor)`, VAE upcast to fp32);
  encode:  utils/generation.py:264-284   (`vae.encode(image)['latent_dist'].mean * 0.18215`),
           utils/generation_sdxl.py:273-276 (img2img `prepare_latents`).
diffusers is a third-party dependency that is absent from /root/reference and from this image (requirements/req.txt:1
pins 0.25.1), so - exactly as for the UNet (oracle/unet_ref.py) - this is a restatement of the published architecture in
plain torch fp32 ops, validated op by op against torch.nn.functional and pinned structurally by the exact parameter
count of the SD VAE (83,653,863) and by the state-dict key / shape layout.  PARITY UNPINNED against real diffusers
outputs: neither the library nor a checkpoint can be run here.

Architecture (SD1.5 `vae` and the SDXL VAE share it; only `scaling_factor` differs: 0.18215 / 0.13025):
  Encoder: conv_in 3->128; DownEncoderBlock2D x4 with channels (128, 256, 512, 512), 2 ResnetBlock2D each (temb=None,
           GroupNorm(32, eps 1e-6) -> SiLU -> conv3x3, 1x1 conv_shortcut when in != out), Downsample2D on the first three
           (F.pad(x, (0,1,0,1)) then conv3x3 stride 2 pad 0); UNetMidBlock2D = resnet, Attention(1 head of 512, GroupNorm
           inside, biased q/k/v/out, residual), resnet; GroupNorm -> SiLU -> conv_out 512->8; quant_conv 1x1 8->8;
           DiagonalGaussianDistribution.mean = first 4 channels.
  Decoder: post_quant_conv 1x1 4->4; conv_in 4->512; mid block; UpDecoderBlock2D x4 with channels (512, 512, 256, 128),
           3 resnets each, Upsample2D (nearest 2x then conv3x3) on the first three; GroupNorm -> SiLU -> conv_out 128->3.
"""
import torch
import torch.nn.functional as F

SD_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
              layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
SDXL_VAE = dict(SD_VAE, scaling_factor=0.13025)
EPS = 1e-6


def _resnet_shapes(p, cin, cout, out):
    out[p + "norm1.weight"] = (cin,); out[p + "norm1.bias"] = (cin,)
    out[p + "conv1.weight"] = (cout, cin, 3, 3); out[p + "conv1.bias"] = (cout,)
    out[p + "norm2.weight"] = (cout,); out[p + "norm2.bias"] = (cout,)
    out[p + "conv2.weight"] = (cout, cout, 3, 3); out[p + "conv2.bias"] = (cout,)
    if cin != cout:
        out[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); out[p + "conv_shortcut.bias"] = (cout,)


def _mid_shapes(p, c, out):
    _resnet_shapes(p + "resnets.0.", c, c, out)
    a = p + "attentions.0."
    out[a + "group_norm.weight"] = (c,); out[a + "group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[a + n + ".weight"] = (c, c); out[a + n + ".bias"] = (c,)
    _resnet_shapes(p + "resnets.1.", c, c, out)


def param_shapes(cfg):
    """diffusers AutoencoderKL state-dict keys -> shapes."""
    ch = list(cfg["block_out_channels"])
    L, zc = cfg["layers_per_block"], cfg["latent_channels"]
    out = {}
    out["encoder.conv_in.weight"] = (ch[0], cfg["in_channels"], 3, 3); out["encoder.conv_in.bias"] = (ch[0],)
    cin = ch[0]
    for i, c in enumerate(ch):
        for j in range(L):
            _resnet_shapes(f"encoder.down_blocks.{i}.resnets.{j}.", cin, c, out)
            cin = c
        if i < len(ch) - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (c, c, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c,)
    _mid_shapes("encoder.mid_block.", ch[-1], out)
    out["encoder.conv_norm_out.weight"] = (ch[-1],); out["encoder.conv_norm_out.bias"] = (ch[-1],)
    out["encoder.conv_out.weight"] = (2 * zc, ch[-1], 3, 3); out["encoder.conv_out.bias"] = (2 * zc,)
    out["quant_conv.weight"] = (2 * zc, 2 * zc, 1, 1); out["quant_conv.bias"] = (2 * zc,)
    out["post_quant_conv.weight"] = (zc, zc, 1, 1); out["post_quant_conv.bias"] = (zc,)
    rev = ch[::-1]
    out["decoder.conv_in.weight"] = (rev[0], zc, 3, 3); out["decoder.conv_in.bias"] = (rev[0],)
    _mid_shapes("decoder.mid_block.", rev[0], out)
    cin = rev[0]
    for i, c in enumerate(rev):
        for j in range(L + 1):
            _resnet_shapes(f"decoder.up_blocks.{i}.resnets.{j}.", cin, c, out)
            cin = c
        if i < len(rev) - 1:
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    out["decoder.conv_norm_out.weight"] = (rev[-1],); out["decoder.conv_norm_out.bias"] = (rev[-1],)
    out["decoder.conv_out.weight"] = (cfg["out_channels"], rev[-1], 3, 3); out["decoder.conv_out.bias"] = (cfg["out_channels"],)
    return out


def count_params(cfg):
    n = 0
    for s in param_shapes(cfg).values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n


def _resnet(w, p, x, groups):
    h = F.silu(F.group_norm(x, groups, w[p + "norm1.weight"], w[p + "norm1.bias"], EPS))
    h = F.conv2d(h, w[p + "conv1.weight"], w[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, w[p + "norm2.weight"], w[p + "norm2.bias"], EPS))
    h = F.conv2d(h, w[p + "conv2.weight"], w[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in w:
        x = F.conv2d(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"])
    return x + h


def _attention(w, p, x, groups):
    """diffusers Attention as the VAE mid block builds it: one head of C channels, biased projections, residual."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, w[p + "group_norm.weight"], w[p + "group_norm.bias"], EPS)
    h = h.reshape(B, C, H * W).transpose(1, 2)
    q = F.linear(h, w[p + "to_q.weight"], w[p + "to_q.bias"])
    k = F.linear(h, w[p + "to_k.weight"], w[p + "to_k.bias"])
    v = F.linear(h, w[p + "to_v.weight"], w[p + "to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1)
    o = F.linear(a @ v, w[p + "to_out.0.weight"], w[p + "to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def _mid(w, p, x, groups):
    x = _resnet(w, p + "resnets.0.", x, groups)
    x = _attention(w, p + "attentions.0.", x, groups)
    return _resnet(w, p + "resnets.1.", x, groups)


def encode_moments(w, cfg, x):
    """image [B,3,H,W] in [-1,1] -> moments [B,8,H/8,W/8] (mean ‖ logvar) = quant_conv(encoder(x))."""
    g, ch, L = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    h = F.conv2d(x, w["encoder.conv_in.weight"], w["encoder.conv_in.bias"], padding=1)
    for i in range(len(ch)):
        for j in range(L):
            h = _resnet(w, f"encoder.down_blocks.{i}.resnets.{j}.", h, g)
        if i < len(ch) - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = F.conv2d(h, w[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         w[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = _mid(w, "encoder.mid_block.", h, g)
    h = F.silu(F.group_norm(h, g, w["encoder.conv_norm_out.weight"], w["encoder.conv_norm_out.bias"], EPS))
    h = F.conv2d(h, w["encoder.conv_out.weight"], w["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, w["quant_conv.weight"], w["quant_conv.bias"])


def encode_mean(w, cfg, x):
    """`vae.encode(x)['latent_dist'].mean` (utils/generation.py:277,282)."""
    return encode_moments(w, cfg, x)[:, :cfg["latent_channels"]]


def decode(w, cfg, z):
    """`vae.decode(z)['sample']`: z [B,4,h,w] (already divided by the scaling factor by the caller) -> [B,3,8h,8w]."""
    g, L = cfg["norm_num_groups"], cfg["layers_per_block"]
    rev = list(cfg["block_out_channels"])[::-1]
    h = F.conv2d(z, w["post_quant_conv.weight"], w["post_quant_conv.bias"])
    h = F.conv2d(h, w["decoder.conv_in.weight"], w["decoder.conv_in.bias"], padding=1)
    h = _mid(w, "decoder.mid_block.", h, g)
    for i in range(len(rev)):
        for j in range(L + 1):
            h = _resnet(w, f"decoder.up_blocks.{i}.resnets.{j}.", h, g)
        if i < len(rev) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, w[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         w[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(F.group_norm(h, g, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], EPS))
    return F.conv2d(h, w["decoder.conv_out.weight"], w["decoder.conv_out.bias"], padding=1)
