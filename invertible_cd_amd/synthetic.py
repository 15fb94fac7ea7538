"""Seeded synthetic weights / inputs in the diffusers state-dict layout (no checkpoints exist offline).

SURVEY.md section 8d: fan-in-scaled normals, variance kept O(1) through the 4 steps; residual-branch output layers are
damped so the fp16 residual stream stays far from overflow.  Every tensor is generated from its own seed
(hash of the key), so the dict is independent of iteration order and can be produced on CPU (parity with the oracle)
or directly on the GPU (bench at full size).
"""
import zlib

import torch

from .unet_config import UNetConfig, LORA_TARGETS

_DAMPED = (".conv2.weight", ".to_out.0.weight", ".ff.net.2.weight", ".proj_out.weight")


def _gen(key, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synthetic_state_dict(cfg: UNetConfig, seed: int = 0, device="cpu", dtype=torch.float32):
    sd = {}
    for key, shape in cfg.state_dict_shapes().items():
        g = _gen(key, seed, device)
        is_norm = ".norm" in key or key.startswith("conv_norm_out")
        if key.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        elif is_norm:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 0.5 if key.endswith(_DAMPED) else 1.0
            t = torch.randn(shape, generator=g, device=device) * (gain / fan_in ** 0.5)
        sd[key] = t.to(dtype)
    return sd


def synthetic_vae_state_dict(cfg, seed: int = 0, device="cpu", dtype=torch.float32):
    """Seeded AutoencoderKL weights (vae.VAEConfig.state_dict_shapes layout), same recipe as the UNet's."""
    sd = {}
    for key, shape in cfg.state_dict_shapes().items():
        g = _gen("vae:" + key, seed, device)
        if key.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        elif "norm" in key:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 0.5 if key.endswith((".conv2.weight", ".to_out.0.weight")) else 1.0
            t = torch.randn(shape, generator=g, device=device) * (gain / fan_in ** 0.5)
        sd[key] = t.to(dtype)
    return sd


def synthetic_clip_state_dict(cfg, with_projection=False, seed: int = 0, device="cpu", dtype=torch.float32, prefix="text_model."):
    """Seeded CLIP text-encoder weights in the checkpoint key layout (clip.CLIPTextConfig.state_dict_shapes)."""
    sd = {}
    for key, shape in cfg.state_dict_shapes(with_projection).items():
        g = _gen("clip:" + key, seed, device)
        if key.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        elif "layer_norm" in key:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif "embedding" in key:
            t = 0.5 * torch.randn(shape, generator=g, device=device)
        else:
            gain = 0.5 if key.endswith(("out_proj.weight", "fc2.weight")) else 1.0
            t = torch.randn(shape, generator=g, device=device) * (gain / shape[-1] ** 0.5)
        sd[(prefix if key != "text_projection.weight" else "") + key] = t.to(dtype)
    return sd


def synthetic_lora(cfg: UNetConfig, seed: int = 1, rank: int = 64, device="cpu", scale: float = 0.5):
    """{module path: (down [r, in(,k,k)], up [out, r(,1,1)])} for every LoRA target of the iCD students."""
    lora = {}
    for key, shape in cfg.state_dict_shapes().items():
        if not key.endswith(".weight"):
            continue
        path = key[: -len(".weight")]
        if not any(path.endswith("." + t) or path.endswith(t) for t in LORA_TARGETS):
            continue
        if ".norm" in path:
            continue
        g = _gen("lora:" + path, seed, device)
        out_c, in_c = shape[0], shape[1]
        r = min(rank, in_c, out_c)
        if len(shape) == 4:
            down = torch.randn((r, in_c, shape[2], shape[3]), generator=g, device=device) / (in_c * shape[2] * shape[3]) ** 0.5
            up = torch.randn((out_c, r, 1, 1), generator=g, device=device) * (scale / r ** 0.5)
        else:
            down = torch.randn((r, in_c), generator=g, device=device) / in_c ** 0.5
            up = torch.randn((out_c, r), generator=g, device=device) * (scale / r ** 0.5)
        lora[path] = (down, up)
    return lora


def synthetic_inputs(cfg: UNetConfig, batch: int, height: int, width: int, seed: int = 0, n_ctx: int = 77, device="cpu"):
    """Latents / context / (SDXL) pooled embeds + time ids, fp32 on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed + 12345)
    out = {
        "latents": torch.randn((batch, cfg.in_channels, height, width), generator=g, device=device),
        "context": torch.randn((batch, n_ctx, cfg.cross_dim), generator=g, device=device),
    }
    if cfg.addition_time_embed_dim:
        out["text_embeds"] = torch.randn((batch, cfg.pooled_dim), generator=g, device=device)
        out["time_ids"] = torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]] * batch, device=device)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Labelled SYNTHETIC stand-ins for the components on either side of the U-Net path (tokenizer, CLIP text encoder,
# VAE: SURVEY.md section 8f ranks 1 and 3 - not rebuilt here, no weights/vocab exist offline).  They let the
# reference-shaped API (runner / invert / init_prompt / latent2image) run end to end on synthetic data.
# ---------------------------------------------------------------------------------------------------------------
class _Encoding:
    def __init__(self, ids):
        self.input_ids = ids


class SyntheticTokenizer:
    """Whitespace tokenizer with CLIP-like framing: [BOS] words... [EOS] padded with EOS to model_max_length = 77."""
    model_max_length = 77
    bos_token_id, eos_token_id = 49406, 49407

    def __init__(self):
        self._words = {}

    def _id(self, word):
        i = zlib.crc32(word.encode()) % 49000 + 1
        self._words.setdefault(i, word)
        return i

    def encode(self, text):
        return [self.bos_token_id] + [self._id(w) for w in text.split(" ") if w] + [self.eos_token_id]

    def decode(self, ids):
        return " ".join(self._words.get(int(i), "") for i in ids)

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        L = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = self.encode(t)[:L]
            ids[-1] = self.eos_token_id
            rows.append(ids + [self.eos_token_id] * (L - len(ids)))
        return _Encoding(torch.tensor(rows, dtype=torch.long))


class SyntheticTextEncoder:
    """ids [B,77] -> (hidden [B,77,dim],): seeded token + position tables (NOT CLIP)."""

    def __init__(self, dim, device="cuda", dtype=torch.float16, seed=7):
        g = torch.Generator().manual_seed(seed)
        self.tok = torch.randn(4096, dim, generator=g).to(device)
        self.pos = (0.5 * torch.randn(77, dim, generator=g)).to(device)
        self.device, self.dtype = torch.device(device), dtype

    def to(self, device=None, dtype=None):
        if dtype is not None:
            self.dtype = dtype
        return self

    def __call__(self, input_ids, **kw):
        ids = input_ids.to(self.device)
        return ((self.tok[ids % 4096] + self.pos[: ids.shape[1]]).to(self.dtype),)


class _LatentDist:
    def __init__(self, mean):
        self.mean = mean

    def sample(self, generator=None):
        return self.mean


class SyntheticVAE:
    """8x down/up-sampling linear stand-in for AutoencoderKL (NOT a VAE): encode = 8x8 mean pool + 3->4 mix, decode = 4->3 mix +
    nearest 8x upsample.  Interface: encode(x)['latent_dist'].mean, decode(z)['sample'], .dtype, .config.scaling_factor."""

    def __init__(self, device="cuda", dtype=torch.float16):
        self.enc = torch.tensor([[0.6, 0.3, 0.1], [-0.3, 0.5, -0.2], [0.2, -0.4, 0.6], [0.5, 0.5, -0.5]]).to(device)
        self.dec = torch.linalg.pinv(self.enc.cpu()).to(device)
        self.dtype = dtype
        import types
        self.config = types.SimpleNamespace(scaling_factor=0.18215)

    def to(self, *a, dtype=None, **k):
        for x in a:
            if isinstance(x, torch.dtype):
                dtype = x
        if dtype is not None:
            self.dtype = dtype
        return self

    def encode(self, x):
        z = torch.nn.functional.avg_pool2d(x.float(), 8)
        z = torch.einsum("lc,bchw->blhw", self.enc.to(z.device), z).to(x.dtype)
        return {"latent_dist": _LatentDist(z)}

    def decode(self, z, return_dict=True):
        im = torch.einsum("cl,blhw->bchw", self.dec.to(z.device), z.float())
        im = torch.nn.functional.interpolate(im, scale_factor=8.0, mode="nearest").to(z.dtype)
        return {"sample": im} if return_dict else (im,)
