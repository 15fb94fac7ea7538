"""Model construction for the iCD path (API mirror of the reference's utils/loading.py).

`load_models(model_id, device, reverse_checkpoint, forward_checkpoint, r=64, w_embed_dim=0, teacher_checkpoint=None,
dtype='fp32') -> (ldm_stable, reverse_cons_model, forward_cons_model)` and `load_models_xl(model_id, reverse_checkpoint,
forward_checkpoint, teacher_checkpoint) -> (stable_pipe, pipe, forw_pipe)` keep the reference signatures
(utils/loading.py:27-90, 93-147).  What they build are the duck-typed pipelines of pipelines.py around native UNets.

Weight sources
  * `model_id` = a local directory in diffusers layout (`unet/diffusion_pytorch_model.safetensors`), or
    "synthetic:sd15" / "synthetic:sdxl" (seeded synthetic weights - there are no checkpoints offline);
  * `teacher_checkpoint`: a torch-saved UNet state dict incl. `time_embedding.cond_proj.weight` (utils/loading.py:52-54);
  * `reverse_checkpoint` / `forward_checkpoint`: peft-keyed LoRA safetensors (`unet.base_model.model.<path>.lora_{A,B}.weight`,
    utils/loading.py:10-23), rank r, alpha 8.  LoRA is FUSED into the base weights at load time, exactly like the
    reference's `load_lora_weights` + `fuse_lora()` - there is no LoRA arithmetic at run time:
        W' = W + (alpha / r) * (B @ A)          (convs flattened to matrices; alpha = 8 -> 0.125 at r = 64)
"""
import dataclasses
import os
import re

import torch

from .pipelines import StableDiffusionPipeline, StableDiffusionXLImg2ImgPipeline, StableDiffusionXLPipeline
from .schedulers import DDIMScheduler
from .synthetic import (SyntheticTextEncoder, SyntheticTokenizer, SyntheticVAE, synthetic_lora, synthetic_state_dict)
from .unet import UNet2DConditionModel
from .unet_config import SD15, SDXL

LORA_ALPHA = 8.0
_PEFT_KEY = re.compile(r"^(?:unet\.)?base_model\.model\.(.+)\.lora_([AB])(?:\.default)?\.weight$")


def parse_peft_lora(tensors):
    """{peft key: tensor} -> {diffusers module path: (down = lora_A, up = lora_B)}."""
    halves = {}
    for key, t in tensors.items():
        m = _PEFT_KEY.match(key)
        if not m:
            raise KeyError(f"unexpected LoRA key '{key}' (want unet.base_model.model.<path>.lora_A|B.weight)")
        halves.setdefault(m.group(1), {})[m.group(2)] = t
    out = {}
    for path, ab in halves.items():
        if set(ab) != {"A", "B"}:
            raise KeyError(f"LoRA module '{path}' is missing lora_A or lora_B")
        out[path] = (ab["A"], ab["B"])
    return out


def fuse_lora(state_dict, lora, alpha=LORA_ALPHA, lora_dtype=torch.float16):
    """Returns a new state dict with every LoRA pair folded into its base weight (fp32 arithmetic, like diffusers)."""
    fused = dict(state_dict)
    for path, (down, up) in lora.items():
        key = path + ".weight"
        if key not in state_dict:
            raise KeyError(f"LoRA targets '{path}' which is not a UNet module")
        W = state_dict[key]
        r = down.shape[0]
        d32 = down.to(lora_dtype).float().flatten(1)
        u32 = up.to(lora_dtype).float().flatten(1)
        delta = (u32.to(W.device) @ d32.to(W.device)).reshape(W.shape)
        fused[key] = (W.float() + (alpha / r) * delta).to(W.dtype)
    return fused


def _load_lora(spec, cfg):
    if spec is None:
        return None
    if isinstance(spec, dict):
        first = next(iter(spec.values()))
        return spec if isinstance(first, tuple) else parse_peft_lora(spec)
    if isinstance(spec, str) and spec.startswith("synthetic"):
        seed = int(spec.split(":")[1]) if ":" in spec else 1
        return synthetic_lora(cfg, seed=seed)
    from safetensors.torch import load_file
    return parse_peft_lora(load_file(spec))


def _load_unet_state(model_id, cfg, teacher_checkpoint):
    if isinstance(model_id, str) and model_id.startswith("synthetic"):
        parts = model_id.split(":")
        sd = synthetic_state_dict(cfg, seed=int(parts[2]) if len(parts) > 2 else 0)
    else:
        from safetensors.torch import load_file
        path = os.path.join(model_id, "unet", "diffusion_pytorch_model.safetensors")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (model_id must be a local diffusers directory or 'synthetic:<arch>')")
        sd = load_file(path)
    if teacher_checkpoint is not None:
        print(f'Embedded model is loading from {teacher_checkpoint}')
        sd = teacher_checkpoint if isinstance(teacher_checkpoint, dict) else torch.load(teacher_checkpoint, map_location="cpu")
    elif cfg.time_cond_proj_dim and "time_embedding.cond_proj.weight" not in sd:
        print('PROVIDE TEACHER')          # the reference prints this and continues with a randomly initialised cond_proj
        g = torch.Generator().manual_seed(0)
        sd["time_embedding.cond_proj.weight"] = torch.randn(cfg.block_out_channels[0], cfg.time_cond_proj_dim, generator=g) \
            * cfg.time_cond_proj_dim ** -0.5
    return sd


def _build_vae(comp, cfg, device, dtype):
    """components['vae_state_dict'] (diffusers AutoencoderKL layout) -> the HIP AutoencoderKL of this package."""
    if "vae" not in comp and comp.get("vae_state_dict") is not None:
        from .vae import AutoencoderKL
        comp["vae"] = AutoencoderKL(cfg, comp["vae_state_dict"], device=device, dtype=dtype)


def _build_text_encoders(comp, device, dtype, xl):
    """components['text_encoder_state_dict'] (and '_2' for SDXL) in transformers' CLIP key layout -> HIP CLIPTextModel."""
    from .clip import CLIPTextModel, CLIP_VIT_L, OPENCLIP_BIGG
    if "text_encoder" not in comp and comp.get("text_encoder_state_dict") is not None:
        comp["text_encoder"] = CLIPTextModel(CLIP_VIT_L, comp["text_encoder_state_dict"], False, device, dtype)
    if xl and "text_encoder_2" not in comp and comp.get("text_encoder_2_state_dict") is not None:
        comp["text_encoder_2"] = CLIPTextModel(OPENCLIP_BIGG, comp["text_encoder_2_state_dict"], True, device, dtype)


def load_models(model_id, device, reverse_checkpoint, forward_checkpoint, r=64, w_embed_dim=0, teacher_checkpoint=None,
                dtype='fp32', components=None, unet_config=None):
    """SD1.5: (ldm_stable, reverse_cons_model, forward_cons_model).  `components` may supply real
    {'vae','tokenizer','text_encoder'} objects or 'vae_state_dict' (AutoencoderKL weights, run on the HIP kernels);
    otherwise labelled synthetic stand-ins are attached.  `unet_config` (not in the reference) overrides the SD1.5
    architecture (reduced-width checkpoints in tests).

    dtype: the reference's default 'fp32' runs diffusers in fp32 (utils/loading.py:34,38).  This executor multiplies fp16 x fp16
    into fp32 accumulators on the matrix cores either way; 'fp32' selects fp32 latents / eps at the UNet boundary and fp32
    boundary-step arithmetic.  In both modes the residual stream carries its rounding error (every x <- x + f(x) chain keeps one bf8
    byte per element beside the fp16 value, so the dominant error term of fp16 activation storage is gone), and the UNet's precision policy
    (unet.UNet2DConditionModel.precision, default "auto") lets the consumers that matter read that byte too inside editing pipelines -
    DESIGN.md section 6 has the measured distance to an fp32 evaluation (eps rel-L2 0.7e-3 for plain generation, 0.4e-3 at the accurate
    level; 1.1e-3 with unet.set_option('residual', 0))."""
    tdtype = torch.float32 if dtype == 'fp32' else torch.float16
    cfg = dataclasses.replace(unet_config or SD15, time_cond_proj_dim=int(w_embed_dim))
    if w_embed_dim > 0:
        print(f'Forward CD is initialized with guidance embedding, dim {w_embed_dim}')
    comp = dict(components or {})
    comp.setdefault("tokenizer", SyntheticTokenizer())
    _build_text_encoders(comp, device, tdtype, xl=False)
    comp.setdefault("text_encoder", SyntheticTextEncoder(cfg.cross_dim, device))
    from .vae import SD_VAE
    _build_vae(comp, SD_VAE, device, tdtype)
    comp.setdefault("vae", SyntheticVAE(device))
    base_sd = _load_unet_state(model_id, cfg, teacher_checkpoint)

    def pipeline(sd):
        unet = UNet2DConditionModel(cfg, sd, device=device, dtype=tdtype)
        sched = DDIMScheduler.sd15()
        return StableDiffusionPipeline(unet, sched, comp["vae"], comp["tokenizer"], comp["text_encoder"], device, tdtype)

    ldm_stable = pipeline(base_sd)
    out = [ldm_stable]
    for name, ckpt in (("Reverse", reverse_checkpoint), ("Forward", forward_checkpoint)):
        if ckpt is None:
            out.append(None)
            continue
        print(f'{name} CD is loading from {ckpt if isinstance(ckpt, str) else "<in-memory LoRA>"}')
        out.append(pipeline(fuse_lora(base_sd, _load_lora(ckpt, cfg), lora_dtype=torch.float16)))     # loading.py:68,82
    return tuple(out)


def load_models_xl(model_id, reverse_checkpoint, forward_checkpoint, teacher_checkpoint, device="cuda", components=None,
                   unet_config=None):
    """SDXL: (stable_pipe, pipe, forw_pipe); fp16 UNets, LoRA fused in fp32 (utils/loading.py:122,141)."""
    cfg = unet_config or SDXL
    comp = dict(components or {})
    from .vae import SDXL_VAE
    _build_vae(comp, SDXL_VAE, device, torch.float16)
    _build_text_encoders(comp, device, torch.float16, xl=True)
    base_sd = _load_unet_state(model_id, cfg, teacher_checkpoint)

    def pipeline(sd, cls):
        unet = UNet2DConditionModel(cfg, sd, device=device, dtype=torch.float16)
        return cls(unet, DDIMScheduler.sdxl(), comp.get("vae"), comp.get("tokenizer"), comp.get("tokenizer_2"),
                   comp.get("text_encoder"), comp.get("text_encoder_2"), device)

    stable_pipe = pipeline(base_sd, StableDiffusionXLImg2ImgPipeline)
    print(f'Reverse CD is loading from {reverse_checkpoint if isinstance(reverse_checkpoint, str) else "<in-memory LoRA>"}')
    pipe = pipeline(fuse_lora(base_sd, _load_lora(reverse_checkpoint, cfg), lora_dtype=torch.float32), StableDiffusionXLPipeline)
    print(f'Forward CD is loading from {forward_checkpoint if isinstance(forward_checkpoint, str) else "<in-memory LoRA>"}')
    forw_pipe = pipeline(fuse_lora(base_sd, _load_lora(forward_checkpoint, cfg), lora_dtype=torch.float32),
                         StableDiffusionXLImg2ImgPipeline)
    return stable_pipe, pipe, forw_pipe
