"""Duck-typed pipeline containers: exactly the attributes the iCD sampler reads from diffusers' pipelines
(SURVEY.md section 8b "Attributes of model/pipe the reference touches").

They hold the native UNet plus the components either side of it: the HIP AutoencoderKL (vae.py) and CLIP text encoders
(clip.py) of this package, or caller-supplied objects; tokenizers need a BPE vocabulary that is not available offline, so
callers pass token ids / `compute_embeddings_fn` (labelled synthetic stand-ins live in synthetic.py).
"""
import types

import torch


class StableDiffusionPipeline:
    """`.unet .vae .tokenizer .text_encoder .scheduler .device .dtype` (utils/generation.py:185-188,256-303,539)."""

    def __init__(self, unet, scheduler, vae=None, tokenizer=None, text_encoder=None, device="cuda", dtype=torch.float16):
        self.unet, self.scheduler, self.vae = unet, scheduler, vae
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.device = torch.device(device)
        self.dtype = dtype

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
            self.unet.to(dtype)
            for m in (self.vae, self.text_encoder):
                if m is not None and hasattr(m, "to"):
                    m.to(dtype=dtype)
        return self


class _ImageProcessor:
    def postprocess(self, image, output_type="pil", do_denormalize=None):
        image = (image / 2 + 0.5).clamp(0, 1)
        if output_type != "pil":
            return image
        from PIL import Image
        arr = (image.cpu().permute(0, 2, 3, 1).float().numpy() * 255).round().astype("uint8")
        return [Image.fromarray(a) for a in arr]

    def __init__(self, vae_scale_factor=8):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height=None, width=None):
        """diffusers VaeImageProcessor.preprocess as running/sdxl/edit.py:198-199 uses it: PIL image(s) / HWC numpy in
        [0,1] / NCHW tensor -> float32 NCHW in [-1,1], size rounded down to a multiple of the VAE scale factor.  Tensors
        that already have 4 latent channels, or that already contain negative values, pass through un-normalised."""
        import numpy as np
        if torch.is_tensor(image) or (isinstance(image, (list, tuple)) and len(image) and torch.is_tensor(image[0])):
            x = image if torch.is_tensor(image) else torch.cat([i if i.dim() == 4 else i[None] for i in image], 0)
            if x.dim() == 3:
                x = x[None]
            if x.shape[1] == 4:
                return x
            h = (height or x.shape[2]) // self.vae_scale_factor * self.vae_scale_factor
            w = (width or x.shape[3]) // self.vae_scale_factor * self.vae_scale_factor
            if (h, w) != tuple(x.shape[2:]):
                x = torch.nn.functional.interpolate(x, size=(h, w))
            return x if float(x.min()) < 0 else 2.0 * x - 1.0
        imgs = list(image) if isinstance(image, (list, tuple)) else [image]
        arrs = []
        for im in imgs:
            if isinstance(im, np.ndarray):
                a = im if im.ndim == 4 else im[None]
            else:                                          # PIL
                w0, h0 = im.size
                h = (height or h0) // self.vae_scale_factor * self.vae_scale_factor
                w = (width or w0) // self.vae_scale_factor * self.vae_scale_factor
                if (w, h) != (w0, h0):
                    from PIL import Image
                    im = im.resize((w, h), resample=Image.LANCZOS)
                a = np.asarray(im.convert("RGB"), dtype=np.float32)[None] / 255.0
            arrs.append(a.astype(np.float32))
        x = torch.from_numpy(np.concatenate(arrs, 0)).permute(0, 3, 1, 2).contiguous()
        return 2.0 * x - 1.0


class StableDiffusionXLPipeline:
    """What utils/generation_sdxl.py:346-347,355,404-420,465-468 and running/sdxl/generate.py:160-161 read."""

    vae_scale_factor = 8

    def __init__(self, unet, scheduler, vae=None, tokenizer=None, tokenizer_2=None, text_encoder=None, text_encoder_2=None,
                 device="cuda"):
        self.unet, self.scheduler, self.vae = unet, scheduler, vae
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self._execution_device = torch.device(device)
        self.device = self._execution_device
        self.image_processor = _ImageProcessor()
        self.force_zeros_for_empty_prompt = True     # SDXL-base's pipeline config (diffusers): see encode_prompt

    def to(self, device=None, dtype=None):
        if device is not None:
            self._execution_device = self.device = torch.device(device)
        return self

    @staticmethod
    def _randn(shape, generator, device, dtype):
        # diffusers.utils.randn_tensor: a CPU generator draws on the CPU (in the target dtype), then moves
        gen_dev = generator.device.type if generator is not None else torch.device(device).type
        if gen_dev == "cpu":
            return torch.randn(shape, generator=generator, device="cpu", dtype=dtype).to(device)
        return torch.randn(shape, generator=generator, device=device, dtype=dtype)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """txt2img: randn [B,4,h/8,w/8] * init_noise_sigma."""
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        latents = self._randn(shape, generator, device, dtype) if latents is None else latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    @torch.no_grad()
    def encode_prompt(self, prompt, device=None, num_images_per_prompt=1, do_classifier_free_guidance=False, negative_prompt=None):
        """The `compute_embeddings_fn=None` branch of the samplers: `pipe.encode_prompt(prompt, device, 1, False)[0]`
        (utils/generation_sdxl.py:242,374), served by the text encoders attached to this pipeline (clip.py on the HIP operators,
        or any object with the transformers call interface).  With both encoders it follows diffusers' SDXL rule - penultimate
        hidden states of the two encoders concatenated ([B, 77, 768 + 1280]), pooled output of the second; with one encoder the
        SD rule - `text_encoder(ids)[0]`.  Returns (prompt_embeds, negative_prompt_embeds, pooled, negative_pooled); the negative
        pair encodes `negative_prompt` (default "") and is None without classifier-free guidance."""
        pairs = [(t, e) for t, e in ((self.tokenizer, self.text_encoder), (self.tokenizer_2, self.text_encoder_2)) if e is not None]
        if not pairs or any(t is None for t, _ in pairs):
            raise RuntimeError("encode_prompt: attach tokenizer(s) and text_encoder(s) to the pipeline (load_models_xl components "
                               "'text_encoder_state_dict' / 'text_encoder_2_state_dict'), or pass compute_embeddings_fn")
        device = torch.device(device) if device is not None else self._execution_device
        texts = [prompt] if isinstance(prompt, str) else list(prompt)

        def run(batch):
            hidden, pooled = [], None
            for tok, enc in pairs:
                ids = tok(batch, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
                if len(pairs) == 1:
                    return enc(ids.to(enc.device))[0].to(device), None
                out = enc(ids.to(enc.device), output_hidden_states=True)
                pooled = out[0]
                hidden.append(out.hidden_states[-2])
            emb = torch.cat(hidden, dim=-1)
            return emb.to(device), pooled.reshape(emb.shape[0], -1).to(device)

        rep = lambda t: None if t is None else t.repeat_interleave(num_images_per_prompt, dim=0)
        emb, pooled = run(texts)
        neg = neg_pooled = None
        if do_classifier_free_guidance:
            if negative_prompt is None and getattr(self, "force_zeros_for_empty_prompt", True):
                # diffusers' SDXL pipelines (config.force_zeros_for_empty_prompt, True for SDXL-base): no negative prompt means ZERO
                # negative embeddings and pooled output, not the encoding of the empty string
                neg, neg_pooled = torch.zeros_like(emb), None if pooled is None else torch.zeros_like(pooled)
            else:
                negs = [negative_prompt or ""] * len(texts) if not isinstance(negative_prompt, (list, tuple)) else list(negative_prompt)
                neg, neg_pooled = run(negs)
        return rep(emb.to(self.unet.dtype)), rep(None if neg is None else neg.to(self.unet.dtype)), rep(pooled), rep(neg_pooled)


class StableDiffusionXLImg2ImgPipeline(StableDiffusionXLPipeline):
    def prepare_latents(self, image, timestep, batch_size, num_images_per_prompt, dtype, device, generator=None,
                        add_noise=True):
        """diffusers 0.25.1 StableDiffusionXLImg2ImgPipeline.prepare_latents, the call of utils/generation_sdxl.py:273-276:
        image (PIL / list / [B,3,H,W] in [-1,1]) -> VAE encode in fp32 -> latent_dist.sample(generator) * scaling_factor
        (4-channel inputs are taken as latents), repeated to batch_size, + add_noise(randn, timestep).  Both draws come
        from `generator` in this order, as in diffusers."""
        if not torch.is_tensor(image):
            image = self.image_processor.preprocess(image)
        image = image.to(device=device, dtype=dtype)
        batch_size = batch_size * num_images_per_prompt
        if image.shape[1] == 4:
            init = image
        else:
            if self.vae is None:
                raise RuntimeError("no VAE attached: pass 4-channel latents or attach a VAE (components['vae_state_dict'])")
            self.vae.to(torch.float32)                       # force_upcast, as diffusers does for the SDXL VAE
            init = self.vae.encode(image.float()).latent_dist.sample(generator)
            self.vae.to(dtype)
            init = init.to(dtype) * self.vae.config.scaling_factor
        if batch_size > init.shape[0]:
            if batch_size % init.shape[0] != 0:
                raise ValueError(f"Cannot duplicate `image` of batch size {init.shape[0]} to {batch_size} text prompts.")
            init = torch.cat([init] * (batch_size // init.shape[0]), dim=0)
        if not add_noise:
            return init
        noise = self._randn(init.shape, generator, device, dtype)
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        return self.scheduler.add_noise(init, noise, t.reshape(-1)[:1])
