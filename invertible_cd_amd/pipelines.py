"""Duck-typed pipeline containers: exactly the attributes the iCD sampler reads from diffusers' pipelines
(SURVEY.md section 8b "Attributes of model/pipe the reference touches").

They hold the native UNet plus the out-of-path components (VAE, tokenizer, text encoder - SURVEY.md section 8f ranks
1 and 3, not rebuilt in this round: pass real ones in, or the labelled synthetic stand-ins of synthetic.py).
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

    def preprocess(self, image):
        return image


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

    def encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance):
        raise NotImplementedError("text encoding is outside the U-Net hot path: pass compute_embeddings_fn")


class StableDiffusionXLImg2ImgPipeline(StableDiffusionXLPipeline):
    def prepare_latents(self, image, timestep, batch_size, num_images_per_prompt, dtype, device, generator=None,
                        add_noise=True):
        """img2img: (VAE-encode unless the input already has 4 latent channels) + add_noise at `timestep`."""
        image = image.to(device=device, dtype=dtype)
        if image.shape[1] == 4:
            init = image
        else:
            if self.vae is None:
                raise RuntimeError("no VAE attached: pass 4-channel latents or attach a VAE (out of the U-Net hot path)")
            init = self.vae.encode(image.float()).latent_dist.sample(generator).to(dtype) * self.vae.config.scaling_factor
        if not add_noise:
            return init
        noise = self._randn(init.shape, generator, device, dtype)
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        return self.scheduler.add_noise(init, noise, t.reshape(-1)[:1])
