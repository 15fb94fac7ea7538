"""SD1.5 iCD sampler / inverter on the native MI355X UNet (API mirror of the reference's utils/generation.py).

Drop-in surface kept: `runner(...)`, `Generator(...)` with `cons_generation`, `cons_inversion`, `get_noise_pred`,
`init_prompt`, `image2latent`, `latent2image`, `ddim_loop`, `ddim_inversion`, `prev_step`, `next_step`, and the helpers
`linear_schedule_old`, `linear_schedule`, `guidance_scale_embedding`, `extract_into_tensor`, `predicted_origin`,
`guided_step`, `latent2image`, `init_latent`, `load_512`.  Semantics (including the reference's quirks, SURVEY.md
section 8a rows a1-a8) are pinned by golden vectors captured from the reference (tests/test_generation_golden.py).

What is different underneath (MI355X-first):
  * the UNet call is one native executor call (no per-op Python), the boundary step is one fused HIP kernel;
  * when `w_embed_dim > 0` the reference computes the CFG-doubled batch and throws the unconditional half away
    (utils/generation.py:245-251).  `Generator.eliminate_dead_uncond` (default True) evaluates the conditional half
    only - identical outputs, half the FLOPs; controllers still see exactly the rows their `forward` saw before;
  * per-step device constants (w-embedding, timestep, boundary coefficients) are cached on the GPU, so the 4-step loop
    issues no host->device copies and never synchronises (the reference syncs at t.item() every step).
"""
from typing import Union

import numpy as np
import torch

from . import p2p


# ----------------------------------------------------------------------------------------------------------- runner
@torch.no_grad()
def runner(model, prompt, controller, solver, is_cons_forward=False, num_inference_steps=50, guidance_scale=7.5,
           generator=None, latent=None, uncond_embeddings=None, start_time=50, return_type='image',
           dynamic_guidance=False, tau1=0.4, tau2=0.6, w_embed_dim=0):
    """utils/generation.py:12-66.  Returns (image | latents, the [1,4,64,64] initial latent)."""
    p2p.register_attention_control(model, controller)
    solver.init_prompt(prompt, None)
    latent, latents = init_latent(latent, model, 512, 512, generator, len(prompt))      # resolution is fixed, :32-34
    model.scheduler.set_timesteps(num_inference_steps)
    dynamic_guidance = tau1 < 1.0                      # the argument is overridden and tau2 is never looked at (:36)
    if is_cons_forward:
        trajectory = solver.cons_generation(latents, guidance_scale=guidance_scale, w_embed_dim=w_embed_dim,
                                            dynamic_guidance=dynamic_guidance, tau1=tau1, tau2=tau2, controller=controller)
    else:
        trajectory = solver.ddim_loop(latents, num_inference_steps, is_forward=False, guidance_scale=guidance_scale,
                                      dynamic_guidance=dynamic_guidance, tau1=tau1, tau2=tau2, w_embed_dim=w_embed_dim,
                                      uncond_embeddings=uncond_embeddings, controller=controller)
    latents = trajectory[-1]
    if return_type == 'image':
        image = latent2image(model.vae, latents.to(model.vae.dtype))
    else:
        image = latents
    return image, latent


# ----------------------------------------------------------------------------------------------------------- schedules
def linear_schedule_old(t, guidance_scale, tau1, tau2):
    """gamma(t/1000) * gs with gamma = 1 below tau1, 0 above tau2, linear in between (utils/generation.py:74-82)."""
    u = t / 1000
    if u <= tau1:
        gamma = 1.0
    elif u >= tau2:
        gamma = 0.0
    else:
        gamma = (tau2 - u) / (tau2 - tau1)
    return gamma * guidance_scale


def linear_schedule(t, guidance_scale, tau1=0.4, tau2=0.8):
    """Classic-CFG variant: gs below tau1, 1.0 above tau2 (utils/generation.py:85-93)."""
    u = t / 1000
    if u <= tau1:
        return guidance_scale
    if u >= tau2:
        return 1.0
    return (tau2 - u) / (tau2 - tau1) * (guidance_scale - 1.0) + 1.0


def guidance_scale_embedding(w, embedding_dim=512, dtype=torch.float32):
    """[sin(1000 w f) || cos(1000 w f)], f_i = exp(-ln(1e4) i / (half - 1))  (utils/generation.py:96-122)."""
    assert len(w.shape) == 1
    w = w * 1000.0
    half_dim = embedding_dim // 2
    step = torch.log(torch.tensor(10000.0)) / (half_dim - 1)
    freqs = torch.exp(torch.arange(half_dim, dtype=dtype) * -step)
    ang = w.to(dtype)[:, None] * freqs[None, :]
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    assert emb.shape == (w.shape[0], embedding_dim)
    return emb


# ----------------------------------------------------------------------------------------------------------- boundary step
def extract_into_tensor(a, t, x_shape):
    b, *_ = t.shape
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def predicted_origin(model_output, timesteps, boundary_timesteps, sample, prediction_type, alphas, sigmas):
    """x0-prediction followed by the jump to the boundary timestep s (utils/generation.py:136-155).

    Generic torch expression (any device); the sampler loops below use the fused HIP kernel icd_x0_step, which is
    bit-identical to this expression evaluated in fp32."""
    sigmas_s = extract_into_tensor(sigmas, boundary_timesteps, sample.shape)
    alphas_s = extract_into_tensor(alphas, boundary_timesteps, sample.shape)
    sigmas_t = extract_into_tensor(sigmas, timesteps, sample.shape)
    alphas_t = extract_into_tensor(alphas, timesteps, sample.shape)
    alphas_s[boundary_timesteps == 0] = 1.0          # hard boundary at s = 0
    sigmas_s[boundary_timesteps == 0] = 0.0
    if prediction_type == "epsilon":
        x0 = (sample - sigmas_t * model_output) / alphas_t
        return alphas_s * x0 + sigmas_s * model_output
    if prediction_type == "v_prediction":
        assert boundary_timesteps == 0, "v_prediction does not support multiple endpoints at the moment"
        return alphas_t * sample - sigmas_t * model_output
    raise ValueError(f"Prediction type {prediction_type} currently not supported.")


def guided_step(noise_prediction_text, noise_pred_uncond, t, guidance_scale, dynamic_guidance=False, tau1=0.4, tau2=0.6):
    if dynamic_guidance:
        if not isinstance(t, int):
            t = t.item()
        guidance_scale = linear_schedule(t, guidance_scale, tau1=tau1, tau2=tau2)
    return noise_pred_uncond + guidance_scale * (noise_prediction_text - noise_pred_uncond)


class _editing:
    """`with unet.editing():` when `on` and the model's UNet is the native one (anything else: no-op)."""

    def __init__(self, model, on):
        ctx = getattr(getattr(model, "unet", None), "editing", None)
        self._ctx = ctx() if (on and ctx is not None) else None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)


# ----------------------------------------------------------------------------------------------------------- Generator
class Generator:
    """Few-step consistency sampler / inverter (utils/generation.py:181-521)."""

    eliminate_dead_uncond = True     # skip the unconditional CFG rows when their output is discarded (w_embed_dim > 0)

    def __init__(self, model, n_steps, noise_scheduler, forward_cons_model=None, reverse_cons_model=None, num_endpoints=1,
                 num_forward_endpoints=1, reverse_timesteps=None, forward_timesteps=None, max_forward_timestep_index=49,
                 start_timestep=19):
        self.model = model
        self.forward_cons_model = forward_cons_model
        self.reverse_cons_model = reverse_cons_model
        self.noise_scheduler = noise_scheduler
        self.n_steps = n_steps
        self.tokenizer = self.model.tokenizer
        self.model.scheduler.set_timesteps(n_steps)
        self.prompt = None
        self.context = None
        self.ddim_timesteps = torch.from_numpy(
            (np.arange(1, n_steps + 1) * (1000 // n_steps)).round().astype(np.int64) - 1).long()
        self.start_timestep = start_timestep
        self._dev_cache = {}

        if reverse_timesteps is None or forward_timesteps is None:
            ends, inv_ends = self._create_forward_inverse_timesteps(num_endpoints, n_steps, max_forward_timestep_index)
            self.reverse_timesteps, self.reverse_boundary_timesteps = inv_ends.flip(0), ends.flip(0)
            ends, inv_ends = self._create_forward_inverse_timesteps(num_forward_endpoints, n_steps, max_forward_timestep_index)
            self.forward_timesteps, self.forward_boundary_timesteps = ends, inv_ends
            self.forward_timesteps[0] = self.start_timestep
        else:
            # the reference reverses the CALLER's list in place (utils/generation.py:507-508); kept for drop-in parity
            reverse_timesteps.reverse()
            rev_boundary = reverse_timesteps[1:] + [0]
            fwd_boundary = forward_timesteps[1:] + [999]
            self.reverse_timesteps = torch.tensor(reverse_timesteps)
            self.reverse_boundary_timesteps = torch.tensor(rev_boundary)
            self.forward_timesteps = torch.tensor(forward_timesteps)
            self.forward_boundary_timesteps = torch.tensor(fwd_boundary)
        print(f"Endpoints reverse CTM: {self.reverse_timesteps}, {self.reverse_boundary_timesteps}")
        print(f"Endpoints forward CTM: {self.forward_timesteps}, {self.forward_boundary_timesteps}")

    def _create_forward_inverse_timesteps(self, num_endpoints, n_steps, max_inverse_timestep_index):
        interval = n_steps // num_endpoints + int(n_steps % num_endpoints > 0)
        idx = torch.arange(interval, n_steps, interval) - 1
        inv_idx = torch.tensor(idx.tolist() + [max_inverse_timestep_index])
        endpoints = torch.tensor([0] + self.ddim_timesteps[idx].tolist())
        return endpoints, self.ddim_timesteps[inv_idx]

    @property
    def scheduler(self):
        return self.model.scheduler

    # ------------------------------------------------------------------ DDIM baselines (utils/generation.py:183-205)
    def prev_step(self, model_output, timestep: int, sample):
        prev_t = timestep - self.scheduler.config.num_train_timesteps // self.scheduler.num_inference_steps
        a_t = self.scheduler.alphas_cumprod[timestep]
        a_prev = self.scheduler.alphas_cumprod[prev_t] if prev_t >= 0 else self.scheduler.final_alpha_cumprod
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * model_output

    def next_step(self, model_output, timestep: int, sample):
        timestep, next_t = min(timestep - self.scheduler.config.num_train_timesteps // self.scheduler.num_inference_steps,
                               999), timestep
        a_t = self.scheduler.alphas_cumprod[timestep] if timestep >= 0 else self.scheduler.final_alpha_cumprod
        a_next = self.scheduler.alphas_cumprod[next_t]
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        return a_next ** 0.5 * x0 + (1 - a_next) ** 0.5 * model_output

    def get_noise_pred_single(self, latents, t, context):
        return self.model.unet(latents, t, encoder_hidden_states=context)["sample"]

    # ------------------------------------------------------------------ the UNet call (utils/generation.py:211-253)
    def _w_vector(self, n_doubled, guidance_scale):
        # [0, 0, 0, gs] iff the CFG-doubled batch is exactly 4, else gs everywhere (utils/generation.py:232-235)
        if n_doubled == 4:
            return (0.0, 0.0, 0.0, float(guidance_scale))
        return (float(guidance_scale),) * n_doubled

    def _cached(self, key, make):
        v = self._dev_cache.get(key)
        if v is None:
            if len(self._dev_cache) > 256:
                self._dev_cache.clear()
            v = self._dev_cache[key] = make()
        return v

    def _can_skip_uncond(self, model):
        unet = model.unet
        if not (self.eliminate_dead_uncond and hasattr(unet, "attn_cond_only")):
            return False
        ctrl = getattr(unet, "attn_controller", None)
        return ctrl is None or isinstance(ctrl, (p2p.AttentionControl, p2p.EmptyControl))

    def get_noise_pred(self, model, latent, t, guidance_scale=1, context=None, w_embed_dim=0, dynamic_guidance=False,
                       tau1=0.4, tau2=0.6):
        if context is None:
            context = self.context
        B = len(latent)
        w_embedding = None
        if w_embed_dim > 0:
            if dynamic_guidance:
                t_item = t if isinstance(t, int) else t.item()
                guidance_scale = linear_schedule_old(t_item, guidance_scale, tau1=tau1, tau2=tau2)
            wv = self._w_vector(2 * B, guidance_scale)
            w_embedding = self._cached(("w", wv, w_embed_dim, str(latent.device), latent.dtype), lambda: guidance_scale_embedding(
                torch.tensor(wv), embedding_dim=w_embed_dim).to(device=latent.device, dtype=latent.dtype))
        unet = model.unet
        if w_embedding is not None and self._can_skip_uncond(model):
            # the unconditional half is dead code on this branch (only `noise_prediction_text` is returned): skip it
            unet.attn_cond_only = True
            try:
                out = unet(latent.to(dtype=unet.dtype), t, timestep_cond=w_embedding[B:].to(dtype=unet.dtype),
                           encoder_hidden_states=context[B:])["sample"]
            finally:
                unet.attn_cond_only = False
            return out
        latents_input = torch.cat([latent] * 2)
        noise_pred = unet(latents_input.to(dtype=unet.dtype), t,
                          timestep_cond=w_embedding.to(dtype=unet.dtype) if w_embed_dim > 0 else None,
                          encoder_hidden_states=context)["sample"]
        noise_pred_uncond, noise_prediction_text = noise_pred.chunk(2)
        if guidance_scale > 1 and w_embedding is None:
            return guided_step(noise_prediction_text, noise_pred_uncond, t, guidance_scale, dynamic_guidance, tau1, tau2)
        return noise_prediction_text

    # ------------------------------------------------------------------ VAE / text plumbing (out of the hot path)
    @torch.no_grad()
    def latent2image(self, latents, return_type='np'):
        latents = 1 / 0.18215 * latents.detach()
        image = self.model.vae.decode(latents.to(dtype=self.model.dtype))['sample']
        if return_type == 'np':
            image = (image / 2 + 0.5).clamp(0, 1)
            image = image.cpu().permute(0, 2, 3, 1).numpy()[0]
            image = (image * 255).astype(np.uint8)
        return image

    @torch.no_grad()
    def image2latent(self, image):
        if type(image) is torch.Tensor and image.dim() == 4:
            return image
        if type(image) is list:
            arr = np.concatenate([np.array(i).reshape(1, 512, 512, 3) for i in image])
            x = (torch.from_numpy(arr).float() / 127.5 - 1).permute(0, 3, 1, 2)
            x = x.to(self.model.device, dtype=self.model.vae.dtype)
        else:
            x = (torch.from_numpy(np.array(image)).float() / 127.5 - 1).permute(2, 0, 1).unsqueeze(0)
            x = x.to(self.model.device, dtype=self.model.dtype)
        return self.model.vae.encode(x)['latent_dist'].mean * 0.18215

    @torch.no_grad()
    def init_prompt(self, prompt, uncond_embeddings=None):
        tok = self.model.tokenizer
        if uncond_embeddings is None:
            ids = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
            uncond_embeddings = self.model.text_encoder(ids.to(self.model.device))[0]
        ids = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids
        text_embeddings = self.model.text_encoder(ids.to(self.model.device))[0]
        self.context = torch.cat([uncond_embeddings.expand(*text_embeddings.shape), text_embeddings])
        self.prompt = prompt

    # ------------------------------------------------------------------ DDIM loops (baselines)
    @torch.no_grad()
    def ddim_loop(self, latent, n_steps, is_forward=True, guidance_scale=1, dynamic_guidance=False, tau1=0.4, tau2=0.6,
                  w_embed_dim=0, uncond_embeddings=None, controller=None):
        all_latent = [latent]
        latent = latent.clone().detach()
        ts = self.model.scheduler.timesteps
        with _editing(self.model, is_forward or dynamic_guidance):
            for i in range(n_steps):
                if uncond_embeddings is not None:
                    self.init_prompt(self.prompt, uncond_embeddings[i])
                t = ts[len(ts) - i - 1] if is_forward else ts[i]
                noise_pred = self.get_noise_pred(model=self.model, latent=latent, t=t, context=None, guidance_scale=guidance_scale,
                                                 dynamic_guidance=dynamic_guidance, w_embed_dim=w_embed_dim, tau1=tau1, tau2=tau2)
                latent = self.next_step(noise_pred, t, latent) if is_forward else self.prev_step(noise_pred, t, latent)
                if controller is not None:
                    latent = controller.step_callback(latent)
                all_latent.append(latent)
        return all_latent

    @torch.no_grad()
    def ddim_inversion(self, image, n_steps=None, guidance_scale=1, dynamic_guidance=False, tau1=0.4, tau2=0.6, w_embed_dim=0):
        n_steps = self.n_steps if n_steps is None else n_steps
        latent = self.image2latent(image)
        image_rec = self.latent2image(latent)
        return image_rec, self.ddim_loop(latent, is_forward=True, guidance_scale=guidance_scale, n_steps=n_steps,
                                         dynamic_guidance=dynamic_guidance, tau1=tau1, tau2=tau2, w_embed_dim=w_embed_dim)

    # ------------------------------------------------------------------ consistency loops (THE hot path)
    def _boundary_step(self, noise_pred, t, s, latent, alpha_schedule, sigma_schedule):
        """predicted_origin for a whole batch at one (t, s) pair - fused HIP kernel on the GPU."""
        ptype = self.model.scheduler.config.prediction_type
        B = len(latent)
        if latent.is_cuda and ptype == "epsilon":
            from . import ops
            ti, si = int(t), int(s)

            def make():
                a_s, s_s = (1.0, 0.0) if si == 0 else (float(alpha_schedule[si]), float(sigma_schedule[si]))
                row = [float(alpha_schedule[ti]), float(sigma_schedule[ti]), a_s, s_s]
                return torch.tensor([row] * B, dtype=torch.float32).to(latent.device)
            coef = self._cached(("coef", ti, si, B, str(latent.device)), make)
            out_dtype = torch.promote_types(torch.promote_types(latent.dtype, noise_pred.dtype), torch.float32)
            return ops.x0_step(latent.contiguous(), noise_pred.contiguous(), coef, out_dtype=out_dtype)
        dev = latent.device
        return predicted_origin(noise_pred, torch.tensor([t] * B, device=dev), torch.tensor([s] * B, device=dev), latent,
                                ptype, alpha_schedule.to(dev), sigma_schedule.to(dev))

    def _schedules(self):
        ac = self.model.scheduler.alphas_cumprod
        return torch.sqrt(ac).cpu(), torch.sqrt(1 - ac).cpu()

    @torch.no_grad()
    def cons_generation(self, latent, guidance_scale=1, dynamic_guidance=False, tau1=0.4, tau2=0.6, w_embed_dim=0,
                        controller=None):
        """Reverse (noise -> data) consistency sampling: one UNet call + one boundary step per (t, s) pair."""
        all_latent = [latent]
        latent = latent.clone().detach()
        alpha_schedule, sigma_schedule = self._schedules()
        # dynamic guidance is the reference's editing schedule (utils/generation.py:74-82): such passes run at the accurate precision
        # level of the native UNet, like the inversion loop and every pass with a controller attached (unet.py: precision policy)
        with _editing(self.reverse_cons_model, dynamic_guidance):
            for t, s in zip(self.reverse_timesteps, self.reverse_boundary_timesteps):
                noise_pred = self.get_noise_pred(model=self.reverse_cons_model, latent=latent, t=t, context=None, tau1=tau1,
                                                 tau2=tau2, w_embed_dim=w_embed_dim, guidance_scale=guidance_scale,
                                                 dynamic_guidance=dynamic_guidance)
                latent = self._boundary_step(noise_pred, t, s, latent, alpha_schedule, sigma_schedule)
                if controller is not None:
                    latent = controller.step_callback(latent)
                all_latent.append(latent)
        return all_latent

    @torch.no_grad()
    def cons_inversion(self, image, guidance_scale=0.0, w_embed_dim=0, seed=0):
        """Forward (data -> noise) consistency inversion from a noised encoding at `start_timestep`."""
        alpha_schedule, sigma_schedule = self._schedules()
        latent = self.image2latent(image)
        noise = torch.randn(latent.shape, generator=torch.Generator().manual_seed(seed)).to(latent.device)
        latent = self.noise_scheduler.add_noise(latent, noise, torch.tensor([self.start_timestep]))
        image_rec = self.latent2image(latent)
        with _editing(self.forward_cons_model, True):    # forward steps amplify the per-evaluation error: accurate precision level
            for t, s in zip(self.forward_timesteps, self.forward_boundary_timesteps):
                noise_pred = self.get_noise_pred(model=self.forward_cons_model, latent=latent, t=t, context=None,
                                                 guidance_scale=guidance_scale, w_embed_dim=w_embed_dim, dynamic_guidance=False)
                latent = self._boundary_step(noise_pred, t, s, latent, alpha_schedule, sigma_schedule)
        return image_rec, [latent]


# ----------------------------------------------------------------------------------------------------------- misc utils
def latent2image(vae, latents):
    image = vae.decode(1 / 0.18215 * latents)['sample']
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.cpu().permute(0, 2, 3, 1).numpy()
    return (image * 255).astype(np.uint8)


def init_latent(latent, model, height, width, generator, batch_size):
    """ONE noise sample, expanded to the batch (utils/generation.py:536-543)."""
    if latent is None:
        latent = torch.randn((1, model.unet.in_channels, height // 8, width // 8), generator=generator)
    latents = latent.expand(batch_size, model.unet.in_channels, height // 8, width // 8).to(model.device)
    return latent, latents


def load_512(image_path, left=0, right=0, top=0, bottom=0):
    from PIL import Image
    image = np.array(Image.open(image_path).convert('RGB'))[:, :, :3]
    return np.array(Image.fromarray(image).resize((512, 512)))


def to_pil_images(images, num_rows=1, offset_ratio=0.02):
    from PIL import Image
    if type(images) is list:
        num_empty = len(images) % num_rows
    elif images.ndim == 4:
        num_empty = images.shape[0] % num_rows
    else:
        images, num_empty = [images], 0
    blank = np.ones(images[0].shape, dtype=np.uint8) * 255
    tiles = [im.astype(np.uint8) for im in images] + [blank] * num_empty
    h, w, _ = tiles[0].shape
    gap = int(h * offset_ratio)
    cols = len(tiles) // num_rows
    canvas = np.ones((h * num_rows + gap * (num_rows - 1), w * cols + gap * (cols - 1), 3), dtype=np.uint8) * 255
    for r in range(num_rows):
        for c in range(cols):
            canvas[r * (h + gap): r * (h + gap) + h, c * (w + gap): c * (w + gap) + w] = tiles[r * cols + c]
    return Image.fromarray(canvas)


def view_images(images, num_rows=1, offset_ratio=0.02):
    img = to_pil_images(images, num_rows, offset_ratio)
    try:
        from IPython.display import display
        display(img)
    except ImportError:
        pass
    return img
