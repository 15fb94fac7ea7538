"""The slice of diffusers' DDIMScheduler the iCD path touches (SURVEY.md section 8b "Scheduler constants").

utils/loading.py:39-40 builds DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
clip_sample=False, set_alpha_to_one=False); the sampler reads `alphas_cumprod`, `final_alpha_cumprod`,
`config.{num_train_timesteps,prediction_type}`, `num_inference_steps`, `timesteps`, and calls `set_timesteps` /
`add_noise` (utils/generation.py:185-188,385-386,427,487).
"""
import types

import numpy as np
import torch


class DDIMScheduler:
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", clip_sample=True,
                 set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps, prediction_type=prediction_type,
                                            steps_offset=steps_offset, clip_sample=clip_sample, beta_start=beta_start,
                                            beta_end=beta_end, beta_schedule=beta_schedule, set_alpha_to_one=set_alpha_to_one)
        self.num_train_timesteps = num_train_timesteps
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def sd15(cls):
        return cls(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False)

    @classmethod
    def sdxl(cls):
        # stabilityai/stable-diffusion-xl-base-1.0 scheduler_config.json: same betas, steps_offset 1, leading spacing
        return cls(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                   set_alpha_to_one=False, steps_offset=1)

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        a = ac[timesteps] ** 0.5
        s = (1 - ac[timesteps]) ** 0.5
        a = a.flatten()
        s = s.flatten()
        while a.dim() < original_samples.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * original_samples + s * noise
