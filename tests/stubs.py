"""Closed-form stand-ins used by the CPU parity tests - the SAME stubs the golden generator drove the reference with
(tests/golden/make_golden.py), so identical inputs reach the reference (then) and the product (now)."""
import types

import numpy as np
import torch


def alphas_cumprod():
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class StubTokenizer:
    model_max_length = 77

    def __init__(self):
        self.vocab, self.words = {}, {}

    def _id(self, w):
        if w not in self.vocab:
            i = len(self.vocab) + 10
            self.vocab[w], self.words[i] = i, w
        return self.vocab[w]

    def encode(self, text):
        return [1] + [self._id(w) for w in text.split(" ") if w != ""] + [2]

    def decode(self, ids):
        return " ".join(self.words.get(i, "") for i in ids)


class StubScheduler:
    def __init__(self, golden_dir=None):
        self.alphas_cumprod = alphas_cumprod()
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.config = types.SimpleNamespace(prediction_type="epsilon", num_train_timesteps=1000)
        self.num_train_timesteps = 1000
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy((np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype(np.int64))

    def add_noise(self, x, noise, t):
        a = self.alphas_cumprod[t] ** 0.5
        s = (1 - self.alphas_cumprod[t]) ** 0.5
        while a.dim() < x.dim():
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return a * x + s * noise


class StubUNet:
    """eps = 0.1 x + 0.01 t/1000 + 0.001 mean(timestep_cond); records its calls."""
    dtype = torch.float32
    in_channels = 4

    def __init__(self):
        self.calls = []

    def named_children(self):
        return []

    def __call__(self, x, t, timestep_cond=None, encoder_hidden_states=None, **kw):
        tt = float(t) if not torch.is_tensor(t) else float(t.item())
        self.calls.append(dict(x=x.clone(), t=tt, cond=None if timestep_cond is None else timestep_cond.clone()))
        eps = 0.1 * x + 0.01 * tt / 1000.0
        if timestep_cond is not None:
            eps = eps + 0.001 * timestep_cond.float().mean(dim=1).reshape(-1, 1, 1, 1)
        if kw.get("return_dict", True) is False:
            return (eps,)
        return {"sample": eps}


class HalfUNet(StubUNet):
    def __call__(self, x, t, timestep_cond=None, encoder_hidden_states=None, **kw):
        out = super().__call__(x, t, timestep_cond, encoder_hidden_states, **kw)["sample"]
        out[: len(out) // 2] *= 0.5
        return {"sample": out}


class StubModel:
    device = torch.device("cpu")
    dtype = torch.float32

    def __init__(self):
        self.scheduler = StubScheduler()
        self.unet = StubUNet()
        self.vae = None
        self.tokenizer = StubTokenizer()


class StubPipe:
    def __init__(self):
        self.unet = StubUNet()
        self.unet.config = types.SimpleNamespace(sample_size=2, in_channels=4)
        self.vae_scale_factor = 8
        self._execution_device = torch.device("cpu")
        self.scheduler = StubScheduler()
        self.vae = types.SimpleNamespace(to=lambda *a, **k: None, config=types.SimpleNamespace(scaling_factor=0.13025),
                                         decode=lambda z, return_dict=False: (z,))
        self.image_processor = types.SimpleNamespace(postprocess=lambda im, output_type, do_denormalize: im)

    def prepare_latents(self, *args, **kw):
        if torch.is_tensor(args[0]):
            image, t, bs, _, dtype, device = args[:6]
            noise = torch.randn(image.shape, generator=kw.get("generator"), dtype=dtype)
            return self.scheduler.add_noise(image, noise, t.reshape(1))
        bs, c, h, w, dtype, device, gen = args[:7]
        return torch.randn((bs, c, h // 8, w // 8), generator=gen, dtype=dtype)


def sdxl_emb_fn(prompts, orig, crop):
    g = torch.Generator().manual_seed(len(prompts) * 100 + len(prompts[0]))
    n = len(prompts)
    return {"prompt_embeds": torch.randn(n, 77, 16, generator=g), "text_embeds": torch.randn(n, 8, generator=g),
            "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * n, dtype=torch.float32)}
