#!/usr/bin/env python3
"""BASELINE config 3 timing: iCD-SD1.5 4-step forward inversion + 4-step reverse pass with a p2p controller, batch 8.

    python tools/edit_bench.py [--batch 8] [--controller store|replace|none] [--reps 5]

Prints one line per phase (inversion, reverse) plus the per-family kernel table of one reverse pass.  Full-size SD1.5,
synthetic weights; 4-D latents go straight through image2latent (no VAE in the timed region).
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--controller", default="store")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()

from invertible_cd_amd import _lib, generation, p2p, synthetic, unet
from invertible_cd_amd.pipelines import StableDiffusionPipeline
from invertible_cd_amd.schedulers import DDIMScheduler
from invertible_cd_amd.unet_config import SD15

dev = "cuda"
sd = synthetic.synthetic_state_dict(SD15, seed=0, device=dev, dtype=torch.float16)
model = StableDiffusionPipeline(unet.UNet2DConditionModel(SD15, sd), DDIMScheduler.sd15(), tokenizer=synthetic.SyntheticTokenizer(),
                                device=dev, dtype=torch.float16)
del sd
solver = generation.Generator(model, 50, DDIMScheduler.sd15(), forward_cons_model=model, reverse_cons_model=model,
                              reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
B = a.batch
g = torch.Generator().manual_seed(453645634)
lat = torch.randn(B, 4, 64, 64, generator=g).to(dev)
solver.context = torch.randn(2 * B, 77, 768, generator=g).to(dev, torch.float16)
solver.latent2image = lambda z, return_type="np": np.zeros((1,))


def make_controller():
    if a.controller == "none":
        return None
    if a.controller == "store":
        return p2p.AttentionStore()
    p2p.tokenizer = synthetic.SyntheticTokenizer()
    p2p.NUM_DDIM_STEPS = 4
    p2p.device = dev
    prompts = ["a cat sitting on a bench"] + ["a dog sitting on a bench"] * (B - 1)
    return p2p.make_controller(prompts, True, 0.5, 0.5)


def inversion():
    return solver.cons_inversion(lat, guidance_scale=0.0, w_embed_dim=512, seed=5)[1][0]


def reverse(start):
    ctrl = make_controller()
    p2p.register_attention_control(model, ctrl)
    out = solver.cons_generation(start, guidance_scale=19.0, w_embed_dim=512, dynamic_guidance=True, tau1=0.8, tau2=0.8,
                                 controller=ctrl)[-1]
    p2p.register_attention_control(model, None)
    return out


def timed(fn, *args):
    fn(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        r = fn(*args)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps, r


t_inv, start = timed(inversion)
t_rev, _ = timed(reverse, start)
print(f"cfg3 B={B} controller={a.controller}: inversion {t_inv * 1e3:.1f} ms, reverse {t_rev * 1e3:.1f} ms, "
      f"{B / (t_inv + t_rev):.2f} edited images/s ({8 * B} UNet sample-evaluations per pass pair)")
_lib.profile_enable(True)
reverse(start)
torch.cuda.synchronize()
for k, v in _lib.profile_read().items():
    if v["launches"]:
        print(f"   {k:12s} {v['launches']:6d} launches {v['ms']:9.3f} ms")
_lib.profile_enable(False)
