#!/usr/bin/env python3
"""AutoencoderKL decode / encode timing at full SD-VAE width (512x512 images <-> 64x64 latents), synthetic weights.

    python tools/vae_bench.py [--batch 16] [--chunk 8] [--reps 3] [--size 64]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import synthetic, vae

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--chunk", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--size", type=int, default=64)
a = ap.parse_args()
sd = synthetic.synthetic_vae_state_dict(vae.SD_VAE, seed=0, device="cuda", dtype=torch.float16)
m = vae.AutoencoderKL(vae.SD_VAE, sd, max_chunk=a.chunk)
del sd
z = torch.randn(a.batch, 4, a.size, a.size, device="cuda")


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps, r


px = a.size * 8
dec_flop = 2.4765e12 * (a.size / 64) ** 2          # per image, 2*MACs of every conv / linear / attention matmul (decoder)
td, img = timed(lambda: m.decode(z)["sample"])
te, _ = timed(lambda: m.encode(img.clamp(-1, 1))["latent_dist"].mean)
print(f"vae decode B={a.batch} {px}x{px}: {td * 1e3:.1f} ms = {td / a.batch * 1e3:.2f} ms/image, {a.batch / td:.1f} images/s, "
      f"~{dec_flop * a.batch / td / 1e12:.0f} TFLOP/s;  encode: {te * 1e3:.1f} ms = {te / a.batch * 1e3:.2f} ms/image")
