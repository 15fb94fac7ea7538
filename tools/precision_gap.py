#!/usr/bin/env python3
"""The 'auto' precision policy's probe (unet.UNet2DConditionModel._probe_plain_level) on the weights the tests and the benchmark run:
gap = rel-L2 between one fast-level and one accurate-level evaluation of the same inputs, and the verdict.

    python tools/precision_gap.py            # on the GPU box
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from invertible_cd_amd import synthetic, unet
from invertible_cd_amd.loading import fuse_lora
from invertible_cd_amd.unet_config import SD15, SDXL


def gap(cfg, sd, B, H, W, seed, t=519, xl=False):
    m = unet.UNet2DConditionModel(cfg, sd)
    inp = synthetic.synthetic_inputs(cfg, B, H, W, seed=seed, device="cuda")
    kw = dict(encoder_hidden_states=inp["context"].half(), timestep_cond=torch.randn(B, 512, device="cuda").half())
    if xl:
        kw["added_cond_kwargs"] = {"text_embeds": inp["text_embeds"].half(), "time_ids": inp["time_ids"]}
    m(inp["latents"].half(), t, **kw)
    return m._auto_gap, m._auto_plain


rows = []
c = SD15.scaled((64, 128, 256, 256), cross_dim=64)
rows.append(("sd15 reduced (64,128,256,256), plain synthetic", gap(c, synthetic.synthetic_state_dict(c, seed=11), 2, 32, 32, 11)))
sd = synthetic.synthetic_state_dict(c, seed=11)
rows.append(("sd15 reduced, LoRA-fused (rank 64, scale 0.5)", gap(c, fuse_lora(sd, synthetic.synthetic_lora(c, seed=1)), 2, 32, 32, 11)))
x = SDXL.scaled((64, 128, 256), cross_dim=128, heads=(2, 4, 8))
sd = synthetic.synthetic_state_dict(x, seed=21)
rows.append(("sdxl reduced (64,128,256), plain synthetic", gap(x, sd, 2, 32, 32, 21, xl=True)))
rows.append(("sdxl reduced, LoRA-fused", gap(x, fuse_lora(sd, synthetic.synthetic_lora(x, seed=1)), 2, 32, 32, 21, xl=True)))
x2 = SDXL.scaled((64, 128, 128), cross_dim=64, heads=(2, 4, 4))
sd = {k: v.half().float() for k, v in synthetic.synthetic_state_dict(x2, seed=21).items()}
rows.append(("sdxl loader-test net (64,128,128), LoRA rank 16 seed 31", gap(x2, fuse_lora(sd, synthetic.synthetic_lora(x2, seed=31, rank=16)), 2, 16, 16, 4, xl=True)))
off = torch.zeros(64); off[[3, 17, 40, 41, 63]] = torch.tensor([3000.0, -9000.0, 16000.0, -24000.0, 12000.0])
sd = synthetic.synthetic_state_dict(c, seed=41); sd["conv_in.bias"] = sd["conv_in.bias"] + off
rows.append(("sd15 reduced, conv_in channels offset by 3000 .. 24000", gap(c, sd, 2, 32, 32, 41)))
if "--no-full" not in sys.argv:
    sd = synthetic.synthetic_state_dict(SD15, seed=0, device="cuda", dtype=torch.float16)
    rows.append(("sd15 FULL width, bench weights (LoRA-fused), B=2 64x64", gap(SD15, fuse_lora(sd, synthetic.synthetic_lora(SD15, seed=1, device="cuda"), lora_dtype=torch.float16), 2, 64, 64, 0)))
    del sd; torch.cuda.empty_cache()
    sd = synthetic.synthetic_state_dict(SDXL, seed=0, device="cuda", dtype=torch.float16)
    rows.append(("sdxl FULL width, bench weights (LoRA-fused), B=2 128x128", gap(SDXL, fuse_lora(sd, synthetic.synthetic_lora(SDXL, seed=1, device="cuda"), lora_dtype=torch.float16), 2, 128, 128, 0, xl=True)))
print(f"# escalation threshold {unet.UNet2DConditionModel.AUTO_ESCALATE_GAP:.2e}")
for name, (g, v) in rows:
    print(f"{name:64s} gap {g:.3e} -> {v}")
