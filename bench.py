#!/usr/bin/env python3
"""bench.py - 4-step iCD images/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--arch sd15|sdxl] [--batch B]

A "step" = one pass of the hot path over one batch of synthetic latents: the Generator.cons_generation loop
(4 U-Net evaluations at the released timesteps + 4 boundary steps) on the configuration BASELINE.json's metric is quoted
on for one GPU: configs[1] "iCD-SD1.5 4-step reverse, batch=32, fp16, 1xMI355X".  Inputs (latents, context, weights) are
resident in HBM before the timed region.  N > 1: one process per GPU (torchrun), the batch is sharded data-parallel
(B per GPU, weak scaling), no collective inside the loop, ONE all-gather (RCCL) of the produced latents at the end of
the timed region.  VAE decode / text encoding are outside this path and not timed.

Extra objects on the JSON line: "roofline" for the dominant kernel family (HIP-event durations recorded by the executor
on the launch stream during the timed region, algorithmic FLOPs) and "cpu_baseline" (the CPU oracle restating the
reference's fp32 diffusers path, config[0]: B=1, CFG-doubled, timed on this host's cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_MFMA_F16 = 2516.6e12          # 256 CU x 2.4 GHz x 4096 flop/clk/CU, dense (MI355X_MICROARCH.md: ~2.5 PF)
PEAK_HBM = 8.0e12
ALGO_TFLOP_PER_SAMPLE_FWD = {"sd15": 0.8033, "sdxl": 6.7612}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--arch", default="sd15", choices=["sd15", "sdxl"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 32 for sd15, 8 for sdxl)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the separate VAE-decode leg (SURVEY section 8f rank 1)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-ref-batching", action="store_true",
                    help="skip the extra CFG-doubled measurement (use under rocprofv3 so kernel stats match the timed region)")
    return ap.parse_args()


def build_sd15(batch, device):
    from invertible_cd_amd import generation, synthetic, unet
    from invertible_cd_amd.pipelines import StableDiffusionPipeline
    from invertible_cd_amd.schedulers import DDIMScheduler
    from invertible_cd_amd.unet_config import SD15
    from invertible_cd_amd.loading import fuse_lora
    # synthetic weights generated directly on this rank's GPU (seeded per tensor), LoRA (rank 64, alpha 8) fused at load
    sd = synthetic.synthetic_state_dict(SD15, seed=0, device=device)
    sd = fuse_lora(sd, synthetic.synthetic_lora(SD15, seed=1, device=device), lora_dtype=torch.float16)
    model = StableDiffusionPipeline(unet.UNet2DConditionModel(SD15, sd, device=device, dtype=torch.float16), DDIMScheduler.sd15(),
                                    tokenizer=synthetic.SyntheticTokenizer(), device=device, dtype=torch.float16)
    solver = generation.Generator(model, 50, DDIMScheduler.sd15(), forward_cons_model=model, reverse_cons_model=model,
                                  reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    g = torch.Generator().manual_seed(453645634)                          # running/sd1.5/launch_generation_iCD_sd1.5.sh:32
    latents = torch.randn(batch, 4, 64, 64, generator=g).to(device)      # B independent samples (throughput run)
    ctx = torch.randn(2 * batch, 77, 768, generator=g).to(device=device, dtype=torch.float16)
    solver.context = ctx

    def step():
        return solver.cons_generation(latents, guidance_scale=7.0, w_embed_dim=512, dynamic_guidance=False, tau1=1.0, tau2=1.0)[-1]
    del sd
    return step, solver, None, SD15


def build_sdxl(batch, device):
    from invertible_cd_amd import generation_sdxl, synthetic, unet
    from invertible_cd_amd.pipelines import StableDiffusionXLPipeline
    from invertible_cd_amd.schedulers import DDIMScheduler
    from invertible_cd_amd.unet_config import SDXL
    sd = synthetic.synthetic_state_dict(SDXL, seed=0, device=device, dtype=torch.float16)
    pipe = StableDiffusionXLPipeline(unet.UNet2DConditionModel(SDXL, sd, device=device, dtype=torch.float16), DDIMScheduler.sdxl(),
                                     device=device)
    inp = synthetic.synthetic_inputs(SDXL, batch, 128, 128, seed=0, device="cpu")
    emb = {"prompt_embeds": inp["context"].to(device, torch.float16), "text_embeds": inp["text_embeds"].to(device, torch.float16),
           "time_ids": inp["time_ids"].to(device)}
    latents = inp["latents"].to(device, torch.float16)
    prompts = ["x"] * batch

    def step():
        return generation_sdxl.sample_deterministic(pipe, prompts, latents=latents, num_inference_steps=4, guidance_scale=7.0,
                                                    is_sdxl=True, timesteps=[249, 499, 699, 999],
                                                    compute_embeddings_fn=lambda p, o, c: dict(emb), return_latent=True)[1]
    pipe.vae = None
    del sd
    return step, None, None, SDXL


def hbm_traffic(arch, batch, family):
    """HBM-side bytes per launch of `family` from the committed PMC summary of this workload (rocprofv3 cannot run inside
    the timed process); None when no summary matches the workload.  See tools/hbm_traffic.py for the corrections."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*hbm_traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("arch") == arch and d.get("per_gpu_batch") == batch and family in d.get("families", {}):
            best = d["families"][family]["traffic_bytes"]           # latest round wins (sorted by name)
    return best


def cpu_baseline(arch, sd, cfg):
    """The reference's CPU path restated (oracle): config[0] = SD1.5, B=1, 4 steps, CFG-doubled UNet batch of 2, fp32."""
    import numpy as np
    from invertible_cd_amd import synthetic
    from oracle import sched_ref, unet_ref
    ocfg = unet_ref.SD15 if arch == "sd15" else unet_ref.SDXL
    if sd is None:                                   # fp32 CPU weights of the same architecture (seeded synthetic)
        sd = synthetic.synthetic_state_dict(cfg, seed=0)
    torch.manual_seed(0)
    ac = sched_ref.alphas_cumprod()
    alpha, sigma = np.sqrt(ac), np.sqrt(1 - ac)
    threads = torch.get_num_threads()
    if arch == "sd15":
        x = torch.randn(1, 4, 64, 64)
        ctx = torch.randn(2, 77, 768)
        wemb = torch.from_numpy(sched_ref.guidance_scale_embedding([7.0, 7.0], 512))
        pairs = list(zip([999, 779, 519, 259], [779, 519, 259, 0]))
        t0 = time.perf_counter()
        for t, s in pairs:
            eps = unet_ref.unet_forward(sd, ocfg, torch.cat([x, x]), t, ctx, timestep_cond=wemb)[1:]
            x = torch.from_numpy(sched_ref.predicted_origin(eps.numpy(), [t], [s], x.numpy(), alpha, sigma))
        dt = time.perf_counter() - t0
        sample = "1 image: SD1.5 B=1, 4 steps, CFG-doubled UNet batch 2 (6.43 TFLOP as the reference executes), fp32 torch CPU"
        return {"value": 1.0 / dt, "unit": "images/sec", "cores": threads, "kind": "port", "sample": sample,
                "seconds": dt, "per_unet_ms": dt / 4 * 1e3}
    x = torch.randn(1, 4, 128, 128)
    ctx = torch.randn(1, 77, 2048)
    added = {"text_embeds": torch.randn(1, 1280), "time_ids": torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]])}
    wemb = torch.from_numpy(sched_ref.guidance_scale_embedding([7.0], 512))
    t0 = time.perf_counter()
    eps = unet_ref.unet_forward(sd, ocfg, x, 999, ctx, timestep_cond=wemb, added_cond=added)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (4 * dt), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "1 of the 4 UNet evaluations of one SDXL image (B=1, 6.76 TFLOP), x4 extrapolated", "seconds": dt,
            "per_unet_ms": dt * 1e3}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    import torch.distributed as dist
    from invertible_cd_amd import _lib, dist_utils
    if world > 1:
        dist_utils.init()
    batch = a.batch or (32 if a.arch == "sd15" else 8)
    step, solver, sd, cfg = (build_sd15 if a.arch == "sd15" else build_sdxl)(batch, device)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fam_table = None
    for i in range(a.warmup):
        if i == a.warmup - 1 and not a.no_profile:      # last warm-up pass: per-family table (every launch recorded)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            step()
            torch.cuda.synchronize()
            fam_table = {k: v for k, v in _lib.profile_read().items() if v["launches"]}
            _lib.profile_enable(False)
        else:
            step()
    sync_all()
    # dominant family = the one carrying the most algorithmic work (stable from run to run, unlike a max over times
    # when two families are within a few per cent of each other)
    dominant = max(fam_table, key=lambda k: (fam_table[k]["flops"], fam_table[k]["ms"])) if fam_table else None
    if not a.no_profile:
        # timed region: HIP events only around the dominant kernel family (keeps event overhead out of `value`)
        _lib.profile_enable(True, only=[dominant] if dominant else None)
    outs = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        outs.append(step())
    local = torch.stack(outs).reshape(-1, *outs[0].shape[1:]).to(torch.float16)
    ids = torch.arange(local.shape[0], device=device, dtype=torch.int64) * world + rank
    gathered, gids = dist_utils.gather_samples(local, ids)                # ONE all-gather at the end (RCCL over xGMI)
    sync_all()
    dt = time.perf_counter() - t0
    prof = None
    if not a.no_profile:
        prof = _lib.profile_read()
        _lib.profile_enable(False)
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert gathered.shape[0] == batch * a.steps * world and bool(torch.isfinite(gathered).all())

    # the same loop with the reference's CFG-doubled batching (uncond rows computed and discarded), for the record
    ref_batching = None
    if solver is not None and rank == 0 and not a.no_ref_batching:
        solver.eliminate_dead_uncond = False
        step(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ref_batching = batch * 2 / (time.perf_counter() - t1)
        solver.eliminate_dead_uncond = True

    # VAE decode of one step's latents, timed separately ("VAE decode ... reported separately"); it is
    # NOT part of `value`.  SD VAE width for both architectures; SDXL latents are 128x128 -> 1024x1024 images.
    vae_leg = None
    if rank == 0 and not a.no_vae:
        from invertible_cd_amd import synthetic, vae as vae_mod
        vcfg = vae_mod.SD_VAE if a.arch == "sd15" else vae_mod.SDXL_VAE
        vsd = synthetic.synthetic_vae_state_dict(vcfg, seed=0, device=device, dtype=torch.float16)
        m = vae_mod.AutoencoderKL(vcfg, vsd, device=device, max_chunk=8 if a.arch == "sd15" else 2)
        del vsd
        lat = (outs[-1].float() / vcfg.scaling_factor).clamp(-30, 30)
        m.decode(lat[:2]); torch.cuda.synchronize()
        tv = time.perf_counter()
        img = m.decode(lat)["sample"]
        torch.cuda.synchronize()
        tv = time.perf_counter() - tv
        ok = bool(torch.isfinite(img).all())
        per_img_unet = dt / (batch * a.steps)
        vae_leg = {"decode_ms_per_image": round(tv / batch * 1e3, 3), "decode_images_per_sec": round(batch / tv, 2),
                   "finite": ok, "images_per_sec_unet_plus_decode_1gpu": round(1.0 / (per_img_unet + tv / batch), 2),
                   "note": "AutoencoderKL decode of one step's latents on the same HIP operators, synthetic weights; not in `value`"}
        del m, img

    if rank != 0:
        return
    images = batch * a.steps * world
    value = images / dt
    algo = ALGO_TFLOP_PER_SAMPLE_FWD[a.arch] * 4e12                       # algorithmic FLOP per image (cond rows only)
    out = {
        "metric": "4-step iCD images/sec (SD1.5 512^2)" if a.arch == "sd15" else "4-step iCD images/sec (SDXL 1024^2)",
        "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": ("iCD-SD1.5 4-step reverse, batch=32/GPU, fp16, 64x64 latents (512x512), w-embedding gs=7, "
                                "timesteps [999,779,519,259]" if a.arch == "sd15" else
                                "iCD-SDXL 4-step reverse, batch=8/GPU, fp16, 128x128 latents (1024x1024), gs=7, timesteps [999,699,499,249]"),
                   "per_gpu_batch": batch, "global_batch": batch * world, "unet_evals_per_step": 4,
                   "dead_uncond_rows_eliminated": a.arch == "sd15", "lora_fused": a.arch == "sd15",
                   "parallelism": f"dp{world}", "collective": "one all-gather of the fp16 latents at the end of the timed region"},
        "per_unet_ms": round(dt / a.steps / 4 * 1e3, 3),
        "end_to_end_algorithmic_tflops_per_gpu": round(value / world * algo / 1e12, 1),
        "end_to_end_frac_of_mfma_peak": round(value / world * algo / PEAK_MFMA_F16, 4),
    }
    if ref_batching is not None:
        out["value_reference_cfg_doubled_batching"] = round(ref_batching, 3)
    if prof is not None:
        fam = fam_table or {k: v for k, v in prof.items() if v["launches"]}
        dom = dominant or max(fam, key=lambda k: fam[k]["ms"])
        d = prof[dom]                                   # measured over the timed region
        if d["flops"] > 0:
            ach = d["flops"] / (d["ms"] * 1e-3)
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach / 1e12, 1), "peak": round(PEAK_MFMA_F16 / 1e12, 1),
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F16, 4)}
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3)
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM, 4)}
        roof.update({"traffic": hbm_traffic(a.arch, batch, dom), "launches": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
                     "algorithmic_per_launch": (d["flops"] or d["bytes"]) / d["launches"]})
        out["roofline"] = roof
        out["kernel_families"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                      "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                                      "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                                  for k, v in fam.items()}
        out["kernel_families_note"] = "per-family table from the last warm-up step; roofline from the timed region"
    if vae_leg is not None:
        out["vae_decode"] = vae_leg
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.arch, sd, cfg)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
