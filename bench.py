#!/usr/bin/env python3
"""bench.py - 4-step iCD images/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--arch sd15|sdxl] [--batch B]

A "step" = one pass of the hot path over one batch of synthetic latents: the Generator.cons_generation loop
(4 U-Net evaluations at the released timesteps + 4 boundary steps) on the configuration BASELINE.json's metric is quoted
on for one GPU: configs[1] "iCD-SD1.5 4-step reverse, batch=32, fp16, 1xMI355X".  Inputs (latents, context, weights) are
resident in HBM before the timed region.  N > 1: one process per GPU (torchrun), the batch is sharded data-parallel
(B per GPU, weak scaling), no collective inside the loop, ONE all-gather (RCCL) of the produced latents at the end of
the timed region (`--gather images`: per-rank VAE decode + uint8 images instead).  VAE decode / text encoding are
outside this path and not part of `value` by default.  `python bench.py --gpus N` without a
launcher re-executes itself under torch.distributed.run with N ranks (rendezvous on a free port of 127.0.0.1).

The default run (no --arch/--batch) carries every BASELINE configuration's per-GPU share on the same JSON line:
  value / config      configs[1]  SD1.5 4-step reverse, 32 images per GPU
  "edit"              configs[2]  SD1.5 4-step inversion + 4-step reverse with p2p.AttentionStore, 8 images per GPU
  "sdxl"              configs[3]  SDXL 4-step reverse, 8 images per GPU (64 over 8 GPUs)
  "sdxl_edit"         configs[4]  SDXL 3-step inversion + 3-step reverse, dynamic guidance tau 0.7, 16 images per GPU (128 over 8)
Extra objects: "roofline" for the dominant kernel family (HIP-event durations recorded by the executor on the launch stream
during timed pass A, algorithmic FLOPs), "event_overhead" (the same K steps timed again with no events at all: `value` comes
from that pass when the events cost more than 1 %), "rccl" (N > 1: world size, library version), per-rank min / max step time,
and "cpu_baseline" (the CPU oracle restating the reference's fp32 diffusers path, config[0]: B=1, CFG-doubled, timed on this
host's cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

_REAL_STDOUT = sys.stdout            # main() prints the JSON line here; __main__ points sys.stdout at stderr for the rest
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
    ap.add_argument("--gather", default="latents", choices=["latents", "images"],
                    help="payload of the ONE end-of-run all-gather: fp16 latents (default; VAE decode is outside the timed "
                         "path, SURVEY 8d) or uint8 images [N,512,512,3] after a per-rank VAE decode inside the timed region "
                         "(running/sd1.5/generate.py:372-383)")
    ap.add_argument("--no-sdxl", action="store_true", help="skip the additional SDXL measurements of the default run")
    ap.add_argument("--no-edit", action="store_true", help="skip the inversion + edit legs (BASELINE configs 3 and 5) of the default run")
    ap.add_argument("--xattn-fusion", type=int, default=2, choices=[0, 1, 2],
                    help="A/B switch (UNet option xattn_fusion): 2 = fused query-projection + cross-attention launch where it measured "
                         "faster (default), 0 = never, 1 = wherever eligible")
    ap.add_argument("--xattn-tile", type=int, default=0, choices=[0, 2, 4, 5, 6],
                    help="A/B: host tile of the fused cross-attention launch (0 planner, 5 = 256 x 256, 6 = 192 x 256 per sample)")
    ap.add_argument("--ln-inline-stats", type=int, default=1, choices=[0, 1],
                    help="A/B switch (UNet option ln_inline_stats): 1 = the GEMM behind a LayerNorm computes its statistics (default), "
                         "0 = a separate statistics pass over the residual stream")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes of one step per architecture after "
                         "the timed legs, N = 1 only); the committed PMC summary of the same workload is reported instead")
    ap.add_argument("--leg", default="all", choices=["all", "edit"],
                    help="internal (the PMC child passes of live_traffic): 'edit' runs ONLY the inversion + edit leg of --arch at --batch")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="independent batches in flight per GPU (host threads x HIP streams x executor replicas over one set of weights; "
                         "InFlight): 1 = the sequential loop of rounds 1-3.  Every UNet call still runs the configuration's batch")
    ap.add_argument("--in-flight-edit", type=int, default=3,
                    help="batches in flight of the SD1.5 edit leg (configs[2]: 8 images per batch, the reference's shipped batch_per_gpu)")
    ap.add_argument("--precision", default="auto", choices=["auto", "fast", "split", "accurate"],
                    help="precision policy of the native UNet (unet.py): auto (default, what a user of the library gets) = the error carry "
                         "for plain generation (one evaluation 0.70-0.85e-3 from fp32; its reverse loops contract that) and the accurate "
                         "level - GroupNorm reads the carry, shortcut / proj_out / sampler convs take hi + lo, 0.40-0.44e-3 - for "
                         "inversion loops, dynamic-guidance passes and passes with a controller, i.e. the edit legs")
    ap.add_argument("--residual", type=int, default=None, choices=[0, 1, 2, 3],
                    help="override the policy with a fixed UNet option residual: 0 = plain fp16 stream (rounds 1-3); 1 = fp32 twin; 2 = fp16 + bf8 "
                         "error carry; 3 = carry + split consumers")
    ap.add_argument("--no-fused-epilogue", action="store_true",
                    help="edit leg: accumulate the attention store with a launch of its own per hooked layer instead of in the probability "
                         "kernel's epilogue (A/B of icd_probs_epilogue)")
    ap.add_argument("--gemm-tune", type=lambda v: int(v, 0), default=0,
                    help="A/B switch (UNet option gemm_tune): ICD_GEMM_TUNE_* planner bits for every GEMM launch, e.g. 0x20000000 = never the "
                         "ping-pong tiles (gemm_pp*.hip): the lockstep tiles of rounds 1-5")
    ap.add_argument("--attn-valu-scale", type=int, default=0, choices=[0, 1],
                    help="A/B switch (UNet option attn_valu_scale): 1 = flash attention applies the softmax offset with an FMA per score "
                         "on the VALU, 0 (default) = the MFMA subtracts it")
    return ap.parse_args()


def free_port():
    """A TCP port nobody listens on right now (the rendezvous of a self-spawned run; a launcher-provided MASTER_PORT is kept)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one process per GPU (RCCL).
    Rendezvous on 127.0.0.1 at $MASTER_PORT if the caller set one, else on a port that is free right now (a fixed default port
    fails on a node with a stale listener); a rank that dies takes the group down (torchrun prints its traceback on stderr)."""
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {n_dev} GPU(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # only when unset: dmabuf IPC for RCCL on this driver stack
    env.setdefault("NCCL_DEBUG", "WARN")                       # RCCL says why when its bootstrap fails
    env.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    port = env.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f"bench.py: the {a.gpus}-rank run failed with exit code {rc} (rendezvous 127.0.0.1:{port}); the failing rank's traceback is above",
              file=sys.stderr)
    raise SystemExit(rc)


class SD15Workload:
    """Full-width SD1.5 + fused LoRA behind generation.Generator: BASELINE configs[1] (reverse, B = 32) and configs[2] (inversion +
    reverse with p2p.AttentionStore, B = 8) share one set of weights."""
    fused_epilogue = True            # --no-fused-epilogue clears it: store += P as launches of its own

    def __init__(self, device, net=None):
        from invertible_cd_amd import generation, synthetic, unet
        from invertible_cd_amd.pipelines import StableDiffusionPipeline
        from invertible_cd_amd.schedulers import DDIMScheduler
        from invertible_cd_amd.unet_config import SD15
        from invertible_cd_amd.loading import fuse_lora
        self.cfg, self.device = SD15, device
        if net is None:
            # synthetic weights generated directly on this rank's GPU (seeded per tensor), LoRA (rank 64, alpha 8) fused at load
            sd = synthetic.synthetic_state_dict(SD15, seed=0, device=device)
            sd = fuse_lora(sd, synthetic.synthetic_lora(SD15, seed=1, device=device), lora_dtype=torch.float16)
            net = unet.UNet2DConditionModel(SD15, sd, device=device, dtype=torch.float16)
            del sd
        self.model = StableDiffusionPipeline(net, DDIMScheduler.sd15(), tokenizer=synthetic.SyntheticTokenizer(), device=device,
                                             dtype=torch.float16)
        self.net = self.model.unet
        self.solver = generation.Generator(self.model, 50, DDIMScheduler.sd15(), forward_cons_model=self.model, reverse_cons_model=self.model,
                                           reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
        self.solver.latent2image = lambda z, return_type="np": None        # cons_inversion's image_rec decode is outside the path (8d)

    def replica(self):
        """The same weights behind a second executor handle / pipeline / Generator: one more batch in flight (UNet2DConditionModel.replica)."""
        return SD15Workload(self.device, net=self.net.replica())

    def inputs(self, batch):
        g = torch.Generator().manual_seed(453645634)                     # running/sd1.5/launch_generation_iCD_sd1.5.sh:32
        latents = torch.randn(batch, 4, 64, 64, generator=g).to(self.device)            # B independent samples (throughput run)
        ctx = torch.randn(2 * batch, 77, 768, generator=g).to(device=self.device, dtype=torch.float16)
        return latents, ctx

    def reverse_step(self, batch):
        latents, ctx = self.inputs(batch)

        def step():
            # a new batch = new prompts: the context K / V projections are computed once per batch (and shared by its 4 evaluations),
            # never carried over from the previous batch although this synthetic run hands in the same tensor
            self.net.reset_context_cache()
            self.solver.context = ctx
            return self.solver.cons_generation(latents, guidance_scale=7.0, w_embed_dim=512, dynamic_guidance=False, tau1=1.0, tau2=1.0)[-1]
        return step

    def edit_step(self, batch):
        """configs[2]: 4-step forward inversion (w = 0) + 4-step reverse (gs 19, tau 0.8) with an AttentionStore registered; a fresh
        controller per pass, as running/sd1.5/edit.py builds one per image group."""
        from invertible_cd_amd import p2p
        latents, ctx = self.inputs(batch)

        def step():
            self.net.reset_context_cache()                  # (see reverse_step)
            self.solver.context = ctx
            start = self.solver.cons_inversion(latents, guidance_scale=0.0, w_embed_dim=512, seed=5)[1][0]
            ctrl = p2p.AttentionStore()
            ctrl.fused_epilogue = self.fused_epilogue
            p2p.register_attention_control(self.model, ctrl)
            try:
                out = self.solver.cons_generation(start, guidance_scale=19.0, w_embed_dim=512, dynamic_guidance=True, tau1=0.8, tau2=0.8,
                                                  controller=ctrl)[-1]
            finally:
                p2p.register_attention_control(self.model, None)
            step.stored = sum(len(v) for v in ctrl.attention_store.values())
            return out
        step.stored = 0
        return step


class SDXLWorkload:
    """Full-width SDXL behind generation_sdxl: configs[3]'s per-GPU share (reverse, 8 images / GPU) and configs[4]'s (3-step inversion
    + 3-step reverse with dynamic guidance, 16 images / GPU) on one set of weights."""

    def __init__(self, device, net=None):
        from invertible_cd_amd import synthetic, unet
        from invertible_cd_amd.pipelines import StableDiffusionXLImg2ImgPipeline, StableDiffusionXLPipeline
        from invertible_cd_amd.schedulers import DDIMScheduler
        from invertible_cd_amd.unet_config import SDXL
        from invertible_cd_amd.loading import fuse_lora
        self.cfg, self.device = SDXL, device
        if net is None:
            # configs[3] / [4] are LoRA models like configs[1] / [2] (utils/loading.py:119-125): rank 64, alpha 8, fused at load
            sd = synthetic.synthetic_state_dict(SDXL, seed=0, device=device, dtype=torch.float16)
            sd = fuse_lora(sd, synthetic.synthetic_lora(SDXL, seed=1, device=device), lora_dtype=torch.float16)
            net = unet.UNet2DConditionModel(SDXL, sd, device=device, dtype=torch.float16)
            del sd
        self.net = net
        self.pipe = StableDiffusionXLPipeline(self.net, DDIMScheduler.sdxl(), device=device)
        self.fwd = StableDiffusionXLImg2ImgPipeline(self.net, DDIMScheduler.sdxl(), device=device)
        self.pipe.vae = None

    def replica(self):
        return SDXLWorkload(self.device, net=self.net.replica())

    def inputs(self, batch):
        from invertible_cd_amd import synthetic
        inp = synthetic.synthetic_inputs(self.cfg, batch, 128, 128, seed=0, device="cpu")
        emb = {"prompt_embeds": inp["context"].to(self.device, torch.float16), "text_embeds": inp["text_embeds"].to(self.device, torch.float16),
               "time_ids": inp["time_ids"].to(self.device)}
        return inp["latents"].to(self.device, torch.float16), emb

    def reverse_step(self, batch):
        from invertible_cd_amd import generation_sdxl
        latents, emb = self.inputs(batch)
        prompts = ["x"] * batch

        def step():
            self.net.reset_context_cache()                  # a new batch of prompts pays for its own context projections
            return generation_sdxl.sample_deterministic(self.pipe, prompts, latents=latents, num_inference_steps=4, guidance_scale=7.0,
                                                        is_sdxl=True, timesteps=[249, 499, 699, 999],
                                                        compute_embeddings_fn=lambda p, o, c: dict(emb), return_latent=True)[1]
        return step

    def edit_step(self, batch):
        """configs[4]: running/sdxl/edit.py:196-226 with the 3-step sets of README.md:59,62: inverse_sample_deterministic (w = 0) then
        sample_deterministic with dynamic guidance gs 19, tau 0.7 (running/sdxl/launch_editing_iCD_sdxl.sh:11-18)."""
        from invertible_cd_amd import generation_sdxl as G
        latents, emb = self.inputs(batch)
        src, dst = ["src"] * batch, ["dst"] * batch
        fn = lambda p, o, c: dict(emb)

        def step():
            self.net.reset_context_cache()
            inv = G.inverse_sample_deterministic(self.fwd, latents, src, num_inference_steps=3, timesteps=[19, 339, 699], guidance_scale=0.0,
                                                 is_sdxl=True, compute_embeddings_fn=fn, seed=3)
            return G.sample_deterministic(self.pipe, dst, latents=inv, num_inference_steps=3, guidance_scale=19.0, is_sdxl=True,
                                          timesteps=[339, 699, 999], compute_embeddings_fn=fn, use_dynamic_guidance=True, tau1=0.7,
                                          tau2=0.7, return_latent=True)[1]
        return step


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
            # latest round wins (sorted by name); the file records the sha of the kernel sources it was measured on
            best = (d["families"][family]["traffic_bytes"], os.path.join("profiles", os.path.basename(f)), d.get("kernels_sha"))
    return best or (None, None, None)


def live_traffic(arch, batch, family, timeout_s=240, leg="all"):
    """roofline.traffic measured IN THIS RUN: two rocprofv3 --pmc child passes (FETCH_SIZE, then WRITE_SIZE - one counter per pass, with
    --kernel-trace only, as MI355X_MICROARCH.md prescribes) over one step of the same leg in a fresh process, summarised per kernel family
    with tools/hbm_traffic.py's own code: bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Returns (bytes per launch, seconds
    spent) or (None, reason): the caller then reports the committed summary instead.  (The counters cannot be read inside the timed
    process; the child runs after the timed legs, on the same GPU, same binary.)"""
    import shutil
    import subprocess
    import tempfile
    t0 = time.perf_counter()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    src = open(os.path.join(ROOT, "tools", "hbm_traffic.py")).read().split("ap = argparse.ArgumentParser()")[0]
    ns = {}
    exec(compile(src, "hbm_traffic_head", "exec"), ns)          # family() and collect(): one source of truth with the offline tool
    tmp = tempfile.mkdtemp(prefix="icd_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    # the child is a single-process run of its own: nothing of the parent's launcher may reach it.  (Under torch.distributed.run the
    # variable TORCHELASTIC_USE_AGENT_STORE makes env:// rendezvous a CLIENT of the agent's store - a child that inherits it waits for
    # a server nobody runs until this function's timeout kills it: 970 s of timeouts on a default run, measured.)
    for k in list(env):
        if k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE",
                 "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING") or k.startswith("TORCHELASTIC_"):
            env.pop(k)
    child = [sys.executable, os.path.abspath(__file__), "--arch", arch, "--batch", str(batch), "--steps", "1", "--warmup", "1", "--in-flight", "1",
             "--no-cpu-baseline", "--no-vae", "--no-ref-batching", "--no-profile", "--no-sdxl", "--no-live-traffic"]
    child += ["--leg", "edit", "--in-flight-edit", "1"] if leg == "edit" else ["--no-edit"]
    try:
        per = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            # counters only on the family's kernels: every other dispatch of the child (building 2.6 G synthetic parameters is thousands of
            # small torch kernels) runs unserialised - the SDXL pass fits the default run that way
            only = ["--kernel-include-regex", "gemm"] if family.startswith("gemm") else []
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p"] + only + ["--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited {r.returncode}"
            agg = ns["collect"](d)
            if family not in agg or not agg[family][0]:
                return None, f"no {family} dispatches in the {counter} pass"
            per[counter] = agg[family][1] / agg[family][0]
        return int((2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024), time.perf_counter() - t0
    except subprocess.TimeoutExpired:
        return None, f"rocprofv3 pass exceeded {timeout_s} s"
    except Exception as e:                                       # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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


def csrc_sha():
    """Digest of the kernel sources the LOADED library was built from - stamped into the .so at build time (icd_build_sha), so it names
    the code that ran, not the files that happen to lie next to it (the binding refuses a library that differs from csrc/)."""
    from invertible_cd_amd import _lib
    return _lib.build_sha()


def to_uint8_images(img):
    """[-1,1] NCHW -> uint8 NHWC, the payload running/sd1.5/generate.py:372-378 gathers."""
    return ((img.float() / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def cuda_sync():
    if torch.cuda.is_available():               # (the gloo world-2 test drives time_leg on CPU tensors)
        torch.cuda.synchronize()


from invertible_cd_amd.inflight import InFlight          # noqa: E402  (N independent batches in flight; part of the product)


def time_leg(step, steps, warmup, batch, device, world, rank, decode=None, events_family=None):
    """W untimed passes, then EXACTLY `steps` timed passes bracketed by barrier + synchronize on both sides, the ONE all-gather of
    the produced samples inside the timed region; MAX over ranks.  `step`: a callable (one batch at a time) or an InFlight (several
    independent batches in flight, K passes in total).  events_family: bracket the launches of that kernel family with HIP events during
    the timed region (the roofline leg; one batch at a time only); None: no events at all.  Returns (seconds, per-rank seconds list,
    event profile or None, last output)."""
    import torch.distributed as dist
    from invertible_cd_amd import _lib, dist_utils
    flight = step if isinstance(step, InFlight) else InFlight([step], device)
    assert not (events_family and len(flight) > 1), "the event profiler serves one batch at a time"

    def sync_all():
        cuda_sync()
        if world > 1:
            dist.barrier()
            cuda_sync()

    if warmup:
        flight.run(warmup * len(flight))
    sync_all()
    if events_family:
        _lib.profile_enable(True, only=[events_family])
    t0 = time.perf_counter()
    outs = flight.run(steps)
    local = torch.stack(outs).reshape(-1, *outs[0].shape[1:]).to(torch.float16)
    if decode is not None:
        local = torch.cat([decode(local[i:i + batch]) for i in range(0, local.shape[0], batch)])
    ids = torch.arange(local.shape[0], device=device, dtype=torch.int64) * world + rank
    # ONE all-gather at the end (RCCL over xGMI); a world of one with a process group still runs the real collective on its GPU
    gathered, gids = dist_utils.gather_samples(local, ids, always_collective=True)
    sync_all()
    dt = time.perf_counter() - t0
    prof = None
    if events_family:
        prof = _lib.profile_read()
        _lib.profile_enable(False)
    per_rank = [dt]
    if world > 1:
        tt = torch.zeros(world, device=device, dtype=torch.float64)
        tt[rank] = dt
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        per_rank = [float(v) for v in tt.tolist()]
        dt = max(per_rank)
    assert gathered.shape[0] == batch * steps * world and bool(torch.equal(gids, torch.arange(batch * steps * world, device=device)))
    assert gathered.dtype == torch.uint8 or bool(torch.isfinite(gathered).all())
    return dt, per_rank, prof, outs[-1]


def family_table(step):
    """One pass with every executor launch bracketed by HIP events: the per-family table, and the dominant family = the one the step
    spends the most TIME in (round 6; rounds 1 - 5 took the one carrying the most algorithmic work, which named gemm_conv while
    gemm_dense was 3 ms larger on the SD1.5 line).  The per-family table on the line carries every family's time and rate either way."""
    from invertible_cd_amd import _lib
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    step()
    torch.cuda.synchronize()
    fam = {k: v for k, v in _lib.profile_read().items() if v["launches"]}
    _lib.profile_enable(False)
    return fam, max(fam, key=lambda k: (fam[k]["ms"], fam[k]["flops"]))


def workload_group(wl, n):
    """[wl, replica, ...]: n executors over wl's weights (built once per workload; options follow wl's at the time of the call)."""
    if not hasattr(wl, "_group"):
        wl._group = [wl]
    while len(wl._group) < n:
        wl._group.append(wl.replica())
    for w in wl._group[1:]:
        for name, value in wl.net._options.items():
            w.net.set_option(name, value)
        w.net.set_precision(wl.net.precision)
    return wl._group[:n]


def set_precision(a, net):
    """The precision policy (default auto) or, with --residual, a fixed residual mode."""
    if a.residual is not None:
        net.set_option("residual", a.residual)
    else:
        net.set_precision(a.precision)


RESID_NAMES = {0: "fp16", 1: "fp16 + fp32 twin", 2: "fp16 + bf8 error carry", 3: "fp16 + bf8 error carry, split consumers"}


def precision_of(a, net):
    """What the line says about the precision of the leg that has just run (read back from the handle the policy drove)."""
    applied = getattr(net, "_applied", None)
    mode = a.residual if a.residual is not None else (applied[0] if applied else None)
    d = {"policy": "fixed (--residual)" if a.residual is not None else a.precision, "residual_stream": RESID_NAMES.get(mode, str(mode))}
    if applied and applied[0] == 3:
        d["split_mask"] = applied[1]
    return d


def run_reverse(a, wl, arch, batch, steps, warmup, device, world, rank, primary):
    """BASELINE configs[1] (SD1.5, the `value` of the line) / configs[3]'s per-GPU share (SDXL): the 4-step reverse loop."""
    import torch.distributed as dist
    from invertible_cd_amd import dist_utils
    wl.net.set_option("ln_inline_stats", a.ln_inline_stats).set_option("xattn_fusion", a.xattn_fusion)     # per-handle A/B switches
    wl.net.set_option("attn_valu_scale", a.attn_valu_scale).set_option("xattn_tile", a.xattn_tile)
    if a.gemm_tune:
        wl.net.set_option("gemm_tune", a.gemm_tune)
    set_precision(a, wl.net)
    step = wl.reverse_step(batch)
    group = workload_group(wl, a.in_flight)              # this workload + its replicas (same weights, own handle / arena / stream)
    flight = InFlight([step] + [w.reverse_step(batch) for w in group[1:]], device)
    vae_m = None
    # the VAE exists only where something decodes: rank 0's separate vae_decode leg, or every rank under --gather images
    if (rank == 0 and not a.no_vae and primary) or a.gather == "images":
        from invertible_cd_amd import synthetic, vae as vae_mod
        vcfg = vae_mod.SD_VAE if arch == "sd15" else vae_mod.SDXL_VAE
        vsd = synthetic.synthetic_vae_state_dict(vcfg, seed=0, device=device, dtype=torch.float16)
        vae_m = vae_mod.AutoencoderKL(vcfg, vsd, device=device, max_chunk=8 if arch == "sd15" else 2)
        del vsd

    def decode_u8(lat):
        return to_uint8_images(vae_m.decode((lat.float() / vae_m.config.scaling_factor).clamp(-30, 30))["sample"])

    decode = decode_u8 if a.gather == "images" else None
    for _ in range(max(0, warmup - 1)):
        step()
    fam_table = dominant = None
    if not a.no_profile:
        fam_table, dominant = family_table(step)          # the last warm-up pass
    elif warmup:
        step()
    if decode is not None:
        decode_u8(step()[:2])
    # Pass A = K steps, one batch at a time, with HIP events around the dominant family (the roofline: per-launch durations of the kernel,
    # undisturbed by a second stream); pass B = K steps with no events at all and `--in-flight` batches in flight.  `value` is pass B -
    # unless one batch is in flight and the events cost less than 1 % (then pass A, and the roofline IS the timed region's).
    dt_ev = prof = None
    if not a.no_profile:
        dt_ev, _, prof, _ = time_leg(step, steps, 0, batch, device, world, rank, decode, events_family=dominant)
    dt_plain, per_rank, _, last = time_leg(flight, steps, 1 if len(flight) > 1 else 0, batch, device, world, rank, decode)
    ev_over = (dt_ev - dt_plain) / dt_plain if dt_ev else None
    use_plain = dt_ev is None or ev_over > 0.01 or len(flight) > 1
    dt = dt_plain if use_plain else dt_ev

    # N > 1: the reference's payload - uint8 images + int64 ids in ONE all-gather (running/sd1.5/generate.py:372-383) -
    # timed on its own after the timed region (the per-rank VAE decode before it is the separate vae_decode leg)
    image_gather = None
    if world > 1 and primary and a.gather != "images":
        # (the payload's content does not matter to the collective: uint8 noise of the image shape, so that no rank builds a VAE for it)
        side = 512 if arch == "sd15" else 1024
        u8 = torch.randint(0, 256, (batch, side, side, 3), device=device, dtype=torch.uint8)
        ids1 = torch.arange(batch, device=device, dtype=torch.int64) * world + rank
        dist_utils.gather_samples(u8, ids1)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        tg = time.perf_counter()
        g8, _ = dist_utils.gather_samples(u8, ids1)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        tg = time.perf_counter() - tg
        image_gather = {"payload": f"uint8 [{batch},{u8.shape[1]},{u8.shape[2]},3] per rank + int64 ids", "ms": round(tg * 1e3, 3),
                        "bytes_per_rank": int(u8.numel()), "gathered_images": int(g8.shape[0])}

    # the same loop with the reference's CFG-doubled batching (uncond rows computed and discarded), for the record
    ref_batching = None
    if arch == "sd15" and rank == 0 and primary and not a.no_ref_batching:
        wl.solver.eliminate_dead_uncond = False
        step(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ref_batching = batch * 2 / (time.perf_counter() - t1)
        wl.solver.eliminate_dead_uncond = True

    # The price of the parity mode: `value` is measured with the error-carried residual stream (UNet option residual = 2: eps within
    # 1e-3 rel-L2 of the fp32 oracle); the same loop on the plain fp16 stream of rounds 1 - 3 (1.0 - 1.2e-3) is timed beside it.
    fp16_stream = None
    prec = precision_of(a, wl.net)
    accurate_level = None
    if a.residual != 0 and not a.no_profile:
        for w in group:
            w.net.set_option("residual", 0)
        n_alt = max(2, steps // 2)
        dt_alt, _, _, _ = time_leg(flight, n_alt, 1, batch, device, world, rank, decode)
        if a.residual is None and a.precision == "auto":
            # ... and the same leg at the accurate level (what the edit legs run): the price of 0.40-0.44e-3 instead of 0.70-0.85e-3
            for w in group:
                w.net.set_precision("accurate")
            dt_acc, _, _, _ = time_leg(flight, n_alt, 1, batch, device, world, rank, decode)
            accurate_level = {"value": round(batch * n_alt * world / dt_acc, 3), "ms_per_step": round(dt_acc / n_alt * 1e3, 3), "steps": n_alt,
                              "note": "the same leg forced to the accurate precision level (residual = 3, every ICD_SPLIT_* bit) that the policy "
                                      "selects for inversion / dynamic-guidance / controller passes: eps rel-L2 vs the fp32 oracle 0.40-0.44e-3"}
        for w in group:
            set_precision(a, w.net)
        flight.run(len(flight))
        fp16_stream = {"value": round(batch * n_alt * world / dt_alt, 3), "ms_per_step": round(dt_alt / n_alt * 1e3, 3), "steps": n_alt,
                       "note": "the same leg with UNet option residual = 0 (plain fp16 residual stream, the mode rounds 1-3 reported): "
                               "eps rel-L2 vs the fp32 oracle 1.0-1.2e-3 instead of 0.7-0.85e-3"}

    # VAE decode of one step's latents, timed separately ("VAE decode ... reported separately"); it is
    # NOT part of `value` unless --gather images.  SDXL latents are 128x128 -> 1024x1024 images.
    vae_leg = None
    if rank == 0 and not a.no_vae and primary and vae_m is not None:
        lat = (last.float() / vae_m.config.scaling_factor).clamp(-30, 30)
        vae_m.decode(lat[:2]); torch.cuda.synchronize()
        tv = time.perf_counter()
        img = vae_m.decode(lat)["sample"]
        torch.cuda.synchronize()
        tv = time.perf_counter() - tv
        ok = bool(torch.isfinite(img).all())
        per_img_unet = dt / (batch * steps)
        vae_leg = {"decode_ms_per_image": round(tv / batch * 1e3, 3), "decode_images_per_sec": round(batch / tv, 2),
                   "finite": ok, "images_per_sec_unet_plus_decode_1gpu": round(1.0 / (per_img_unet + tv / batch), 2),
                   "note": "AutoencoderKL decode of one step's latents on the same HIP operators, synthetic weights; not in `value`"}
        del img
    del vae_m
    if rank != 0:
        return None
    images = batch * steps * world
    value = images / dt
    algo = ALGO_TFLOP_PER_SAMPLE_FWD[arch] * 4e12                       # algorithmic FLOP per image (cond rows only)
    payload = "uint8 images after a per-rank VAE decode" if a.gather == "images" else "fp16 latents"
    out = {
        "metric": "4-step iCD images/sec (SD1.5 512^2)" if arch == "sd15" else "4-step iCD images/sec (SDXL 1024^2)",
        "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": ("iCD-SD1.5 4-step reverse, batch=32/GPU, fp16, 64x64 latents (512x512), w-embedding gs=7, "
                                "timesteps [999,779,519,259]" if arch == "sd15" else
                                "iCD-SDXL 4-step reverse, batch=8/GPU, fp16, 128x128 latents (1024x1024), gs=7, timesteps [999,699,499,249]"),
                   "per_gpu_batch": batch, "global_batch": batch * world, "unet_evals_per_step": 4,
                   "dead_uncond_rows_eliminated": arch == "sd15", "lora_fused": True,
                   "precision": prec,
                   "in_flight_batches": len(flight),
                   "parallelism": f"dp{world}", "collective": f"one all-gather (RCCL) of the {payload} + int64 ids at the end of the timed region"},
        "per_unet_ms": round(dt / steps / 4 * 1e3, 3),             # throughput time of one evaluation (wall / evaluations)
        "ms_per_step_per_rank": {"min": round(min(per_rank) / steps * 1e3, 3), "max": round(max(per_rank) / steps * 1e3, 3),
                                 "ranks": len(per_rank)},
        "end_to_end_algorithmic_tflops_per_gpu": round(value / world * algo / 1e12, 1),
        "end_to_end_frac_of_mfma_peak": round(value / world * algo / PEAK_MFMA_F16, 4),
    }
    if dt_ev is not None and len(flight) == 1:
        out["event_overhead"] = {"ms_per_step_with_events": round(dt_ev / steps * 1e3, 3), "ms_per_step_without_events": round(dt_plain / steps * 1e3, 3),
                                 "frac": round(ev_over, 4), "value_from": "pass without events" if use_plain else "pass with events",
                                 "note": "pass A: K steps with HIP events around the dominant family (the roofline); pass B: K steps with none"}
    elif dt_ev is not None:
        out["value_one_batch_at_a_time"] = round(images / dt_ev, 3)       # top level too: the figure comparable with rounds 1 - 3 (advisor r4)
        out["one_batch_at_a_time"] = {"value": round(images / dt_ev, 3), "ms_per_step": round(dt_ev / steps * 1e3, 3),
                                      "note": f"pass A: the sequential loop (one batch in flight) with HIP events around the dominant family - the "
                                              f"roofline leg; `value` is pass B: the same K steps with {len(flight)} independent batches in flight "
                                              f"(InFlight: host threads x HIP streams x executor replicas over one set of weights), no events"}
    if fp16_stream is not None:
        out["fp16_stream"] = fp16_stream
    if accurate_level is not None:
        out["accurate_level"] = accurate_level
    if ref_batching is not None:
        out["value_reference_cfg_doubled_batching"] = round(ref_batching, 3)
    if image_gather is not None:
        out["image_gather"] = image_gather
    if prof is not None:
        fam = fam_table
        d = prof[dominant]                              # measured over timed pass A
        if d["flops"] > 0:
            ach = d["flops"] / (d["ms"] * 1e-3)
            roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach / 1e12, 1), "peak": round(PEAK_MFMA_F16 / 1e12, 1),
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F16, 4)}
            if d.get("flops_executed") and abs(d["flops_executed"] - d["flops"]) > 1e-6 * d["flops"]:
                # algorithmic = the operator as the reference computes it (Upsample2D's conv3x3 on the upsampled map: 9 taps); the phase
                # form issues 4/9 of that to the matrix cores (the split-operand launches of the accurate level issue more)
                roof["executed"] = round(d["flops_executed"] / (d["ms"] * 1e-3) / 1e12, 1)
                roof["executed_frac"] = round(d["flops_executed"] / (d["ms"] * 1e-3) / PEAK_MFMA_F16, 4)
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3)
            roof = {"bound": "hbm", "kernel": dominant, "achieved": round(ach / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM, 4)}
        traffic, tsrc, tsha = hbm_traffic(arch, batch, dominant)
        roof.update({"traffic": traffic, "traffic_source": tsrc, "traffic_kernels_sha": tsha, "kernels_sha": csrc_sha(),
                     "launches": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
                     "algorithmic_per_launch": (d["flops"] or d["bytes"]) / d["launches"],
                     "timed_pass": "A (K steps with events around this family)"})
        out["roofline"] = roof
        out["kernel_families"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                      "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                                      "tflops_executed": round(v.get("flops_executed", 0.0) / (v["ms"] * 1e-3) / 1e12, 1) if v.get("flops_executed") else None,
                                      "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                                  for k, v in fam.items()}
        out["kernel_families_note"] = "per-family table from the last warm-up step; roofline from timed pass A"
    if vae_leg is not None:
        out["vae_decode"] = vae_leg
    return out


def run_edit(a, wl, arch, batch, steps, warmup, device, world, rank):
    """BASELINE configs[2] (SD1.5: 4-step inversion + 4-step reverse with p2p.AttentionStore, 8 images / GPU) and configs[4]'s per-GPU
    share (SDXL: 3 + 3 steps, dynamic guidance tau 0.7, 16 images / GPU): edited images / s, no events in the timed region."""
    set_precision(a, wl.net)
    if a.no_fused_epilogue:
        type(wl).fused_epilogue = False
    step = wl.edit_step(batch)
    n_fl = a.in_flight_edit if arch == "sd15" else a.in_flight
    group = workload_group(wl, n_fl)
    steps_fns = [step] + [w.edit_step(batch) for w in group[1:]]
    flight = InFlight(steps_fns, device)
    steps = (steps + n_fl - 1) // n_fl * n_fl            # whole rounds of the batches in flight
    # roofline of this leg: its dominant kernel family, HIP events on the launch stream over two sequential passes (one batch at a time)
    roof = seq = None
    if not a.no_profile:
        step()
        fam, dominant = family_table(step)
        n_ev = 2
        dt_ev, _, prof, _ = time_leg(step, n_ev, 0, batch, device, world, rank, events_family=dominant)
        d = prof[dominant]
        if d["flops"] > 0 and d["ms"] > 0:
            ach = d["flops"] / (d["ms"] * 1e-3)
            roof = {"bound": "mfma", "kernel": dominant, "achieved": round(ach / 1e12, 1), "peak": round(PEAK_MFMA_F16 / 1e12, 1), "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_MFMA_F16, 4), "launches": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
                    "algorithmic_per_launch": d["flops"] / d["launches"], "traffic": None, "kernels_sha": csrc_sha(),
                    "timed_pass": f"{n_ev} sequential passes with events around this family (not the passes `value` times)"}
            if d.get("flops_executed") and abs(d["flops_executed"] - d["flops"]) > 1e-6 * d["flops"]:
                roof["executed"] = round(d["flops_executed"] / (d["ms"] * 1e-3) / 1e12, 1)
                roof["executed_frac"] = round(d["flops_executed"] / (d["ms"] * 1e-3) / PEAK_MFMA_F16, 4)
        seq = {"value": round(batch * n_ev * world / dt_ev, 3), "ms_per_step": round(dt_ev / n_ev * 1e3, 3),
               "note": "the sequential loop (one batch in flight), with the events of the roofline pass"}
    dt, per_rank, _, _ = time_leg(flight, steps, warmup, batch, device, world, rank)
    step.stored = max(f.stored for f in steps_fns) if arch == "sd15" else 0
    if rank != 0:
        return None
    evals = 8 if arch == "sd15" else 6
    algo = ALGO_TFLOP_PER_SAMPLE_FWD[arch] * evals * 1e12
    value = batch * steps * world / dt
    out = {"metric": "iCD edited images/sec (inversion + reverse)", "value": round(value, 3), "unit": "images/sec", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
           "ms_per_step_per_rank": {"min": round(min(per_rank) / steps * 1e3, 3), "max": round(max(per_rank) / steps * 1e3, 3), "ranks": len(per_rank)},
           "config": {"workload": ("iCD-SD1.5 4-step forward inversion (w=0) + 4-step reverse (gs=19, tau=0.8) with p2p.AttentionStore registered, "
                                   "batch=8/GPU, fp16, 64x64 latents" if arch == "sd15" else
                                   "iCD-SDXL 3-step forward inversion (w=0) + 3-step reverse (gs=19, dynamic guidance tau=0.7), batch=16/GPU, fp16, "
                                   "128x128 latents, timesteps fwd [19,339,699] rev [999,699,339]"),
                      "per_gpu_batch": batch, "global_batch": batch * world, "unet_evals_per_step": evals, "parallelism": f"dp{world}",
                      "in_flight_batches": len(flight),
                      "precision": precision_of(a, wl.net),
                      **({"attention_store_accumulate": "probability kernel epilogue" if getattr(wl, "fused_epilogue", False) else
                          "one accumulate launch per hooked layer"} if arch == "sd15" else {})},
           "end_to_end_algorithmic_tflops_per_gpu": round(value / world * algo / 1e12, 1),
           "end_to_end_frac_of_mfma_peak": round(value / world * algo / PEAK_MFMA_F16, 4)}
    if arch == "sd15":
        out["attention_store_tensors_per_pass"] = int(step.stored)
    if roof is not None:
        out["roofline"] = roof
        out["one_batch_at_a_time"] = seq
    return out


def main():
    a = parse()
    if os.environ.get("ICD_BENCH_WATCHDOG"):            # debugging aid: dump every thread's stack to stderr every N seconds (a hang names itself)
        import atexit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["ICD_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)
        t_start = time.time()
        atexit.register(lambda: print(f"[bench watchdog] interpreter exit after {time.time() - t_start:.1f} s", file=sys.stderr, flush=True))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a, sys.argv[1:])                # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    import torch.distributed as dist
    from invertible_cd_amd import dist_utils
    rccl = None
    try:
        # N = 1 too: a process group of one rank on backend "nccl" - the end-of-run all-gather of the timed region then goes through
        # RCCL on every N the driver runs (a failure to bring RCCL up at N = 1 is reported on the line, not fatal; at N > 1 it is fatal)
        dist_utils.init("nccl")                     # RCCL
        assert dist.get_world_size() == a.gpus and dist.get_backend() == "nccl"
        # first contact with the fabric before anything is timed: every rank contributes its id to one all-gather
        probe = [torch.zeros(1, device=device, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(probe, torch.tensor([rank], device=device, dtype=torch.int64))
        assert [int(p) for p in probe] == list(range(world))
        ver = torch.cuda.nccl.version()
        rccl = {"rccl_world_size": dist.get_world_size(), "rccl_version": ".".join(str(v) for v in ver) if isinstance(ver, tuple) else str(ver),
                "backend": dist.get_backend(), "devices": world, "all_gather_executed": True}
    except Exception as e:                          # noqa: BLE001
        if world > 1:
            raise
        rccl = {"rccl_world_size": 0, "error": f"{type(e).__name__}: {e}"[:300], "all_gather_executed": False}
        if dist.is_initialized():
            dist.destroy_process_group()
    batch = a.batch or (32 if a.arch == "sd15" else 8)
    default_run = a.arch == "sd15" and not a.batch
    wl = (SD15Workload if a.arch == "sd15" else SDXLWorkload)(device)
    if a.leg == "edit":                                 # child pass of live_traffic(leg="edit"): the edit leg alone, then out
        e = run_edit(a, wl, a.arch, a.batch or (8 if a.arch == "sd15" else 16), 1, 1, device, world, rank)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(e), file=_REAL_STDOUT, flush=True)
        return
    out = run_reverse(a, wl, a.arch, batch, a.steps, a.warmup, device, world, rank, primary=True)
    # The default run carries every BASELINE configuration's per-GPU share on the same JSON line, inside the driver's clock:
    # `value` stays configs[1]; "edit" = configs[2] (B = 8, inversion + reverse with AttentionStore), "sdxl" = configs[3]'s 8 images
    # per GPU, "sdxl_edit" = configs[4]'s 16 images per GPU (3 + 3 steps, dynamic guidance).
    edit = sdxl = sdxl_edit = None
    if default_run and not a.no_edit:
        edit = run_edit(a, wl, "sd15", 8, max(1, min(a.steps, 5)), 1, device, world, rank)
    cfg_for_cpu = wl.cfg
    if default_run and not a.no_sdxl:
        del wl
        torch.cuda.empty_cache()
        wx = SDXLWorkload(device)
        saved = a.no_vae
        a.no_vae = True
        sdxl = run_reverse(a, wx, "sdxl", 8, max(1, min(a.steps, 8)), max(1, min(a.warmup, 2)), device, world, rank, primary=False)
        a.no_vae = saved
        if not a.no_edit:
            sdxl_edit = run_edit(a, wx, "sdxl", 16, max(1, min(a.steps, 3)), 1, device, world, rank)
        del wx
        torch.cuda.empty_cache()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    if rccl is not None:
        out["rccl"] = rccl
    if edit is not None:
        out["edit"] = edit
    if sdxl is not None:
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "per_unet_ms", "ms_per_step_per_rank", "config",
                "end_to_end_algorithmic_tflops_per_gpu", "end_to_end_frac_of_mfma_peak", "event_overhead", "roofline", "kernel_families",
                "fp16_stream", "one_batch_at_a_time")
        out["sdxl"] = {k: sdxl[k] for k in keep if k in sdxl}
    if sdxl_edit is not None:
        out["sdxl_edit"] = sdxl_edit
    if world == 1 and not a.no_live_traffic and not a.no_profile and out.get("roofline"):
        # roofline.traffic of the PRIMARY line measured in this run (child rocprofv3 passes, after every timed leg, ~17 s for SD1.5); the
        # committed summary stays as the fallback, and as the source of the "sdxl" object's figure (building the 2.6 G-parameter workload
        # under the counter profiler takes minutes - too long for the default run)
        roof = out["roofline"]
        t_bytes, info = live_traffic(a.arch, batch, roof["kernel"], timeout_s=120 if a.arch == "sd15" else 600)
        if t_bytes:
            roof.update({"traffic": t_bytes, "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of one step, this run",
                         "traffic_kernels_sha": roof["kernels_sha"], "traffic_seconds": round(info, 1)})
        else:
            roof["traffic_live_error"] = info
        # ... and of the "sdxl" object (its own pair of child passes over one SDXL step; the committed summary stays the fallback)
        xr = out.get("sdxl", {}).get("roofline")
        if default_run and xr:
            t_bytes, info = live_traffic("sdxl", 8, xr["kernel"], timeout_s=300)
            if t_bytes:
                xr.update({"traffic": t_bytes, "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of one step, this run",
                           "traffic_kernels_sha": xr["kernels_sha"], "traffic_seconds": round(info, 1)})
            else:
                xr["traffic_live_error"] = info
        # ... and of the two edit legs (round 6): child passes that run the edit leg alone (--leg edit), counters on the family's kernels
        for key, arch_e, b_e, lim in (("edit", "sd15", 8, 150), ("sdxl_edit", "sdxl", 16, 400)):
            er = out.get(key, {}).get("roofline") if default_run else None
            if er:
                t_bytes, info = live_traffic(arch_e, b_e, er["kernel"], timeout_s=lim, leg="edit")
                if t_bytes:
                    er.update({"traffic": t_bytes, "traffic_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of one edit step, this run",
                               "traffic_kernels_sha": er["kernels_sha"], "traffic_seconds": round(info, 1)})
                else:
                    er["traffic_live_error"] = info
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.arch, None, cfg_for_cpu)
    print(json.dumps(out), file=_REAL_STDOUT, flush=True)


if __name__ == "__main__":
    # the contract is ONE JSON line on stdout: everything else the stack prints (the reference-shaped Generator announces its
    # endpoint tables like utils/generation.py does) goes to stderr
    # - at the file-descriptor level too: native libraries write to fd 1 directly (RCCL prints its version banner there when the
    # process group comes up), which would put extra lines in front of the JSON line the driver parses.
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    main()
