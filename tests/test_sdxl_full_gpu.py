"""BASELINE configs 4 and 5 at their per-GPU size: full-width SDXL UNet (2.57 G parameters), 128x128 latents.

  * config 4: 4-step reverse, 8 images per GPU (utils/generation_sdxl.py:324-473, running/sdxl/generate.py:160-203)
  * config 5: 3-step forward inversion + 3-step reverse with dynamic guidance tau = 0.7, 16 images per GPU
    (running/sdxl/edit.py:196-226, running/sdxl/launch_editing_iCD_sdxl.sh:11-18)
The CPU oracle cannot run these sizes in seconds, so they are checked through size-independent properties (bit
reproducibility, batch independence = exact data-parallel sharding, the dynamic-guidance w sequence, finiteness); one
full-width forward at 64x64 is compared with the oracle, and the image -> latent entry of the inversion is exercised
with a full-width SDXL VAE at 1024x1024 from a tensor and from a PIL image.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def sdxl():
    from invertible_cd_amd import synthetic, unet
    from invertible_cd_amd.pipelines import StableDiffusionXLImg2ImgPipeline, StableDiffusionXLPipeline
    from invertible_cd_amd.schedulers import DDIMScheduler
    from invertible_cd_amd.unet_config import SDXL
    sd = synthetic.synthetic_state_dict(SDXL, seed=0, device="cuda", dtype=torch.float16)
    u = unet.UNet2DConditionModel(SDXL, sd, device="cuda", dtype=torch.float16)
    del sd
    torch.cuda.empty_cache()
    pipe = StableDiffusionXLPipeline(u, DDIMScheduler.sdxl(), device="cuda")
    fwd = StableDiffusionXLImg2ImgPipeline(u, DDIMScheduler.sdxl(), device="cuda")      # same weights: properties only
    return SDXL, pipe, fwd


def _embeds(cfg, B, seed):
    from invertible_cd_amd import synthetic
    inp = synthetic.synthetic_inputs(cfg, B, 128, 128, seed=seed, device="cpu")
    emb = {"prompt_embeds": inp["context"].cuda().half(), "text_embeds": inp["text_embeds"].cuda().half(),
           "time_ids": inp["time_ids"].cuda()}
    return inp["latents"].cuda().half(), emb


def _fn(emb, rows=None):
    def f(prompts, sizes, crops):
        e = {k: (v if rows is None else v[rows]) for k, v in emb.items()}
        assert e["prompt_embeds"].shape[0] == len(prompts)
        return dict(e)
    return f


def test_config4_sdxl_b8_4step_reverse_properties(sdxl):
    from invertible_cd_amd import generation_sdxl as G
    cfg, pipe, _ = sdxl
    B = 8
    lat, emb = _embeds(cfg, B, seed=11)
    run = lambda l, f, n: G.sample_deterministic(pipe, ["x"] * n, latents=l, num_inference_steps=4, guidance_scale=7.0, is_sdxl=True,
                                                 timesteps=[249, 499, 699, 999], compute_embeddings_fn=f, return_latent=True)[1]
    a = run(lat, _fn(emb), B)
    b = run(lat, _fn(emb), B)
    assert a.shape == (B, 4, 128, 128) and a.dtype == torch.float16 and torch.isfinite(a).all()
    assert torch.equal(a, b)                                                   # bit reproducible
    assert float(a.float().std()) > 0.05
    sub = run(lat[2:4].contiguous(), _fn(emb, slice(2, 4)), 2)                 # rows 2..3 alone == rows 2..3 of the batch
    e = rel_l2(sub, a[2:4])
    print(f"[sdxl cfg4] batch independence rel-L2 = {e:.3e}")
    assert e < 2e-3                                                            # different tile plans: fp rounding only


def test_config5_sdxl_b16_3plus3_dynamic_guidance(sdxl, monkeypatch):
    from invertible_cd_amd import generation_sdxl as G
    cfg, pipe, fwd = sdxl
    B = 16
    lat, emb = _embeds(cfg, B, seed=12)
    seen = []
    orig = G._w_embedding
    monkeypatch.setattr(G, "_w_embedding", lambda vals, dev, dt: (seen.append(tuple(float(v) for v in vals)), orig(vals, dev, dt))[1])

    def edit(l, f, n):
        inv = G.inverse_sample_deterministic(fwd, l, ["src"] * n, num_inference_steps=3, timesteps=[19, 339, 699],
                                             guidance_scale=0.0, is_sdxl=True, compute_embeddings_fn=f, seed=3)
        out = G.sample_deterministic(pipe, ["dst"] * n, latents=inv, num_inference_steps=3, guidance_scale=19.0, is_sdxl=True,
                                     timesteps=[339, 699, 999], compute_embeddings_fn=f, use_dynamic_guidance=True, tau1=0.7,
                                     tau2=0.7, return_latent=True)[1]
        return inv, out

    inv, out = edit(lat, _fn(emb), B)
    assert inv.shape == out.shape == (B, 4, 128, 128) and torch.isfinite(inv).all() and torch.isfinite(out).all()
    # w sequence: inversion w = 0 for the batch; reverse: the static embedding (gs) is built first, then per step
    # (t = 999, 699, 339 with tau 0.7): 0, 19, 19 (step rule of a5)
    assert seen[0] == (0.0,) * B and seen[1] == (19.0,) * B
    assert [s[0] for s in seen[2:]] == [0.0, 19.0, 19.0] and all(len(s) == B for s in seen)
    inv2, out2 = edit(lat, _fn(emb), B)
    assert torch.equal(inv, inv2) and torch.equal(out, out2)
    # batch independence: the CPU noise of prepare_latents is drawn per call for the whole batch, so compare the reverse
    # half on identical inverted latents
    sub = G.sample_deterministic(pipe, ["dst"] * 2, latents=inv[5:7].contiguous(), num_inference_steps=3, guidance_scale=19.0,
                                 is_sdxl=True, timesteps=[339, 699, 999], compute_embeddings_fn=_fn(emb, slice(5, 7)),
                                 use_dynamic_guidance=True, tau1=0.7, tau2=0.7, return_latent=True)[1]
    e = rel_l2(sub, out[5:7])
    print(f"[sdxl cfg5] batch independence rel-L2 = {e:.3e}")
    assert e < 4e-3                                  # measured 2.5e-3: three steps amplify the plan-dependent rounding (loop error vs oracle 1.9e-3)


def test_sdxl_inversion_from_image_tensor_and_pil(sdxl):
    """running/sdxl/edit.py:196-207: PIL image -> image_processor.preprocess -> inverse_sample_deterministic ->
    img2img prepare_latents (fp32 VAE encode, latent_dist.sample(generator) * 0.13025, add_noise at t = 19)."""
    from PIL import Image
    from invertible_cd_amd import generation_sdxl as G
    from invertible_cd_amd import synthetic, vae
    cfg, _, fwd = sdxl
    v = vae.AutoencoderKL(vae.SDXL_VAE, synthetic.synthetic_vae_state_dict(vae.SDXL_VAE, seed=0, device="cuda", dtype=torch.float16),
                          max_chunk=1)
    fwd.vae = v
    try:
        rng = np.random.default_rng(0)
        pil = Image.fromarray(rng.integers(0, 255, (600, 800, 3), dtype=np.uint8)).resize((1024, 1024))
        x = fwd.image_processor.preprocess(pil)
        assert x.shape == (1, 3, 1024, 1024) and x.dtype == torch.float32 and -1.0 <= float(x.min()) < float(x.max()) <= 1.0
        ref = torch.from_numpy(np.asarray(pil, dtype=np.float32) / 255.0).permute(2, 0, 1)[None] * 2 - 1
        assert torch.equal(x, ref)
        _, emb = _embeds(cfg, 1, seed=13)
        kw = dict(num_inference_steps=3, timesteps=[19, 339, 699], guidance_scale=0.0, is_sdxl=True,
                  compute_embeddings_fn=_fn(emb), seed=5, return_start_latent=True)
        lat_a, start_a = G.inverse_sample_deterministic(fwd, x, ["a photo"], **kw)
        assert lat_a.shape == start_a.shape == (1, 4, 128, 128) and lat_a.dtype == torch.float16
        assert torch.isfinite(lat_a).all() and torch.isfinite(start_a).all()
        # the start latent is add_noise(sample * 0.13025, randn) with both draws from Generator().manual_seed(seed) on the CPU
        v.to(torch.float32)
        d = v.encode(x.cuda().half().float())["latent_dist"]
        v.to(torch.float16)
        g = torch.Generator().manual_seed(5)
        init = (d.mean.cpu() + d.std.cpu() * torch.randn(d.mean.shape, generator=g)).half() * 0.13025
        noise = torch.randn(init.shape, generator=g, dtype=torch.float16)
        want = fwd.scheduler.add_noise(init.cuda(), noise.cuda(), torch.tensor([19]))      # fp16 arithmetic, as diffusers
        assert torch.equal(start_a, want)
        # a PIL input takes the same path
        lat_b, start_b = G.inverse_sample_deterministic(fwd, pil, ["a photo"], **kw)
        assert torch.equal(start_a, start_b) and torch.equal(lat_a, lat_b)
    finally:
        fwd.vae = None


def test_fused_query_projection_cross_attention_inside_the_unet(sdxl):
    """The north-star kernel forced on for every eligible layer of the executor (option xattn_fusion = 1), full-width SDXL at a
    64x64 latent, B = 2: the 1024-token layers have C = 640 (not a multiple of 256) and the 256-token C = 1280 layers only make 10
    blocks of 256 x 256, so ALL 70 launches are hosted on the 128-wide tile of gemm.hip here (asserted from the planner records).
    The 256 x 256 host (xattn_epilogue_big) is pinned against the oracle at 128x128 latents by
    test_full_sdxl_forward_128x128_b2_on_the_benchmarked_tiles.  Same result as projection + attention (mode 0) up to the fp16
    rounding order of q."""
    from invertible_cd_amd import _lib, synthetic
    cfg, pipe, _ = sdxl
    inp = synthetic.synthetic_inputs(cfg, 2, 64, 64, seed=21, device="cpu")
    kw = dict(encoder_hidden_states=inp["context"].cuda().half(), timestep_cond=torch.randn(2, 512, generator=torch.Generator().manual_seed(1)).cuda().half(),
              added_cond_kwargs={"text_embeds": inp["text_embeds"].cuda().half(), "time_ids": inp["time_ids"].cuda()})
    x = inp["latents"].cuda().half()
    pipe.unet.set_option("xattn_fusion", 0)
    _lib.profile_enable(True)
    try:
        base = pipe.unet(x, 499, **kw).sample
        pipe.unet.set_option("xattn_fusion", 1)
        fused = pipe.unet(x, 499, **kw).sample
        torch.cuda.synchronize()
        fam = _lib.profile_read()
        plans = [p for p in _lib.profile_plans() if p["family"] == "xattn_fused"]
    finally:
        pipe.unet.set_option("xattn_fusion", 2)  # default: only where it measured faster
        _lib.profile_enable(False)
    assert fam["xattn_fused"]["launches"] == 70                  # every cross-attention layer of SDXL took the fused kernel
    assert len(plans) == 70 and all(p["xattn"] and not p["big"] and p["tile"] == (128, 128) for p in plans)
    e = rel_l2(fused, base)
    print(f"[sdxl fused xattn in the executor, 128-wide host] rel-L2 vs two launches = {e:.3e}")
    # two fp16 evaluation orders of the same graph: their distance is bounded by the fp16 floor of the graph (2.0-2.6e-3 at
    # this width, see test_unet_gpu), measured 0.9-1.0e-3
    assert torch.isfinite(fused).all() and e < 2e-3


def test_full_sdxl_forward_128x128_b2_on_the_benchmarked_tiles():
    """The code path bench.py times for SDXL, against the oracle: full width, B = 2 at 128x128 latents (13.5 TFLOP of CPU oracle).
    From the planner records: the GEMM flops run on the 256-wide tiles of gemm_big.hip, the first GEMM behind every LayerNorm
    takes the statistics from its main loop, and - under option xattn_fusion = 1 - the sixty 1024-token C = 1280 cross-attention
    layers run as the epilogue of the 192 x 256 query-projection tile (xattn_epilogue_big; six m-tiles per sample x 5 = 60 blocks each), the ten
    4096-token C = 640 layers on the 128-wide host.  Default options (mode 2 fuses nothing at B = 2) and the fused variant are both
    compared with the fp32 oracle and with the fp16-torch floor."""
    from test_unet_gpu import _run_case
    from invertible_cd_amd.unet_config import SDXL
    torch.cuda.empty_cache()

    def check(name, plans):
        if name in ("accurate", "fast"):          # (the split-operand launches of the accurate level: checked against the oracle only)
            return
        gem = [p for p in plans if p["family"] in ("gemm_dense", "gemm_conv", "xattn_fused")]
        fl = lambda ps: sum(2.0 * p["M"] * p["N"] * p["K"] for p in ps)
        big = [p for p in gem if p["big"]]
        inline = [p for p in gem if p["ln_inline"]]
        xa = [p for p in plans if p["family"] == "xattn_fused"]
        print(f"[plans {name}] {len(gem)} GEMM launches, {100 * fl(big) / fl(gem):.1f} % of their flops on gemm_big tiles "
              f"{sorted({p['tile'] for p in big})}, {len(inline)} with in-loop LayerNorm statistics, {len(xa)} fused cross-attention launches")
        # planner at this size (icd_gemm_plan): every to_qk (70) and the 4096-token to_q (10) take the statistics in their main
        # loop; the 1024-token N = 1280 projections (M = 2048: 64 blocks at best) stay on the 128-wide tiles
        assert fl(big) / fl(gem) > 0.8 and len(inline) >= (70 if name is None else 130)
        assert {(128, 320), (192, 256), (256, 256)} <= {p["tile"] for p in big}
        if name is None:
            assert len(xa) == 0
        else:
            on_big = [p for p in xa if p["big"] and p["tile"] == (192, 256) and p["N"] == 1280 and p["M"] == 2048]
            on_128 = [p for p in xa if not p["big"] and p["N"] == 640 and p["M"] == 8192]
            assert len(xa) == 70 and len(on_big) == 60 and len(on_128) == 10 and all(p["xattn"] for p in xa)
            assert all(p["ln_inline"] for p in on_big)            # ... with the LayerNorm statistics from the same main loop

    # ... and the accurate level of the precision policy (residual = 3, every ICD_SPLIT_* bit: what the inversion / edit loops run) on the
    # same tiles: < 0.6e-3 (last variant: an explicit residual option switches the policy of the handle off)
    r = _run_case(SDXL, B=2, H=128, W=128, t=699, seed=9, tol=1e-3,
                  variants={"xattn_everywhere": {"xattn_fusion": 1}, "accurate": {"xattn_fusion": 2, "residual": 3, "split_mask": 1023},
                            "fast": {"residual": 2}}, check_plans=check, cache_tag="sdxl_full_128x128_b2_eps")
    # (round 6: the default run follows the 'auto' policy, whose probe picks the level for these weights; the two levels are pinned explicitly)
    print(f"[sdxl B=2 128x128] fast level {r['fast'][0]:.3e} -> accurate level {r['accurate'][0]:.3e}; default (auto policy) {r[None][0]:.3e}")
    assert r["accurate"][0] < 0.6e-3 and r["accurate"][0] < 0.7 * r["fast"][0]


def test_full_width_sdxl_loops_64x64_b1_meet_1e3():
    """SURVEY 8c at full width for the SDXL loops (round 6; rounds 2 - 5 checked configs 4 / 5 against the oracle at widths (64, 128, 256)
    only): the 2.57 G-parameter UNet - 140 attention modules, transformer depth 10 - on 64 x 64 latents, B = 1.
      (a) config 4's loop: 4-step reverse (utils/generation_sdxl.py:324-473), gs = 7, the fast level of the precision policy;
      (b) config 5's forward leg: 3-step inversion (utils/generation_sdxl.py:204-310), w = 0, the accurate level (`with unet.editing()`).
    Both within 1e-3 rel-L2 of the fp32 oracle loop with fp16 latents between steps (7 full-width evaluations: ~12 TFLOP of CPU oracle)."""
    import test_sampler_gpu as T
    E = T._env()
    X = E["generation_sdxl"]
    cfg = E["SDXL"]
    torch.cuda.empty_cache()
    sd = {k: v.half().float() for k, v in E["synthetic"].synthetic_state_dict(cfg, seed=31).items()}
    B, H, W = 1, 64, 64
    inp = E["synthetic"].synthetic_inputs(cfg, B, H, W, seed=31)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    added = {"text_embeds": inp["text_embeds"].half().float(), "time_ids": inp["time_ids"]}
    u = E["unet"].UNet2DConditionModel(cfg, sd, dtype=torch.float16)
    pipe = E["StableDiffusionXLPipeline"](u, E["DDIMScheduler"].sdxl())
    fpipe = E["StableDiffusionXLImg2ImgPipeline"](u, E["DDIMScheduler"].sdxl())
    emb = lambda p, o, c: {"prompt_embeds": ctx.cuda().half(), "text_embeds": added["text_embeds"].cuda().half(),
                           "time_ids": added["time_ids"].cuda()}
    _, out = X.sample_deterministic(pipe, ["x"] * B, latents=lat.cuda().half(), num_inference_steps=4, guidance_scale=7.0, is_sdxl=True,
                                    timesteps=[249, 499, 699, 999], compute_embeddings_fn=emb, return_latent=True)
    ref = T._oracle_loop_xl(E, sd, cfg, lat.clone(), ctx, list(zip([999, 699, 499, 249], [699, 499, 249, 0])), [[7.0] * B] * 4, added,
                            cache_tag="sdxl_full_64x64_b1_reverse")
    e4 = rel_l2(out, ref)
    print(f"[sdxl FULL width 64x64 B=1, 4-step reverse] rel-L2 = {e4:.3e}  (probe gap {u._auto_gap}, plain level {u._auto_plain})")
    assert e4 < 1e-3
    fwd, start = X.inverse_sample_deterministic(fpipe, lat.cuda().half(), ["x"] * B, num_inference_steps=3, timesteps=[19, 339, 699],
                                                guidance_scale=0.0, is_sdxl=True, compute_embeddings_fn=emb, seed=3, return_start_latent=True)
    ref_f = T._oracle_loop_xl(E, sd, cfg, start.float().cpu(), ctx, list(zip([19, 339, 699], [339, 699, 999])), [[0.0] * B] * 3, added,
                              cache_tag="sdxl_full_64x64_b1_forward")
    e5 = rel_l2(fwd, ref_f)
    print(f"[sdxl FULL width 64x64 B=1, 3-step forward] rel-L2 = {e5:.3e}")
    assert e5 < 1e-3
