"""End-to-end parity of the sampler paths on the GPU against the CPU oracle (UNet restatement + boundary step), including
every attention-store tensor, plus size-independent properties at BASELINE.json's full sizes.

BASELINE configs covered here (reduced widths for the oracle-checked cases, full size for the property cases):
  cfg 2  iCD-SD1.5 4-step reverse (w-embedding, dead-uncond elimination on/off)
  cfg 3  iCD-SD1.5 4-step forward inversion + 4-step reverse edit with p2p controllers (AttentionStore / AttentionReplace)
  cfg 4  iCD-SDXL 4-step reverse
  cfg 5  iCD-SDXL 3-step forward + 3-step reverse with dynamic guidance (tau < 1)
Tolerances (round 5): the north star's 1e-3 rel-L2 is asserted on EVERY loop - final latents of the reverse loops, every attention-store
tensor of a reverse pass (reduced and full width), and since this round also the FORWARD (inversion) loops, the replace edit (latents and
edited store tensors) and SDXL's forward + dynamic-guidance reverse.  The forward loops amplify the per-evaluation error (every step adds
its own eps error times (sigma_s - alpha_s sigma_t / alpha_t) ~ 0.7 on top of the propagated one: ~2.4 x after 3 - 4 steps), so they need
one evaluation at ~0.4e-3: the 'accurate' level of the UNet's precision policy (unet.py; residual mode 3: GroupNorm reads the stream's
error carry, the shortcut / proj_out / sampler GEMMs take hi + lo; tests/error_budget_sim.py and profiles/r05_error_budget*.txt have the
budget), which the samplers select for inversion loops, dynamic-guidance passes and every pass with a controller attached.  Plain
generation runs at the 'fast' level (the error carry of round 4, one evaluation 0.70 - 0.85e-3), whose reverse loops contract the error
(3e-4 after 4 steps).  For scale: an fp16 diffusers-style pipeline (the oracle graph in fp16 torch) sits at 2 - 2.6e-3 per evaluation.
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

STORE_BAR = 1e-3
REV_T, REV_S = [999, 779, 519, 259], [779, 519, 259, 0]
FWD_T, FWD_S = [19, 259, 519, 779], [259, 519, 779, 999]


def _env():
    from invertible_cd_amd import generation, generation_sdxl, p2p, synthetic, unet
    from invertible_cd_amd.pipelines import StableDiffusionPipeline, StableDiffusionXLImg2ImgPipeline, StableDiffusionXLPipeline
    from invertible_cd_amd.schedulers import DDIMScheduler
    from invertible_cd_amd.unet_config import SD15, SDXL
    from oracle import sched_ref, unet_ref
    return locals()


def _ocfg(unet_ref, cfg):
    o = dict(unet_ref.SD15 if cfg.addition_time_embed_dim == 0 else unet_ref.SDXL)
    o.update(block_out_channels=cfg.block_out_channels, cross_dim=cfg.cross_dim, num_heads=cfg.num_heads)
    if cfg.addition_time_embed_dim:
        o["add_in_dim"] = cfg.add_in_dim
    return o


def _tables(sched_ref):
    ac = sched_ref.alphas_cumprod()
    return np.sqrt(ac), np.sqrt(1 - ac)


def _fused(E, cfg, sd, seed):
    """LoRA-fused weights as bench.py builds them (synthetic rank-64 LoRA -> loading.fuse_lora, utils/loading.py:64-71,119-125), rounded to the
    fp16 the executor stores: BOTH sides (oracle and GPU) get these, so the loop is checked on the weight statistics the benchmark runs
    (the fused update widens the activations; the fusion arithmetic itself is test_loading_gpu.py's business)."""
    from invertible_cd_amd.loading import fuse_lora
    fused = fuse_lora(sd, E["synthetic"].synthetic_lora(cfg, seed=seed + 100), lora_dtype=torch.float16)
    return {k: v.half().float() for k, v in fused.items()}


def _sd15_setup(E, B, H, W, seed, full=False, lora=False):
    cfg = E["SD15"] if full else E["SD15"].scaled((64, 128, 256, 256), cross_dim=64)
    sd = {k: v.half().float() for k, v in E["synthetic"].synthetic_state_dict(cfg, seed=seed).items()}
    if lora:
        sd = _fused(E, cfg, sd, seed)
    inp = E["synthetic"].synthetic_inputs(cfg, B, H, W, seed=seed)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    model = E["StableDiffusionPipeline"](E["unet"].UNet2DConditionModel(cfg, sd, dtype=torch.float16), E["DDIMScheduler"].sd15(),
                                         tokenizer=E["synthetic"].SyntheticTokenizer(), device="cuda", dtype=torch.float16)
    solver = E["generation"].Generator(model, 50, E["DDIMScheduler"].sd15(), forward_cons_model=model, reverse_cons_model=model,
                                       reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    solver.context = torch.cat([torch.zeros_like(ctx), ctx]).cuda().half()
    return cfg, sd, lat, ctx, model, solver


def _oracle_loop(E, sd, cfg, x, ctx, pairs, w_vals, controller=None, added=None, cache_tag=None):
    """cond rows only (the unconditional rows never influence the output when w_embed_dim > 0).  cache_tag (full-width cases without a
    controller): the result may come from a committed fixture keyed by a hash of everything it depends on (tests/oracle_cache.py)."""
    if cache_tag is not None and controller is None:
        import oracle_cache
        return oracle_cache.lookup(cache_tag, sd, [cfg.name, x, ctx, pairs, w_vals, added],
                                   lambda: _oracle_loop(E, sd, cfg, x, ctx, pairs, w_vals, added=added))
    S, U = E["sched_ref"], E["unet_ref"]
    alpha, sigma = _tables(S)
    ocfg = _ocfg(U, cfg)
    B = x.shape[0]
    hook = None
    if controller is not None:
        hook = lambda p, is_cross, place: controller.call_cond_only(p, is_cross, place)
    for (t, s), w in zip(pairs, w_vals):
        wemb = torch.from_numpy(S.guidance_scale_embedding(w, 512)).half().float()
        eps = U.unet_forward(sd, ocfg, x.half().float(), t, ctx, timestep_cond=wemb, added_cond=added, hook=hook).half().float()
        x = torch.from_numpy(S.predicted_origin(eps.numpy(), [t] * B, [s] * B, x.numpy(), alpha, sigma))
        if controller is not None:
            x = controller.step_callback(x)
    return x


@pytest.mark.parametrize("eliminate,lora", [(True, False), (False, False), (True, True)])
def test_sd15_reverse_with_attention_store(eliminate, lora):
    """cfg 2/3: 4-step reverse, AttentionStore registered; latents AND every stored probability tensor vs the oracle.
    lora (round 6): the same loop on LoRA-fused weights, as the benchmark (and every shipped iCD checkpoint) runs."""
    E = _env()
    p2p = E["p2p"]
    B, H, W, gs = 3, 32, 32, 7.0
    cfg, sd, lat, ctx, model, solver = _sd15_setup(E, B, H, W, seed=11, lora=lora)
    solver.eliminate_dead_uncond = eliminate
    store = p2p.AttentionStore()
    p2p.register_attention_control(model, store)
    assert store.num_att_layers == 32
    outs = solver.cons_generation(lat.cuda(), guidance_scale=gs, w_embed_dim=512, dynamic_guidance=False, tau1=1.0, tau2=1.0,
                                  controller=store)
    assert len(outs) == 5 and outs[-1].dtype == torch.float32          # fp32 latent chain, as in the reference
    ref_store = p2p.AttentionStore()
    ref_store.num_att_layers = 32
    ref = _oracle_loop(E, sd, cfg, lat.clone(), ctx, list(zip(REV_T, REV_S)), [[gs] * B] * 4, controller=ref_store)
    err = rel_l2(outs[-1], ref)
    print(f"[sd15 reverse, eliminate={eliminate}, lora={lora}] rel-L2(latents) = {err:.3e}")
    assert err < 1e-3                                # measured 2.9e-4 (4.1e-4 on the plain fp16 stream of rounds 1-3)
    assert store.cur_step == ref_store.cur_step == 4
    n_checked, worst = 0, 0.0
    for key, refs in ref_store.attention_store.items():
        got = store.attention_store[key]
        assert len(got) == len(refs), key
        for g, r in zip(got, refs):
            gc = g                                   # both modes store the conditional rows only (utils/p2p.py:106-107)
            assert tuple(gc.shape) == tuple(r.shape), (key, gc.shape, r.shape)
            e = rel_l2(gc, r)
            worst = max(worst, e)
            assert e < 1e-3, (key, e)              # the north star's bar on every stored tensor (worst measured: 8.3e-4)
            n_checked += 1
    print(f"[sd15 reverse, eliminate={eliminate}, lora={lora}] worst attention-store tensor rel-L2 = {worst:.3e}")
    # latent 32x32 -> query counts 1024,1024,256,256,64,64 down; 16 mid; up 64x3,256x3,1024x3: all <= 32^2 -> all 32 stored
    assert n_checked == 32


@pytest.mark.slow
def test_full_width_sd15_64x64_reverse_store_and_inversion_meet_1e3():
    """SURVEY 8c(3) at full width: the 859.7 M-parameter SD1.5 UNet on 64 x 64 latents (512 x 512 images), B = 2, the benchmarked
    residual mode.  (a) 4-step reverse with an AttentionStore: final latents and EVERY stored tensor (22 per pass at this size: the 32^2,
    16^2 and 8^2 layers) within 1e-3 of the fp32 oracle loop; (b) the 4-step forward inversion, within 1e-3 too since round 5 (the accurate level).
    2 x 6.4 TFLOP of CPU oracle."""
    E = _env()
    p2p = E["p2p"]
    B, H, W, gs = 2, 64, 64, 7.0
    cfg, sd, lat, ctx, model, solver = _sd15_setup(E, B, H, W, seed=15, full=True)
    store = p2p.AttentionStore()
    p2p.register_attention_control(model, store)
    outs = solver.cons_generation(lat.cuda(), guidance_scale=gs, w_embed_dim=512, dynamic_guidance=False, tau1=1.0, tau2=1.0,
                                  controller=store)
    ref_store = p2p.AttentionStore()
    ref_store.num_att_layers = 32
    # B == 2 -> the CFG-doubled batch of 4 takes the [0, 0, 0, gs] branch of the w vector (utils/generation.py:232-233): cond rows [0, gs]
    ref = _oracle_loop(E, sd, cfg, lat.clone(), ctx, list(zip(REV_T, REV_S)), [[0.0, gs]] * 4, controller=ref_store)
    err = rel_l2(outs[-1], ref)
    worst, n = 0.0, 0
    for key, refs in ref_store.attention_store.items():
        got = store.attention_store[key]
        assert len(got) == len(refs), key
        for g, r in zip(got, refs):
            e = rel_l2(g, r)
            worst, n = max(worst, e), n + 1
            assert e < 1e-3, (key, tuple(r.shape), e)
    print(f"[sd15 full width 64x64 reverse] rel-L2(latents) = {err:.3e}, worst of {n} attention-store tensors = {worst:.3e}")
    assert err < 1e-3 and n == 22
    p2p.register_attention_control(model, None)
    del store, ref_store
    solver.latent2image = lambda z, return_type="np": np.zeros((1,))
    _, inv = solver.cons_inversion(lat.cuda(), guidance_scale=0.0, w_embed_dim=512, seed=5)
    alpha, sigma = _tables(E["sched_ref"])
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(5))
    x0 = float(alpha[19]) * lat + float(sigma[19]) * noise
    ref_inv = _oracle_loop(E, sd, cfg, x0.clone(), ctx, list(zip(FWD_T, FWD_S)), [[0.0, 0.0]] * 4, cache_tag="sd15_full_64x64_b2_inversion")
    e_inv = rel_l2(inv[0], ref_inv)
    print(f"[sd15 full width 64x64 inversion] rel-L2 = {e_inv:.3e}")
    assert e_inv < 1e-3


@pytest.mark.parametrize("lora", [False, True])
def test_sd15_inversion_then_replace_edit(lora):
    """cfg 3: consistency inversion (forward model, w = 0) then a 2-prompt AttentionReplace edit on the reverse model.
    lora (round 6): on LoRA-fused weights."""
    E = _env()
    p2p = E["p2p"]
    B, H, W = 2, 32, 32
    cfg, sd, lat, ctx, model, solver = _sd15_setup(E, B, H, W, seed=12, lora=lora)
    # ---- inversion: 4D latents pass straight through image2latent; add_noise at t=19 with the CPU generator(seed)
    solver.latent2image = lambda z, return_type="np": np.zeros((1,))
    img = lat.cuda()
    _, inv = solver.cons_inversion(img, guidance_scale=0.0, w_embed_dim=512, seed=5)
    S = E["sched_ref"]
    alpha, sigma = _tables(S)
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(5))
    x0 = float(alpha[19]) * lat + float(sigma[19]) * noise
    # B == 2 -> CFG-doubled batch of 4 -> w vector [0,0,0,gs] with gs = 0 -> all zeros
    ref_inv = _oracle_loop(E, sd, cfg, x0.clone(), ctx, list(zip(FWD_T, FWD_S)), [[0.0, 0.0]] * 4)
    e_inv = rel_l2(inv[0], ref_inv)
    print(f"[sd15 inversion, lora={lora}] rel-L2 = {e_inv:.3e}")
    assert e_inv < 1e-3                              # round 5, accurate level: 6.6e-4 (1.03e-3 on the carry alone, 1.37e-3 on the plain fp16 stream)
    # ---- edit: replace controller (cross 0.5 / self 0.5), dynamic guidance tau = 0.8, gs = 19
    p2p.tokenizer = E["synthetic"].SyntheticTokenizer()
    p2p.NUM_DDIM_STEPS = 4
    prompts = ["a cat sitting on a bench", "a dog sitting on a bench"]
    p2p.device = "cuda"
    ctrl = p2p.make_controller(prompts, True, 0.5, 0.5)
    p2p.device = "cpu"
    ref_ctrl = p2p.make_controller(prompts, True, 0.5, 0.5)
    ref_ctrl.num_att_layers = 32
    p2p.register_attention_control(model, ctrl)
    start = ref_inv.clone()                         # same starting latents on both sides
    outs = solver.cons_generation(start.cuda(), guidance_scale=19.0, w_embed_dim=512, dynamic_guidance=True, tau1=0.8, tau2=0.8,
                                  controller=ctrl)
    # w: 0 at t=999 (> tau), then [0, gs] for the cond half of the 4-row doubled batch (utils/generation.py:232-233 quirk)
    ws = [[0.0, 0.0]] + [[0.0, 19.0]] * 3
    ref = _oracle_loop(E, sd, cfg, start.clone(), ctx, list(zip(REV_T, REV_S)), ws, controller=ref_ctrl)
    e = rel_l2(outs[-1], ref)
    print(f"[sd15 replace edit, lora={lora}] rel-L2 = {e:.3e}")
    assert e < 1e-3                                  # round 5: 7.7e-4 (1.20e-3 on the carry alone; gs = 19, edited probabilities)
    assert ctrl.cur_step == 4
    worst, errs = 0.0, []
    for key, refs in ref_ctrl.attention_store.items():
        for i, (g, r) in enumerate(zip(ctrl.attention_store[key], refs)):
            errs.append((rel_l2(g, r), key, i, tuple(r.shape)))
    errs.sort(reverse=True)
    worst = errs[0][0]
    print(f"[sd15 replace edit, lora={lora}] worst attention-store tensor rel-L2 = {worst:.3e}; top: " +
          ", ".join(f"{k}[{i}]{sh} {e:.2e}" for e, k, i, sh in errs[:6]))
    for e, k, i, sh in errs:
        assert e < STORE_BAR, (k, i, sh, e)


@pytest.mark.parametrize("lora", [False, True])
def test_sdxl_reverse_and_dynamic_edit_pipeline(lora):
    """cfg 4 + cfg 5: SDXL 4-step reverse; 3-step forward + 3-step reverse with dynamic guidance (tau = 0.7).
    lora (round 6): on LoRA-fused weights (utils/loading.py:119-125)."""
    E = _env()
    X = E["generation_sdxl"]
    cfg = E["SDXL"].scaled((64, 128, 256), cross_dim=128, heads=(2, 4, 8))
    sd = {k: v.half().float() for k, v in E["synthetic"].synthetic_state_dict(cfg, seed=21).items()}
    if lora:
        sd = _fused(E, cfg, sd, 21)
    B, H, W = 2, 32, 32
    inp = E["synthetic"].synthetic_inputs(cfg, B, H, W, seed=21)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    added = {"text_embeds": inp["text_embeds"].half().float(), "time_ids": inp["time_ids"]}
    u = E["unet"].UNet2DConditionModel(cfg, sd, dtype=torch.float16)
    pipe = E["StableDiffusionXLPipeline"](u, E["DDIMScheduler"].sdxl())
    fpipe = E["StableDiffusionXLImg2ImgPipeline"](u, E["DDIMScheduler"].sdxl())
    emb = lambda p, o, c: {"prompt_embeds": ctx.cuda().half(), "text_embeds": added["text_embeds"].cuda().half(),
                           "time_ids": added["time_ids"].cuda()}
    # cfg 4
    _, out = X.sample_deterministic(pipe, ["x"] * B, latents=lat.cuda().half(), num_inference_steps=4, guidance_scale=7.0, is_sdxl=True,
                                    timesteps=[249, 499, 699, 999], compute_embeddings_fn=emb, return_latent=True)
    assert out.dtype == torch.float16
    pairs = list(zip([999, 699, 499, 249], [699, 499, 249, 0]))
    ref = _oracle_loop_xl(E, sd, cfg, lat.clone(), ctx, pairs, [[7.0] * B] * 4, added)
    e4 = rel_l2(out, ref)
    print(f"[sdxl reverse, lora={lora}] rel-L2 = {e4:.3e}")
    assert e4 < 1e-3                                 # measured 4.9e-4 (6.1e-4 before the carry)
    # cfg 5: forward 3 steps (w = 0) from noised latents at t = 19, then reverse 3 steps with tau = 0.7, gs = 19
    fwd, start = X.inverse_sample_deterministic(fpipe, lat.cuda().half(), ["x"] * B, num_inference_steps=3, timesteps=[19, 339, 699],
                                                guidance_scale=0.0, is_sdxl=True, compute_embeddings_fn=emb, seed=3,
                                                return_start_latent=True)
    S = E["sched_ref"]
    alpha, sigma = _tables(S)
    noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float16).float()
    x0 = start.float().cpu()                         # add_noise at t = 19 is scheduler plumbing (fp16 arithmetic as in diffusers)
    assert rel_l2(x0, float(alpha[19]) * lat + float(sigma[19]) * noise) < 2e-3
    ref_f = _oracle_loop_xl(E, sd, cfg, x0, ctx, list(zip([19, 339, 699], [339, 699, 999])), [[0.0] * B] * 3, added)
    e5f = rel_l2(fwd, ref_f)
    print(f"[sdxl forward, lora={lora}] rel-L2 = {e5f:.3e}")
    assert e5f < 1e-3                                # round 5, accurate level (1.45e-3 on the carry alone): three forward steps amplify
    _, rev = X.sample_deterministic(pipe, ["x"] * B, latents=ref_f.cuda().half(), num_inference_steps=3, guidance_scale=19.0,
                                    is_sdxl=True, timesteps=[339, 699, 999], compute_embeddings_fn=emb, return_latent=True,
                                    use_dynamic_guidance=True, tau1=0.7, tau2=0.7)
    ws = [[S.linear_schedule_old(t, 19.0, 0.7, 0.7)] * B for t in (999, 699, 339)]
    assert [w[0] for w in ws] == [0.0, 19.0, 19.0]
    ref_r = _oracle_loop_xl(E, sd, cfg, ref_f.half().float(), ctx, list(zip([999, 699, 339], [699, 339, 0])), ws, added)
    e5r = rel_l2(rev, ref_r)
    print(f"[sdxl dynamic reverse, lora={lora}] rel-L2 = {e5r:.3e}")
    assert e5r < 1e-3                                # round 5, accurate level (1.31e-3 on the carry alone; gs = 19 from t = 999)


def _oracle_loop_xl(E, sd, cfg, x, ctx, pairs, w_vals, added, cache_tag=None):
    """SDXL keeps fp16 latents between steps (utils/generation_sdxl.py:463).  cache_tag: as _oracle_loop."""
    if cache_tag is not None:
        import oracle_cache
        return oracle_cache.lookup(cache_tag, sd, [cfg.name, x, ctx, pairs, w_vals, added],
                                   lambda: _oracle_loop_xl(E, sd, cfg, x, ctx, pairs, w_vals, added))
    S, U = E["sched_ref"], E["unet_ref"]
    alpha, sigma = _tables(S)
    ocfg = _ocfg(U, cfg)
    B = x.shape[0]
    for (t, s), w in zip(pairs, w_vals):
        wemb = torch.from_numpy(S.guidance_scale_embedding(w, 512)).half().float()
        eps = U.unet_forward(sd, ocfg, x, t, ctx, timestep_cond=wemb, added_cond=added).half().float()
        x = torch.from_numpy(S.predicted_origin(eps.numpy(), [t] * B, [s] * B, x.numpy(), alpha, sigma)).half().float()
    return x


# ------------------------------------------------------------------------------------------ full-size properties
@pytest.mark.slow
def test_full_size_sd15_properties():
    """BASELINE cfg 2 shape (SD1.5, B=32, 64x64 latents, 4 steps): determinism, batch independence, store structure."""
    E = _env()
    p2p, cfg = E["p2p"], E["SD15"]
    sd = E["synthetic"].synthetic_state_dict(cfg, seed=0, device="cuda", dtype=torch.float16)
    model = E["StableDiffusionPipeline"](E["unet"].UNet2DConditionModel(cfg, sd), E["DDIMScheduler"].sd15(),
                                         tokenizer=E["synthetic"].SyntheticTokenizer(), device="cuda", dtype=torch.float16)
    del sd
    solver = E["generation"].Generator(model, 50, E["DDIMScheduler"].sd15(), forward_cons_model=model, reverse_cons_model=model,
                                       reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    B = 32
    g = torch.Generator().manual_seed(453645634)
    lat = torch.randn(B, 4, 64, 64, generator=g).cuda()
    ctx = torch.randn(2 * B, 77, 768, generator=g).cuda().half()
    solver.context = ctx
    run = lambda l: solver.cons_generation(l, guidance_scale=7.0, w_embed_dim=512, dynamic_guidance=False, tau1=1.0, tau2=1.0)[-1]
    a, b = run(lat), run(lat)
    assert torch.isfinite(a).all() and torch.equal(a, b)                      # run-to-run bit reproducibility
    # sample i does not depend on the rest of the batch (data-parallel sharding is exact): rerun rows 8..15 alone
    solver.context = torch.cat([ctx[8:16], ctx[B + 8:B + 16]])
    sub = run(lat[8:16].contiguous())
    assert rel_l2(sub, a[8:16]) < 2e-3                                        # different tile plans -> fp rounding only
    # cfg 3 shape: B = 8 with AttentionStore: 11 cross + 11 self stored per step with the reference's query counts
    solver.context = torch.cat([ctx[:8], ctx[B:B + 8]])
    store = p2p.AttentionStore()
    p2p.register_attention_control(model, store)
    solver.cons_generation(lat[:8].contiguous(), guidance_scale=7.0, w_embed_dim=512, dynamic_guidance=False, tau1=1.0, tau2=1.0,
                           controller=store)
    assert store.cur_step == 4
    q = {k: [t.shape[1] for t in v] for k, v in store.attention_store.items()}
    assert q["down_cross"] == q["down_self"] == [1024, 1024, 256, 256]
    assert q["mid_cross"] == q["mid_self"] == [64]
    assert q["up_cross"] == q["up_self"] == [256, 256, 256, 1024, 1024, 1024]
    for k, v in store.attention_store.items():
        for t in v:
            assert t.shape[0] == 8 * 8 and t.shape[2] == (77 if "cross" in k else t.shape[1])
            rows = (t.float() / 4).sum(-1)                                    # accumulated over 4 steps; each row a softmax
            assert float((rows - 1).abs().max()) < 5e-3
    p2p.register_attention_control(model, None)
    # cfg 3's FORWARD leg at its size: 4-step consistency inversion of 8 latents (w = 0, utils/generation.py:414-451), then the
    # reverse pass with an AttentionStore from the inverted latents - the pair bench.py times as "edit"
    solver.latent2image = lambda z, return_type="np": None
    inv = lambda l: solver.cons_inversion(l, guidance_scale=0.0, w_embed_dim=512, seed=5)[1][0]
    i1, i2 = inv(lat[:8].contiguous()), inv(lat[:8].contiguous())
    assert i1.shape == (8, 4, 64, 64) and torch.isfinite(i1).all() and torch.equal(i1, i2) and float(i1.float().std()) > 0.5
    # the CPU noise of a 2-sample call is the head of the 8-sample draw (same seed, one stream): rows 0..1 alone == rows 0..1 of the batch
    solver.context = torch.cat([ctx[:2], ctx[B:B + 2]])
    assert rel_l2(inv(lat[:2].contiguous()), i1[:2]) < 3e-3
    solver.context = torch.cat([ctx[:8], ctx[B:B + 8]])
    store = p2p.AttentionStore()
    p2p.register_attention_control(model, store)
    out = solver.cons_generation(i1, guidance_scale=19.0, w_embed_dim=512, dynamic_guidance=True, tau1=0.8, tau2=0.8, controller=store)[-1]
    p2p.register_attention_control(model, None)
    assert out.shape == (8, 4, 64, 64) and torch.isfinite(out).all() and store.cur_step == 4
    assert sum(len(v) for v in store.attention_store.values()) == 22


def test_boundary_step_round_trip_full_size():
    """predicted_origin(t -> s) followed by (s -> t) with the same eps is the identity (B=32 x 4x64x64, and 128x128)."""
    from invertible_cd_amd import ops
    from oracle import sched_ref
    ac = sched_ref.alphas_cumprod()
    al, si = np.sqrt(ac), np.sqrt(1 - ac)
    for shape in ((32, 4, 64, 64), (8, 4, 128, 128)):
        g = torch.Generator().manual_seed(1)
        x, eps = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
        B = shape[0]
        for t, s in ((999, 779), (519, 259), (19, 259)):
            fwd = torch.tensor([[al[t], si[t], al[s], si[s]]] * B, dtype=torch.float32)
            bwd = torch.tensor([[al[s], si[s], al[t], si[t]]] * B, dtype=torch.float32)
            y = ops.x0_step(x, eps, fwd)
            back = ops.x0_step(y, eps, bwd)
            assert rel_l2(back, x) < 1e-5


def test_ddim_baseline_loops_match_oracle():
    """The DDIM baselines that share the UNet (utils/generation.py:183-205,305-371): 3 reverse DDIM steps
    with classic classifier-free guidance (w_embed_dim = 0 -> `guided_step`, the CFG-doubled batch really is evaluated) and
    3 forward (inversion) steps at guidance 1, against the oracle UNet + the DDIM update written out."""
    E = _env()
    B, H, W, g = 2, 16, 16, 7.5
    cfg, sd, lat, ctx, model, solver = _sd15_setup(E, B, H, W, seed=31)
    U = E["unet_ref"]
    ocfg = _ocfg(U, cfg)
    ac = model.scheduler.alphas_cumprod.double().numpy()
    ts = [int(t) for t in model.scheduler.timesteps]
    assert len(ts) == 50 and ts[0] - ts[1] == 20          # 50-step DDIM grid (980.. or 981.. depending on steps_offset)
    ctx2 = torch.cat([torch.zeros_like(ctx), ctx])

    def ddim_update(x, eps, t_from, t_to):
        a_f = ac[t_from] if t_from >= 0 else float(model.scheduler.final_alpha_cumprod)
        a_t = ac[t_to] if t_to >= 0 else float(model.scheduler.final_alpha_cumprod)
        x0 = (x - (1 - a_f) ** 0.5 * eps) / a_f ** 0.5
        return a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * eps

    # ---- reverse: the first three grid points, each t -> t - 20
    outs = solver.ddim_loop(lat.cuda(), 3, is_forward=False, guidance_scale=g)
    x = lat.clone()
    for t in ts[:3]:
        e2 = U.unet_forward(sd, ocfg, torch.cat([x, x]).half().float(), t, ctx2).half().float()
        eps = e2[:B] + g * (e2[B:] - e2[:B])
        x = ddim_update(x.double(), eps.double(), t, t - 20).float()
    e_rev = rel_l2(outs[-1], x)
    # ---- forward (inversion): the last three grid points in ascending order, stepping from t - 20 (final alpha below 0) up
    # to t, guidance 1 (cond prediction only)
    outs_f = solver.ddim_loop(lat.cuda(), 3, is_forward=True, guidance_scale=1)
    x = lat.clone()
    for t in ts[::-1][:3]:
        e2 = U.unet_forward(sd, ocfg, torch.cat([x, x]).half().float(), t, ctx2).half().float()
        eps = e2[:B] + 1.0 * (e2[B:] - e2[:B])
        x = ddim_update(x.double(), eps.double(), min(t - 20, 999), t).float()
    e_fwd = rel_l2(outs_f[-1], x)
    print(f"[sd15 DDIM baseline] reverse (CFG 7.5, 3 steps) rel-L2 = {e_rev:.3e}  forward (3 steps) = {e_fwd:.3e}")
    assert len(outs) == len(outs_f) == 4
    assert e_rev < 3e-3 and e_fwd < 2e-3


@pytest.mark.parametrize("kind", ["store", "replace", "refine_reweight_blend"])
def test_fused_probability_epilogue_matches_the_separate_passes_bit_for_bit(kind):
    """Section 8(f) rank 2 as written: for the reference's shipped controllers the edit operator, the self-attention replacement and
    store += P ride in the probability kernel's epilogue (icd_probs_epilogue via the hook's phase 2; p2p.HookAdapter) - one pass over P on
    hooked layers.  The same loop with `controller.fused_epilogue = False` (probabilities -> icd_p2p_cross_edit / row copy -> P.V, one
    accumulate launch per step) must give the same bits: latents of every step and every stored tensor.  The executor's launch records show
    that the fused run issued no accumulate / edit pass of its own."""
    E = _env()
    p2p = E["p2p"]
    from invertible_cd_amd import p2p as P
    B, H, W = (3, 32, 32) if kind == "store" else (2, 32, 32)
    cfg, sd, lat, ctx, model, solver = _sd15_setup(E, B, H, W, seed=23)
    p2p.tokenizer = E["synthetic"].SyntheticTokenizer()
    p2p.NUM_DDIM_STEPS = 4
    prompts = ["a cat sitting on a bench", "a dog sitting on a bench"]

    def make():
        if kind == "store":
            return p2p.AttentionStore()
        p2p.device = "cuda"
        try:
            if kind == "replace":
                return p2p.make_controller(prompts, True, 0.5, 0.5)
            return p2p.make_controller(prompts, False, {"default_": 0.6, "dog": (0.0, 0.3)}, 0.4, blend_words=(("cat",), ("dog",)),
                                       equilizer_params={"words": ("dog",), "values": (2.0,)})
        finally:
            p2p.device = "cpu"

    results = []
    for fused in (True, False):
        ctrl = make()
        ctrl.fused_epilogue = fused
        p2p.register_attention_control(model, ctrl)
        seen = {"epi": 0}
        orig = P.HookAdapter._plan_epilogue

        def counting(self, *a, **k):
            e = orig(self, *a, **k)
            seen["epi"] += e is not None
            return e
        P.HookAdapter._plan_epilogue = counting
        try:
            outs = solver.cons_generation(lat.cuda(), guidance_scale=19.0 if kind != "store" else 7.0, w_embed_dim=512,
                                          dynamic_guidance=kind != "store", tau1=0.8, tau2=0.8, controller=ctrl)
        finally:
            P.HookAdapter._plan_epilogue = orig
        torch.cuda.synchronize()
        results.append((outs, {k: [t.clone() for t in v] for k, v in ctrl.attention_store.items()}, seen["epi"], ctrl.cur_step))
        p2p.register_attention_control(model, None)
    (o1, s1, n1, c1), (o0, s0, n0, c0) = results
    print(f"[fused epilogue, {kind}] layers with an epilogue: fused run {n1}, separate passes {n0}")
    assert c1 == c0 == 4 and n0 == 0
    assert n1 >= (3 * 32 if kind == "store" else 4 * 16)          # store: every layer of steps 2-4; edits: at least every cross layer
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    assert s1.keys() == s0.keys()
    for k in s1:
        assert len(s1[k]) == len(s0[k]) and all(torch.equal(a, b) for a, b in zip(s1[k], s0[k])), k


def test_probabilities_are_materialised_for_the_conditional_half_of_a_cfg_batch_only():
    """Round 5: the reference's controllers work on attn[h // 2:] of a [uncond; cond] batch (utils/p2p.py:153-155), so the executor writes P
    for the conditional samples alone (hook return 2) and the unconditional half takes the fused kernel.  Against the same loop with
    `controller.cond_rows_only = False` (the whole batch materialised): every stored tensor and every latent has the same bits - the
    conditional rows go through the same kernels, and the guidance-distilled loop discards the unconditional outputs, the only ones that
    changed kernels - while the probability kernels move half the bytes (the executor's launch records)."""
    E = _env()
    p2p = E["p2p"]
    from invertible_cd_amd import _lib
    cfg, sd, lat, ctx, model, solver = _sd15_setup(E, 2, 32, 32, seed=29)
    p2p.tokenizer = E["synthetic"].SyntheticTokenizer()
    p2p.NUM_DDIM_STEPS = 4
    solver.eliminate_dead_uncond = False       # the reference's batching: every evaluation sees the [uncond; cond] batch (the default of this
    res = []                                   # package never computes the dead unconditional rows of the guidance-distilled loops at all)
    for half in (True, False):
        ctrl = p2p.AttentionStore()
        ctrl.cond_rows_only = half
        p2p.register_attention_control(model, ctrl)
        _lib.profile_enable(True)
        try:
            outs = solver.cons_generation(lat.cuda(), guidance_scale=7.0, w_embed_dim=512, dynamic_guidance=False, tau1=1.0, tau2=1.0, controller=ctrl)
            torch.cuda.synchronize()
            fam = _lib.profile_read()
        finally:
            _lib.profile_enable(False)
            p2p.register_attention_control(model, None)
        res.append((outs, {k: [t.clone() for t in v] for k, v in ctrl.attention_store.items()}, fam))
    (o1, s1, f1), (o0, s0, f0) = res
    assert s1.keys() == s0.keys() and sum(len(v) for v in s1.values()) > 0
    for k in s1:
        assert all(torch.equal(a, b) for a, b in zip(s1[k], s0[k])), k
    e = max(rel_l2(a, b.float().cpu()) for a, b in zip(o1, o0))
    print(f"[conditional half only] latents vs whole-batch materialisation rel-L2 = {e:.2e}; probability kernel bytes "
          f"{f1['softmax']['bytes'] / 1e6:.1f} MB vs {f0['softmax']['bytes'] / 1e6:.1f} MB")
    assert e == 0.0
    assert abs(f1["softmax"]["bytes"] / f0["softmax"]["bytes"] - 0.5) < 0.02
    assert f1["attn_fused"]["launches"] > f0["attn_fused"]["launches"]
