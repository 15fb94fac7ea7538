"""Prompt -> image through the reference-shaped API with every stage on the HIP operators: SyntheticTokenizer ids ->
CLIPTextModel (clip.py) -> 4-step consistency loop with an AttentionStore controller (unet.py + p2p.py) -> AutoencoderKL
decode (vae.py).  Full SD1.5 sizes, seeded synthetic weights (no checkpoints offline): checks the plumbing between the
rows of SURVEY.md section 8 (shapes, dtypes, determinism, controller bookkeeping), not image quality."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_prompt_to_image_end_to_end():
    from invertible_cd_amd import clip, generation, p2p, synthetic, vae
    from invertible_cd_amd.loading import load_models
    from invertible_cd_amd.schedulers import DDIMScheduler
    comp = {"vae_state_dict": synthetic.synthetic_vae_state_dict(vae.SD_VAE, seed=0, device="cuda", dtype=torch.float16),
            "text_encoder_state_dict": synthetic.synthetic_clip_state_dict(clip.CLIP_VIT_L, seed=0)}
    ldm, rev, fwd = load_models("synthetic:sd15", "cuda", reverse_checkpoint="synthetic:1", forward_checkpoint=None,
                                w_embed_dim=512, dtype="fp16", components=comp)
    assert isinstance(rev.vae, vae.AutoencoderKL) and isinstance(rev.text_encoder, clip.CLIPTextModel) and fwd is None
    solver = generation.Generator(ldm, 50, DDIMScheduler.sd15(), forward_cons_model=rev, reverse_cons_model=rev,
                                  reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    prompts = ["a photo of a cat sitting on a bench", "a photo of a dog sitting on a bench", "a red car"]

    def run():
        store = p2p.AttentionStore()
        img, lat = generation.runner(model=rev, prompt=prompts, controller=store, solver=solver, is_cons_forward=True,
                                     generator=torch.Generator().manual_seed(3), guidance_scale=7.0, tau1=1.0, tau2=1.0,
                                     w_embed_dim=512, return_type="image")
        return img, lat, store

    img, lat, store = run()
    assert isinstance(img, np.ndarray) and img.shape == (3, 512, 512, 3) and img.dtype == np.uint8
    assert lat.shape == (1, 4, 64, 64)                      # the shared initial latent (utils/generation.py:538-542)
    assert solver.context.shape == (6, 77, 768)             # cat([uncond.expand, cond]) of the CLIP hidden states
    assert store.cur_step == 4 and len(store.attention_store["down_cross"]) == 4
    img2, _, _ = run()
    assert np.array_equal(img, img2)                        # same seed -> bit-identical image
    assert img.std() > 0                                    # not a constant image
    # latents instead of images, no controller
    latents, _ = generation.runner(model=rev, prompt=prompts[:1], controller=None, solver=solver, is_cons_forward=True,
                                   generator=torch.Generator().manual_seed(3), guidance_scale=7.0, tau1=1.0, tau2=1.0,
                                   w_embed_dim=512, return_type="latent")
    assert latents.shape == (1, 4, 64, 64) and torch.isfinite(latents).all()


def test_image_file_inversion_then_edit(tmp_path):
    """The reference's editing flow (running/sd1.5/edit.py:353-455): image file -> inversion.invert (VAE encode + 4-step
    forward consistency inversion) -> make_controller (Replace + Reweight + LocalBlend) -> runner (4-step reverse with
    dynamic guidance) -> decoded images.  Full SD1.5 sizes, synthetic weights; checks plumbing and invariants."""
    from PIL import Image
    from invertible_cd_amd import clip, generation, inversion, p2p, synthetic, vae
    from invertible_cd_amd.loading import load_models
    from invertible_cd_amd.schedulers import DDIMScheduler
    rng = np.random.default_rng(0)
    path = str(tmp_path / "img.png")
    Image.fromarray(rng.integers(0, 255, (300, 400, 3), dtype=np.uint8)).save(path)          # load_512 resizes to 512x512
    comp = {"vae_state_dict": synthetic.synthetic_vae_state_dict(vae.SD_VAE, seed=0, device="cuda", dtype=torch.float16),
            "text_encoder_state_dict": synthetic.synthetic_clip_state_dict(clip.CLIP_VIT_L, seed=0)}
    ldm, rev, fwd = load_models("synthetic:sd15", "cuda", reverse_checkpoint="synthetic:1", forward_checkpoint="synthetic:2",
                                w_embed_dim=512, dtype="fp16", components=comp)
    solver = generation.Generator(ldm, 50, DDIMScheduler.sd15(), forward_cons_model=fwd, reverse_cons_model=rev,
                                  reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    src, dst = "a photo of a cat sitting on a bench", "a photo of a dog sitting on a bench"
    (image_gt, image_rec), latent, uncond = inversion.invert(solver, stop_step=50, is_cons_inversion=True, inv_guidance_scale=0.0,
                                                             w_embed_dim=512, image_path=path, prompt=src, seed=1)
    assert image_gt.shape == (512, 512, 3) and image_rec.shape == (512, 512, 3) and uncond is None
    assert latent.shape == (1, 4, 64, 64) and torch.isfinite(latent).all()
    p2p.tokenizer, p2p.device, p2p.NUM_DDIM_STEPS = rev.tokenizer, "cuda", 4
    ctrl = p2p.make_controller([src, dst], True, 0.5, 0.5, blend_words=(("cat",), ("dog",)),
                               equilizer_params={"words": ("dog",), "values": (2.0,)})
    images, _ = generation.runner(model=rev, prompt=[src, dst], controller=ctrl, solver=solver, is_cons_forward=True,
                                  latent=latent, guidance_scale=19.0, tau1=0.8, tau2=0.8, w_embed_dim=512)
    p2p.device = "cpu"
    assert images.shape == (2, 512, 512, 3) and images.dtype == np.uint8
    assert ctrl.cur_step == 4 and isinstance(ctrl, p2p.AttentionReweight)
    assert not np.array_equal(images[0], images[1])          # the edit prompt changed the second image
    # NPI branch of invert: the conditional embedding repeated n_steps times (utils/inversion.py:101-102)
    _, _, npi = inversion.invert(solver, stop_step=50, is_cons_inversion=True, inv_guidance_scale=0.0, w_embed_dim=512,
                                 image_path=path, prompt=src, do_npi=True, seed=1)
    assert len(npi) == solver.n_steps and npi[0].shape == (1, 77, 768)
    with pytest.raises(NotImplementedError):
        inversion.invert(solver, stop_step=50, is_cons_inversion=True, w_embed_dim=512, image_path=path, prompt=src, do_nti=True)


def test_sdxl_prompts_to_pil_images():
    """SDXL flow of running/sdxl/generate.py at reduced width: two CLIP encoders (clip.py, the second with projection) ->
    compute_embeddings (penultimate hidden states concatenated, pooled text_embeds, time_ids) -> sample_deterministic
    (4 steps) -> AutoencoderKL decode (the reference upcasts the VAE: `.to(float32)`) -> PIL images."""
    from invertible_cd_amd import clip, generation_sdxl, synthetic, unet, vae
    from invertible_cd_amd.pipelines import StableDiffusionXLPipeline
    from invertible_cd_amd.schedulers import DDIMScheduler
    from invertible_cd_amd.unet_config import SDXL
    cfg = SDXL.scaled((64, 128, 256), cross_dim=128, heads=(2, 4, 8))
    pooled_dim = cfg.add_in_dim - 6 * cfg.addition_time_embed_dim
    c1 = clip.CLIPTextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1)
    c2 = clip.CLIPTextConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=1,
                             hidden_act="gelu", projection_dim=pooled_dim)
    enc1 = clip.CLIPTextModel(c1, synthetic.synthetic_clip_state_dict(c1, seed=1))
    enc2 = clip.CLIPTextModel(c2, synthetic.synthetic_clip_state_dict(c2, True, seed=2), with_projection=True)
    tok = synthetic.SyntheticTokenizer()
    vcfg = vae.SDXL_VAE.scaled((32, 64, 64, 64))
    v = vae.AutoencoderKL(vcfg, synthetic.synthetic_vae_state_dict(vcfg, seed=3))
    u = unet.UNet2DConditionModel(cfg, synthetic.synthetic_state_dict(cfg, seed=4), dtype=torch.float16)
    pipe = StableDiffusionXLPipeline(u, DDIMScheduler.sdxl(), vae=v, tokenizer=tok, tokenizer_2=tok, text_encoder=enc1,
                                     text_encoder_2=enc2)
    prompts = ["a photo of a cat", "a dog on a bench"]
    emb_fn = lambda p, o, c: generation_sdxl.compute_embeddings(p, o, c, 0, [enc1, enc2], [tok, tok], is_train=False)
    e = emb_fn(prompts, [(1024, 1024)] * 2, [(0, 0)] * 2)
    assert e["prompt_embeds"].shape == (2, 77, 128) and e["text_embeds"].shape == (2, pooled_dim) and e["time_ids"].shape == (2, 6)
    # pipe.encode_prompt - the samplers' compute_embeddings_fn=None branch (utils/generation_sdxl.py:242,374) - is served by the same
    # encoders: [0] is the concatenated penultimate hidden states, [2] the pooled output of the second encoder
    pe, ne, pooled, npooled = pipe.encode_prompt(prompts, "cuda", 1, False)
    assert torch.equal(pe, e["prompt_embeds"].to(pe.dtype)) and torch.equal(pooled, e["text_embeds"]) and ne is None and npooled is None
    pe2, ne2, _, np2 = pipe.encode_prompt(prompts[0], "cuda", 2, True)
    assert pe2.shape == (2, 77, 128) and torch.equal(pe2[0], pe[0]) and torch.equal(pe2[1], pe[0])
    assert ne2.shape == pe2.shape and np2.shape == (2, pooled_dim) and not torch.equal(ne2, pe2)
    bare = StableDiffusionXLPipeline(u, DDIMScheduler.sdxl())
    with pytest.raises(RuntimeError, match="text_encoder"):
        bare.encode_prompt(prompts, "cuda", 1, False)
    lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(0)).cuda().half()
    images, latents = generation_sdxl.sample_deterministic(pipe, prompts, latents=lat, num_inference_steps=4, guidance_scale=7.0,
                                                           is_sdxl=True, timesteps=[249, 499, 699, 999],
                                                           compute_embeddings_fn=emb_fn, return_latent=True)
    assert len(images) == 2 and images[0].size == (128, 128) and latents.shape == (2, 4, 16, 16)
    assert v.dtype == torch.float32                          # the reference's upcast request reached the VAE object
    assert np.asarray(images[0]).std() > 0 and not np.array_equal(np.asarray(images[0]), np.asarray(images[1]))
