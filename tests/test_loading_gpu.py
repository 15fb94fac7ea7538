"""On-disk checkpoint round trip of the loaders (utils/loading.py:10-23,39-90,100-147): a diffusers-layout directory
(`unet/diffusion_pytorch_model.safetensors`), a torch-saved teacher state dict with `time_embedding.cond_proj.weight`, and
peft-keyed LoRA `.safetensors` (`unet.base_model.model.<path>.lora_{A,B}.weight`) are written to tmp_path, loaded through
`load_models` / `load_models_xl`, and the resulting native UNets are compared with the oracle running the oracle-fused
weights (`oracle.unet_ref.fuse_lora`, W' = W + (8 / r) * up @ down).  Reduced-width architectures (`unet_config=`)."""
import os

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _write_checkpoints(tmp_path, cfg, with_cond_in_base):
    from safetensors.torch import save_file
    from invertible_cd_amd import synthetic
    full = synthetic.synthetic_state_dict(cfg, seed=21)
    full = {k: v.half().float() for k, v in full.items()}
    base = dict(full)
    if not with_cond_in_base:
        base.pop("time_embedding.cond_proj.weight")                 # the stock SD UNet has no w-embedding projection
    root = tmp_path / "model"
    os.makedirs(root / "unet")
    save_file({k: v.half().contiguous() for k, v in base.items()}, str(root / "unet" / "diffusion_pytorch_model.safetensors"))
    teacher = tmp_path / "teacher.pt"
    torch.save({k: v.half() for k, v in full.items()}, str(teacher))
    paths = {}
    loras = {}
    for name, seed in (("reverse", 31), ("forward", 32)):
        lora = synthetic.synthetic_lora(cfg, seed=seed, rank=16)
        lora = {p: (d.half().float(), u.half().float()) for p, (d, u) in lora.items()}
        peft = {}
        for p, (down, up) in lora.items():
            peft[f"unet.base_model.model.{p}.lora_A.weight"] = down.half().contiguous()
            peft[f"unet.base_model.model.{p}.lora_B.weight"] = up.half().contiguous()
        f = tmp_path / f"{name}.safetensors"
        save_file(peft, str(f))
        paths[name], loras[name] = str(f), lora
    return str(root), str(teacher), paths, loras, full


def _check(model_unet, cfg, weights, seed, xl):
    from invertible_cd_amd import synthetic
    from oracle import unet_ref
    from test_unet_gpu import _oracle_cfg
    inp = synthetic.synthetic_inputs(cfg, 2, 16, 16, seed=seed)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    cond = torch.randn(2, cfg.time_cond_proj_dim, generator=torch.Generator().manual_seed(seed)).half().float()
    added = {"text_embeds": inp["text_embeds"].half().float(), "time_ids": inp["time_ids"]} if xl else None
    ref = unet_ref.unet_forward(weights, _oracle_cfg(unet_ref, cfg), lat, 519, ctx, timestep_cond=cond, added_cond=added)
    got = model_unet(lat.cuda().half(), 519, encoder_hidden_states=ctx.cuda().half(), timestep_cond=cond.cuda().half(),
                     added_cond_kwargs=None if added is None else {k: v.cuda() for k, v in added.items()}).sample
    w16 = {k: v.cuda().half() for k, v in weights.items()}          # fp16-torch noise floor on the same weights
    flo = unet_ref.unet_forward(w16, _oracle_cfg(unet_ref, cfg), lat, 519, ctx, timestep_cond=cond, added_cond=added)
    return rel_l2(got, ref), rel_l2(flo.float().cpu(), ref)


def test_load_models_from_disk_matches_oracle_fused_weights(tmp_path):
    from invertible_cd_amd.loading import load_models
    from invertible_cd_amd.unet_config import SD15
    from oracle import unet_ref
    cfg = SD15.scaled((64, 64, 128, 128), cross_dim=64, heads=(2, 2, 4, 4))
    root, teacher, paths, loras, full = _write_checkpoints(tmp_path, cfg, with_cond_in_base=False)
    ldm, rev, fwd = load_models(root, "cuda", paths["reverse"], paths["forward"], r=16, w_embed_dim=512, teacher_checkpoint=teacher,
                                dtype="fp16", unet_config=cfg)
    errs = {"teacher": _check(ldm.unet, cfg, full, 1, False)}
    for name, pipe in (("reverse", rev), ("forward", fwd)):
        # the loader fuses fp16-rounded LoRA factors in fp32 and the executor stores fp16 weights: the oracle gets the same
        fused = {k: v.half().float() for k, v in unet_ref.fuse_lora(full, loras[name]).items()}
        errs[name] = _check(pipe.unet, cfg, fused, 2, False)
        errs[name + "_vs_unfused"] = _check(pipe.unet, cfg, full, 2, False)
    print("[loader sd15] (error, fp16-torch floor):", {k: f"{e:.3e}/{f:.3e}" for k, (e, f) in errs.items()})
    for k in ("teacher", "reverse", "forward"):
        assert errs[k][0] <= 1.5 * errs[k][1] + 1e-4 and errs[k][0] < 1e-3, (k, errs[k])       # the north star's bar on ONE evaluation (round 6:
        # the 'auto' policy probes the checkpoint - unet._probe_plain_level - and runs plain generation at the accurate level where the fast one is too far off)
    assert errs["reverse_vs_unfused"][0] > 1e-2 and errs["forward_vs_unfused"][0] > 1e-2     # the LoRA really changed the function


def test_load_models_xl_from_disk_matches_oracle_fused_weights(tmp_path):
    from invertible_cd_amd.loading import load_models_xl
    from invertible_cd_amd.pipelines import StableDiffusionXLImg2ImgPipeline
    from invertible_cd_amd.unet_config import SDXL
    from oracle import unet_ref
    cfg = SDXL.scaled((64, 128, 128), cross_dim=64, heads=(2, 4, 4))
    root, teacher, paths, loras, full = _write_checkpoints(tmp_path, cfg, with_cond_in_base=True)
    stable, pipe, forw = load_models_xl(root, paths["reverse"], paths["forward"], teacher, unet_config=cfg)
    assert isinstance(forw, StableDiffusionXLImg2ImgPipeline) and isinstance(stable, StableDiffusionXLImg2ImgPipeline)
    fused_r = {k: v.half().float() for k, v in unet_ref.fuse_lora(full, loras["reverse"]).items()}
    fused_f = {k: v.half().float() for k, v in unet_ref.fuse_lora(full, loras["forward"]).items()}
    res = {"teacher": _check(stable.unet, cfg, full, 3, True), "reverse": _check(pipe.unet, cfg, fused_r, 4, True),
           "forward": _check(forw.unet, cfg, fused_f, 4, True), "reverse_vs_unfused": _check(pipe.unet, cfg, full, 4, True)}
    print("[loader sdxl] (error, fp16-torch floor):", {k: f"{e:.3e}/{f:.3e}" for k, (e, f) in res.items()})
    print("[loader sdxl] precision probe (gap, plain level):", {n: (f"{m.unet._auto_gap:.3e}", m.unet._auto_plain) for n, m in (("teacher", stable), ("reverse", pipe), ("forward", forw))})
    for k in ("teacher", "reverse", "forward"):
        assert res[k][0] <= 1.5 * res[k][1] + 1e-4 and res[k][0] < 1e-3, (k, res[k])         # round 6 (was 1.2e-3: the fused LoRA widens the activations,
        # 1.04e-3 at the fast level - the probe of the 'auto' policy now sees that on the checkpoint itself and escalates)
    assert res["reverse_vs_unfused"][0] > 1e-2


def test_missing_files_and_bad_keys_raise(tmp_path):
    from safetensors.torch import save_file
    from invertible_cd_amd.loading import load_models, parse_peft_lora
    from invertible_cd_amd.unet_config import SD15
    cfg = SD15.scaled((64, 64, 128, 128), cross_dim=64, heads=(2, 2, 4, 4))
    with pytest.raises(FileNotFoundError):
        load_models(str(tmp_path / "nope"), "cuda", None, None, w_embed_dim=512, dtype="fp16", unet_config=cfg)
    with pytest.raises(KeyError):
        parse_peft_lora({"unet.something.else": torch.zeros(1)})
    with pytest.raises(KeyError):
        parse_peft_lora({"unet.base_model.model.mid_block.attentions.0.proj_in.lora_A.weight": torch.zeros(4, 8)})
