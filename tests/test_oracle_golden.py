"""Pins the CPU oracle (oracle/sched_ref.py, oracle/unet_ref.py) against vectors captured from the reference itself
(tests/golden/make_golden.py imports /root/reference in the build container)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sched_ref as S
from oracle import unet_ref as U


def _load(golden_dir, name):
    p = os.path.join(golden_dir, name)
    return json.load(open(p)) if name.endswith(".json") else np.load(p)


def test_alphas_cumprod_matches_reference_schedule(golden_dir):
    g = _load(golden_dir, "alphas_cumprod.npz")
    ac = S.alphas_cumprod()
    # cumprod association differs between numpy and torch in the last bits: compare to 1e-6 relative
    np.testing.assert_allclose(ac, g["alphas_cumprod"], rtol=2e-6, atol=0)
    probe = _load(golden_dir, "timesteps.json")["alphas_cumprod_probe"]
    for k, v in probe.items():
        assert abs(ac[int(k)] - v) <= 2e-6 * v


def test_timestep_tables(golden_dir):
    g = _load(golden_dir, "timesteps.json")
    for name in ("set1", "set2", "set3"):
        e = g[name]
        rt, rb, ft, fb = S.generator_tables(list(e["reverse_in"]), list(e["forward_in"]))
        assert rt.tolist() == e["reverse_timesteps"] and rb.tolist() == e["reverse_boundary"]
        assert ft.tolist() == e["forward_timesteps"] and fb.tolist() == e["forward_boundary"]
    for ne in (1, 2, 3, 4):
        e = g[f"default_{ne}"]
        rt, rb, ft, fb = S.generator_tables(num_endpoints=ne, num_forward_endpoints=ne)
        assert rt.tolist() == e["reverse_timesteps"] and rb.tolist() == e["reverse_boundary"]
        assert ft.tolist() == e["forward_timesteps"] and fb.tolist() == e["forward_boundary"]
        d = g[f"ddimsolver_{ne}"]
        ep, iep = S.default_endpoints(ne)
        assert ep.tolist() == d["endpoints"] and iep.tolist() == d["inverse_endpoints"]
        assert S.ddim_timesteps().tolist() == d["ddim_timesteps"]
    # released set 1 (README.md:57,60)
    assert g["set1"]["reverse_timesteps"] == [999, 779, 519, 259] and g["set1"]["reverse_boundary"] == [779, 519, 259, 0]
    assert g["set1"]["forward_timesteps"] == [19, 259, 519, 779] and g["set1"]["forward_boundary"] == [259, 519, 779, 999]
    assert S.sdxl_reverse_tables([249, 499, 699, 999])[1].tolist() == [699, 499, 249, 0]
    assert S.sdxl_forward_tables([19, 249, 499, 699])[1].tolist() == [249, 499, 699, 999]


def test_guidance_scale_embedding(golden_dir):
    g = _load(golden_dir, "wembed.npz")
    for dim, key in ((512, "emb512"), (256, "emb256"), (33, "emb33")):
        out = S.guidance_scale_embedding(g["w"], dim)
        assert out.shape == g[key].shape
        # sin/cos of arguments up to 1.9e4 rad: libm vs torch differ by a few fp32 ulps of the ARGUMENT
        np.testing.assert_allclose(out, g[key], rtol=0, atol=3e-3)
    np.testing.assert_array_equal(g["emb512"], g["emb512_xl"])       # the SDXL duplicate is the same function


def test_linear_schedules(golden_dir):
    for row in _load(golden_dir, "schedules.json"):
        assert S.linear_schedule_old(row["t"], row["gs"], row["tau1"], row["tau2"]) == pytest.approx(row["old"], abs=1e-12)
        assert row["old"] == row["old_xl"]
        assert S.linear_schedule(row["t"], row["gs"], row["tau1"], row["tau2"]) == pytest.approx(row["new"], abs=1e-12)
    # tau1 == tau2 is a step function: gs at or below tau, 0 above
    assert S.linear_schedule_old(999, 19.0, 0.8, 0.8) == 0.0 and S.linear_schedule_old(779, 19.0, 0.8, 0.8) == 19.0


def test_predicted_origin_bit_exact(golden_dir):
    g = _load(golden_dir, "predicted_origin.npz")
    tab = _load(golden_dir, "alphas_cumprod.npz")
    alpha, sigma = tab["alpha_table"], tab["sigma_table"]
    for (t, s), ref, ref_xl in zip(g["pairs"].tolist(), g["out"], g["out_xl"]):
        out = S.predicted_origin(g["eps"], [t, t], [s, s], g["x"], alpha, sigma)
        np.testing.assert_array_equal(out, ref)
        np.testing.assert_array_equal(ref, ref_xl)
    mixed = S.predicted_origin(g["eps"], g["mixed_t"], g["mixed_s"], g["x"], alpha, sigma)
    np.testing.assert_array_equal(mixed, g["mixed"])
    v = S.predicted_origin(g["eps"], [999, 999], [0, 0], g["x"], alpha, sigma, prediction_type="v_prediction")
    np.testing.assert_array_equal(v, g["vpred"])
    with pytest.raises(ValueError):
        S.predicted_origin(g["eps"], [1, 1], [0, 0], g["x"], alpha, sigma, prediction_type="sample")


def test_w_vector_rule_and_sharding(golden_dir):
    g = _load(golden_dir, "sd15_loops.npz")
    # utils/generation.py:232-235 as observed through the recorded timestep_cond of the reference run
    emb0, emb19 = S.guidance_scale_embedding([0.0], 512)[0], S.guidance_scale_embedding([19.0], 512)[0]
    cond = g["rev_B2_gs19_tau8_cond"]          # [step, 2B = 4, 512]; step 0 is t = 999 (> tau -> w = 0 everywhere)
    np.testing.assert_allclose(cond[1, 3], emb19, atol=3e-3)
    np.testing.assert_allclose(cond[1, 0], emb0, atol=3e-3)       # 2B == 4 -> [0, 0, 0, gs]
    assert S.w_vector_sd15(4, 19.0).tolist() == [0.0, 0.0, 0.0, 19.0] and S.w_vector_sd15(6, 7.0).tolist() == [7.0] * 6
    for e in _load(golden_dir, "sharding.json"):
        assert S.prepare_val_prompts(e["N"], e["bs"], e["W"], e["rank"]) == e["index"]


def test_unet_oracle_architecture_pins():
    """Exact parameter totals of the published UNets (+163 840 for the 512->320 cond_proj) and key layout."""
    assert U.count_params(U.SD15) == 859_520_964 + 163_840
    assert U.count_params(U.SDXL) == 2_567_463_684 + 163_840
    sh = U.param_shapes(U.SD15)
    assert sh["time_embedding.cond_proj.weight"] == (320, 512)
    assert sh["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    assert sh["up_blocks.1.resnets.0.conv1.weight"] == (1280, 2560, 3, 3)
    assert sh["up_blocks.3.resnets.2.conv_shortcut.weight"] == (320, 640, 1, 1)
    assert "down_blocks.3.attentions.0.norm.weight" not in sh and "up_blocks.0.attentions.0.norm.weight" not in sh
    xl = U.param_shapes(U.SDXL)
    assert xl["add_embedding.linear_1.weight"] == (1280, 2816)
    assert xl["down_blocks.2.attentions.1.transformer_blocks.9.ff.net.0.proj.weight"] == (10240, 1280)
    assert xl["mid_block.attentions.0.proj_in.weight"] == (1280, 1280)          # use_linear_projection
    assert "down_blocks.0.attentions.0.norm.weight" not in xl


def test_unet_oracle_runs_and_hook_order():
    cfg = U.tiny(U.SD15, (32, 32, 64, 64), cross_dim=16)
    torch.manual_seed(0)
    w = {k: torch.randn(s) * 0.05 if not k.endswith("norm.weight") and ".norm" not in k else torch.ones(s)
         for k, s in U.param_shapes(cfg).items()}
    calls = []

    def hook(p, is_cross, place):
        calls.append((place, is_cross, p.shape[1]))
        assert torch.allclose(p.sum(-1), torch.ones(p.shape[:2]), atol=1e-5)
        return p
    x = torch.randn(2, 4, 16, 16)
    eps = U.unet_forward(w, cfg, x, 999, torch.randn(2, 5, 16), timestep_cond=torch.randn(2, 512), hook=hook)
    assert eps.shape == x.shape and torch.isfinite(eps).all()
    assert len(calls) == 32                                        # SD1.5 topology: 16 self + 16 cross
    assert [c[0] for c in calls] == ["down"] * 12 + ["mid"] * 2 + ["up"] * 18
    assert [c[1] for c in calls] == [False, True] * 16
    assert [c[2] for c in calls[:12:2]] == [256, 256, 64, 64, 16, 16]


def test_fuse_lora_matches_definition():
    W = {"m.weight": torch.randn(6, 4, 3, 3), "l.weight": torch.randn(5, 7)}
    lora = {"m": (torch.randn(2, 4, 3, 3), torch.randn(6, 2, 1, 1)), "l": (torch.randn(3, 7), torch.randn(5, 3))}
    out = U.fuse_lora(W, lora, alpha=8.0)
    ref = W["l.weight"] + (8.0 / 3) * lora["l"][1] @ lora["l"][0]
    assert torch.allclose(out["l.weight"], ref, atol=1e-6)
    conv_delta = torch.einsum("or,rikl->oikl", lora["m"][1][:, :, 0, 0], lora["m"][0])
    assert torch.allclose(out["m.weight"], W["m.weight"] + 4.0 * conv_delta, atol=1e-5)


def test_vae_oracle_architecture_pins():
    """AutoencoderKL restatement: exact parameter count of the SD VAE, key layout shared with the product module."""
    import torch
    from oracle import vae_ref
    from invertible_cd_amd.vae import SD_VAE, SDXL_VAE
    assert vae_ref.count_params(vae_ref.SD_VAE) == 83_653_863
    shapes = vae_ref.param_shapes(vae_ref.SD_VAE)
    assert len(shapes) == 248
    assert {k: tuple(v) for k, v in SD_VAE.state_dict_shapes().items()} == {k: tuple(v) for k, v in shapes.items()}
    assert SDXL_VAE.scaling_factor == vae_ref.SDXL_VAE["scaling_factor"] == 0.13025
    assert shapes["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1)
    assert shapes["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in shapes
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in shapes
    # runs end to end at reduced width; encode halves the resolution three times, decode doubles it three times
    cfg = dict(vae_ref.SD_VAE, block_out_channels=(32, 32, 64, 64))
    g = torch.Generator().manual_seed(0)
    w = {k: torch.randn(s, generator=g) * 0.05 for k, s in vae_ref.param_shapes(cfg).items()}
    img = vae_ref.decode(w, cfg, torch.randn(1, 4, 4, 6, generator=g))
    assert img.shape == (1, 3, 32, 48)
    assert vae_ref.encode_mean(w, cfg, img).shape == (1, 4, 4, 6)
    # the asymmetric Downsample2D padding: bottom/right only
    x = torch.zeros(1, 32, 4, 4); x[0, 0, 0, 0] = 1.0
    wd = torch.zeros(32, 32, 3, 3); wd[0, 0, 0, 0] = 1.0
    y = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (0, 1, 0, 1)), wd, stride=2)
    assert y[0, 0, 0, 0] == 1.0 and y.shape[-1] == 2


def test_vae_host_algebra_folds_are_exact():
    """pack_vae_state_dict folds post_quant_conv into conv_in (bias on a ones channel), quant_conv into conv_out (mean rows)
    and the V bias into the output projection - checked against the oracle's unfused arithmetic in fp64 on CPU."""
    import torch
    from oracle import vae_ref
    from invertible_cd_amd import synthetic
    from invertible_cd_amd.vae import SD_VAE, pack_vae_state_dict
    F = torch.nn.functional
    cfg = SD_VAE.scaled((32, 32, 64, 64))
    sd = synthetic.synthetic_vae_state_dict(cfg, seed=11)
    P = pack_vae_state_dict(cfg, sd, device="cpu")
    g = torch.Generator().manual_seed(1)
    z = torch.randn(2, 4, 5, 7, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.conv2d(z, sd["post_quant_conv.weight"].double(), sd["post_quant_conv.bias"].double()),
                   sd["decoder.conv_in.weight"].double(), sd["decoder.conv_in.bias"].double(), padding=1)
    z8 = torch.zeros(2, 8, 5, 7, dtype=torch.float64); z8[:, :4] = z; z8[:, 4] = 1.0
    w8 = P["decoder.conv_in.weight"].double().reshape(-1, 3, 3, 8).permute(0, 3, 1, 2)
    got = F.conv2d(z8, w8, P["decoder.conv_in.bias"].double(), padding=1)
    assert float((got - ref).abs().max()) < 2e-3                       # only the fp16 rounding of the packed weights
    h = torch.randn(2, 64, 5, 7, generator=g, dtype=torch.float64)
    mom = F.conv2d(F.conv2d(h, sd["encoder.conv_out.weight"].double(), sd["encoder.conv_out.bias"].double(), padding=1),
                   sd["quant_conv.weight"].double(), sd["quant_conv.bias"].double())[:, :4]
    w4 = P["encoder.conv_out_mean.weight"].double().reshape(4, 3, 3, 64).permute(0, 3, 1, 2)
    assert float((F.conv2d(h, w4, P["encoder.conv_out_mean.bias"].double(), padding=1) - mom).abs().max()) < 2e-3
    a = "decoder.mid_block.attentions.0."
    fused = sd[a + "to_out.0.weight"].double() @ sd[a + "to_v.bias"].double() + sd[a + "to_out.0.bias"].double()
    assert float((P[a + "to_out.bias"].double() - fused).abs().max()) < 1e-6


def test_clip_layout_matches_transformers_classes():
    """The text-encoder key layout / parameter counts of the product module equal those of the transformers classes the
    reference instantiates (utils/loading.py:41,108-112): 123 060 480 (CLIP ViT-L) and 694 659 840 (OpenCLIP bigG + projection)."""
    import math
    from oracle import clip_ref
    from invertible_cd_amd import clip
    for cfg, proj, n in ((clip.CLIP_VIT_L, False, 123_060_480), (clip.OPENCLIP_BIGG, True, 694_659_840)):
        shapes = cfg.state_dict_shapes(proj)
        assert sum(math.prod(s) for s in shapes.values()) == n
        assert clip_ref.state_dict_keys(cfg.to_dict(), proj) == {k: tuple(v) for k, v in shapes.items()}
