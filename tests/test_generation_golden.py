"""Host logic of the sampler (invertible_cd_amd.generation / generation_sdxl) against vectors captured from the
reference's utils/generation.py / utils/generation_sdxl.py driven with the same closed-form stub UNet: identical call
order, timesteps, w-embeddings and BIT-IDENTICAL latents (all fp32 CPU arithmetic here)."""
import json
import os

import numpy as np
import pytest
import torch

from invertible_cd_amd import generation as G
from invertible_cd_amd import generation_sdxl as X
from stubs import HalfUNet, StubModel, StubPipe, StubScheduler, sdxl_emb_fn


@pytest.fixture(scope="module")
def g15(golden_dir):
    return np.load(os.path.join(golden_dir, "sd15_loops.npz"))


@pytest.fixture(scope="module")
def gxl(golden_dir):
    return np.load(os.path.join(golden_dir, "sdxl_loops.npz"))


def _solver(model=None):
    m = model or StubModel()
    s = G.Generator(m, 50, StubScheduler(), forward_cons_model=m, reverse_cons_model=m,
                    reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    return m, s


def test_generator_tables_and_caller_list_mutation(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "timesteps.json")))
    for name in ("set1", "set2", "set3"):
        e = g[name]
        rev, fwd = list(e["reverse_in"]), list(e["forward_in"])
        s = G.Generator(StubModel(), 50, StubScheduler(), reverse_timesteps=rev, forward_timesteps=fwd)
        assert s.reverse_timesteps.tolist() == e["reverse_timesteps"]
        assert s.reverse_boundary_timesteps.tolist() == e["reverse_boundary"]
        assert s.forward_timesteps.tolist() == e["forward_timesteps"]
        assert s.forward_boundary_timesteps.tolist() == e["forward_boundary"]
        assert rev == e["caller_reverse_list_after"] and fwd == e["caller_forward_list_after"]   # in-place reverse quirk
    for ne in (1, 2, 3, 4):
        e = g[f"default_{ne}"]
        s = G.Generator(StubModel(), 50, StubScheduler(), num_endpoints=ne, num_forward_endpoints=ne)
        assert s.reverse_timesteps.tolist() == e["reverse_timesteps"]
        assert s.reverse_boundary_timesteps.tolist() == e["reverse_boundary"]
        assert s.forward_timesteps.tolist() == e["forward_timesteps"]
        assert s.forward_boundary_timesteps.tolist() == e["forward_boundary"]
        d = g[f"ddimsolver_{ne}"]
        sv = X.DDIMSolver(StubScheduler().alphas_cumprod.numpy(), num_endpoints=ne, num_inverse_endpoints=ne)
        assert sv.endpoints.tolist() == d["endpoints"] and sv.inverse_endpoints.tolist() == d["inverse_endpoints"]
    sv = X.DDIMSolver(StubScheduler().alphas_cumprod.numpy(), num_endpoints=4, num_inverse_endpoints=4,
                      endpoints="0,249,499,699", inverse_endpoints="249,499,699,999")
    assert sv.endpoints.tolist() == g["ddimsolver_explicit"]["endpoints"]


def test_helpers_bit_exact(golden_dir):
    w = np.load(os.path.join(golden_dir, "wembed.npz"))
    for dim, key in ((512, "emb512"), (256, "emb256"), (33, "emb33")):
        assert torch.equal(G.guidance_scale_embedding(torch.from_numpy(w["w"]), dim), torch.from_numpy(w[key]))
        assert torch.equal(X.guidance_scale_embedding(torch.from_numpy(w["w"]), dim), torch.from_numpy(w[key]))
    for row in json.load(open(os.path.join(golden_dir, "schedules.json"))):
        args = (row["t"], row["gs"], row["tau1"], row["tau2"])
        assert G.linear_schedule_old(*args) == row["old"] and X.linear_schedule_old(*args) == row["old_xl"]
        assert G.linear_schedule(*args) == row["new"]
    po = np.load(os.path.join(golden_dir, "predicted_origin.npz"))
    tab = np.load(os.path.join(golden_dir, "alphas_cumprod.npz"))
    alpha, sigma = torch.from_numpy(tab["alpha_table"]), torch.from_numpy(tab["sigma_table"])
    x, eps = torch.from_numpy(po["x"]), torch.from_numpy(po["eps"])
    for (t, s), ref in zip(po["pairs"].tolist(), po["out"]):
        for mod in (G, X):
            out = mod.predicted_origin(eps, torch.tensor([t, t]), torch.tensor([s, s]), x, "epsilon", alpha, sigma)
            assert torch.equal(out, torch.from_numpy(ref))
    with pytest.raises(ValueError, match="currently not supported"):
        G.predicted_origin(eps, torch.tensor([1, 1]), torch.tensor([0, 0]), x, "sample", alpha, sigma)


@pytest.mark.parametrize("B,gs,tau", [(3, 19.0, 0.8), (2, 19.0, 0.8), (2, 7.0, 1.0), (1, 7.0, 0.7)])
def test_cons_generation_matches_reference(g15, B, gs, tau):
    tag = f"rev_B{B}_gs{int(gs)}_tau{int(tau * 10)}"
    m, s = _solver()
    s.context = torch.zeros(2 * B, 77, 8)
    lat = torch.from_numpy(g15[tag + "_in"])
    outs = s.cons_generation(lat, guidance_scale=gs, w_embed_dim=512, dynamic_guidance=tau < 1.0, tau1=tau, tau2=tau)
    assert len(outs) == 5 and torch.equal(torch.stack(outs), torch.from_numpy(g15[tag + "_out"]))
    assert [c["t"] for c in m.unet.calls] == g15[tag + "_t"].tolist() == [999.0, 779.0, 519.0, 259.0]
    assert torch.equal(torch.stack([c["cond"] for c in m.unet.calls]), torch.from_numpy(g15[tag + "_cond"]))
    assert torch.equal(torch.stack([c["x"] for c in m.unet.calls]), torch.from_numpy(g15[tag + "_x"]))   # CFG-doubled input


def test_classic_cfg_branch(g15):
    m, s = _solver()
    m.unet = HalfUNet()
    s.context = torch.zeros(4, 77, 8)
    outs = s.cons_generation(torch.from_numpy(g15["cfg_in"]), guidance_scale=7.5, w_embed_dim=0, dynamic_guidance=True,
                             tau1=0.4, tau2=0.8)
    assert torch.equal(torch.stack(outs), torch.from_numpy(g15["cfg_out"]))


def test_cons_inversion_matches_reference(g15):
    m, s = _solver()
    s.context = torch.zeros(4, 77, 8)
    s.latent2image = lambda z, return_type="np": np.zeros((1,))
    _, out = s.cons_inversion(torch.from_numpy(g15["inv_in"]), guidance_scale=0.0, w_embed_dim=512, seed=5)
    assert torch.equal(out[0], torch.from_numpy(g15["inv_out"]))
    assert [c["t"] for c in m.unet.calls] == g15["inv_t"].tolist() == [19.0, 259.0, 519.0, 779.0]
    assert torch.equal(torch.stack([c["cond"] for c in m.unet.calls]), torch.from_numpy(g15["inv_cond"]))
    assert torch.equal(m.unet.calls[0]["x"], torch.from_numpy(g15["inv_x0"]))          # add_noise at t = 19 with seed 5


def test_init_latent_and_runner(g15):
    m, s = _solver()
    one, batch = G.init_latent(None, m, 64, 64, torch.Generator().manual_seed(11), 3)
    assert torch.equal(one, torch.from_numpy(g15["init_latent_one"])) and torch.equal(batch, torch.from_numpy(g15["init_latent_batch"]))
    assert batch.stride(0) == 0                                                          # ONE sample expanded to the batch
    ctx = torch.randn(6, 77, 8, generator=torch.Generator().manual_seed(3))

    def init_prompt(prompt, unc=None):
        s.context, s.prompt = ctx, prompt
    s.init_prompt = init_prompt
    img, lat = G.runner(model=m, prompt=["a", "b", "c"], controller=None, solver=s, is_cons_forward=True,
                        num_inference_steps=50, guidance_scale=19.0, generator=torch.Generator().manual_seed(21), latent=None,
                        return_type="latent", dynamic_guidance=False, tau1=0.8, tau2=0.8, w_embed_dim=512)
    assert torch.equal(lat, torch.from_numpy(g15["runner_latent"])) and torch.equal(img, torch.from_numpy(g15["runner_out"]))
    assert [c["t"] for c in m.unet.calls] == g15["runner_t"].tolist()
    assert torch.equal(torch.stack([c["cond"][:, 0] for c in m.unet.calls]), torch.from_numpy(g15["runner_w_first_col"]))
    assert m.scheduler.num_inference_steps == 50


def test_sdxl_reverse_static(gxl):
    p = StubPipe()
    ts = [249, 499, 699, 999]
    img, lat = X.sample_deterministic(p, ["aa", "bb", "cc"], num_inference_steps=4, generator=torch.Generator().manual_seed(0),
                                      guidance_scale=7.0, is_sdxl=True, timesteps=ts, compute_embeddings_fn=sdxl_emb_fn,
                                      return_latent=True)
    assert ts == [249, 499, 699, 999]
    assert torch.equal(lat, torch.from_numpy(gxl["rev_B3_out"]))
    assert [c["t"] for c in p.unet.calls] == gxl["rev_B3_t"].tolist() == [999.0, 699.0, 499.0, 249.0]
    assert torch.equal(torch.stack([c["x"] for c in p.unet.calls]), torch.from_numpy(gxl["rev_B3_x"]))
    assert torch.equal(torch.stack([c["cond"] for c in p.unet.calls]), torch.from_numpy(gxl["rev_B3_cond"]))


def test_sdxl_reverse_dynamic_and_default_timesteps(gxl):
    p = StubPipe()
    img, lat = X.sample_deterministic(p, ["aa"], num_inference_steps=3, generator=torch.Generator().manual_seed(1),
                                      guidance_scale=19.0, is_sdxl=True, timesteps=[339, 699, 999],
                                      compute_embeddings_fn=sdxl_emb_fn, return_latent=True, use_dynamic_guidance=True, tau1=0.7,
                                      tau2=0.7, amplify_prompt=["zzzz"])
    assert torch.equal(lat, torch.from_numpy(gxl["dyn_B1_out"]))
    assert torch.equal(torch.stack([c["cond"] for c in p.unet.calls]), torch.from_numpy(gxl["dyn_B1_cond"]))
    assert str(gxl["dyn_B2_error"]) != ""           # the reference raises for B > 1 ...
    p2 = StubPipe()                                  # ... the product applies the same scalar rule to every sample
    X.sample_deterministic(p2, ["aa", "bb"], num_inference_steps=3, generator=torch.Generator().manual_seed(1), guidance_scale=19.0,
                           is_sdxl=True, timesteps=[339, 699, 999], compute_embeddings_fn=sdxl_emb_fn, use_dynamic_guidance=True,
                           tau1=0.7, tau2=0.7)
    for c2, c1 in zip(p2.unet.calls, p.unet.calls):
        assert torch.equal(c2["cond"][0], c1["cond"][0]) and torch.equal(c2["cond"][1], c1["cond"][0])
    p = StubPipe()
    img, lat = X.sample_deterministic(p, ["aa", "bb"], num_inference_steps=4, generator=torch.Generator().manual_seed(2),
                                      guidance_scale=7.0, is_sdxl=True, timesteps=None, compute_embeddings_fn=sdxl_emb_fn,
                                      return_latent=True)
    assert torch.equal(lat, torch.from_numpy(gxl["revdef_out"])) and [c["t"] for c in p.unet.calls] == gxl["revdef_t"].tolist()


def test_sdxl_forward(gxl):
    p = StubPipe()
    lat, start = X.inverse_sample_deterministic(p, torch.from_numpy(gxl["fwd_in"]), ["aa", "bb"], num_inference_steps=4,
                                                timesteps=[19, 249, 499, 699], guidance_scale=0.0, is_sdxl=True,
                                                compute_embeddings_fn=sdxl_emb_fn, seed=3, return_start_latent=True)
    assert torch.equal(start, torch.from_numpy(gxl["fwd_start"])) and torch.equal(lat, torch.from_numpy(gxl["fwd_out"]))
    assert [c["t"] for c in p.unet.calls] == gxl["fwd_t"].tolist() == [19.0, 249.0, 499.0, 699.0]
    assert torch.equal(torch.stack([c["cond"] for c in p.unet.calls]), torch.from_numpy(gxl["fwd_cond"]))
