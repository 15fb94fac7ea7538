"""p2p controllers (invertible_cd_amd.p2p / seq_aligner) against vectors captured from the reference's utils/p2p.py and
utils/seq_aligner.py with the same whitespace stub tokenizer and the same seeded probability tensors.

Big tensors are compared through the fixture's strided subsample + float64 (sum, sum of squares) of the full tensor."""
import json
import os

import numpy as np
import pytest
import torch

from invertible_cd_amd import p2p, seq_aligner
from stubs import StubTokenizer

HEADS = 2


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "p2p.npz"))


@pytest.fixture(autouse=True)
def _globals():
    p2p.tokenizer = StubTokenizer()
    p2p.device = "cpu"
    p2p.NUM_DDIM_STEPS = 4
    p2p.LOW_RESOURCE = False
    yield


def _walk(g):
    return list(zip(g["walk_place"].tolist(), [bool(x) for x in g["walk_cross"]], g["walk_n"].tolist()))


def _probs(gen, bh, n, m):
    return torch.softmax(torch.randn(bh, n, m, generator=gen) * 2.0, dim=-1)


def _check(g, name, t):
    t = t.float()
    stats = g[name + "__stats"]
    assert float(t.double().sum()) == pytest.approx(stats[0], rel=1e-9), name
    assert float((t.double() ** 2).sum()) == pytest.approx(stats[1], rel=1e-9), name
    sr, sc = g[name + "__stride"]
    np.testing.assert_allclose(t[:, ::sr, ::sc].numpy(), g[name], rtol=1e-6, atol=1e-8, err_msg=name)


def _run(g, prefix, controller, n_prompts, seed, latents=None, steps=4, cond_only=False):
    walk = _walk(g)
    controller.num_att_layers = len(walk)
    gen = torch.Generator().manual_seed(seed)
    changed, lat_out, i = [], [], 0
    for step in range(steps):
        for place, is_cross, n in walk:
            m = 77 if is_cross else n
            P = _probs(gen, 2 * n_prompts * HEADS, n, m)
            before = P.clone()
            if cond_only:          # the executor's dead-uncond elimination: the controller gets the cond rows only
                half = P.shape[0] // 2
                R = controller.call_cond_only(P[half:], is_cross, place)
                assert R.data_ptr() == P[half:].data_ptr()
            else:
                R = controller(P, is_cross, place)
                assert R is P                                    # in-place contract of utils/p2p.py:101-113
            assert torch.equal(P[: P.shape[0] // 2], before[: P.shape[0] // 2])     # unconditional rows untouched
            ch = not torch.equal(P, before)
            changed.append(int(ch))
            if ch:
                _check(g, f"{prefix}_out{i}", P)
            i += 1
        if latents is not None:
            latents = controller.step_callback(latents)
            lat_out.append(latents.clone())
    assert changed == g[f"{prefix}_changed"].tolist()
    assert controller.cur_step == int(g[f"{prefix}_cur_step"])
    for key, lst in controller.attention_store.items():
        assert len(lst) == int(g[f"{prefix}_storelen_{key}"])
        for j, t in enumerate(lst):
            _check(g, f"{prefix}_store_{key}_{j}", t)
    return lat_out


@pytest.mark.parametrize("cond_only", [False, True])
def test_attention_store(g, cond_only):
    c = p2p.AttentionStore()
    _run(g, "store", c, 1, 100, cond_only=cond_only)
    _check(g, "store_avg_down_cross_0", c.get_average_attention()["down_cross"][0])
    assert not c.needs_probs(False, 4096, "down") and c.needs_probs(True, 1024, "up")
    # stored tensors are VIEWS of the probability buffers handed to the controller (no clone)
    c2 = p2p.AttentionStore(); c2.num_att_layers = 1
    P = torch.rand(4, 64, 77)
    c2(P, True, "mid")
    assert c2.attention_store["mid_cross"][0].data_ptr() == P[2:].data_ptr()


PROMPTS = ["a cat sitting on a bench", "a dog sitting on a bench"]
PROMPTS_R = ["a cat sitting on a bench", "a fluffy cat sitting on a red bench"]


@pytest.mark.parametrize("cond_only", [False, True])
def test_attention_replace(g, cond_only):
    c = p2p.make_controller(PROMPTS, True, 0.5, 0.5)
    assert isinstance(c, p2p.AttentionReplace)
    assert torch.equal(c.cross_replace_alpha, torch.from_numpy(g["replace_alpha"]))
    assert torch.equal(c.mapper, torch.from_numpy(g["replace_mapper"]))
    assert list(c.num_self_replace) == g["replace_num_self"].tolist()
    _run(g, "replace", c, 2, 101, cond_only=cond_only)


def test_attention_refine_with_local_blend(g):
    c = p2p.make_controller(PROMPTS_R, False, {"default_": 0.8, "fluffy": (0.0, 0.4)}, 0.4, blend_words=(("cat",), ("cat",)))
    assert isinstance(c, p2p.AttentionRefine) and c.local_blend is not None
    assert torch.equal(c.cross_replace_alpha, torch.from_numpy(g["refine_alpha"]))
    assert torch.equal(c.mapper, torch.from_numpy(g["refine_mapper"])) and torch.equal(c.alphas, torch.from_numpy(g["refine_alphas"]))
    assert torch.equal(c.local_blend.alpha_layers, torch.from_numpy(g["refine_lb_alpha_layers"]))
    lat = _run(g, "refine", c, 2, 102, latents=torch.from_numpy(g["refine_lat_in"]))
    np.testing.assert_allclose(torch.stack(lat).numpy(), g["refine_latents"], rtol=1e-6, atol=1e-7)
    assert torch.equal(lat[-1][0], torch.from_numpy(g["refine_lat_in"])[0])         # base prompt row never blended


def test_attention_reweight_chained(g):
    c = p2p.make_controller(PROMPTS, True, 0.6, 0.2, equilizer_params={"words": ("dog",), "values": (3.0,)})
    assert isinstance(c, p2p.AttentionReweight) and isinstance(c.prev_controller, p2p.AttentionReplace)
    assert torch.equal(c.equalizer, torch.from_numpy(g["reweight_equalizer"]))
    _run(g, "reweight", c, 2, 103)


def test_three_prompt_group(g):
    c = p2p.make_controller(["a cat sitting on a bench", "a dog sitting on a bench", "a cat sitting on a sofa"], True, 0.5, 0.25)
    _run(g, "replace3", c, 3, 104, steps=2)


def test_helpers(g):
    tok = p2p.tokenizer
    assert p2p.get_word_inds(PROMPTS[0], "cat", tok).tolist() == g["word_inds_cat"].tolist()
    assert p2p.get_word_inds(PROMPTS[0], 2, tok).tolist() == g["word_inds_2"].tolist()
    assert torch.equal(p2p.get_equalizer(PROMPTS_R[1], ("fluffy", "red"), (2.0, 0.5)), torch.from_numpy(g["equalizer_multi"]))
    a = p2p.get_time_words_attention_alpha(PROMPTS_R, 4, {"default_": (0.1, 0.9), "red": 0.3}, tok)
    assert torch.equal(a, torch.from_numpy(g["time_words_alpha"]))
    sr = p2p.SpatialReplace(0.5)
    sr.cur_step = 0
    assert sr.stop_inject == int(g["spatial_stop"])
    assert torch.equal(sr.step_callback(torch.from_numpy(g["spatial_in"])).contiguous(), torch.from_numpy(g["spatial_out0"]))
    e = p2p.EmptyControl()
    x = torch.rand(2, 3)
    assert e(x, True, "up") is x and e.step_callback(x) is x and not e.needs_probs(True, 64, "mid")


def test_register_attention_control_counts_layers():
    class FakeUNet:
        num_attention_layers = 32
        attn_controller = "stale"
    model = type("M", (), {"unet": FakeUNet()})()
    c = p2p.AttentionStore()
    p2p.register_attention_control(model, c)
    assert c.num_att_layers == 32 and model.unet.attn_controller is c
    p2p.register_attention_control(model, None)
    assert model.unet.attn_controller is None


def test_hook_adapter_tick_vs_materialise():
    c = p2p.AttentionStore()
    c.num_att_layers = 3
    ad = p2p.HookAdapter(c, cond_only=True, dev="cpu")
    assert ad.query(0, False, "down", 4, 4096, 4096, 4096) is None and c.cur_att_layer == 1      # fused + tick
    buf = ad.query(1, True, "down", 4, 256, 77, 80)
    assert buf.shape == (4, 256, 80) and c.cur_att_layer == 1
    buf.zero_(); buf[:, :, :77] = 1 / 77
    ad.probs_ready(1, True, "down")
    assert c.cur_att_layer == 2 and c.step_store["down_cross"][0].shape == (4, 256, 77)
    assert c.step_store["down_cross"][0].data_ptr() == buf.data_ptr()
    b2 = ad.query(2, False, "mid", 4, 64, 64, 64)
    ad.probs_ready(2, False, "mid")
    assert c.cur_step == 1 and c.cur_att_layer == 0 and len(c.attention_store["mid_self"]) == 1   # between_steps fired
    # a foreign callable always gets the probabilities
    seen = []
    ad2 = p2p.HookAdapter(lambda p, is_cross, place: seen.append((tuple(p.shape), is_cross, place)) or p, False, "cpu")
    assert ad2.query(0, False, "up", 2, 4096, 4096, 4096) is not None
    ad2.probs_ready(0, False, "up")
    assert seen == [((2, 4096, 4096), False, "up")]


def test_fused_epilogue_is_planned_only_when_the_samples_are_the_controllers_prompts():
    """The probability kernel's epilogue indexes the edit operators by SAMPLE; the controller (like utils/p2p.py:192-194) groups rows by
    heads = rows / batch_size.  The two agree only when the conditional samples of the call are exactly the controller's prompts: a
    2-prompt controller on a 4-latent cond-only batch (or an unknown batch) must fall back to the separate passes."""
    prompts = ["a cat on a bench", "a dog on a bench"]
    c = p2p.AttentionReplace(prompts, 4, cross_replace_steps=0.8, self_replace_steps=0.6)
    heads = 2
    for batch, cond_only, want in [(2, True, True), (4, True, False), (4, False, True), (8, False, False), (0, True, False), (3, True, False)]:
        ad = p2p.HookAdapter(c, cond_only=cond_only, dev="cpu", batch=batch)
        bh = max(batch, 2) * heads
        epi = ad._plan_epilogue(True, "down", bh, 256, 77, 80)
        assert (epi is not None and bool(epi.edit_At)) == want, (batch, cond_only)
        if want:
            assert epi.edit_count == 1 and epi.first_cond_row == (0 if cond_only else bh // 2)
        epi = ad._plan_epilogue(False, "down", bh, 256, 256, 256)                       # self-attention replacement (step 0 is inside its window)
        assert (epi is not None and epi.self_from_base == 1) == want, (batch, cond_only)


def test_seq_aligner(golden_dir):
    g = np.load(os.path.join(golden_dir, "seq_aligner.npz"))
    pr = json.load(open(os.path.join(golden_dir, "seq_aligner_prompts.json")))
    tok = StubTokenizer()
    for i, (a, b) in enumerate(pr["refine"]):
        m, al = seq_aligner.get_refinement_mapper([a, b], tok)
        assert torch.equal(m, torch.from_numpy(g[f"refine_mapper_{i}"])) and torch.equal(al, torch.from_numpy(g[f"refine_alphas_{i}"]))
    for i, (a, b) in enumerate(pr["replace"]):
        assert torch.equal(seq_aligner.get_replacement_mapper([a, b], tok), torch.from_numpy(g[f"replace_mapper_{i}"]))
    m3 = seq_aligner.get_replacement_mapper(["a cat on a bench", "a dog on a bench", "a cat on a sofa"], tok)
    assert torch.equal(m3, torch.from_numpy(g["replace_mapper_3prompts"]))
    assert str(g["replace_unequal_error"]) == "ValueError"
    with pytest.raises(ValueError, match="same length"):
        seq_aligner.get_replacement_mapper(["a cat", "a big cat"], tok)
