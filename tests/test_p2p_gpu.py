"""p2p edit controllers on the GPU (fused in-place HIP kernel csrc/p2p.hip for the cross-attention edit, torch for the
self-attention broadcast / LocalBlend) against the vectors captured from the reference's utils/p2p.py - the same goldens the
CPU tests match bit for bit.  Probabilities are fp16 on the device, so comparisons use atol 3e-3 on values in [0, 1]."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from invertible_cd_amd import p2p
from stubs import StubTokenizer

pytestmark = pytest.mark.gpu
HEADS = 2


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "p2p.npz"))


@pytest.fixture(autouse=True)
def _globals():
    p2p.tokenizer = StubTokenizer()
    p2p.device = "cuda"
    p2p.NUM_DDIM_STEPS = 4
    p2p.LOW_RESOURCE = False
    yield
    p2p.device = "cpu"


def _check(g, name, t):
    t = t.float().cpu()
    stats = g[name + "__stats"]
    assert float(t.double().sum()) == pytest.approx(stats[0], rel=2e-3), name
    assert float((t.double() ** 2).sum()) == pytest.approx(stats[1], rel=4e-3), name
    sr, sc = g[name + "__stride"]
    np.testing.assert_allclose(t[:, ::sr, ::sc].numpy(), g[name], rtol=0, atol=3e-3, err_msg=name)


def _run(g, prefix, controller, n_prompts, seed, steps=4, cond_only=False, padded=False):
    walk = list(zip(g["walk_place"].tolist(), [bool(x) for x in g["walk_cross"]], g["walk_n"].tolist()))
    controller.num_att_layers = len(walk)
    gen = torch.Generator().manual_seed(seed)
    changed, i = [], 0
    for step in range(steps):
        for place, is_cross, n in walk:
            m = 77 if is_cross else n
            P32 = torch.softmax(torch.randn(2 * n_prompts * HEADS, n, m, generator=gen) * 2.0, dim=-1)
            if padded and is_cross:                              # the executor's layout: row stride 80, 77 live columns
                buf = torch.zeros(P32.shape[0], n, 80, dtype=torch.float16, device="cuda")
                buf[:, :, :77] = P32.half()
                P = buf[:, :, :77]
            else:
                P = P32.half().cuda()
            before = P.clone()
            half = P.shape[0] // 2
            if cond_only:
                R = controller.call_cond_only(P[half:], is_cross, place)
                assert R.data_ptr() == P[half:].data_ptr()
            else:
                R = controller(P, is_cross, place)
                assert R is P
            assert torch.equal(P[:half], before[:half])
            ch = not torch.equal(P, before)
            changed.append(int(ch))
            if ch:
                _check(g, f"{prefix}_out{i}", P)
            i += 1
    assert changed == g[f"{prefix}_changed"].tolist()
    assert controller.cur_step == int(g[f"{prefix}_cur_step"])
    for key, lst in controller.attention_store.items():
        assert len(lst) == int(g[f"{prefix}_storelen_{key}"])
        for j, t in enumerate(lst):
            _check(g, f"{prefix}_store_{key}_{j}", t)


PROMPTS = ["a cat sitting on a bench", "a dog sitting on a bench"]


@pytest.mark.parametrize("cond_only,padded", [(False, False), (True, True)])
def test_replace_on_device(g, cond_only, padded):
    c = p2p.make_controller(PROMPTS, True, 0.5, 0.5)
    _run(g, "replace", c, 2, 101, cond_only=cond_only, padded=padded)


def test_refine_on_device(g):
    c = p2p.make_controller(["a cat sitting on a bench", "a fluffy cat sitting on a red bench"], False,
                            {"default_": 0.8, "fluffy": (0.0, 0.4)}, 0.4)
    _run(g, "refine", c, 2, 102)


def test_reweight_chained_on_device(g):
    c = p2p.make_controller(PROMPTS, True, 0.6, 0.2, equilizer_params={"words": ("dog",), "values": (3.0,)})
    _run(g, "reweight", c, 2, 103)


def test_three_prompt_group_on_device(g):
    c = p2p.make_controller(["a cat sitting on a bench", "a dog sitting on a bench", "a cat sitting on a sofa"], True, 0.5, 0.25)
    _run(g, "replace3", c, 3, 104, steps=2)


def test_local_blend_values_on_device(g):
    """LocalBlend (utils/p2p.py:18-70) on the device: the fp16 probability walk runs on the GPU (fused cross-edit kernel,
    device-side store accumulation) and the word-localised latent blend is applied to device latents after every step;
    the blended latents are compared with the vectors captured from the reference.  The mask is a threshold of fp16 maps,
    so a handful of border pixels may flip: at most 0.5 % of the elements may differ, and none in the base prompt's row."""
    c = p2p.make_controller(["a cat sitting on a bench", "a fluffy cat sitting on a red bench"], False,
                            {"default_": 0.8, "fluffy": (0.0, 0.4)}, 0.4, blend_words=(("cat",), ("cat",)))
    assert c.local_blend is not None and c.local_blend.alpha_layers.is_cuda
    walk = list(zip(g["walk_place"].tolist(), [bool(x) for x in g["walk_cross"]], g["walk_n"].tolist()))
    c.num_att_layers = len(walk)
    gen = torch.Generator().manual_seed(102)
    lat = torch.from_numpy(g["refine_lat_in"]).cuda()
    outs = []
    for step in range(4):
        for place, is_cross, n in walk:
            m = 77 if is_cross else n
            P = torch.softmax(torch.randn(2 * 2 * HEADS, n, m, generator=gen) * 2.0, dim=-1).half().cuda()
            assert c(P, is_cross, place) is P
        lat = c.step_callback(lat)
        assert lat.is_cuda
        outs.append(lat.float().cpu())
    got, want = torch.stack(outs).numpy(), g["refine_latents"]
    assert got.shape == want.shape
    bad = np.abs(got - want) > 1e-5
    print(f"[local blend on device] mismatching elements: {bad.mean() * 100:.4f} %")
    assert bad.mean() < 5e-3
    assert not bad[:, 0].any()                                   # base prompt row is never blended
    assert np.abs(got - g["refine_lat_in"][None]).max() > 0      # and the blend did change the edited row


def test_local_blend_kernel_matches_the_torch_expression():
    """icd_local_blend (one launch) against the torch expression of utils/p2p.py:18-44 evaluated on the same device tensors:
    with and without substruct words, fp16 and fp32 latents, 3 prompts.  The two sum the word-weighted maps in different orders,
    so a border pixel of the thresholded mask may flip: <= 0.2 % of the elements may differ, none in the base row; wherever the
    masks agree the blend is bit-identical."""
    from invertible_cd_amd import ops
    gen = torch.Generator().manual_seed(7)
    P, H_, res = 3, 8, 16
    yy, xx = torch.meshgrid(torch.arange(res).float(), torch.arange(res).float(), indexing="ij")
    layers = []
    for li in range(5):                                  # word columns carry a spatial blob (else the normalised heat is flat)
        m = torch.softmax(torch.randn(P * H_, res * res, 77, generator=gen) * 3.0, dim=-1)
        for w, (cy, cx) in ((2, (4.0, 5.0)), (5, (11.0, 9.0)), (7, (8.0, 3.0))):
            blob = torch.exp(-((yy - cy - 0.3 * li) ** 2 + (xx - cx) ** 2) / 8.0).reshape(1, res * res)
            m[:, :, w] = m[:, :, w] * (0.05 + blob) * (1 + 0.1 * torch.rand(P * H_, 1, generator=gen))
        layers.append(m.half().cuda())
    alpha = torch.zeros(P, 77); alpha[:, [2, 5]] = 1
    sub = torch.zeros(P, 77); sub[:, [7]] = 1
    for x_dtype in (torch.float16, torch.float32):
        x = torch.randn(P, 4, 64, 64, generator=gen).to(x_dtype).cuda()
        for a_sub in (None, sub):
            got = ops.local_blend(layers, alpha, a_sub, 0.3, 0.3, x)
            maps = torch.cat([m.reshape(P, -1, 1, res, res, 77) for m in layers], dim=1)

            def get_mask(al, use_pool, th):
                heat = (maps * al.cuda().reshape(P, 1, 1, 1, 1, 77)).sum(-1).mean(1)
                if use_pool:
                    heat = F.max_pool2d(heat, kernel_size=3, stride=1, padding=1)
                heat = F.interpolate(heat, size=x.shape[2:])
                heat = heat / heat.amax(dim=(2, 3), keepdim=True)
                on = heat.gt(th)
                return on[:1] + on
            mask = get_mask(alpha, True, 0.3)
            if a_sub is not None:
                mask = mask * ~get_mask(a_sub, False, 0.3)
            want = x[:1] + mask.float() * (x - x[:1])
            assert got.dtype == want.dtype == torch.float32 and got.shape == want.shape
            bad = (got != want)
            assert bad.float().mean() < 2e-3 and not bad[0].any()
            assert 0.02 < mask.float().mean() < 0.98                              # the test exercises both branches of the mask


def test_accumulate_multi_is_torch_inplace_add():
    """icd_accumulate_multi: one launch adds 32 + 5 step maps into the store with torch's fp16 rounding (ragged sizes, tails)."""
    from invertible_cd_amd import ops
    gen = torch.Generator().manual_seed(8)
    shapes = [(16, 256, 77), (16, 1024, 77), (16, 256, 256), (3, 5, 7)] * 9 + [(1, 1, 1)]
    dst = [torch.rand(s, generator=gen).half().cuda() for s in shapes]
    src = [torch.rand(s, generator=gen).half().cuda() for s in shapes]
    want = [d.clone() for d in dst]
    for w, s_ in zip(want, src):
        w += s_
    ops.accumulate_multi(dst, src)
    for d, w in zip(dst, want):
        assert torch.equal(d, w)
