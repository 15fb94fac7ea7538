"""CLIP text encoders on the HIP kernels vs the real transformers classes (oracle/clip_ref.py) - SURVEY.md section 8f rank 3.

The oracle here is the third-party implementation itself (transformers CLIPTextModel / CLIPTextModelWithProjection) with
the same seeded weights, fp32 on the CPU.  Tolerance: rel-L2 <= 3e-3 on hidden states / pooled outputs (fp16 storage).
"""
import dataclasses

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _ids(B, T, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, vocab - 2, (B, T), generator=g)
    ids[:, 0] = vocab - 2                                      # <|startoftext|>
    for b in range(B):
        n = int(torch.randint(2, T - 1, (1,), generator=g))
        ids[b, n:] = vocab - 1                                 # <|endoftext|> then padding with the same id (max id)
    return ids


def _run(cfg, proj, B, seed):
    from invertible_cd_amd import clip, synthetic
    from oracle import clip_ref
    sd = {k: v.half().float() for k, v in synthetic.synthetic_clip_state_dict(cfg, proj, seed=seed).items()}
    ids = _ids(B, cfg.max_position_embeddings, cfg.vocab_size, seed)
    ref = clip_ref.forward(cfg.to_dict(), sd, ids, with_projection=proj)
    m = clip.CLIPTextModel(cfg, sd, with_projection=proj)
    out = m(ids.cuda(), output_hidden_states=True)
    return ref, out, m, ids


@pytest.mark.parametrize("act,proj,eos", [("quick_gelu", False, 2), ("gelu", True, 2), ("gelu", True, 999)])
def test_reduced_width_matches_transformers(act, proj, eos):
    from invertible_cd_amd import clip
    cfg = clip.CLIPTextConfig(vocab_size=1000, hidden_size=256, intermediate_size=1024, num_hidden_layers=3,
                              num_attention_heads=4, hidden_act=act, projection_dim=192, eos_token_id=eos)
    ref, out, m, ids = _run(cfg, proj, B=3, seed=21)
    assert len(out.hidden_states) == 4
    for i, (g, r) in enumerate(zip(out.hidden_states, ref["hidden_states"])):
        e = rel_l2(g.float().cpu(), r)
        assert e < 3e-3, (i, e)
    e = rel_l2(out.last_hidden_state.float().cpu(), ref["last_hidden_state"])
    print(f"[clip {act} proj={proj} eos={eos}] last_hidden_state rel-L2 = {e:.3e}")
    assert e < 3e-3
    if proj:
        assert out[0] is out.text_embeds and rel_l2(out.text_embeds.float().cpu(), ref["text_embeds"]) < 3e-3
    else:
        assert out[0] is out.last_hidden_state and rel_l2(out.pooler_output.float().cpu(), ref["pooled"]) < 3e-3


def test_full_clip_vit_l():
    """SD1.5's text encoder at full size: `text_encoder(ids)[0]` -> [B, 77, 768] (utils/generation.py:293,301)."""
    from invertible_cd_amd import clip
    ref, out, m, ids = _run(clip.CLIP_VIT_L, False, B=2, seed=5)
    got = m(ids.cuda())[0]
    assert got.shape == (2, 77, 768) and got.dtype == torch.float16
    e = rel_l2(got.float().cpu(), ref["last_hidden_state"])
    e2 = rel_l2(out.hidden_states[-2].float().cpu(), ref["hidden_states"][-2])
    print(f"[clip ViT-L full] last_hidden_state rel-L2 = {e:.3e}, hidden_states[-2] = {e2:.3e}")
    assert e < 1e-3 and e2 < 1e-3                        # round 6: fp32 twin of the residual stream (1.07e-3 / 1.03e-3 on the plain fp16 stream)


def test_bigg_width_four_layers_with_projection():
    """SDXL's second encoder at full width (1280, 20 heads, gelu, projection), 4 of its 32 layers (CPU-oracle time)."""
    from invertible_cd_amd import clip
    cfg = dataclasses.replace(clip.OPENCLIP_BIGG, num_hidden_layers=4)
    ref, out, m, ids = _run(cfg, True, B=2, seed=6)
    e = rel_l2(out.hidden_states[-2].float().cpu(), ref["hidden_states"][-2])
    e2 = rel_l2(out[0].float().cpu(), ref["text_embeds"])
    print(f"[clip bigG width] hidden_states[-2] rel-L2 = {e:.3e}, text_embeds = {e2:.3e}")
    assert e < 1e-3 and e2 < 1e-3 and out[0].shape == (2, 1280)


def test_interface_errors():
    from invertible_cd_amd import clip, synthetic
    cfg = clip.CLIPTextConfig(vocab_size=100, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1)
    sd = synthetic.synthetic_clip_state_dict(cfg, seed=1)
    m = clip.CLIPTextModel(cfg, sd)
    with pytest.raises(IndexError):
        m(torch.full((1, 77), 100))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 78, dtype=torch.int64))
    bad = dict(sd); bad.pop("text_model.final_layer_norm.bias")
    with pytest.raises(KeyError):
        clip.CLIPTextModel(cfg, bad)
    assert m(torch.ones(2, 77, dtype=torch.int64)).hidden_states is None
    assert m.to(torch.float32)(torch.ones(1, 77, dtype=torch.int64))[0].dtype == torch.float32


def test_causal_attention_kernel_against_torch():
    """icd_attention_fused_ex with ICD_ATTN_CAUSAL on its own: ragged 77 keys and a multi-tile 200-token sequence."""
    from invertible_cd_amd import ops
    for (B, H, N, d) in [(2, 3, 77, 64), (1, 2, 200, 40)]:
        g = torch.Generator().manual_seed(N)
        q, k, v = (torch.randn(B * N, H * d, generator=g).half() for _ in range(3))
        ld = (N + 7) // 8 * 8
        vt = torch.zeros(B, H * d, ld, dtype=torch.float16)
        vt[:, :, :N] = v.reshape(B, N, H * d).transpose(1, 2)
        out = ops.attention_fused(q.cuda(), k.cuda(), vt.cuda(), B, H, N, N, d, d ** -0.5, causal=True).float().cpu()
        qh, kh, vh = (t.float().reshape(B, N, H, d).permute(0, 2, 1, 3) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) * d ** -0.5
        s = s.masked_fill(torch.triu(torch.ones(N, N, dtype=torch.bool), 1), float("-inf"))
        ref = (torch.softmax(s, -1) @ vh).permute(0, 2, 1, 3).reshape(B * N, H * d)
        assert rel_l2(out, ref) < 2e-3, (N, d)
