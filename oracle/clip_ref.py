"""CPU oracle for the CLIP text encoders in front of the iCD path - TEST INFRASTRUCTURE ONLY.

Only tests/ (and bench / smoke legs, if any) may import this module; the product path (invertible_cd_amd/clip.py) never
does.

Unlike the UNet / VAE oracles this is NOT a restatement: the reference's text encoders are `transformers` classes
(`CLIPTextModel` for SD1.5 and SDXL's first encoder, `CLIPTextModelWithProjection` for SDXL's second;
utils/loading.py:41,108-112) and `transformers` is installed in this image, so the oracle IS the third-party
implementation, instantiated from a config with caller-supplied weights and run in fp32 on the CPU.  Version here:
transformers 5.x (the reference does not pin one, requirements/req.txt); the architecture of these classes has been
stable since 4.2x.  PARITY PINNED by construction for the architecture; real checkpoints cannot be loaded (no network).

Call sites restated by the product: `text_encoder(ids)[0]` (utils/generation.py:293,301) and
`text_encoder(ids, output_hidden_states=True)` -> `[0]`, `.hidden_states[-2]` (utils/generation_sdxl.py:31-44).
"""
import torch


def build(cfg: dict, state_dict: dict, with_projection: bool):
    """cfg: keys of transformers.CLIPTextConfig; state_dict: checkpoint-style keys ('text_model.' prefix optional)."""
    import transformers
    tc = transformers.CLIPTextConfig(**cfg)
    tc._attn_implementation = "eager"
    model = (transformers.CLIPTextModelWithProjection if with_projection else transformers.CLIPTextModel)(tc).eval().float()
    own = model.state_dict()
    canon = lambda k: k[len("text_model."):] if k.startswith("text_model.") else k
    src = {canon(k): v for k, v in state_dict.items()}
    new = {}
    for k, v in own.items():
        ck = canon(k)
        if ck in src:
            assert tuple(src[ck].shape) == tuple(v.shape), (k, tuple(src[ck].shape), tuple(v.shape))
            new[k] = src[ck].to(torch.float32)
        elif ck.endswith("position_ids"):
            new[k] = v
        else:
            raise KeyError(f"oracle/clip_ref: state dict lacks {ck}")
    model.load_state_dict(new, strict=True)
    return model


@torch.no_grad()
def forward(cfg: dict, state_dict: dict, input_ids, with_projection=False):
    """-> dict(last_hidden_state [B,T,C], pooled [B,C], text_embeds [B,P] | None, hidden_states tuple of L+1 [B,T,C])."""
    m = build(cfg, state_dict, with_projection)
    out = m(input_ids, output_hidden_states=True)
    if with_projection:
        return dict(last_hidden_state=out.last_hidden_state, pooled=None, text_embeds=out.text_embeds,
                    hidden_states=tuple(out.hidden_states))
    return dict(last_hidden_state=out.last_hidden_state, pooled=out.pooler_output, text_embeds=None,
                hidden_states=tuple(out.hidden_states))


def state_dict_keys(cfg: dict, with_projection: bool):
    """canonical (prefix-free) keys -> shapes of the transformers model, for the layout test."""
    import transformers
    tc = transformers.CLIPTextConfig(**cfg)
    with torch.device("meta"):
        model = (transformers.CLIPTextModelWithProjection if with_projection else transformers.CLIPTextModel)(tc)
    canon = lambda k: k[len("text_model."):] if k.startswith("text_model.") else k
    return {canon(k): tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("position_ids")}
