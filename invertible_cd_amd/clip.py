"""CLIP text encoders (the step in front of the iCD path) on the HIP kernels of this package - SURVEY.md section 8f rank 3.

Drop-in for what the reference calls on `model.text_encoder` / `pipe.text_encoder(_2)` (transformers classes there):
    text_encoder(input_ids)[0]                                   utils/generation.py:293,301      -> [B, 77, 768]
    out = text_encoder(input_ids, output_hidden_states=True)     utils/generation_sdxl.py:31-44
    out[0] (pooled text_embeds of the projection model), out.hidden_states[-2]
    .device, .dtype, .config

Architecture (transformers CLIPTextModel / CLIPTextModelWithProjection; oracle/clip_ref.py runs the real classes):
token + position embeddings (`icd_embed_tokens`), N pre-LayerNorm blocks of causal multi-head self-attention (head dim 64;
`icd_attention_fused_ex` with ICD_ATTN_CAUSAL, q/k from one fused biased GEMM, V^T from the transposing GEMM epilogue) and a
biased MLP (`icd_gemm` -> `icd_activation` quick_gelu | gelu -> `icd_gemm` + residual), final LayerNorm, EOS pooling, optional
bias-free projection.  The V bias is folded into the output projection's bias (softmax rows sum to one, also under the
causal mask).  fp16 storage, fp32 accumulation.  Tokenizers need a vocabulary that is not available offline: callers pass
token ids (synthetic.SyntheticTokenizer produces ids of the right shape).
"""
from dataclasses import dataclass, asdict
from types import SimpleNamespace

import torch

from . import ops


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    projection_dim: int = 768
    eos_token_id: int = 2            # legacy value of the released checkpoints: pooling takes argmax(input_ids)
    bos_token_id: int = 0
    pad_token_id: int = 1

    def to_dict(self):
        return asdict(self)

    def state_dict_shapes(self, with_projection=False):
        C, I, V, T = self.hidden_size, self.intermediate_size, self.vocab_size, self.max_position_embeddings
        out = {"embeddings.token_embedding.weight": (V, C), "embeddings.position_embedding.weight": (T, C)}
        for i in range(self.num_hidden_layers):
            p = f"encoder.layers.{i}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                out[p + f"self_attn.{n}.weight"] = (C, C); out[p + f"self_attn.{n}.bias"] = (C,)
            for n in ("layer_norm1", "layer_norm2"):
                out[p + n + ".weight"] = (C,); out[p + n + ".bias"] = (C,)
            out[p + "mlp.fc1.weight"] = (I, C); out[p + "mlp.fc1.bias"] = (I,)
            out[p + "mlp.fc2.weight"] = (C, I); out[p + "mlp.fc2.bias"] = (C,)
        out["final_layer_norm.weight"] = (C,); out["final_layer_norm.bias"] = (C,)
        if with_projection:
            out["text_projection.weight"] = (self.projection_dim, C)
        return out


CLIP_VIT_L = CLIPTextConfig()                                                        # SD1.5 / SDXL text_encoder
OPENCLIP_BIGG = CLIPTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                               hidden_act="gelu", projection_dim=1280)               # SDXL text_encoder_2 (with projection)
_ACT = {"quick_gelu": ops.ACT_QUICK_GELU, "gelu": ops.ACT_GELU}


class TextEncoderOutput(tuple):
    """Indexable like the transformers ModelOutput the reference indexes ([0]) with the attributes it reads."""

    def __new__(cls, first, **fields):
        self = super().__new__(cls, (first,) + tuple(v for v in fields.values() if v is not None))
        self.__dict__.update(fields)
        return self


def _canon(sd):
    return {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}


class CLIPTextModel:
    def __init__(self, cfg: CLIPTextConfig, state_dict, with_projection=False, device="cuda", dtype=torch.float16):
        if cfg.hidden_size % cfg.num_attention_heads or cfg.hidden_size // cfg.num_attention_heads > 160 \
                or (cfg.hidden_size // cfg.num_attention_heads) % 8 or cfg.hidden_size % 8 or cfg.intermediate_size % 8:
            raise ValueError("CLIPTextModel: head dim must be a multiple of 8 and <= 160, widths multiples of 8")
        if cfg.hidden_act not in _ACT:
            raise ValueError(f"CLIPTextModel: unsupported hidden_act {cfg.hidden_act!r}")
        self.cfg, self.with_projection = cfg, with_projection
        self.device, self.dtype = torch.device(device), dtype
        self.config = SimpleNamespace(**cfg.to_dict())
        sd = _canon(state_dict)
        want = cfg.state_dict_shapes(with_projection)
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"CLIP text state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        for k, shp in want.items():
            if tuple(sd[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {tuple(shp)}, got {tuple(sd[k].shape)}")
        f32 = lambda k: sd[k].detach().to("cpu", torch.float32)
        half = lambda t: t.to(device=device, dtype=torch.float16).contiguous()
        full = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
        w = {"tok": half(f32("embeddings.token_embedding.weight")), "pos": half(f32("embeddings.position_embedding.weight"))}
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            w[p + "qk.w"] = half(torch.cat([f32(a + "q_proj.weight"), f32(a + "k_proj.weight")]))
            w[p + "qk.b"] = full(torch.cat([f32(a + "q_proj.bias"), f32(a + "k_proj.bias")]))
            w[p + "v.w"] = half(f32(a + "v_proj.weight"))
            wo = f32(a + "out_proj.weight")
            w[p + "o.w"] = half(wo)
            w[p + "o.b"] = full(wo @ f32(a + "v_proj.bias") + f32(a + "out_proj.bias"))
            for n in ("layer_norm1", "layer_norm2"):
                w[p + n + ".w"], w[p + n + ".b"] = full(f32(p + n + ".weight")), full(f32(p + n + ".bias"))
            for n in ("fc1", "fc2"):
                w[p + n + ".w"], w[p + n + ".b"] = half(f32(p + f"mlp.{n}.weight")), full(f32(p + f"mlp.{n}.bias"))
        w["ln_f.w"], w["ln_f.b"] = full(f32("final_layer_norm.weight")), full(f32("final_layer_norm.bias"))
        if with_projection:
            w["proj.w"] = half(f32("text_projection.weight"))
        self.w = w

    def to(self, *args, **kw):
        for a in list(args) + [kw.get("dtype")]:
            if isinstance(a, torch.dtype):
                self.dtype = a
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def __call__(self, input_ids, output_hidden_states=False, **unused):
        cfg, w = self.cfg, self.w
        if input_ids.dim() != 2 or input_ids.shape[1] > cfg.max_position_embeddings:
            raise ValueError(f"CLIPTextModel: input_ids must be [B, T <= {cfg.max_position_embeddings}], got {tuple(input_ids.shape)}")
        ids = input_ids.to(self.device, torch.int64).contiguous()
        if int(ids.min()) < 0 or int(ids.max()) >= cfg.vocab_size:
            raise IndexError("CLIPTextModel: token id out of range")
        B, T = ids.shape
        C, H = cfg.hidden_size, cfg.num_attention_heads
        d, ld = C // H, (T + 7) // 8 * 8
        x = ops.embed_tokens(ids, w["tok"], w["pos"])
        hs = [x]
        # Round 6: the residual stream keeps an fp32 twin (icd_gemm_desc.out_f32 + an fp32 `resid`): the 2 x num_hidden_layers adds x <- x + f(x)
        # accumulate in fp32, the fp16 copy is what LayerNorm and the returned hidden states read (ViT-L 1.07e-3 -> < 1e-3 against transformers;
        # the text encoders run once per prompt, outside every timed loop)
        x32 = None

        def add(f, wk, bk, x, x32):
            n32 = torch.empty(x.shape, device=x.device, dtype=torch.float32)
            return ops.gemm(f, w[wk], w[bk], resid=x if x32 is None else x32, out32=n32), n32
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{i}."
            h = ops.layernorm(x, w[p + "layer_norm1.w"], w[p + "layer_norm1.b"], cfg.layer_norm_eps)
            qk = ops.gemm(h, w[p + "qk.w"], w[p + "qk.b"])
            vt = ops.project_vt(h, w[p + "v.w"], B, T, ld)
            o = ops.attention_fused(qk[:, :C], qk[:, C:], vt, B, H, T, T, d, d ** -0.5, causal=True)
            x, x32 = add(o, p + "o.w", p + "o.b", x, x32)
            h = ops.layernorm(x, w[p + "layer_norm2.w"], w[p + "layer_norm2.b"], cfg.layer_norm_eps)
            f = ops.activation(ops.gemm(h, w[p + "fc1.w"], w[p + "fc1.b"]), _ACT[cfg.hidden_act])
            x, x32 = add(f, p + "fc2.w", p + "fc2.b", x, x32)
            hs.append(x)
        last = ops.layernorm(x, w["ln_f.w"], w["ln_f.b"], cfg.layer_norm_eps).reshape(B, T, C)
        if cfg.eos_token_id == 2:                               # transformers: legacy configs pool at argmax(input_ids)
            eos = ids.argmax(dim=-1)
        else:
            eos = (ids == cfg.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=self.device), eos].contiguous()
        cast = lambda t: t.to(self.dtype)
        hidden = tuple(cast(h.reshape(B, T, C)) for h in hs) if output_hidden_states else None
        if self.with_projection:
            embeds = ops.gemm(pooled, w["proj.w"])
            return TextEncoderOutput(cast(embeds), text_embeds=cast(embeds), last_hidden_state=cast(last), hidden_states=hidden)
        return TextEncoderOutput(cast(last), last_hidden_state=cast(last), pooler_output=cast(pooled), hidden_states=hidden)
