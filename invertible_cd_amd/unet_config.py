"""Architecture descriptions of the two UNet2DConditionModel variants the iCD path evaluates.

SD1.5 (utils/loading.py:48-51, `time_cond_proj_dim=w_embed_dim`) and SDXL (utils/loading.py:100-103).  The numbers are
the published diffusers configs of runwayml/stable-diffusion-v1-5 and stabilityai/stable-diffusion-xl-base-1.0
(SURVEY.md section 8a rows a12/a13); `state_dict_shapes` reproduces the diffusers key layout so real checkpoints
(teacher .pt state-dicts, utils/loading.py:52-54,105-108) bind by name.
"""
from dataclasses import dataclass, replace
from typing import Dict, Optional, Tuple


@dataclass(frozen=True)
class UNetConfig:
    name: str
    block_out_channels: Tuple[int, ...]
    down_has_attn: Tuple[bool, ...]
    up_has_attn: Tuple[bool, ...]
    transformer_layers: Tuple[int, ...]        # per down level
    num_heads: Tuple[int, ...]                 # per down level
    cross_dim: int
    use_linear_projection: bool
    time_cond_proj_dim: int = 512
    addition_time_embed_dim: int = 0
    pooled_dim: int = 0                        # SDXL text_embeds width
    layers_per_block: int = 2
    in_channels: int = 4
    out_channels: int = 4
    norm_groups: int = 32
    sample_size: int = 64

    @property
    def num_levels(self):
        return len(self.block_out_channels)

    @property
    def temb_dim(self):
        return 4 * self.block_out_channels[0]

    @property
    def add_in_dim(self):
        return self.pooled_dim + 6 * self.addition_time_embed_dim if self.addition_time_embed_dim else 0

    def resnet_names(self):
        """Resnets in execution order with (in_channels, out_channels) - also the order of time_emb_proj_cat."""
        ch, L = self.block_out_channels, self.num_levels
        out, prev = [], ch[0]
        for i in range(L):
            for j in range(self.layers_per_block):
                out.append((f"down_blocks.{i}.resnets.{j}", prev if j == 0 else ch[i], ch[i]))
            prev = ch[i]
        out.append(("mid_block.resnets.0", ch[-1], ch[-1]))
        out.append(("mid_block.resnets.1", ch[-1], ch[-1]))
        prev = ch[-1]
        for i in range(L):
            lvl = L - 1 - i
            co, below = ch[lvl], ch[max(lvl - 1, 0)]
            for j in range(self.layers_per_block + 1):
                skip = below if j == self.layers_per_block else co
                out.append((f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else co) + skip, co))
            prev = co
        return out

    def transformer_names(self):
        """Transformer2DModel instances in execution order: (prefix, channels, depth, heads, place)."""
        ch, L = self.block_out_channels, self.num_levels
        out = []
        for i in range(L):
            if self.down_has_attn[i]:
                for j in range(self.layers_per_block):
                    out.append((f"down_blocks.{i}.attentions.{j}", ch[i], self.transformer_layers[i], self.num_heads[i], "down"))
        out.append(("mid_block.attentions.0", ch[-1], self.transformer_layers[-1], self.num_heads[-1], "mid"))
        for i in range(L):
            lvl = L - 1 - i
            if self.up_has_attn[i]:
                for j in range(self.layers_per_block + 1):
                    out.append((f"up_blocks.{i}.attentions.{j}", ch[lvl], self.transformer_layers[lvl], self.num_heads[lvl], "up"))
        return out

    @property
    def num_attention_layers(self):
        return sum(2 * depth for _, _, depth, _, _ in self.transformer_names())

    def state_dict_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """diffusers state-dict key -> shape (every parameter of the UNet incl. time_embedding.cond_proj)."""
        sh: Dict[str, Tuple[int, ...]] = {}
        ch, temb, X = self.block_out_channels, self.temb_dim, self.cross_dim

        def affine(p, c):
            sh[p + ".weight"], sh[p + ".bias"] = (c,), (c,)

        def dense(p, o, i, bias=True):
            sh[p + ".weight"] = (o, i)
            if bias:
                sh[p + ".bias"] = (o,)

        def conv(p, o, i, k):
            sh[p + ".weight"], sh[p + ".bias"] = (o, i, k, k), (o,)

        conv("conv_in", ch[0], self.in_channels, 3)
        dense("time_embedding.linear_1", temb, ch[0])
        dense("time_embedding.linear_2", temb, temb)
        if self.time_cond_proj_dim:
            dense("time_embedding.cond_proj", ch[0], self.time_cond_proj_dim, bias=False)
        if self.add_in_dim:
            dense("add_embedding.linear_1", temb, self.add_in_dim)
            dense("add_embedding.linear_2", temb, temb)
        for p, ci, co in self.resnet_names():
            affine(p + ".norm1", ci)
            conv(p + ".conv1", co, ci, 3)
            dense(p + ".time_emb_proj", co, temb)
            affine(p + ".norm2", co)
            conv(p + ".conv2", co, co, 3)
            if ci != co:
                conv(p + ".conv_shortcut", co, ci, 1)
        for p, c, depth, _, _ in self.transformer_names():
            affine(p + ".norm", c)
            for q in ("proj_in", "proj_out"):
                if self.use_linear_projection:
                    dense(f"{p}.{q}", c, c)
                else:
                    conv(f"{p}.{q}", c, c, 1)
            for k in range(depth):
                b = f"{p}.transformer_blocks.{k}"
                for n in ("norm1", "norm2", "norm3"):
                    affine(f"{b}.{n}", c)
                for a, kv_in in (("attn1", c), ("attn2", X)):
                    dense(f"{b}.{a}.to_q", c, c, bias=False)
                    dense(f"{b}.{a}.to_k", c, kv_in, bias=False)
                    dense(f"{b}.{a}.to_v", c, kv_in, bias=False)
                    dense(f"{b}.{a}.to_out.0", c, c)
                dense(f"{b}.ff.net.0.proj", 8 * c, c)
                dense(f"{b}.ff.net.2", c, 4 * c)
        for i in range(self.num_levels - 1):
            conv(f"down_blocks.{i}.downsamplers.0.conv", ch[i], ch[i], 3)
            c_up = ch[self.num_levels - 1 - i]
            conv(f"up_blocks.{i}.upsamplers.0.conv", c_up, c_up, 3)
        affine("conv_norm_out", ch[0])
        conv("conv_out", self.out_channels, ch[0], 3)
        return sh

    def num_parameters(self):
        n = 0
        for s in self.state_dict_shapes().values():
            k = 1
            for d in s:
                k *= d
            n += k
        return n

    def scaled(self, channels, cross_dim=64, heads=None, pooled_dim=64, name=None) -> "UNetConfig":
        """Same topology, reduced widths - for fast tests."""
        return replace(self, block_out_channels=tuple(channels), cross_dim=cross_dim,
                       num_heads=tuple(heads) if heads else self.num_heads,
                       pooled_dim=pooled_dim if self.addition_time_embed_dim else 0,
                       name=name or self.name + "_tiny")


SD15 = UNetConfig(
    name="sd15", block_out_channels=(320, 640, 1280, 1280), down_has_attn=(True, True, True, False),
    up_has_attn=(False, True, True, True), transformer_layers=(1, 1, 1, 1), num_heads=(8, 8, 8, 8), cross_dim=768,
    use_linear_projection=False, sample_size=64)

SDXL = UNetConfig(
    name="sdxl", block_out_channels=(320, 640, 1280), down_has_attn=(False, True, True), up_has_attn=(True, True, False),
    transformer_layers=(1, 2, 10), num_heads=(5, 10, 20), cross_dim=2048, use_linear_projection=True,
    addition_time_embed_dim=256, pooled_dim=1280, sample_size=128)

# LoRA targets of the iCD students (training/train_icd_sd15_lora.py:617-632)
LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0", "proj_in", "proj_out", "ff.net.0.proj", "ff.net.2", "conv1", "conv2",
                "conv_shortcut", "downsamplers.0.conv", "upsamplers.0.conv", "time_emb_proj")


def get_config(name: str) -> UNetConfig:
    return {"sd15": SD15, "sdxl": SDXL}[name]
