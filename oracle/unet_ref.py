"""ORACLE (test infrastructure, not product code) - CPU fp32 restatement of the UNet2DConditionModel forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

PARITY UNPINNED for this file: the UNet arithmetic of the reference lives in the third-party package
diffusers==0.25.1 (requirements/req.txt:1), which is NOT vendored under /root/reference, is not installed in the
build container, and cannot be fetched (no network).  The reference's own call sites are
utils/generation.py:208,241-244 (SD1.5) and utils/generation_sdxl.py:288-295,445-453 (SDXL); the only in-repo
statement of the attention arithmetic is the p2p hook utils/p2p.py:299-352, which this file follows exactly
(to_q/to_k/to_v -> head_to_batch_dim -> baddbmm(scale) -> softmax(-1) -> controller -> bmm -> to_out).
Everything else restates the published diffusers 0.25.1 architecture (SURVEY.md section 8a rows a12/a13 and
Appendix B) with stock torch.nn.functional ops; architecture fidelity is pinned by the exact parameter counts
(859 520 964 / 2 567 463 684 + 163 840 for cond_proj) and the diffusers state-dict key/shape layout
(tests/test_oracle_golden.py).

Functional style: weights are a flat dict {diffusers key: fp32 tensor}; layout NCHW / [B,N,C] as in the reference.
"""
import math

import torch
import torch.nn.functional as F


SD15 = dict(
    name="sd15", in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
    down_has_attn=(True, True, True, False), up_has_attn=(False, True, True, True), layers_per_block=2,
    transformer_layers=(1, 1, 1, 1), num_heads=(8, 8, 8, 8), cross_dim=768, use_linear_projection=False,
    time_cond_proj_dim=512, addition_time_embed_dim=None, add_in_dim=None, norm_groups=32,
)
SDXL = dict(
    name="sdxl", in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280),
    down_has_attn=(False, True, True), up_has_attn=(True, True, False), layers_per_block=2,
    transformer_layers=(1, 2, 10), num_heads=(5, 10, 20), cross_dim=2048, use_linear_projection=True,
    time_cond_proj_dim=512, addition_time_embed_dim=256, add_in_dim=2816, norm_groups=32,
)


def tiny(cfg, channels, cross_dim=64, heads=None):
    """Reduced-width variant of a config (same topology) for fast CPU tests."""
    c = dict(cfg)
    c["block_out_channels"] = tuple(channels)
    c["cross_dim"] = cross_dim
    if heads is not None:
        c["num_heads"] = tuple(heads)
    if cfg["add_in_dim"] is not None:
        c["pooled_dim"] = 64
        c["add_in_dim"] = 64 + 6 * cfg["addition_time_embed_dim"]
    c["name"] = cfg["name"] + "_tiny"
    return c


# --------------------------------------------------------------------------------------------- shapes
def param_shapes(cfg):
    """diffusers state-dict keys -> shapes for the UNet described by cfg (incl. time_embedding.cond_proj)."""
    ch = cfg["block_out_channels"]
    temb = ch[0] * 4
    sh = {}

    def lin(name, o, i, bias=True):
        sh[name + ".weight"] = (o, i)
        if bias:
            sh[name + ".bias"] = (o,)

    def conv(name, o, i, k):
        sh[name + ".weight"] = (o, i, k, k)
        sh[name + ".bias"] = (o,)

    def norm(name, c):
        sh[name + ".weight"] = (c,)
        sh[name + ".bias"] = (c,)

    def resnet(p, i, o):
        norm(p + ".norm1", i)
        conv(p + ".conv1", o, i, 3)
        lin(p + ".time_emb_proj", o, temb)
        norm(p + ".norm2", o)
        conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".conv_shortcut", o, i, 1)

    def transformer(p, c, depth):
        norm(p + ".norm", c)
        if cfg["use_linear_projection"]:
            lin(p + ".proj_in", c, c)
            lin(p + ".proj_out", c, c)
        else:
            conv(p + ".proj_in", c, c, 1)
            conv(p + ".proj_out", c, c, 1)
        for k in range(depth):
            b = f"{p}.transformer_blocks.{k}"
            norm(b + ".norm1", c)
            lin(b + ".attn1.to_q", c, c, False)
            lin(b + ".attn1.to_k", c, c, False)
            lin(b + ".attn1.to_v", c, c, False)
            lin(b + ".attn1.to_out.0", c, c)
            norm(b + ".norm2", c)
            lin(b + ".attn2.to_q", c, c, False)
            lin(b + ".attn2.to_k", c, cfg["cross_dim"], False)
            lin(b + ".attn2.to_v", c, cfg["cross_dim"], False)
            lin(b + ".attn2.to_out.0", c, c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", 8 * c, c)
            lin(b + ".ff.net.2", c, 4 * c)

    conv("conv_in", ch[0], cfg["in_channels"], 3)
    lin("time_embedding.linear_1", temb, ch[0])
    lin("time_embedding.linear_2", temb, temb)
    if cfg["time_cond_proj_dim"]:
        lin("time_embedding.cond_proj", ch[0], cfg["time_cond_proj_dim"], False)
    if cfg["add_in_dim"]:
        lin("add_embedding.linear_1", temb, cfg["add_in_dim"])
        lin("add_embedding.linear_2", temb, temb)
    nlev = len(ch)
    out_c = ch[0]
    for i in range(nlev):
        in_c, out_c = out_c, ch[i]
        for j in range(cfg["layers_per_block"]):
            resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
            if cfg["down_has_attn"][i]:
                transformer(f"down_blocks.{i}.attentions.{j}", out_c, cfg["transformer_layers"][i])
        if i < nlev - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    mid = ch[-1]
    resnet("mid_block.resnets.0", mid, mid)
    transformer("mid_block.attentions.0", mid, cfg["transformer_layers"][-1])
    resnet("mid_block.resnets.1", mid, mid)
    rev = list(reversed(ch))
    rev_depth = list(reversed(cfg["transformer_layers"]))
    prev = rev[0]
    for i in range(nlev):
        out_c = rev[i]
        in_c = rev[min(i + 1, nlev - 1)]
        for j in range(cfg["layers_per_block"] + 1):
            skip_c = in_c if j == cfg["layers_per_block"] else out_c
            res_in = prev if j == 0 else out_c
            resnet(f"up_blocks.{i}.resnets.{j}", res_in + skip_c, out_c)
            if cfg["up_has_attn"][i]:
                transformer(f"up_blocks.{i}.attentions.{j}", out_c, rev_depth[i])
        if i < nlev - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
        prev = out_c
    norm("conv_norm_out", ch[0])
    conv("conv_out", cfg["out_channels"], ch[0], 3)
    return sh


def count_params(cfg):
    return sum(math.prod(s) for s in param_shapes(cfg).values())


# --------------------------------------------------------------------------------------------- ops
def sinusoid(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos || sin], fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    ang = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _resnet(w, p, x, emb, groups):
    h = F.silu(F.group_norm(x, groups, w[p + ".norm1.weight"], w[p + ".norm1.bias"], eps=1e-5))
    h = F.conv2d(h, w[p + ".conv1.weight"], w[p + ".conv1.bias"], padding=1)
    tb = F.linear(F.silu(emb), w[p + ".time_emb_proj.weight"], w[p + ".time_emb_proj.bias"])
    h = h + tb[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, w[p + ".norm2.weight"], w[p + ".norm2.bias"], eps=1e-5))
    h = F.conv2d(h, w[p + ".conv2.weight"], w[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in w:
        x = F.conv2d(x, w[p + ".conv_shortcut.weight"], w[p + ".conv_shortcut.bias"])
    return x + h


def _attention(w, p, x, ctx, heads, is_cross, place, hook):
    """utils/p2p.py:299-352 with the Attention defaults of diffusers 0.25.1 (no group_norm, no mask,
    residual_connection=False, rescale_output_factor=1, upcast_attention=False)."""
    src = ctx if is_cross else x
    q = F.linear(x, w[p + ".to_q.weight"])
    k = F.linear(src, w[p + ".to_k.weight"])
    v = F.linear(src, w[p + ".to_v.weight"])
    B, N, C = q.shape
    d = C // heads

    def h2b(t):                                   # head_to_batch_dim: row index = b*heads + h
        return t.reshape(B, -1, heads, d).permute(0, 2, 1, 3).reshape(B * heads, -1, d)
    q, k, v = h2b(q), h2b(k), h2b(v)
    probs = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (d ** -0.5), dim=-1)
    if hook is not None:
        probs = hook(probs, is_cross, place)
    o = torch.bmm(probs, v)
    o = o.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    return F.linear(o, w[p + ".to_out.0.weight"], w[p + ".to_out.0.bias"])


def _transformer(w, p, x, ctx, heads, depth, linear_proj, groups, place, hook):
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, w[p + ".norm.weight"], w[p + ".norm.bias"], eps=1e-6)
    if linear_proj:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = F.linear(h, w[p + ".proj_in.weight"], w[p + ".proj_in.bias"])
    else:
        h = F.conv2d(h, w[p + ".proj_in.weight"], w[p + ".proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    for k in range(depth):
        b = f"{p}.transformer_blocks.{k}"
        n = F.layer_norm(h, (C,), w[b + ".norm1.weight"], w[b + ".norm1.bias"], eps=1e-5)
        h = h + _attention(w, b + ".attn1", n, None, heads, False, place, hook)
        n = F.layer_norm(h, (C,), w[b + ".norm2.weight"], w[b + ".norm2.bias"], eps=1e-5)
        h = h + _attention(w, b + ".attn2", n, ctx, heads, True, place, hook)
        n = F.layer_norm(h, (C,), w[b + ".norm3.weight"], w[b + ".norm3.bias"], eps=1e-5)
        g = F.linear(n, w[b + ".ff.net.0.proj.weight"], w[b + ".ff.net.0.proj.bias"])
        val, gate = g.chunk(2, dim=-1)
        h = h + F.linear(val * F.gelu(gate), w[b + ".ff.net.2.weight"], w[b + ".ff.net.2.bias"])
    if linear_proj:
        h = F.linear(h, w[p + ".proj_out.weight"], w[p + ".proj_out.bias"])
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = F.conv2d(h, w[p + ".proj_out.weight"], w[p + ".proj_out.bias"])
    return h + res


@torch.no_grad()
def time_embedding(w, cfg, t, batch, timestep_cond=None, added_cond=None):
    """Timesteps -> (+cond_proj(w_emb)) -> Linear -> SiLU -> Linear  (+ SDXL text_time add_embedding)."""
    ch0 = cfg["block_out_channels"][0]
    ref = w["time_embedding.linear_1.weight"]          # fp32 on the CPU for the oracle proper; fp16 on a GPU for the noise floor
    dt, dev = ref.dtype, ref.device
    tt = torch.as_tensor(t, dtype=torch.float32).reshape(-1).cpu()
    tt = tt.expand(batch) if tt.numel() == 1 else tt
    e = sinusoid(tt, ch0).to(dev, dt)                  # diffusers: time_proj in fp32, then cast to the sample dtype
    if timestep_cond is not None:
        e = e + F.linear(timestep_cond.to(dev, dt), w["time_embedding.cond_proj.weight"])
    e = F.linear(e, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"])
    e = F.linear(F.silu(e), w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])
    if cfg["add_in_dim"]:
        tid = added_cond["time_ids"].float().cpu()
        te = sinusoid(tid.flatten(), cfg["addition_time_embed_dim"]).reshape(batch, -1).to(dev, dt)
        a = torch.cat([added_cond["text_embeds"].to(dev, dt), te], dim=-1)
        a = F.linear(a, w["add_embedding.linear_1.weight"], w["add_embedding.linear_1.bias"])
        a = F.linear(F.silu(a), w["add_embedding.linear_2.weight"], w["add_embedding.linear_2.bias"])
        e = e + a
    return e


@torch.no_grad()
def unet_forward(w, cfg, sample, t, encoder_hidden_states, timestep_cond=None, added_cond=None, hook=None,
                 taps=None):
    """eps = UNet(sample[B,4,H,W], t, ctx[B,77,D], w_emb[B,512], {text_embeds,time_ids}) in fp32 on CPU.

    The arithmetic runs in the dtype / on the device of the weight dict `w`: fp32 CPU tensors give the oracle; the parity
    tests also call it once with fp16 weights on the GPU (stock torch fp16 ops, like the reference's fp16 diffusers path) to
    measure how far an fp16 pipeline sits from fp32 by construction - the "noise floor" printed next to the product's error.

    hook(probs[B*heads,N,M], is_cross, place) is called once per Attention module in module-execution order
    (self then cross per block; down -> mid -> up), exactly where utils/p2p.py:336 calls the controller.
    taps: optional dict filled with named intermediate activations (for kernel-level parity tests).
    """
    wref = w["conv_in.weight"]                         # compute dtype / device follow the weights (fp32 CPU = the oracle;
    x = sample.to(wref.device, wref.dtype)             # fp16 weights on a GPU = the "fp16 torch" noise-floor run of the tests)
    ctx = encoder_hidden_states.to(wref.device, wref.dtype)
    B = x.shape[0]
    G = cfg["norm_groups"]
    ch = cfg["block_out_channels"]
    nlev = len(ch)
    emb = time_embedding(w, cfg, t, B, timestep_cond, added_cond)
    if taps is not None:
        taps["emb"] = emb
    h = F.conv2d(x, w["conv_in.weight"], w["conv_in.bias"], padding=1)
    skips = [h]
    for i in range(nlev):
        for j in range(cfg["layers_per_block"]):
            h = _resnet(w, f"down_blocks.{i}.resnets.{j}", h, emb, G)
            if cfg["down_has_attn"][i]:
                h = _transformer(w, f"down_blocks.{i}.attentions.{j}", h, ctx, cfg["num_heads"][i],
                                 cfg["transformer_layers"][i], cfg["use_linear_projection"], G, "down", hook)
            skips.append(h)
        if i < nlev - 1:
            h = F.conv2d(h, w[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         w[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(h)
        if taps is not None:
            taps[f"down{i}"] = h
    h = _resnet(w, "mid_block.resnets.0", h, emb, G)
    h = _transformer(w, "mid_block.attentions.0", h, ctx, cfg["num_heads"][-1], cfg["transformer_layers"][-1],
                     cfg["use_linear_projection"], G, "mid", hook)
    h = _resnet(w, "mid_block.resnets.1", h, emb, G)
    if taps is not None:
        taps["mid"] = h
    rev_heads = list(reversed(cfg["num_heads"]))
    rev_depth = list(reversed(cfg["transformer_layers"]))
    for i in range(nlev):
        for j in range(cfg["layers_per_block"] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = _resnet(w, f"up_blocks.{i}.resnets.{j}", h, emb, G)
            if cfg["up_has_attn"][i]:
                h = _transformer(w, f"up_blocks.{i}.attentions.{j}", h, ctx, rev_heads[i], rev_depth[i],
                                 cfg["use_linear_projection"], G, "up", hook)
        if i < nlev - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, w[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                         w[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None:
            taps[f"up{i}"] = h
    h = F.silu(F.group_norm(h, G, w["conv_norm_out.weight"], w["conv_norm_out.bias"], eps=1e-5))
    return F.conv2d(h, w["conv_out.weight"], w["conv_out.bias"], padding=1)


def fuse_lora(w, lora, alpha=8.0):
    """diffusers fuse_lora (lora_scale=1): W' = W + (alpha/r) * up @ down, convs flattened to matrices.

    lora: {diffusers module path: (down[r, in(,k,k)], up[out, r(,1,1)])}; rank 64, alpha 8 -> 0.125
    (utils/loading.py:10-23, training/train_icd_sd15_lora.py:617-633).
    """
    out = dict(w)
    for path, (down, up) in lora.items():
        W = w[path + ".weight"].float()
        r = down.shape[0]
        delta = (up.float().flatten(1) @ down.float().flatten(1)).reshape(W.shape)
        out[path + ".weight"] = W + (alpha / r) * delta
    return out
