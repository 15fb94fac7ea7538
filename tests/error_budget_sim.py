"""CPU error budget of one UNet evaluation (test infrastructure; run as a script, imports oracle/).

The product stores activations in fp16 and feeds fp16 operands to the matrix cores; the oracle (oracle/unet_ref.py) is fp32 end to
end.  This file restates the oracle's graph with a rounding at EVERY site where the executor (csrc/runtime.hip) rounds, each site
tagged with a class, so that the rel-L2 of eps against the fp32 evaluation can be attributed:

  stream   the x <- x + f(x) chain as stored (fp16 + bf8 error carry, residual mode 2)
  sread    consumers of the stream read its fp16 part only (GroupNorm inputs, the A operand of the LayerNorm-folded projections,
           proj_out, the shortcut conv, the skip concat)
  gn       GroupNorm(+SiLU) outputs (conv / proj_in operands)
  c1       conv1 (+time bias) output of a ResnetBlock2D (GroupNorm 2 input)
  q k v    projection outputs (self and cross attention; k / v of the context GEMMs too)
  p        probabilities as the P.V operand (un-normalised exp2, row sum from the same rounded values)
  o        attention output (to_out operand)
  ff       GEGLU hidden tensor (ff.net.2 operand)
  wfold    fp16 rounding of LayerNorm-folded weights W * gamma (and the query scale d^-1/2 log2 e)
  temb     the time-embedding MLP chain and the per-resnet time biases
  misc     non-chain conv outputs (down / up samplers)

Usage:  python tests/error_budget_sim.py [sd15|sdxl] [--full]      prints: all sites on, each class knocked out, each class alone.
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import unet_ref  # noqa: E402

ALL = ("stream", "sread", "gn", "c1", "q", "k", "v", "p", "o", "ff", "wfold", "temb", "misc")


def h16(x):
    return x.half().float()


def carry(x):
    """fp16 + bf8(e5m2) of the rounding error at scale 2^14 (gemm_common.h carry_of8 / carry_add8)."""
    hi = x.half().float()
    c = ((x - hi) * 16384.0).to(torch.float8_e5m2).float() / 16384.0
    return hi, hi + c


class View:
    """A stored tensor of the residual chain: [0] operand view, [1] residual view, [2] GroupNorm view; .rd(kind) = the view a consumer
    of sub-class `kind` (sr_gn, sr_ln, sr_po, sr_sc, sr_conv, sr_skip) reads: fp16 part only unless that sub-class is switched off."""

    def __init__(self, rd, full, sim):
        self.h, self.full, self.sim = rd, full, sim

    def rd(self, kind):
        return self.full if kind in self.sim.extra.get("exact_reads", ()) else self.h

    def __getitem__(self, i):
        return (self.rd("sr_op"), self.full, self.rd("sr_gn"))[i]


class Plain(View):
    def __init__(self, t):
        self.h = self.full = t
        self.sim = None

    def rd(self, kind):
        return self.h


class Sim:
    def __init__(self, w, cfg, on, stream_mode="carry", extra=None):
        self.w, self.cfg, self.on, self.stream_mode = w, cfg, set(on), stream_mode
        self.extra = extra or {}

    def r(self, x, cls):
        return h16(x) if cls in self.on else x

    def store(self, x):
        """A tensor of the residual chain: returns (what consumers read, what the next add reads)."""
        if "stream" not in self.on:
            full = x
        elif self.stream_mode == "carry":
            full = carry(x)[1]
        else:
            full = h16(x)
        rd = h16(full) if "sread" in self.on else full
        return View(rd, full, self)

    # stream tensors are triples (operand view, residual view, groupnorm view)
    def resnet(self, p, x, emb_s, G, x_cat=None):
        w = self
        W = self.w
        xin_gn = x.rd("sr_gn") if x_cat is None else torch.cat([x.rd("sr_gn"), x_cat.rd("sr_skip")], 1)
        xin_op = x.rd("sr_sc") if x_cat is None else torch.cat([x.rd("sr_sc"), x_cat.rd("sr_skip")], 1)
        a1 = self.r(F.silu(F.group_norm(xin_gn, G, W[p + ".norm1.weight"], W[p + ".norm1.bias"], eps=1e-5)), "gn")
        tb = F.linear(emb_s, W[p + ".time_emb_proj.weight"], W[p + ".time_emb_proj.bias"])
        tb = tb if self.extra.get("temb_f32_out") else self.r(tb, "temb")
        h1 = F.conv2d(a1, W[p + ".conv1.weight"], W[p + ".conv1.bias"], padding=1) + tb[:, :, None, None]
        if self.extra.get("c1_carry"):
            h1 = carry(h1)[1] if "c1" in self.on else h1
            if "sr_gn" not in self.extra.get("exact_reads", ()):
                h1 = h16(h1)
        else:
            h1 = self.r(h1, "c1")
        a2 = self.r(F.silu(F.group_norm(h1, G, W[p + ".norm2.weight"], W[p + ".norm2.bias"], eps=1e-5)), "gn")
        h2 = F.conv2d(a2, W[p + ".conv2.weight"], W[p + ".conv2.bias"], padding=1)
        if (p + ".conv_shortcut.weight") in W:
            sc = self.store(F.conv2d(xin_op, W[p + ".conv_shortcut.weight"], W[p + ".conv_shortcut.bias"]))[1]
        else:
            sc = x[1]
        return self.store(sc + h2)

    def attention(self, p, h, ln, ctx, heads, is_cross):
        W = self.w
        g, be = W[ln + ".weight"], W[ln + ".bias"]
        x = h.rd("sr_ln")
        C = x.shape[-1]
        d = C // heads
        mean = x.mean(-1, keepdim=True)
        rstd = (x.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        xn = (x - mean) * rstd
        qs = d ** -0.5 * 1.4426950408889634

        def folded(key, scale=1.0):
            wt = W[p + key + ".weight"] * scale
            return self.r(wt * g[None, :], "wfold"), wt @ be

        wq, bq = folded(".to_q", qs)
        q = self.r(F.linear(xn, wq, bq), "q")
        if is_cross:
            k = self.r(F.linear(ctx, W[p + ".to_k.weight"]), "k")
            v = self.r(F.linear(ctx, W[p + ".to_v.weight"]), "v")
            vb = 0.0
        else:
            wk, bk = folded(".to_k")
            wv, bv = folded(".to_v")
            k = self.r(F.linear(xn, wk, bk), "k")
            v = self.r(F.linear(xn, wv), "v")                  # W_v beta rides in to_out's bias
            vb = bv
        B, N, _ = q.shape

        def h2b(t):
            return t.reshape(B, -1, heads, d).permute(0, 2, 1, 3).reshape(B * heads, -1, d)
        q, k, v = h2b(q), h2b(k), h2b(v)
        s = torch.bmm(q, k.transpose(1, 2))                    # base-2 exponent
        pr = torch.exp2(s - s.amax(-1, keepdim=True))
        pr = self.r(pr, "p")
        o = torch.bmm(pr, v) / pr.sum(-1, keepdim=True)
        o = o.reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
        o = self.r(o, "o")
        return F.linear(o, W[p + ".to_out.0.weight"], W[p + ".to_out.0.bias"] + (F.linear(vb, W[p + ".to_out.0.weight"]) if not is_cross else 0.0))

    def transformer(self, p, x, ctx, heads, depth, linear_proj, G):
        W = self.w
        B, C, H, Wd = x.h.shape
        a = self.r(F.group_norm(x.rd("sr_gn"), G, W[p + ".norm.weight"], W[p + ".norm.bias"], eps=1e-6), "gn")
        a = a.permute(0, 2, 3, 1).reshape(B, H * Wd, C)
        wpi = W[p + ".proj_in.weight"].reshape(C, C)
        h = self.store(F.linear(a, wpi, W[p + ".proj_in.bias"]))
        for kb in range(depth):
            b = f"{p}.transformer_blocks.{kb}"
            h = self.store(h[1] + self.attention(b + ".attn1", h, b + ".norm1", None, heads, False))
            h = self.store(h[1] + self.attention(b + ".attn2", h, b + ".norm2", ctx, heads, True))
            g3, b3 = W[b + ".norm3.weight"], W[b + ".norm3.bias"]
            xx = h.rd("sr_ln")
            xn = (xx - xx.mean(-1, keepdim=True)) * (xx.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
            wf = W[b + ".ff.net.0.proj.weight"]
            gg = F.linear(xn, self.r(wf * g3[None, :], "wfold"), wf @ b3 + W[b + ".ff.net.0.proj.bias"])
            val, gate = gg.chunk(2, dim=-1)
            hid = self.r(val * F.gelu(gate), "ff")
            h = self.store(h[1] + F.linear(hid, W[b + ".ff.net.2.weight"], W[b + ".ff.net.2.bias"]))
        wpo = W[p + ".proj_out.weight"].reshape(C, C)
        o = F.linear(h.rd("sr_po"), wpo, W[p + ".proj_out.bias"]).reshape(B, H, Wd, C).permute(0, 3, 1, 2)
        return self.store(o + x[1])

    @torch.no_grad()
    def forward(self, sample, t, ctx, timestep_cond=None, added_cond=None):
        W, cfg = self.w, self.cfg
        B = sample.shape[0]
        G = cfg["norm_groups"]
        ch = cfg["block_out_channels"]
        nlev = len(ch)
        T = lambda x: self.r(x, "temb")  # noqa: E731
        tt = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
        tt = tt.expand(B) if tt.numel() == 1 else tt
        e = T(unet_ref.sinusoid(tt, ch[0]))
        if timestep_cond is not None:
            e = T(e + F.linear(timestep_cond, W["time_embedding.cond_proj.weight"]))
        e = T(F.silu(T(F.linear(e, W["time_embedding.linear_1.weight"], W["time_embedding.linear_1.bias"]))))
        e = T(F.linear(e, W["time_embedding.linear_2.weight"], W["time_embedding.linear_2.bias"]))
        if cfg["add_in_dim"]:
            te = T(unet_ref.sinusoid(added_cond["time_ids"].float().flatten(), cfg["addition_time_embed_dim"]).reshape(B, -1))
            a = torch.cat([added_cond["text_embeds"], te], -1)
            a = T(F.silu(T(F.linear(a, W["add_embedding.linear_1.weight"], W["add_embedding.linear_1.bias"]))))
            e = T(e + F.linear(a, W["add_embedding.linear_2.weight"], W["add_embedding.linear_2.bias"]))
        emb_s = T(F.silu(e))
        h = self.store(F.conv2d(sample, W["conv_in.weight"], W["conv_in.bias"], padding=1))
        skips = [h]
        for i in range(nlev):
            for j in range(cfg["layers_per_block"]):
                h = self.resnet(f"down_blocks.{i}.resnets.{j}", h, emb_s, G)
                if cfg["down_has_attn"][i]:
                    h = self.transformer(f"down_blocks.{i}.attentions.{j}", h, ctx, cfg["num_heads"][i], cfg["transformer_layers"][i],
                                         cfg["use_linear_projection"], G)
                skips.append(h)
            if i < nlev - 1:
                dn = F.conv2d(h.rd("sr_down") if "sr_down" in self.extra.get("exact_reads", ()) else h.rd("sr_conv"), W[f"down_blocks.{i}.downsamplers.0.conv.weight"], W[f"down_blocks.{i}.downsamplers.0.conv.bias"],
                              stride=2, padding=1)
                if ch[i + 1] == ch[i]:
                    h = self.store(dn)
                else:
                    h = self.store(dn) if self.extra.get("misc_carry") and "misc" in self.on else Plain(self.r(dn, "misc"))
                skips.append(h)
        h = self.resnet("mid_block.resnets.0", h, emb_s, G)
        h = self.transformer("mid_block.attentions.0", h, ctx, cfg["num_heads"][-1], cfg["transformer_layers"][-1],
                             cfg["use_linear_projection"], G)
        h = self.resnet("mid_block.resnets.1", h, emb_s, G)
        rev_heads = list(reversed(cfg["num_heads"]))
        rev_depth = list(reversed(cfg["transformer_layers"]))
        for i in range(nlev):
            for j in range(cfg["layers_per_block"] + 1):
                h = self.resnet(f"up_blocks.{i}.resnets.{j}", h, emb_s, G, x_cat=skips.pop())
                if cfg["up_has_attn"][i]:
                    h = self.transformer(f"up_blocks.{i}.attentions.{j}", h, ctx, rev_heads[i], rev_depth[i],
                                         cfg["use_linear_projection"], G)
            if i < nlev - 1:
                u = F.interpolate(h.rd("sr_conv"), scale_factor=2.0, mode="nearest")
                u = F.conv2d(u, W[f"up_blocks.{i}.upsamplers.0.conv.weight"], W[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
                h = self.store(u) if self.extra.get("misc_carry") and "misc" in self.on else Plain(self.r(u, "misc"))
        a = self.r(F.silu(F.group_norm(h.rd("sr_gn"), G, W["conv_norm_out.weight"], W["conv_norm_out.bias"], eps=1e-5)), "gn")
        return F.conv2d(a, W["conv_out.weight"], W["conv_out.bias"], padding=1)


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def case(which, full=False, seed=31, B=2, H=32, t=779):
    from invertible_cd_amd import synthetic, unet_config as uc
    if which == "sd15":
        cfg = uc.SD15 if full else uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
        base = unet_ref.SD15
    else:
        cfg = uc.SDXL if full else uc.SDXL.scaled((64, 128, 256), cross_dim=128, heads=(2, 4, 8))
        base = unet_ref.SDXL
    o = dict(base)
    o["block_out_channels"], o["cross_dim"], o["num_heads"] = cfg.block_out_channels, cfg.cross_dim, cfg.num_heads
    if cfg.addition_time_embed_dim:
        o["add_in_dim"] = cfg.add_in_dim
    sd = {k: v.half().float() for k, v in synthetic.synthetic_state_dict(cfg, seed=seed).items()}
    inp = synthetic.synthetic_inputs(cfg, B, H, H, seed=seed)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    cond = torch.randn(B, cfg.time_cond_proj_dim, generator=torch.Generator().manual_seed(seed + 5)).half().float()
    added = None
    if cfg.addition_time_embed_dim:
        added = {"text_embeds": inp["text_embeds"].half().float(), "time_ids": inp["time_ids"]}
    return sd, o, lat, t, ctx, cond, added


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "sd15"
    full = "--full" in sys.argv
    sd, o, lat, t, ctx, cond, added = case(which, full)
    ref = unet_ref.unet_forward(sd, o, lat, t, ctx, timestep_cond=cond, added_cond=added)

    def run(on, **kw):
        return rel_l2(Sim(sd, o, on, **kw).forward(lat, t, ctx, cond, added), ref)

    print(f"# {which} {'full' if full else 'reduced'} width, B={lat.shape[0]} {lat.shape[-1]}^2 t={t}; rel-L2 of eps vs the fp32 oracle")
    print(f"no rounding (graph check)             {run(()):.3e}")
    base = run(ALL)
    print(f"all sites, carried stream (product)   {base:.3e}")
    print(f"all sites, plain fp16 stream          {run(ALL, stream_mode='fp16'):.3e}")
    rows = []
    for c in ALL:
        off = run([x for x in ALL if x != c])
        alone = run([c] if c != "sread" else ["sread", "stream"])
        rows.append((c, off, alone))
    print("class     without it   share of variance   alone")
    for c, off, alone in rows:
        print(f"{c:8s}  {off:.3e}    {1 - (off / base) ** 2:6.1%}             {alone:.3e}")
    print("stream reads by consumer (that consumer alone reads fp16 + carry):")
    for k in ("sr_gn", "sr_skip", "sr_ln", "sr_po", "sr_sc", "sr_conv"):
        e = run(ALL, extra={"exact_reads": (k,)})
        print(f"  {k:8s} {e:.3e}   share {1 - (e / base) ** 2:6.1%}")
    combos = [
        ("GN reads carry (incl. skips)", dict(exact_reads=("sr_gn", "sr_skip"))),
        ("GN + shortcut conv split", dict(exact_reads=("sr_gn", "sr_skip", "sr_sc"))),
        ("GN + shortcut + proj_out split", dict(exact_reads=("sr_gn", "sr_skip", "sr_sc", "sr_po"))),
        ("GN + shortcut + proj_out + misc carried", dict(exact_reads=("sr_gn", "sr_skip", "sr_sc", "sr_po"), misc_carry=True)),
        ("... + conv1 carried", dict(exact_reads=("sr_gn", "sr_skip", "sr_sc", "sr_po"), misc_carry=True, c1_carry=True)),
        ("... + time biases fp32", dict(exact_reads=("sr_gn", "sr_skip", "sr_sc", "sr_po"), misc_carry=True, c1_carry=True, temb_f32_out=True)),
        ("... + samplers split", dict(exact_reads=("sr_gn", "sr_skip", "sr_sc", "sr_po", "sr_conv"), misc_carry=True, c1_carry=True, temb_f32_out=True)),
        ("GN + shortcut + proj_out + misc carried + temb fp32 (no c1)", dict(exact_reads=("sr_gn", "sr_skip", "sr_sc", "sr_po"), misc_carry=True, temb_f32_out=True)),
    ]
    for name, ex in combos:
        e = run(ALL, extra=ex)
        print(f"{name:60s} {e:.3e}  x{e / base:.2f}")
    e = run([c for c in ALL if c != "temb"], extra=combos[-1][1])
    print(f"{'last + whole time-embedding path exact':60s} {e:.3e}  x{e / base:.2f}")


def loops():
    """The forward (inversion) loops of tests/test_sampler_gpu.py with the simulated product UNet in place of the GPU: how a
    per-evaluation improvement carries over to the loop bars."""
    import numpy as np
    from oracle import sched_ref as S
    ac = S.alphas_cumprod()
    alpha, sigma = np.sqrt(ac), np.sqrt(1 - ac)
    split = dict(exact_reads=("sr_gn", "sr_skip", "sr_sc", "sr_po", "sr_down"), misc_carry=True, c1_carry=True)
    accurate = dict(split, exact_reads=split["exact_reads"] + ("sr_conv",))
    no_temb = [c for c in ALL if c != "temb"]
    variants = [("carry only (round 4)", ALL, {}), ("GroupNorm reads the carry", ALL, dict(exact_reads=("sr_gn", "sr_skip"))),
                ("split level (mask default)", no_temb, split), ("accurate level (every bit)", no_temb, accurate)]
    for which, pairs, seed, xl in (("sd15", list(zip([19, 259, 519, 779], [259, 519, 779, 999])), 12, False),
                                   ("sdxl", list(zip([19, 339, 699], [339, 699, 999])), 21, True)):
        sd, o, lat, _, ctx, _, added = case(which, False, seed=seed)
        B = lat.shape[0]
        noise = torch.randn(lat.shape, generator=torch.Generator().manual_seed(5))
        x0 = float(alpha[19]) * lat + float(sigma[19]) * noise
        wemb = torch.from_numpy(S.guidance_scale_embedding([0.0] * B, 512)).half().float()

        def loop(fn):
            x = x0.clone()
            for t, s in pairs:
                eps = fn(x.half().float(), t).half().float()
                x = torch.from_numpy(S.predicted_origin(eps.numpy(), [t] * B, [s] * B, x.numpy(), alpha, sigma))
                if xl:
                    x = x.half().float()
            return x
        ref = loop(lambda x, t: unet_ref.unet_forward(sd, o, x, t, ctx, timestep_cond=wemb, added_cond=added))
        for name, on, ex in variants:
            got = loop(lambda x, t: Sim(sd, o, on, extra=ex).forward(x, t, ctx, wemb, added))
            print(f"[{which} forward loop, {len(pairs)} steps] {name:28s} rel-L2 = {rel_l2(got, ref):.3e}")


if __name__ == "__main__":
    loops() if "--loops" in sys.argv else main()
