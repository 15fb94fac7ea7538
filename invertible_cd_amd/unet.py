"""UNet2DConditionModel - the duck type the iCD sampler calls, backed by the native MI355X executor.

Mirrors what the reference touches on diffusers' class: callable
`unet(sample, t, encoder_hidden_states=, timestep_cond=, added_cond_kwargs=, return_dict=)` returning an object with
`.sample` / `["sample"]` / `[0]`, plus `.dtype`, `.in_channels`, `.config`, `.named_children()`, `.to()`.
Reference call sites: utils/generation.py:208,241-244; utils/generation_sdxl.py:288-295,445-453.

All arithmetic happens in libicd_amd.so (HIP, gfx950); torch is used for device memory, streams and the weight
re-layout at load time.  There is no CPU fallback: constructing this class without the built extension or without a GPU
raises.
"""
import ctypes as C
import types

import torch

from . import _lib
from .ops import geglu_perm
from .unet_config import UNetConfig

PLACES = ("down", "mid", "up")


class UNetOutput:
    """Stands in for diffusers' UNet2DConditionOutput: `.sample`, ["sample"], [0]."""

    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, k):
        if k == "sample" or k == 0:
            return self.sample
        raise KeyError(k)

    def __iter__(self):
        return iter((self.sample,))


def upsample_phase_weights(w):
    """[O, I, 3, 3] -> four [O, 4, I] tap-summed weights, phase 2 py + px, taps (dy, dx) in {0, 1}^2 row-major: nearest-2x upsampling
    followed by conv3x3 (pad 1) equals, at output pixel (2y + py, 2x + px), a 2 x 2 conv over input pixels (y + py - 1 + dy, x + px - 1 + dx)
    whose weights are the sums of the 3 x 3 taps that land on the same input pixel (ky -> input row y + floor((py + ky - 1) / 2))."""
    sets = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}          # (phase, d) -> 3 x 3 tap indices
    w = w if w.dtype == torch.float64 else w.float()
    out = []
    for py in (0, 1):
        for px in (0, 1):
            taps = []
            for dy in (0, 1):
                for dx in (0, 1):
                    taps.append(sum(w[:, :, ky, kx] for ky in sets[(py, dy)] for kx in sets[(px, dx)]))
            out.append(torch.stack(taps, 1))                                         # [O, 4, I]
    return out


def pack_state_dict(cfg: UNetConfig, sd, device):
    """diffusers-layout state dict -> the packed tensors the native executor binds (fp16 weights, fp32 bias/norm).

      conv [O,I,k,k]      -> [O, k*k*I]  (tap-major K of the implicit GEMM; 1x1 convs become plain [O,I])
      attn1.to_q/to_k     -> attn1.to_qk [2C, C]   (one GEMM, q|k column blocks; q rows carry d^-1/2 * log2(e), as attn2.to_q does)
      norm1/2/3 (LayerNorm) -> folded into to_qk / to_v / attn2.to_q / ff.net.0.proj: gamma into the weight columns,
                             W beta into the bias, plus the row sums the epilogue's mean correction needs
      ff.net.0.proj       -> rows interleaved 32 value / 32 gate so GEGLU is applied in the GEMM epilogue
      *.time_emb_proj     -> ONE [sum(Cout), 4*ch0] matrix in execution order (all resnets' time biases in one GEMM)
      attn2.to_k / to_v   -> attn2_k_cat / attn2_v_cat [sum(C), cross_dim] in execution order (two GEMMs per forward)
    """
    packed = {}

    def w16(t):
        return t.to(device=device, dtype=torch.float16).contiguous()

    w16_ = w16

    def f32(t):
        return t.to(device=device, dtype=torch.float32).contiguous()

    consumed = set()

    def take(k):
        consumed.add(k)
        return sd[k]

    def conv(name, split=False):
        w = take(name + ".weight")
        packed[name + ".weight"] = w16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)) if w.dim() == 4 else w16(w)
        packed[name + ".bias"] = f32(take(name + ".bias"))
        if split:
            split2(name, w)

    def split2(name, w):
        """`.weight2` of a split-operand consumer (residual mode 3): the operand is [x | lo] per tap (lo = what the fp16 rounding of the
        carried stream lost), the weight [W | W] per tap - W (hi + lo) on the unchanged two-source loader."""
        w4 = w.reshape(w.shape[0], w.shape[1], -1).permute(0, 2, 1)                      # [O, taps, I]
        packed[name + ".weight2"] = w16(torch.cat([w4, w4], 2).reshape(w.shape[0], -1))

    def dense(name, bias=True, split=False):
        w = take(name + ".weight")
        packed[name + ".weight"] = w16(w.reshape(w.shape[0], -1))
        if bias:
            packed[name + ".bias"] = f32(take(name + ".bias"))
        if split:            # the precise time-embedding path (ICD_SPLIT_TEMB): the operand is [hi | lo] of an fp32 activation
            w2 = w.reshape(w.shape[0], -1)
            packed[name + ".weight2"] = w16(torch.cat([w2, w2], 1))

    def affine(name):
        packed[name + ".weight"] = f32(take(name + ".weight"))
        packed[name + ".bias"] = f32(take(name + ".bias"))

    conv("conv_in")
    w_in = sd["conv_in.weight"].to(device)                       # [C0, 4, 3, 3] -> [C0, 3,3, 8] (channels 4..7 zero): GEMM path
    w8 = torch.zeros((w_in.shape[0], 3, 3, 8), device=device, dtype=torch.float32)
    w8[..., :4] = w_in.permute(0, 2, 3, 1).float()
    packed["conv_in.weight8"] = w16(w8.reshape(w_in.shape[0], 72))
    dense("time_embedding.linear_1", split=True)
    dense("time_embedding.linear_2", split=True)
    if cfg.time_cond_proj_dim:
        dense("time_embedding.cond_proj", bias=False)
    if cfg.add_in_dim:
        dense("add_embedding.linear_1")
        dense("add_embedding.linear_2", split=True)
        wa = sd["add_embedding.linear_1.weight"]                 # input = cat([text_embeds (fp16: exact), time sinusoids (fp32: [hi | lo])])
        packed["add_embedding.linear_1.weight2"] = w16(torch.cat([wa, wa[:, cfg.pooled_dim:]], 1))
    tw, tb = [], []
    for p, ci, co in cfg.resnet_names():
        affine(p + ".norm1")
        conv(p + ".conv1")
        affine(p + ".norm2")
        conv(p + ".conv2")
        if ci != co:
            conv(p + ".conv_shortcut", split=True)
        tw.append(take(p + ".time_emb_proj.weight"))
        tb.append(take(p + ".time_emb_proj.bias"))
    packed["time_emb_proj_cat.weight"] = w16(torch.cat([t.to(device) for t in tw], 0))
    packed["time_emb_proj_cat.weight2"] = torch.cat([packed["time_emb_proj_cat.weight"]] * 2, 1).contiguous()
    packed["time_emb_proj_cat.bias"] = f32(torch.cat([t.to(device) for t in tb], 0))
    kcat, vcat = [], []
    for p, c, depth, heads, _ in cfg.transformer_names():
        # softmax(d^-1/2 q.k) = exp2(q'.k - max) / sum with q' = (d^-1/2 log2 e) q: the factor is folded into the query projections
        # (weights, LayerNorm column sums and biases alike, before the single fp16 rounding of the weights), so the attention
        # kernels use q.k as a base-2 exponent directly - no scale multiply per score on the VALU (attention.hip MODE 1 / 2)
        qscale = (c // heads) ** -0.5 * 1.4426950408889634
        affine(p + ".norm")
        dense(p + ".proj_in")
        dense(p + ".proj_out")
        split2(p + ".proj_out", sd[p + ".proj_out.weight"])
        perm = geglu_perm(4 * c).to(device)
        for k in range(depth):
            b = f"{p}.transformer_blocks.{k}"
            # LayerNorm folded into the projections that consume it (executor: icd_layernorm_stats + icd_gemm ln_stats):
            #   LN(x) W^T + bias = rstd * (x (W*gamma)^T - mean * rowsum(W*gamma)) + (W beta + bias)
            # `.weight` = fp16(W * gamma), `.lnsum` = row sums of exactly those fp16 values, `.lnbias` / `.bias` = W beta (+ bias)
            ln = {n: (take(f"{b}.{n}.weight").to(device).float(), take(f"{b}.{n}.bias").to(device).float())
                  for n in ("norm1", "norm2", "norm3")}

            def fold(name, w, norm, bias=None, bias_key="lnbias"):
                g, be = ln[norm]
                w = w.to(device).float()
                w16 = w16_(w * g[None, :])
                packed[name + ".weight"] = w16
                packed[name + ".lnsum"] = f32(w16.float().sum(1))
                packed[name + "." + bias_key] = f32(w @ be + (0 if bias is None else bias.to(device).float()))
                return w @ be

            fold(f"{b}.attn1.to_qk", torch.cat([take(f"{b}.attn1.to_q.weight").to(device).float() * qscale,
                                                take(f"{b}.attn1.to_k.weight").to(device).float()], 0), "norm1")
            tv = fold(f"{b}.attn1.to_v", take(f"{b}.attn1.to_v.weight"), "norm1")           # W_v beta: a constant over the keys,
            del packed[f"{b}.attn1.to_v.lnbias"]                                                # (not applied by the to_v GEMM itself)
            wo = take(f"{b}.attn1.to_out.0.weight").to(device).float()                          # softmax rows sum to one ->
            packed[f"{b}.attn1.to_out.0.weight"] = w16(wo)                                      # it moves into to_out's bias
            packed[f"{b}.attn1.to_out.0.bias"] = f32(take(f"{b}.attn1.to_out.0.bias").to(device).float() + wo @ tv)
            fold(f"{b}.attn2.to_q", take(f"{b}.attn2.to_q.weight").to(device).float() * qscale, "norm2")
            kcat.append(take(f"{b}.attn2.to_k.weight"))        # every cross-attention K / V projection of the UNet
            vcat.append(take(f"{b}.attn2.to_v.weight"))        # is batched into one GEMM per forward (context-only)
            dense(f"{b}.attn2.to_out.0")
            fold(f"{b}.ff.net.0.proj", take(f"{b}.ff.net.0.proj.weight").to(device)[perm], "norm3",
                 bias=take(f"{b}.ff.net.0.proj.bias").to(device)[perm], bias_key="bias")
            dense(f"{b}.ff.net.2")
    packed["attn2_k_cat.weight"] = w16(torch.cat([t.to(device) for t in kcat], 0))
    packed["attn2_v_cat.weight"] = w16(torch.cat([t.to(device) for t in vcat], 0))
    for i in range(cfg.num_levels - 1):
        conv(f"down_blocks.{i}.downsamplers.0.conv", split=True)
        conv(f"up_blocks.{i}.upsamplers.0.conv", split=True)
        for ph, wp in enumerate(upsample_phase_weights(sd[f"up_blocks.{i}.upsamplers.0.conv.weight"].to(device))):
            packed[f"up_blocks.{i}.upsamplers.0.conv.phase.{ph}"] = w16(wp.reshape(wp.shape[0], -1))
            hi = wp.half().float()                                                   # the accurate level: [W_hi | W_hi | W_lo] per tap against
            lo = wp - hi                                                             # [h | lo | h] - the products of the exact tap sums
            packed[f"up_blocks.{i}.upsamplers.0.conv.phase3.{ph}"] = w16(torch.cat([hi, hi, lo], 2).reshape(wp.shape[0], -1))
    affine("conv_norm_out")
    conv("conv_out")
    missing = set(cfg.state_dict_shapes()) - consumed
    if missing:
        raise KeyError(f"state dict keys not consumed by the packer: {sorted(missing)[:5]}")
    return packed


class UNet2DConditionModel:
    def __init__(self, cfg: UNetConfig, state_dict, device="cuda", dtype=torch.float16):
        if not torch.cuda.is_available():
            raise RuntimeError("invertible_cd_amd.UNet2DConditionModel needs an MI355X (no CPU fallback on the product path)")
        self._lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = dtype                       # I/O dtype seen by the sampler (compute is fp16 x fp16 -> fp32 accumulate)
        self.in_channels = cfg.in_channels
        self.config = types.SimpleNamespace(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                                            sample_size=cfg.sample_size, time_cond_proj_dim=cfg.time_cond_proj_dim,
                                            cross_attention_dim=cfg.cross_dim, block_out_channels=cfg.block_out_channels,
                                            addition_time_embed_dim=cfg.addition_time_embed_dim or None)
        shapes = cfg.state_dict_shapes()
        for k, s in shapes.items():
            if k not in state_dict:
                raise KeyError(f"state dict is missing '{k}'")
            if tuple(state_dict[k].shape) != tuple(s):
                raise ValueError(f"state dict tensor '{k}' has shape {tuple(state_dict[k].shape)}, expected {s}")
        self._packed = pack_state_dict(cfg, state_dict, self.device)
        self._options = {}
        self._create_handle()

    def _create_handle(self):
        """A native executor handle bound to self._packed, plus the per-handle host state (arena, caches, plugin slot)."""
        cfg = self.cfg
        c = _lib.UNetConfig()
        c.in_channels, c.out_channels, c.num_levels = cfg.in_channels, cfg.out_channels, cfg.num_levels
        for i in range(cfg.num_levels):
            c.block_out_channels[i] = cfg.block_out_channels[i]
            c.down_has_attn[i] = int(cfg.down_has_attn[i])
            c.up_has_attn[i] = int(cfg.up_has_attn[i])
            c.transformer_layers[i] = cfg.transformer_layers[i]
            c.num_heads[i] = cfg.num_heads[i]
        c.layers_per_block, c.cross_dim = cfg.layers_per_block, cfg.cross_dim
        c.use_linear_projection = int(cfg.use_linear_projection)
        c.time_cond_proj_dim, c.addition_time_embed_dim = cfg.time_cond_proj_dim, cfg.addition_time_embed_dim
        c.add_in_dim, c.norm_groups = cfg.add_in_dim, cfg.norm_groups
        h = C.c_void_p()
        _lib.check(self._lib.icd_unet_create(C.byref(c), C.byref(h)), "icd_unet_create")
        self._h = h
        for name, t in self._packed.items():
            _lib.check(self._lib.icd_unet_set_tensor(h, name.encode(), C.c_void_p(t.data_ptr()),
                                                     0 if t.dtype == torch.float16 else 1, t.numel()), "icd_unet_set_tensor")
        _lib.check(self._lib.icd_unet_finalize(h), "icd_unet_finalize")
        self.num_attention_layers = self._lib.icd_unet_num_attention_layers(h)
        assert self.num_attention_layers == cfg.num_attention_layers
        self._ws = None
        self._ws_key = None
        self._ws_mode = 0
        self._ws_pool = {}         # (B, H, W, n_ctx) -> (arena, probs_mode, {(level, probs_mode): bytes needed}): ONE arena per shape, sized
                                   # to the largest precision level seen (the accurate level's layout is a superset of the fast one's)
        self._applied = None       # (residual mode, split mask) last sent to the native handle by the precision policy
        # p2p plugin state (set by p2p.register_attention_control / Generator.get_noise_pred)
        self.attn_controller = None
        self.attn_cond_only = False
        self._t_cache = {}
        self._live = []
        # cross-attention K / V^T of the last context (icd_unet_io.kv_cache): a sampling loop passes the same context tensor at
        # every step, so steps 2..n skip the two context projections.  The cached context is kept referenced - its storage
        # cannot be recycled under the cache - and is recognised by (data_ptr, shape, version counter): an in-place update
        # of the tensor bumps the counter and refills the cache.
        # (ICD_AMD_PRECISION=fast|split|accurate|auto: process-wide default of the policy, e.g. "accurate" for checkpoints whose residual stream
        #  has large-magnitude channels, DESIGN.md section 6 "Streams that are not O(1)")
        self.precision = getattr(self, "precision", None) or self._env_precision()
        self._inverting = 0
        # 'auto', plain generation: the level the PROBE chose for these weights (None = not probed yet; see _probe_plain_level)
        self._auto_plain = getattr(self, "_auto_plain", None)
        self._auto_gap = getattr(self, "_auto_gap", None)
        self.kv_cache_enabled = True
        self._kv = None            # (ctx tensor, version, cache buffer, stream)

    def replica(self):
        """A second executor over the SAME packed weights: its own native handle, arena, caches and plugin slot, the options of this
        one.  For a second batch in flight on another HIP stream / host thread (independent batches overlap their launch ramps and
        tails on the CUs one batch leaves idle: +4 % at 32 images, +23 % at the reference's shipped batch of 8; bench.py
        --in-flight) - the library keeps no process-wide execution state, so handles on different streams do not interact."""
        r = object.__new__(type(self))
        for k in ("_lib", "cfg", "device", "dtype", "in_channels", "config", "_packed"):
            setattr(r, k, getattr(self, k))
        r._options = {}
        r.precision = self.precision
        r._auto_plain, r._auto_gap = self._auto_plain, self._auto_gap      # the probe's verdict belongs to the weights
        r._create_handle()
        for name, value in self._options.items():
            r.set_option(name, value)
        r.precision = self.precision
        return r

    OPTIONS = {"xattn_fusion": _lib.ICD_UNET_OPT_XATTN_FUSION, "ln_inline_stats": _lib.ICD_UNET_OPT_LN_INLINE_STATS,
               "xattn_tile": _lib.ICD_UNET_OPT_XATTN_TILE, "attn_valu_scale": _lib.ICD_UNET_OPT_ATTN_VALU_SCALE,
               "residual": _lib.ICD_UNET_OPT_RESIDUAL_MODE, "residual_f32": _lib.ICD_UNET_OPT_RESIDUAL_MODE,
               "split_mask": _lib.ICD_UNET_OPT_SPLIT_MASK, "upsample_phases": _lib.ICD_UNET_OPT_UPSAMPLE_PHASES,
               "gemm_tune": _lib.ICD_UNET_OPT_GEMM_TUNE}

    # ------------------------------------------------------------------ precision policy
    # Numerical precision of the residual stream and of its consumers (icd_unet options residual / split_mask):
    #   "fast"      error carry (mode 2): one evaluation 0.70 - 0.85e-3 from an fp32 evaluation - enough for the REVERSE loops, which
    #               contract the per-step error (4-step generation: latents 3e-4, every attention-store tensor < 1e-3);
    #   "accurate"  carry + split consumers incl. the upsampler convs (mode 3, ICD_SPLIT_ACCURATE = every bit): 0.39 - 0.41e-3, +11 % / +5 % time
    #               (SD1.5 / SDXL) - what the FORWARD (inversion) loops, which amplify the per-step error by ~2.4 x over 3 - 4 steps, and
    #               the edit passes behind them need to stay inside 1e-3;
    #   "auto"      (default) accurate inside EDITING pipelines - the evaluations a sampler wraps in `with unet.editing():` (inversion
    #               loops, reverse passes under dynamic guidance: the reference uses both for editing only) and every evaluation with an
    #               attention controller attached (edit / store passes) - fast for plain text-to-image generation;
    #   None        leave the options alone (set_option('residual' | 'split_mask') switches to this).
    PRECISION = {"fast": (_lib.ICD_RESIDUAL_CARRY, _lib.ICD_SPLIT_DEFAULT), "accurate": (_lib.ICD_RESIDUAL_SPLIT, _lib.ICD_SPLIT_ACCURATE),
                 "split": (_lib.ICD_RESIDUAL_SPLIT, _lib.ICD_SPLIT_DEFAULT)}

    @classmethod
    def _env_precision(cls):
        import os
        level = os.environ.get("ICD_AMD_PRECISION", "auto")
        if level != "auto" and level not in cls.PRECISION:
            raise ValueError(f"ICD_AMD_PRECISION must be one of auto, fast, split, accurate (got {level!r})")
        return level

    def set_precision(self, level):
        if level not in (None, "auto") and level not in self.PRECISION:
            raise ValueError(f"precision must be one of auto, fast, split, accurate or None (got {level!r})")
        self.precision = level
        return self

    def editing(self):
        """Context manager marking the evaluations inside as part of an editing pipeline (inversion steps, dynamic-guidance reverse
        passes) for the 'auto' precision policy."""
        unet = self

        class _Inv:
            def __enter__(self_):
                unet._inverting += 1

            def __exit__(self_, *exc):
                unet._inverting -= 1
        return _Inv()

    def _apply_precision(self):
        level = getattr(self, "precision", "auto")
        if level is None:
            return
        if level == "auto":
            level = "accurate" if (self._inverting > 0 or self.attn_controller is not None) else (self._auto_plain or "fast")
        want = self.PRECISION[level]
        if self._applied != want:
            _lib.check(self._lib.icd_unet_set_option(self._h, _lib.ICD_UNET_OPT_RESIDUAL_MODE, want[0]), "icd_unet_set_option(residual)")
            _lib.check(self._lib.icd_unet_set_option(self._h, _lib.ICD_UNET_OPT_SPLIT_MASK, want[1]), "icd_unet_set_option(split_mask)")
            self._applied = want
            self._ws_key = None
            self._kv = None                      # (the cached context projections carry their error byte only at the split levels)

    # 'auto' is data-aware (round 6): whether the FAST level is good enough for plain generation depends on the checkpoint - residual
    # streams with channels beyond ~2^11 (real SD checkpoints have them; a fused LoRA widens the activations further) push one fast
    # evaluation past 1e-3 of the fp32 result while the accurate level stays at ~0.4e-3.  The first plain evaluation of a model therefore
    # runs BOTH levels on its own inputs and compares them (one device->host scalar, once per set of weights; replicas inherit the verdict):
    # the gap between the two IS the fast level's excess error, measured on the data instead of predicted from a proxy such as the
    # stream's absolute maximum.  Above AUTO_ESCALATE_GAP the policy uses the accurate level for plain generation too.
    # (measured gaps, tools/precision_gap.py -> profiles/r06_precision_gap.txt: 0.73 - 0.92e-3 on every synthetic checkpoint of the tests and of
    #  the benchmark, plain or LoRA-fused - the fast level's own 0.70 - 0.85e-3 and the accurate level's 0.40e-3 in quadrature; 1.09e-3 on the
    #  loader test's rank-16 LoRA net whose fast evaluation is 1.04e-3 from the oracle; 1.9e-3 with conv_in channels offset by thousands)
    AUTO_ESCALATE_GAP = 1.0e-3

    def _probe_plain_level(self, args, kwargs):
        saved = self.precision
        try:
            self.precision = "accurate"
            e_acc = self.__call__(*args, **kwargs)
            self.precision = "fast"
            e_fast = self.__call__(*args, **kwargs)
        finally:
            self.precision = saved
        a, f = (e_acc.sample if hasattr(e_acc, "sample") else e_acc[0]), (e_fast.sample if hasattr(e_fast, "sample") else e_fast[0])
        gap = float((f.float() - a.float()).norm() / a.float().norm().clamp_min(1e-30))
        self._auto_gap = gap
        self._auto_plain = "fast" if gap <= self.AUTO_ESCALATE_GAP else "accurate"      # (NaN compares false: accurate)
        return e_fast if self._auto_plain == "fast" else e_acc

    def set_option(self, name, value):
        """Per-handle execution option (icd_unet_set_option): 'xattn_fusion' 0 / 1 / 2, 'ln_inline_stats' 0 / 1, 'xattn_tile' 0 / 2 / 4 / 5 / 6,
        'residual' 0 fp16 stream / 1 fp32 twin / 2 error carry / 3 carry + split consumers (default; 'residual_f32' is the round-3
        name of the same option).  A/B
        tuning and tests; nothing is process-wide."""
        _lib.check(self._lib.icd_unet_set_option(self._h, self.OPTIONS[name], int(value)), f"icd_unet_set_option({name})")
        name = {"residual_f32": "residual"}.get(name, name)      # aliases are recorded under one key: a replica replays the final state
        self._options.pop(name, None)
        self._options[name] = int(value)
        if name in ("residual", "split_mask"):
            self.precision = None                # an explicit setting switches the policy off
            self._applied = None
            self._kv = None
            self._ws_pool.clear()
            self._ws_key = None                  # the arena holds the twins / carries of the residual stream: size it again
        return self

    # ------------------------------------------------------------------ duck-typed nn.Module surface
    def named_children(self):
        return []

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.dtype):
                self.dtype = a
        return self

    def eval(self):
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.icd_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ forward
    def _workspace(self, B, H, W, n_ctx, probs_mode):
        """Arena for one forward, sized by the materialisation rule of the attached controller (0 none, 1 the shipped
        controllers' rule, 2 any layer): only grows, so switching controllers does not thrash the allocator.  The precision levels of
        one shape SHARE an arena (round 6; the 'auto' policy alternates between them inside one editing pipeline, and two arenas of
        multiple GB each could tip a near-capacity run over): it is sized to the largest level asked for so far."""
        shape = (B, H, W, n_ctx)
        key = shape + (self._applied or (None, None))
        if self._ws_key != key or probs_mode > self._ws_mode:
            hit = self._ws_pool.get(shape)
            need_key = (self._applied, probs_mode)
            need = hit[2].get(need_key) if hit is not None else None
            if need is None:
                need = self._lib.icd_unet_workspace_bytes_ex(self._h, B, H, W, n_ctx, probs_mode)
                if need <= 0:
                    raise RuntimeError("icd_unet_workspace_bytes failed")
            if hit is not None and hit[0].numel() >= need:
                hit[2][need_key] = need
                self._ws, self._ws_mode = hit[0], probs_mode
            else:
                sizes = dict(hit[2]) if hit is not None else {}
                sizes[need_key] = need
                self._ws = None
                self._ws_pool.clear()                # one shape at a time, not a history of shapes; the old arena goes back to the allocator first
                hit = None
                self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
                self._ws_mode = probs_mode
                self._ws_pool[shape] = (self._ws, probs_mode, sizes)
            self._ws_key = key
        return self._ws

    def _make_hook(self, errors, batch=0):
        ctrl = self.attn_controller
        if ctrl is None:
            return _lib.ATTN_HOOK(0), 0
        from . import p2p
        adapter = p2p.HookAdapter(ctrl, self.attn_cond_only, self.device, batch=batch)
        live = self._live

        def hook(user, phase, layer, is_cross, place, bh, nq, nk, ld, probs_pp):
            try:
                if phase == _lib.ICD_HOOK_QUERY:
                    buf = adapter.query(layer, bool(is_cross), PLACES[place], bh, nq, nk, ld)
                    if buf is None:
                        return 0
                    live.append(buf)
                    probs_pp[0] = buf.data_ptr()
                    if adapter.epilogue is not None:         # the controller's work on P rides in the kernel's epilogue
                        probs_pp[1] = C.addressof(adapter.epilogue)
                    return 2 if adapter.half else 1          # 2: `buf` holds the second half of the batch only (the conditional rows)
                adapter.probs_ready(layer, bool(is_cross), PLACES[place])
                return 0
            except BaseException as e:       # never let an exception cross the C boundary
                errors.append(e)
                return -1
        return _lib.ATTN_HOOK(hook), adapter.probs_mode

    def reset_context_cache(self):
        """Forget the cached context K / V projections: the next forward recomputes them whatever tensor it is handed.  (The cache
        keys on the context tensor's identity and version counter; a caller that rewrites ONE tensor object through a raw pointer - a
        ctypes kernel of this library writing into it as an `out=` buffer, a DLPack alias - or a benchmark that wants every batch to
        pay for its projections calls this at the batch boundary.  In-place torch ops bump the version counter and need no call.)"""
        self._kv = None

    @staticmethod
    def _ctx_version(ctx):
        """Version counter of the context tensor, or None where torch does not track one (inference tensors: reading `_version`
        raises) - None never hits the cache."""
        try:
            if ctx.is_inference():
                return None
            return ctx._version
        except RuntimeError:
            return None

    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states=None, class_labels=None, timestep_cond=None,
                 attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=True, **kwargs):
        if encoder_hidden_states is None:
            raise ValueError("encoder_hidden_states is required")
        if (self.precision == "auto" and self._auto_plain is None and self._inverting == 0 and self.attn_controller is None
                and str(self.device).startswith("cuda") and not torch.cuda.is_current_stream_capturing()):
            return self._probe_plain_level((sample, timestep), dict(encoder_hidden_states=encoder_hidden_states, class_labels=class_labels,
                                           timestep_cond=timestep_cond, attention_mask=attention_mask, cross_attention_kwargs=cross_attention_kwargs,
                                           added_cond_kwargs=added_cond_kwargs, return_dict=return_dict, **kwargs))
        if sample.dim() != 4 or sample.shape[1] != self.cfg.in_channels:
            raise ValueError(f"sample must be [B,{self.cfg.in_channels},H,W], got {tuple(sample.shape)}")
        dev = self.device
        io_dtype = sample.dtype if sample.dtype in (torch.float16, torch.float32) else torch.float16
        x = sample.to(device=dev, dtype=io_dtype).contiguous()
        B, _, H, W = x.shape
        if torch.is_tensor(timestep) and timestep.is_cuda:
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
            t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()
        elif torch.is_tensor(timestep) and timestep.numel() > 1:
            t = timestep.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        else:
            # scalar host timestep (the loops pass CPU ints): one cached device tensor per (value, B) - no host->device
            # copy per step, and nothing pageable inside a hipGraph capture
            key = (float(timestep), B)
            t = self._t_cache.get(key)
            if t is None:
                if len(self._t_cache) > 4096:
                    self._t_cache.clear()
                t = self._t_cache[key] = torch.full((B,), key[0], device=dev, dtype=torch.float32)
        ctx = encoder_hidden_states.to(device=dev, dtype=torch.float16).contiguous()
        if ctx.shape[0] != B or ctx.shape[2] != self.cfg.cross_dim:
            raise ValueError(f"encoder_hidden_states must be [{B}, n, {self.cfg.cross_dim}], got {tuple(ctx.shape)}")
        n_ctx = ctx.shape[1]
        io = _lib.UNetIO()
        keep = [x, t, ctx]
        if timestep_cond is not None:
            if not self.cfg.time_cond_proj_dim:
                raise ValueError("this UNet has no time_cond_proj (w-embedding) input")
            cond = timestep_cond.to(device=dev, dtype=torch.float16).contiguous()
            if tuple(cond.shape) != (B, self.cfg.time_cond_proj_dim):
                raise ValueError(f"timestep_cond must be [{B}, {self.cfg.time_cond_proj_dim}]")
            keep.append(cond)
            io.timestep_cond = cond.data_ptr()
        if self.cfg.addition_time_embed_dim:
            if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
                raise ValueError("SDXL UNet needs added_cond_kwargs={'text_embeds','time_ids'}")
            te = added_cond_kwargs["text_embeds"].to(device=dev, dtype=torch.float16).contiguous()
            ti = added_cond_kwargs["time_ids"].to(device=dev, dtype=torch.float32).contiguous()
            keep += [te, ti]
            io.text_embeds, io.time_ids = te.data_ptr(), ti.data_ptr()
        eps = torch.empty_like(x)
        self._apply_precision()
        errors = []
        self._live.clear()
        hook, probs_mode = self._make_hook(errors, B)
        ws = self._workspace(B, H, W, n_ctx, probs_mode)
        io.sample, io.timesteps, io.context, io.eps = x.data_ptr(), t.data_ptr(), ctx.data_ptr(), eps.data_ptr()
        io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel()
        io.batch, io.H, io.W, io.n_ctx = B, H, W, n_ctx
        io.sample_is_f32 = int(io_dtype == torch.float32)
        io.hook = hook
        if self.kv_cache_enabled:
            kv, stream_id = self._kv, torch.cuda.current_stream().cuda_stream
            ver = self._ctx_version(ctx)
            # a hit needs the SAME storage, layout and version counter as the forward that filled the cache, on the same stream (filled and
            # read in stream order).  Tensors without a version counter (created under torch.inference_mode()) never hit.  The key cannot
            # see writes made through raw pointers - e.g. this library's own kernels writing into an `out=` tensor that is later handed
            # in as the context: such callers call reset_context_cache() (see its docstring).
            hit = (kv is not None and ver is not None and kv[0].data_ptr() == ctx.data_ptr() and kv[0].shape == ctx.shape
                   and kv[0].stride() == ctx.stride() and kv[1] == ver and kv[3] == stream_id)
            if not hit:
                nbytes = self._lib.icd_unet_kv_cache_bytes(self._h, B, n_ctx)
                buf = kv[2] if kv is not None and kv[2].numel() == nbytes else torch.empty((nbytes,), dtype=torch.uint8, device=dev)
                self._kv = kv = (ctx, ver, buf, stream_id)
            io.kv_cache, io.kv_cache_bytes, io.kv_cache_valid = kv[2].data_ptr(), kv[2].numel(), int(hit)
        rc = self._lib.icd_unet_forward(self._h, C.byref(io), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        # the probability buffers handed in by the hook stay referenced until the NEXT call (cleared at its start): the launch is
        # asynchronous, and freeing them here would rely on every later consumer of the allocator running on this same stream
        if rc != 0 or errors:
            self._kv = None                      # a failed forward (also one aborted by a hook's exception) may have left the cache half written
        if errors:
            raise errors[0]
        _lib.check(rc, "icd_unet_forward")
        if not return_dict:
            return (eps,)
        return UNetOutput(eps)

    forward = __call__
