"""Prompt-to-prompt attention-control plugin for the native UNet (API mirror of the reference's utils/p2p.py).

Same module globals (NUM_DDIM_STEPS / tokenizer / device / LOW_RESOURCE / MAX_NUM_WORDS, utils/p2p.py:9-13), same classes
(AttentionStore, AttentionReplace, AttentionRefine, AttentionReweight, LocalBlend, EmptyControl, SpatialReplace) and the
same entry points (register_attention_control, make_controller, get_equalizer, get_word_inds,
get_time_words_attention_alpha).  Semantics are pinned by golden vectors captured from the reference
(tests/golden/p2p.npz, tests/test_p2p_golden.py).

What is different underneath: the reference monkey-patches every diffusers `Attention.forward` (utils/p2p.py:291-386).
Here `register_attention_control` hands the controller to the native executor, which calls back (C ABI hook,
include/icd_amd.h) once per Attention module in module-execution order; `HookAdapter` decides per layer whether the
probabilities must be materialised at all:
  * layers the controller does not read or edit (`needs_probs` False, e.g. N > 32^2 self-attention for every shipped
    controller, utils/p2p.py:147,184-188) run the fused flash kernel and only tick the controller's counters;
  * when the sampler eliminated the dead unconditional half of the CFG batch (utils/generation.py:247-251 discards it
    when w_embed_dim > 0) the controller's `forward` - which in the reference only ever sees the conditional half,
    utils/p2p.py:106-107 - is applied to the whole (cond-only) tensor.
"""
import abc
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as nnf

from . import seq_aligner

MAX_NUM_WORDS = 77
LOW_RESOURCE = False
NUM_DDIM_STEPS = 50
device = "cuda"
tokenizer = None


# ------------------------------------------------------------------------------------------- latent blending
class LocalBlend:
    """Word-localised latent blending from the accumulated 16x16 cross-attention maps (utils/p2p.py:18-70)."""

    def __init__(self, prompts: List[str], words, substruct_words=None, start_blend=0.2, th=(.3, .3)):
        self.alpha_layers = self._word_mask(prompts, words).to(device)
        self.substruct_layers = None if substruct_words is None else self._word_mask(prompts, substruct_words).to(device)
        self.start_blend = int(start_blend * NUM_DDIM_STEPS)
        self.counter = 0
        self.th = th

    @staticmethod
    def _word_mask(prompts, words):
        mask = torch.zeros(len(prompts), 1, 1, 1, 1, MAX_NUM_WORDS)
        for i, (prompt, ws) in enumerate(zip(prompts, words)):
            for w in ([ws] if isinstance(ws, str) else ws):
                mask[i, :, :, :, :, get_word_inds(prompt, w, tokenizer)] = 1
        return mask

    @staticmethod
    def _on_device(x_t, layers):
        """The HIP kernel takes the accumulated store tensors as they are: fp16 on the GPU, [P*heads, 256, 77] with evenly
        strided rows.  Anything else (CPU goldens, foreign dtypes) takes the torch expression below."""
        if not (x_t.is_cuda and x_t.dim() == 4 and x_t.dtype in (torch.float16, torch.float32)):
            return False
        for m in layers:
            if not (m.is_cuda and m.dtype == torch.float16 and m.dim() == 3 and m.shape[1] == 256 and m.shape[2] == MAX_NUM_WORDS
                    and m.stride(2) == 1 and m.stride(1) == layers[0].stride(1) and m.stride(0) == 256 * m.stride(1)):
                return False
        return len(layers) <= 8

    def get_mask(self, maps, alpha, use_pool, x_t):
        """Word-weighted mean over layers x heads of the 16x16 maps -> (3x3 max-pool) -> nearest resize to the latent ->
        per-prompt max normalisation -> threshold; every prompt's mask is OR-ed with the base prompt's (utils/p2p.py:20-31)."""
        heat = (maps * alpha).sum(-1).mean(1)                              # [P, 1, 16, 16]
        if use_pool:
            heat = nnf.max_pool2d(heat, kernel_size=3, stride=1, padding=1)
        heat = nnf.interpolate(heat, size=(x_t.shape[2:]))
        heat = heat / heat.amax(dim=(2, 3), keepdim=True)
        on = heat.gt(self.th[1 - int(use_pool)])
        return on[:1] + on

    def __call__(self, x_t, attention_store):
        """`x_t[e] <- x_t[0] + mask[e] * (x_t[e] - x_t[0])` from the accumulated 16x16 cross-attention maps of
        down_cross[2:4] + up_cross[:3] (utils/p2p.py:33-44)."""
        self.counter += 1
        if self.counter <= self.start_blend:
            return x_t
        n_prompts = self.alpha_layers.shape[0]
        layers = attention_store["down_cross"][2:4] + attention_store["up_cross"][:3]
        if self._on_device(x_t, layers):
            # one HIP launch (icd_local_blend): word-weighted mean of the maps, pool, normalise, threshold, OR with the base
            # prompt's mask, substruct words, nearest resize and the blend itself
            from . import ops
            sub = None if self.substruct_layers is None else self.substruct_layers.reshape(n_prompts, MAX_NUM_WORDS)
            return ops.local_blend(layers, self.alpha_layers.reshape(n_prompts, MAX_NUM_WORDS), sub, self.th[0], self.th[1],
                                   x_t.contiguous())
        maps = torch.cat([m.reshape(n_prompts, -1, 1, 16, 16, MAX_NUM_WORDS) for m in layers], dim=1)
        mask = self.get_mask(maps, self.alpha_layers, True, x_t)
        if self.substruct_layers is not None:
            mask = mask * ~self.get_mask(maps, self.substruct_layers, False, x_t)
        base = x_t[:1]
        return base + mask.float() * (x_t - base)


# ------------------------------------------------------------------------------------------- controllers
class EmptyControl:
    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    def needs_probs(self, is_cross, n_queries, place_in_unet):
        return False

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        return attn


class AttentionControl(abc.ABC):
    """Counter logic of utils/p2p.py:85-122.  `forward` receives the conditional rows only."""

    def __init__(self):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0

    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    @property
    def num_uncond_att_layers(self):
        return self.num_att_layers if LOW_RESOURCE else 0

    @abc.abstractmethod
    def forward(self, attn, is_cross: bool, place_in_unet: str):
        raise NotImplementedError

    def needs_probs(self, is_cross, n_queries, place_in_unet):
        """May `forward` read or modify the probabilities of such a layer?  Unknown subclasses: always."""
        return True

    def _advance(self):
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()

    def __call__(self, attn, is_cross: bool, place_in_unet: str):
        """CFG-doubled layout [uncond rows ; cond rows] (utils/p2p.py:101-113): edit the second half in place."""
        if self.cur_att_layer >= self.num_uncond_att_layers:
            if LOW_RESOURCE:
                attn = self.forward(attn, is_cross, place_in_unet)
            else:
                half = attn.shape[0] // 2
                attn[half:] = self.forward(attn[half:], is_cross, place_in_unet)
        self._advance()
        return attn

    def call_cond_only(self, attn, is_cross: bool, place_in_unet: str):
        """The whole tensor is the conditional half (dead unconditional rows were never computed)."""
        if self.cur_att_layer >= self.num_uncond_att_layers:
            new = self.forward(attn, is_cross, place_in_unet)
            if new is not attn:
                attn.copy_(new)
        self._advance()
        return attn

    def tick(self):
        """A layer whose probabilities this controller neither reads nor edits was executed (fused kernel)."""
        self._advance()

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


class SpatialReplace(EmptyControl):
    def __init__(self, stop_inject: float):
        super().__init__()
        self.stop_inject = int((1 - stop_inject) * NUM_DDIM_STEPS)

    def step_callback(self, x_t):
        if self.cur_step < self.stop_inject:
            x_t = x_t[:1].expand(x_t.shape[0], *x_t.shape[1:])
        return x_t


class AttentionStore(AttentionControl):
    """Keeps (views of) the conditional probabilities of every layer with <= 32^2 queries and sums them over steps
    (utils/p2p.py:138-173)."""

    STORE_MAX_QUERIES = 32 ** 2

    def __init__(self):
        super().__init__()
        self.step_store = self.get_empty_store()
        self.attention_store = {}
        self._fused_acc = set()        # (key, index) of this step's maps the probability kernel has already added to attention_store

    @staticmethod
    def get_empty_store():
        return {f"{place}_{kind}": [] for kind in ("cross", "self") for place in ("down", "mid", "up")}

    def needs_probs(self, is_cross, n_queries, place_in_unet):
        return n_queries <= self.STORE_MAX_QUERIES

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        if attn.shape[1] <= self.STORE_MAX_QUERIES:
            self.step_store[f"{place_in_unet}_{'cross' if is_cross else 'self'}"].append(attn)
        return attn

    def between_steps(self):
        if not self.attention_store:
            self.attention_store = self.step_store
        else:
            fused = self._fused_acc            # added by the probability kernel's epilogue (HookAdapter, icd_probs_epilogue.acc)
            pairs = [(t, self.step_store[key][i]) for key, acc in self.attention_store.items() for i, t in enumerate(acc)
                     if (key, i) not in fused]
            if pairs and all(t.is_cuda and t.dtype == torch.float16 and t.is_contiguous() and s.is_cuda and s.dtype == torch.float16
                             and s.is_contiguous() and t.shape == s.shape and t.data_ptr() % 16 == 0 and s.data_ptr() % 16 == 0
                             for t, s in pairs):
                from . import ops                  # one launch for all stored maps (icd_accumulate_multi), same rounding as `t += s`
                ops.accumulate_multi([t for t, _ in pairs], [s for _, s in pairs])
            else:
                for t, s in pairs:
                    t += s
        self.step_store = self.get_empty_store()
        self._fused_acc = set()

    def get_average_attention(self):
        return {key: [t / self.cur_step for t in ts] for key, ts in self.attention_store.items()}

    def reset(self):
        super().reset()
        self.step_store = self.get_empty_store()
        self.attention_store = {}
        self._fused_acc = set()
        self._epi_acc_ok = {}


class AttentionControlEdit(AttentionStore, abc.ABC):
    """Cross/self attention injection between a base prompt (row group 0) and its edits (utils/p2p.py:176-221)."""

    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, local_blend: Optional[LocalBlend]):
        super().__init__()
        self.batch_size = len(prompts)
        self.cross_replace_alpha = get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer).to(device)
        if type(self_replace_steps) is float:
            self_replace_steps = 0, self_replace_steps
        self.num_self_replace = int(num_steps * self_replace_steps[0]), int(num_steps * self_replace_steps[1])
        self.local_blend = local_blend

    def needs_probs(self, is_cross, n_queries, place_in_unet):
        return is_cross or n_queries <= self.STORE_MAX_QUERIES

    def step_callback(self, x_t):
        return x_t if self.local_blend is None else self.local_blend(x_t, self.attention_store)

    def replace_self_attention(self, attn_base, att_replace, place_in_unet):
        if att_replace.shape[2] > self.STORE_MAX_QUERIES:
            return att_replace
        return attn_base.unsqueeze(0).expand(att_replace.shape[0], *attn_base.shape)

    @abc.abstractmethod
    def replace_cross_attention(self, attn_base, att_replace):
        raise NotImplementedError

    # ---- the same edit as one linear operator per step (device path: ops.p2p_cross_edit, csrc/p2p.hip) -------------
    def cross_edit_terms(self, dev):
        """(A0 [P-1, T, T], D0 [P-1, T]) fp32 on `dev` with  replace_cross_attention(base, cur)[b] == base . A0[b] + D0[b] * cur[b]."""
        raise NotImplementedError

    def cross_edit_operator(self, step, dev):
        """The packed operator (ops.p2p_pack_operator) of  a_t * replace_cross_attention(base, cur) + (1 - a_t) * cur  for
        this step, cached on `dev`."""
        key = (int(step), str(dev))
        cache = self.__dict__.setdefault("_edit_ops", {})
        if key not in cache:                 # built on `dev` with torch ops: no host round trip (a sync inside the UNet's
            A0, D0 = self.cross_edit_terms(dev)                           # hook callbacks would serialise CPU and GPU)
            a = self.cross_replace_alpha[step].reshape(self.batch_size - 1, -1).to(dev, torch.float32)       # [P-1, T]
            from . import ops
            cache[key] = ops.p2p_pack_operator(A0 * a[:, None, :], D0 * a + (1.0 - a))
        return cache[key]

    def forward(self, attn, is_cross: bool, place_in_unet: str):
        super().forward(attn, is_cross, place_in_unet)          # stores a VIEW: the stored tensor sees the edit below
        if getattr(self, "_kernel_edit", False):                # the probability kernel's epilogue has applied this layer's edit already
            return attn
        in_self_window = self.num_self_replace[0] <= self.cur_step < self.num_self_replace[1]
        if not (is_cross or in_self_window):
            return attn
        if is_cross and attn.is_cuda:
            from . import ops                                    # fused in-place HIP kernel (matrix cores, fp32 accumulate)
            if ops.p2p_cross_edit_supported(attn):               # the executor's 80-column rows; else the torch path below
                At, Dp = self.cross_edit_operator(self.cur_step, attn.device)
                return ops.p2p_cross_edit(attn, self.batch_size, At, Dp)
        heads = attn.shape[0] // self.batch_size
        grouped = attn.reshape(self.batch_size, heads, *attn.shape[1:])
        base, edits = grouped[0], grouped[1:]
        if is_cross:
            a = self.cross_replace_alpha[self.cur_step]
            grouped[1:] = self.replace_cross_attention(base, edits) * a + (1 - a) * edits
        else:
            grouped[1:] = self.replace_self_attention(base, edits, place_in_unet)
        return grouped.reshape(self.batch_size * heads, *grouped.shape[2:])


class AttentionReplace(AttentionControlEdit):
    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, local_blend: Optional[LocalBlend] = None):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend)
        self.mapper = seq_aligner.get_replacement_mapper(prompts, tokenizer).to(device)

    def replace_cross_attention(self, attn_base, att_replace):
        return torch.einsum("hpw,bwn->bhpn", attn_base, self.mapper.to(attn_base.dtype))

    def cross_edit_terms(self, dev):
        m = self.mapper.to(dev, torch.float32)
        return m, torch.zeros(m.shape[0], m.shape[2], device=dev)


class AttentionRefine(AttentionControlEdit):
    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, local_blend: Optional[LocalBlend] = None):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend)
        mapper, alphas = seq_aligner.get_refinement_mapper(prompts, tokenizer)
        self.mapper = mapper.to(device)
        self.alphas = alphas.to(device).reshape(alphas.shape[0], 1, 1, alphas.shape[1])

    def replace_cross_attention(self, attn_base, att_replace):
        gathered = attn_base[:, :, self.mapper].permute(2, 0, 1, 3)
        return gathered * self.alphas + att_replace * (1 - self.alphas)

    def cross_edit_terms(self, dev):
        idx = self.mapper.to(dev, torch.int64)                                 # [P-1, T]; -1 indexes the last token
        al = self.alphas.reshape(idx.shape[0], -1).to(dev, torch.float32)     # [P-1, T]
        T = idx.shape[1]
        A0 = torch.zeros(idx.shape[0], T, T, device=dev)
        A0.scatter_(1, (idx % T)[:, None, :], al[:, None, :])                 # A0[b, idx[b, n], n] = alphas[b, n]
        return A0, 1.0 - al


class AttentionReweight(AttentionControlEdit):
    def __init__(self, prompts, num_steps: int, cross_replace_steps, self_replace_steps, equalizer,
                 local_blend: Optional[LocalBlend] = None, controller: Optional[AttentionControlEdit] = None):
        super().__init__(prompts, num_steps, cross_replace_steps, self_replace_steps, local_blend)
        self.equalizer = equalizer.to(device)
        self.prev_controller = controller
        self.attn = []

    def replace_cross_attention(self, attn_base, att_replace):
        if self.prev_controller is not None:
            attn_base = self.prev_controller.replace_cross_attention(attn_base, att_replace)
        return attn_base[None, :, :, :] * self.equalizer[:, None, None, :]

    def cross_edit_terms(self, dev):
        eq = self.equalizer.to(dev, torch.float32)                            # [P-1, T]
        if self.prev_controller is not None:
            A0, D0 = self.prev_controller.cross_edit_terms(dev)
        else:
            T = eq.shape[1]
            A0, D0 = torch.eye(T, device=dev).expand(eq.shape[0], T, T), torch.zeros(eq.shape[0], T, device=dev)
        return A0 * eq[:, None, :], D0 * eq


def make_controller(prompts: List[str], is_replace_controller: bool, cross_replace_steps, self_replace_steps,
                    blend_words=None, equilizer_params=None) -> AttentionControlEdit:
    """utils/p2p.py:272-289 (argument names kept, incl. `equilizer_params`)."""
    lb = None if blend_words is None else LocalBlend(prompts, blend_words, start_blend=0.0, th=(0.3, 0.3))
    cls = AttentionReplace if is_replace_controller else AttentionRefine
    controller = cls(prompts, NUM_DDIM_STEPS, cross_replace_steps=cross_replace_steps, self_replace_steps=self_replace_steps,
                     local_blend=lb)
    if equilizer_params is not None:
        eq = get_equalizer(prompts[1], equilizer_params["words"], equilizer_params["values"])
        controller = AttentionReweight(prompts, NUM_DDIM_STEPS, cross_replace_steps=cross_replace_steps,
                                       self_replace_steps=self_replace_steps, equalizer=eq, local_blend=lb, controller=controller)
    return controller


# ------------------------------------------------------------------------------------------- registration
class DummyController:
    """What the reference installs for `controller=None` (utils/p2p.py:356-365)."""

    def __init__(self):
        self.num_att_layers = 0

    def __call__(self, *args):
        return args[0]


def register_attention_control(model, controller):
    """Attach `controller` to `model.unet` (utils/p2p.py:291-386).  Sets controller.num_att_layers to the number of
    Attention modules (32 for SD1.5, 140 for SDXL), exactly as the reference's module walk does."""
    if controller is None:
        controller = DummyController()
    unet = model.unet
    if hasattr(unet, "num_attention_layers"):
        unet.attn_controller = None if isinstance(controller, DummyController) else controller
        controller.num_att_layers = unet.num_attention_layers
    else:        # a foreign UNet object (e.g. a stub in tests): nothing to hook, the reference would count 0 modules
        controller.num_att_layers = 0


class HookAdapter:
    """Bridges the executor's C callback (query / probs-ready per Attention module) to a controller object.

    Constraint on what a controller may return for SELF-attention layers: every row of P must still sum to one.  The
    executor folds norm1's beta through attn1.to_v into attn1.to_out's bias at load time (unet.pack_state_dict: P (V + 1 w^T)
    = P V + 1 w^T only when P 1 = 1), so an edit that rescales or un-normalises self-attention rows would see the reference's
    P (V + W_v beta) replaced by P V + W_o W_v beta.  Every shipped controller satisfies it (self-attention edits copy whole
    softmax rows, utils/p2p.py:183-188; Reweight only touches cross-attention, whose V carries no LayerNorm).  Cross-attention
    probabilities are unconstrained."""

    def __init__(self, controller, cond_only: bool, dev, batch: int = 0):
        self.c = controller
        self.cond_only = cond_only
        self.dev = dev
        self.native = self._trusts_needs_probs(controller)
        # arena sizing hint for the executor (icd_unet_workspace_bytes_ex): 1 = the shipped controllers' rule, 2 = any layer
        self.probs_mode = 1 if self.native else 2
        self.pending = None
        # round 5: for the shipped controllers (exactly these classes) what they do to P - accumulate into the store, copy the base prompt's
        # self-attention rows, the cross-attention edit operator - rides in the probability kernel's epilogue (icd_probs_epilogue): one pass
        # over P instead of three.  `controller.fused_epilogue = False` keeps the separate passes (A/B; bit-identical results).
        self.fuse = (self.native and type(controller) in (AttentionStore, AttentionReplace, AttentionRefine, AttentionReweight)
                     and getattr(controller, "fused_epilogue", True) and not LOW_RESOURCE and str(dev).startswith("cuda"))
        self.epilogue = None
        self._epi_refs = None
        self._epi_acc_key = None
        self._epi_edit = False
        # round 5: a controller with the reference's __call__ touches the second half of a [uncond; cond] batch only (utils/p2p.py:153-155).
        # The executor then writes P for those samples alone and runs the unconditional half through the fused kernel (hook return 2):
        # half the probabilities written, read back by P.V and held by the caching allocator.  `controller.cond_rows_only = False` keeps
        # the whole batch materialised (A/B; the conditional rows and every stored tensor are the same bits either way).
        self.half_ok = (self.native and not cond_only and not LOW_RESOURCE and isinstance(controller, AttentionControl)
                        and type(controller).__call__ is AttentionControl.__call__ and batch > 0 and batch % 2 == 0
                        and getattr(controller, "cond_rows_only", True) and str(dev).startswith("cuda"))
        self.half = False
        self.batch = batch                         # samples of the UNet batch ([uncond; cond] when not cond_only); 0 = unknown

    @staticmethod
    def _trusts_needs_probs(c):
        """`needs_probs` may only short-cut a layer when it describes the `forward` that will actually run: the shipped
        forwards (this module), or a subclass that overrides `forward` AND states its own `needs_probs` at or below it in
        the MRO.  A reference-style subclass that only overrides `forward` gets every layer, as utils/p2p.py:336 gives it."""
        if isinstance(c, EmptyControl):
            return True
        if not isinstance(c, AttentionControl):
            return False
        mro = type(c).__mro__
        fwd_owner = next(k for k in mro if "forward" in k.__dict__)
        np_owner = next(k for k in mro if "needs_probs" in k.__dict__)
        if fwd_owner.__module__ == __name__:
            return True
        return mro.index(np_owner) <= mro.index(fwd_owner)

    def query(self, layer, is_cross, place, bh, nq, nk, ld):
        c = self.c
        if self.native and not c.needs_probs(is_cross, nq, place):
            if isinstance(c, AttentionControl):
                c.tick()
            return None
        # a FRESH buffer per layer call: AttentionStore keeps views of it alive across steps (utils/p2p.py:148,153-157)
        self.half = self.half_ok and bh % 2 == 0
        rows = bh // 2 if self.half else bh
        buf = torch.empty((rows, nq, ld), dtype=torch.float16, device=self.dev)
        self.pending = buf[:, :, :nk]
        self.epilogue = self._plan_epilogue(is_cross, place, rows, nq, nk, ld) if self.fuse else None
        return buf

    def _plan_epilogue(self, is_cross, place, bh, nq, nk, ld):
        """icd_probs_epilogue of this layer call, or None: which of the controller's operations on P the kernel performs itself."""
        from . import _lib
        c = self.c
        self._epi_refs, self._epi_acc_key, self._epi_edit = None, None, False
        first = 0 if (self.cond_only or self.half) else bh // 2
        rows = bh - first
        epi = _lib.ProbsEpilogue()
        epi.first_cond_row = first
        refs = []
        if isinstance(c, AttentionControlEdit):
            # the kernel indexes the edit operators by SAMPLE (prompt jp = sample - first conditional sample, jp - 1 operators), while the
            # controller - like the reference (utils/p2p.py:192-194) - groups the rows by heads = rows / batch_size: the two agree only when
            # the conditional samples of this call are exactly the controller's prompts.  Anything else (more latents than prompts, an
            # unknown batch) keeps the separate passes.
            cond = (self.batch if self.cond_only else self.batch // 2) if self.batch > 0 else 0
            if rows % c.batch_size != 0 or c.batch_size < 2 or cond != c.batch_size:
                return None
            epi.edit_count = c.batch_size - 1
            if is_cross:
                if not (nk <= 80 and ld >= 80 and ld % 8 == 0):
                    return None                      # the torch / icd_p2p_cross_edit path handles it (and then the store add must follow it)
                At, Dp = c.cross_edit_operator(c.cur_step, self.dev)
                epi.edit_At, epi.edit_D = At.data_ptr(), Dp.data_ptr()
                refs += [At, Dp]
                self._epi_edit = True
            elif c.num_self_replace[0] <= c.cur_step < c.num_self_replace[1]:
                epi.self_from_base = 1               # (materialised self-attention layers are the <= 32^2-query ones: always replaced)
                self._epi_edit = True
        if nq <= c.STORE_MAX_QUERIES and c.attention_store:
            key = f"{place}_{'cross' if is_cross else 'self'}"
            idx = len(c.step_store[key])
            acc = c.attention_store.get(key, [])
            if idx < len(acc):
                t = acc[idx]
                seen = c.__dict__.setdefault("_epi_acc_ok", {})      # geometry checked once per stored tensor, not once per step
                ok = seen.get((key, idx))
                if ok is None or ok[0] is not t:
                    ok = (t, t.is_cuda and t.dtype == torch.float16 and tuple(t.shape) == (rows, nq, nk) and t.stride() == (nq * ld, ld, 1)
                          and t.data_ptr() % 16 == 0, t.data_ptr() if t.is_cuda else 0)
                    seen[(key, idx)] = ok
                if ok[1]:
                    epi.acc = ok[2]
                    refs.append(t)
                    self._epi_acc_key = (key, idx)
        if not (self._epi_edit or self._epi_acc_key):
            return None
        self._epi_refs = refs
        return epi

    def probs_ready(self, layer, is_cross, place):
        view, self.pending = self.pending, None
        c = self.c
        fused, self.epilogue = self.epilogue is not None, None
        if fused:                                    # tell the controller what the kernel has done to this layer's P
            c._kernel_edit = self._epi_edit
            if self._epi_acc_key is not None:
                c._fused_acc.add(self._epi_acc_key)
        try:
            if (self.cond_only or self.half) and isinstance(c, AttentionControl):
                c.call_cond_only(view, is_cross, place)
                return
            out = c(view, is_cross, place)
            if out is not None and out is not view:
                view.copy_(out)
        finally:
            if fused:
                c._kernel_edit = False


# ------------------------------------------------------------------------------------------- word / schedule helpers
def get_word_inds(text: str, word_place, tokenizer):
    """Token positions (1-based, after BOS) of a word given by value or by index (utils/p2p.py:422-440): one
    implementation, shared with the aligner."""
    return seq_aligner.get_word_inds(text, word_place, tokenizer)


def _step_gate(bounds, n_rows):
    """0/1 column over the n_rows = num_steps + 1 schedule rows: 1 on [int(lo * n_rows), int(hi * n_rows))."""
    lo, hi = (0.0, bounds) if type(bounds) is float else bounds
    rows = torch.arange(n_rows)
    return ((rows >= int(lo * n_rows)) & (rows < int(hi * n_rows))).float()


def update_alpha_time_word(alpha, bounds, prompt_ind: int, word_inds: Optional[torch.Tensor] = None):
    """Set alpha[:, prompt_ind, word_inds] to the step gate of `bounds` (utils/p2p.py:388-399); all words when None."""
    gate = _step_gate(bounds, alpha.shape[0])
    cols = slice(None) if word_inds is None else torch.as_tensor(np.asarray(word_inds), dtype=torch.int64)
    alpha[:, prompt_ind, cols] = gate[:, None]
    return alpha


def get_time_words_attention_alpha(prompts, num_steps, cross_replace_steps, tokenizer, max_num_words=77):
    """[num_steps+1, n_edits, 1, 1, 77] step x word gate of the cross-attention injection (utils/p2p.py:402-420): the
    "default_" window for every token, overridden per word where a word-specific window is given."""
    windows = dict(cross_replace_steps) if type(cross_replace_steps) is dict else {"default_": cross_replace_steps}
    if type(cross_replace_steps) is dict:
        cross_replace_steps.setdefault("default_", (0., 1.))          # the reference adds the key to the caller's dict
    windows.setdefault("default_", (0., 1.))
    n_edits, rows = len(prompts) - 1, num_steps + 1
    alpha = _step_gate(windows["default_"], rows)[:, None, None].repeat(1, n_edits, max_num_words)
    for word, bounds in windows.items():
        if word == "default_":
            continue
        gate = _step_gate(bounds, rows)
        for e in range(n_edits):
            cols = get_word_inds(prompts[e + 1], word, tokenizer)
            if len(cols) > 0:
                alpha[:, e, torch.as_tensor(cols, dtype=torch.int64)] = gate[:, None]
    return alpha.reshape(rows, n_edits, 1, 1, max_num_words)


def get_equalizer(text: str, word_select, values):
    """[1, 77] per-token scale: `values[k]` on the tokens of word `word_select[k]`, 1 elsewhere (utils/p2p.py:443-453)."""
    if type(word_select) is int or type(word_select) is str:
        word_select = (word_select,)
    eq = torch.ones(1, 77)
    for word, val in zip(word_select, values):
        eq[:, get_word_inds(text, word, tokenizer)] = val
    return eq
