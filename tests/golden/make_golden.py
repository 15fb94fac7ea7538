#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference in the build container.

Runs only where /root/reference exists (never on the GPU box).  It executes the reference's own
Python (utils/generation.py, utils/generation_sdxl.py, utils/p2p.py, utils/seq_aligner.py and one
function of running/sd1.5/generate.py) on seeded synthetic inputs and stores INPUTS + OUTPUTS as
small .npz/.json fixtures next to this script.  No reference source text is stored - only data.

    python3 -B tests/golden/make_golden.py

The fixtures pin:
  a2  Generator.__init__ timestep / boundary tables          -> timesteps.json
  a4  guidance_scale_embedding                               -> wembed.npz
  a5  linear_schedule_old / linear_schedule                  -> schedules.json
  a6  predicted_origin                                       -> predicted_origin.npz
  a3/a7/a8  get_noise_pred / cons_generation / cons_inversion with a closed-form stub UNet
                                                             -> sd15_loops.npz
  a9/a10/a11 sample_deterministic / inverse_sample_deterministic / DDIMSolver (stub pipe)
                                                             -> sdxl_loops.npz
  a15-a18 p2p controllers + LocalBlend (stub whitespace tokenizer) -> p2p_*.npz
  seq_aligner mappers                                        -> seq_aligner.npz
  a19 prepare_val_prompts partitions                         -> sharding.json
"""
import ast
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

if not os.path.isdir(REF):
    sys.exit("reference tree not present - golden vectors can only be regenerated in the build container")

sys.dont_write_bytecode = True
# IPython.display is only used for notebook display (utils/generation.py:6,620)
_ip = types.ModuleType("IPython")
_ipd = types.ModuleType("IPython.display")
_ipd.display = lambda *a, **k: None
_ip.display = _ipd
sys.modules.setdefault("IPython", _ip)
sys.modules.setdefault("IPython.display", _ipd)
sys.path.insert(0, REF)

from utils import generation as rgen          # noqa: E402
from utils import generation_sdxl as rxl      # noqa: E402
from utils import p2p as rp2p                 # noqa: E402
from utils import seq_aligner as rsa          # noqa: E402


def alphas_cumprod():
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class StubTokenizer:
    """Whitespace tokenizer: encode -> [bos] + ids + [eos]; decode([id]) -> word."""
    model_max_length = 77

    def __init__(self):
        self.vocab = {}
        self.words = {}

    def _id(self, w):
        if w not in self.vocab:
            i = len(self.vocab) + 10
            self.vocab[w] = i
            self.words[i] = w
        return self.vocab[w]

    def encode(self, text):
        return [1] + [self._id(w) for w in text.split(" ") if w != ""] + [2]

    def decode(self, ids):
        return " ".join(self.words.get(i, "") for i in ids)


class _Cfg:
    prediction_type = "epsilon"
    num_train_timesteps = 1000


class StubScheduler:
    def __init__(self):
        self.alphas_cumprod = alphas_cumprod()
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.config = _Cfg()
        self.num_train_timesteps = 1000
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        step = 1000 // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * step).round()[::-1].copy().astype(np.int64))

    def add_noise(self, x, noise, t):
        a = self.alphas_cumprod[t] ** 0.5
        s = (1 - self.alphas_cumprod[t]) ** 0.5
        while a.dim() < x.dim():
            a = a.unsqueeze(-1)
            s = s.unsqueeze(-1)
        return a * x + s * noise


class StubUNet:
    """eps = 0.1*x + 0.01*t/1000 + 0.001*mean(timestep_cond) (closed form), records its calls."""
    dtype = torch.float32
    in_channels = 4

    def __init__(self):
        self.calls = []

    def named_children(self):
        return []

    def __call__(self, x, t, timestep_cond=None, encoder_hidden_states=None, **kw):
        tt = float(t) if not torch.is_tensor(t) else float(t.item())
        self.calls.append(dict(x=x.clone(), t=tt,
                               cond=None if timestep_cond is None else timestep_cond.clone()))
        eps = 0.1 * x + 0.01 * tt / 1000.0
        if timestep_cond is not None:
            eps = eps + 0.001 * timestep_cond.float().mean(dim=1).reshape(-1, 1, 1, 1)
        if kw.get("return_dict", True) is False:
            return (eps,)
        return {"sample": eps}


class StubVAE:
    dtype = torch.float32

    def encode(self, x):
        # 8x8 average pool of the first 4 "channels" -> [B,4,h/8,w/8]
        z = torch.nn.functional.avg_pool2d(torch.cat([x, x[:, :1]], 1), 8)
        return {"latent_dist": types.SimpleNamespace(mean=z)}

    def decode(self, z):
        return {"sample": torch.nn.functional.interpolate(z[:, :3], scale_factor=8)}


class StubModel:
    device = torch.device("cpu")
    dtype = torch.float32

    def __init__(self):
        self.scheduler = StubScheduler()
        self.unet = StubUNet()
        self.vae = StubVAE()
        self.tokenizer = StubTokenizer()


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items()})


def jdump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
    print("wrote", name)


# ----------------------------------------------------------------------------------------------
def gold_timesteps():
    res = {}
    sets = {
        "set1": ([259, 519, 779, 999], [19, 259, 519, 779]),
        "set2": ([249, 499, 699, 999], [19, 249, 499, 699]),
        "set3": ([339, 699, 999], [19, 339, 699]),
    }
    for name, (rev, fwd) in sets.items():
        rev_in, fwd_in = list(rev), list(fwd)
        g = rgen.Generator(StubModel(), 50, StubScheduler(), reverse_timesteps=rev_in, forward_timesteps=fwd_in)
        res[name] = dict(
            reverse_in=rev, forward_in=fwd,
            reverse_timesteps=g.reverse_timesteps.tolist(),
            reverse_boundary=g.reverse_boundary_timesteps.tolist(),
            forward_timesteps=g.forward_timesteps.tolist(),
            forward_boundary=g.forward_boundary_timesteps.tolist(),
            caller_reverse_list_after=rev_in, caller_forward_list_after=fwd_in,
        )
    for ne in (1, 2, 3, 4):
        g = rgen.Generator(StubModel(), 50, StubScheduler(), num_endpoints=ne, num_forward_endpoints=ne)
        res[f"default_{ne}"] = dict(
            reverse_timesteps=g.reverse_timesteps.tolist(),
            reverse_boundary=g.reverse_boundary_timesteps.tolist(),
            forward_timesteps=g.forward_timesteps.tolist(),
            forward_boundary=g.forward_boundary_timesteps.tolist(),
        )
    # DDIMSolver endpoints (utils/generation_sdxl.py:135-177)
    ac = alphas_cumprod().numpy()
    for ne in (1, 2, 3, 4):
        s = rxl.DDIMSolver(ac, timesteps=1000, ddim_timesteps=50, num_endpoints=ne, num_inverse_endpoints=ne)
        res[f"ddimsolver_{ne}"] = dict(endpoints=s.endpoints.tolist(), inverse_endpoints=s.inverse_endpoints.tolist(),
                                       ddim_timesteps=s.ddim_timesteps.tolist())
    s = rxl.DDIMSolver(ac, num_endpoints=4, num_inverse_endpoints=4, endpoints="0,249,499,699",
                       inverse_endpoints="249,499,699,999")
    res["ddimsolver_explicit"] = dict(endpoints=s.endpoints.tolist(), inverse_endpoints=s.inverse_endpoints.tolist())
    res["alphas_cumprod_probe"] = {str(i): float(ac[i]) for i in (0, 19, 249, 259, 339, 499, 519, 699, 779, 999)}
    jdump("timesteps.json", res)
    # torch's CPU sqrt is not correctly rounded and differs between CPU vendors: pin the tables the reference used
    act = alphas_cumprod()
    npz("alphas_cumprod.npz", alphas_cumprod=ac, alpha_table=torch.sqrt(act), sigma_table=torch.sqrt(1 - act))


def gold_wembed():
    w = torch.tensor([0.0, 7.0, 19.0, 1.5, 7.5])
    npz("wembed.npz", w=w, emb512=rgen.guidance_scale_embedding(w, 512),
        emb512_xl=rxl.guidance_scale_embedding(w, 512), emb256=rgen.guidance_scale_embedding(w, 256),
        emb33=rgen.guidance_scale_embedding(w, 33))


def gold_schedules():
    rows = []
    for t in (999, 779, 519, 259, 699, 499, 249, 339, 19, 800, 700, 400):
        for gs in (7.0, 19.0, 1.0):
            for (t1, t2) in ((0.7, 0.7), (0.8, 0.8), (1.0, 1.0), (0.4, 0.6), (0.4, 0.8)):
                rows.append(dict(t=t, gs=gs, tau1=t1, tau2=t2,
                                 old=float(rgen.linear_schedule_old(t, gs, t1, t2)),
                                 old_xl=float(rxl.linear_schedule_old(t, gs, t1, t2)),
                                 new=float(rgen.linear_schedule(t, gs, t1, t2))))
    jdump("schedules.json", rows)


def gold_predicted_origin():
    g = torch.Generator().manual_seed(1234)
    ac = alphas_cumprod()
    alpha, sigma = torch.sqrt(ac), torch.sqrt(1 - ac)
    x = torch.randn(2, 4, 8, 8, generator=g)
    eps = torch.randn(2, 4, 8, 8, generator=g)
    pairs = [(999, 779), (779, 519), (519, 259), (259, 0), (19, 259), (259, 519), (519, 779), (779, 999),
             (999, 699), (699, 499), (499, 249), (249, 0), (19, 249), (699, 999), (339, 0), (19, 339), (339, 699)]
    outs, outs_xl = [], []
    for t, s in pairs:
        tt, ss = torch.tensor([t, t]), torch.tensor([s, s])
        outs.append(rgen.predicted_origin(eps, tt, ss, x, "epsilon", alpha, sigma))
        outs_xl.append(rxl.predicted_origin(eps, tt, ss, x, "epsilon", alpha, sigma))
    # mixed per-sample (t, s) incl. one s == 0 row
    tt, ss = torch.tensor([999, 259]), torch.tensor([779, 0])
    mixed = rgen.predicted_origin(eps, tt, ss, x, "epsilon", alpha, sigma)
    vpred = rgen.predicted_origin(eps, torch.tensor([999, 999]), torch.tensor([0]), x, "v_prediction", alpha, sigma)
    # fp16 inputs, fp32 tables (what the fp16 pipelines execute)
    o16 = rgen.predicted_origin(eps.half(), torch.tensor([779, 779]), torch.tensor([519, 519]), x.half(), "epsilon",
                                alpha, sigma)
    npz("predicted_origin.npz", x=x, eps=eps, pairs=np.array(pairs), out=torch.stack(outs), out_xl=torch.stack(outs_xl),
        mixed_t=tt, mixed_s=ss, mixed=mixed, vpred=vpred, out_fp16in=o16.float(),
        out_fp16in_dtype=str(o16.dtype))


def gold_sd15_loops():
    arrs = {}
    for B, gs, tau in ((3, 19.0, 0.8), (2, 19.0, 0.8), (2, 7.0, 1.0), (1, 7.0, 0.7)):
        m = StubModel()
        G = rgen.Generator(m, 50, StubScheduler(), forward_cons_model=m, reverse_cons_model=m,
                           reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
        gen = torch.Generator().manual_seed(453645634 + B)
        G.context = torch.randn(2 * B, 77, 8, generator=gen)
        lat = torch.randn(B, 4, 8, 8, generator=gen)
        dyn = tau < 1.0
        outs = G.cons_generation(lat, guidance_scale=gs, w_embed_dim=512, dynamic_guidance=dyn, tau1=tau, tau2=tau,
                                 controller=None)
        tag = f"rev_B{B}_gs{int(gs)}_tau{int(tau * 10)}"
        arrs[tag + "_in"] = lat
        arrs[tag + "_out"] = torch.stack(outs)
        arrs[tag + "_t"] = np.array([c["t"] for c in m.unet.calls])
        arrs[tag + "_cond"] = torch.stack([c["cond"] for c in m.unet.calls])
        arrs[tag + "_x"] = torch.stack([c["x"] for c in m.unet.calls])
    # classic CFG branch (w_embed_dim == 0): guided_step with linear_schedule
    m = StubModel()
    G = rgen.Generator(m, 50, StubScheduler(), forward_cons_model=m, reverse_cons_model=m,
                       reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    gen = torch.Generator().manual_seed(7)
    G.context = torch.randn(4, 77, 8, generator=gen)
    lat = torch.randn(2, 4, 8, 8, generator=gen)

    class HalfUNet(StubUNet):
        def __call__(self, x, t, timestep_cond=None, encoder_hidden_states=None, **kw):
            out = super().__call__(x, t, timestep_cond, encoder_hidden_states, **kw)["sample"]
            out[: len(out) // 2] *= 0.5       # make uncond != cond so CFG is visible
            return {"sample": out}
    m.unet = HalfUNet()
    outs = G.cons_generation(lat, guidance_scale=7.5, w_embed_dim=0, dynamic_guidance=True, tau1=0.4, tau2=0.8)
    arrs["cfg_in"], arrs["cfg_out"] = lat, torch.stack(outs)

    # cons_inversion (utils/generation.py:414-451): 4D tensor image -> image2latent passes it through
    m = StubModel()
    G = rgen.Generator(m, 50, StubScheduler(), forward_cons_model=m, reverse_cons_model=m,
                       reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    gen = torch.Generator().manual_seed(99)
    G.context = torch.randn(4, 77, 8, generator=gen)
    lat0 = torch.randn(2, 4, 8, 8, generator=gen)
    G.latent2image = lambda z, return_type="np": np.zeros((1,))
    _, out = G.cons_inversion(lat0, guidance_scale=0.0, w_embed_dim=512, seed=5)
    arrs["inv_in"], arrs["inv_out"] = lat0, out[0]
    arrs["inv_t"] = np.array([c["t"] for c in m.unet.calls])
    arrs["inv_cond"] = torch.stack([c["cond"] for c in m.unet.calls])
    arrs["inv_x0"] = m.unet.calls[0]["x"]
    arrs["inv_noise"] = torch.randn(lat0.shape, generator=torch.Generator().manual_seed(5))

    # init_latent (utils/generation.py:536-543): one sample expanded to the batch
    l1, lB = rgen.init_latent(None, m, 64, 64, torch.Generator().manual_seed(11), 3)
    arrs["init_latent_one"], arrs["init_latent_batch"] = l1, lB.contiguous()

    # runner() end to end with the stub model (return_type='latent')
    m = StubModel()
    G = rgen.Generator(m, 50, StubScheduler(), forward_cons_model=m, reverse_cons_model=m,
                       reverse_timesteps=[259, 519, 779, 999], forward_timesteps=[19, 259, 519, 779])
    ctx = torch.randn(6, 77, 8, generator=torch.Generator().manual_seed(3))

    def _init_prompt(prompt, unc=None):
        G.context = ctx
        G.prompt = prompt
    G.init_prompt = _init_prompt
    # runner hard-codes 512x512 -> 64x64 latents
    img, lat = rgen.runner(model=m, prompt=["a", "b", "c"], controller=None, solver=G, is_cons_forward=True,
                           num_inference_steps=50, guidance_scale=19.0, generator=torch.Generator().manual_seed(21),
                           latent=None, return_type="latent", dynamic_guidance=False, tau1=0.8, tau2=0.8,
                           w_embed_dim=512)
    arrs["runner_latent"], arrs["runner_out"] = lat, img
    arrs["runner_t"] = np.array([c["t"] for c in m.unet.calls])
    arrs["runner_w_first_col"] = torch.stack([c["cond"][:, 0] for c in m.unet.calls])
    npz("sd15_loops.npz", **arrs)


class StubPipe:
    """What utils/generation_sdxl.py touches on a diffusers SDXL pipeline."""

    def __init__(self):
        self.unet = StubUNet()
        self.unet.config = types.SimpleNamespace(sample_size=2, in_channels=4)
        self.vae_scale_factor = 8
        self._execution_device = torch.device("cpu")
        self.scheduler = StubScheduler()
        self.vae = types.SimpleNamespace(
            to=lambda *a, **k: None, config=types.SimpleNamespace(scaling_factor=0.13025),
            decode=lambda z, return_dict=False: (z,))
        self.image_processor = types.SimpleNamespace(postprocess=lambda im, output_type, do_denormalize: im)

    def prepare_latents(self, *args, **kw):
        if torch.is_tensor(args[0]):          # img2img signature (image, timestep, bs, n, dtype, device, generator)
            image, t, bs, _, dtype, device = args[:6]
            gen = kw.get("generator")
            noise = torch.randn(image.shape, generator=gen, dtype=dtype)
            return self.scheduler.add_noise(image, noise, t.reshape(1))
        bs, c, h, w, dtype, device, gen = args[:7]
        return torch.randn((bs, c, h // 8, w // 8), generator=gen, dtype=dtype)


def gold_sdxl_loops():
    arrs = {}

    def emb_fn(prompts, orig, crop):
        g = torch.Generator().manual_seed(len(prompts) * 100 + len(prompts[0]))
        n = len(prompts)
        return {"prompt_embeds": torch.randn(n, 77, 16, generator=g), "text_embeds": torch.randn(n, 8, generator=g),
                "time_ids": torch.tensor([[1024, 1024, 0, 0, 1024, 1024]] * n, dtype=torch.float32)}

    # static guidance, B = 3, 4-step set
    p = StubPipe()
    img, lat = rxl.sample_deterministic(p, ["aa", "bb", "cc"], num_inference_steps=4,
                                        generator=torch.Generator().manual_seed(0), guidance_scale=7.0, is_sdxl=True,
                                        timesteps=[249, 499, 699, 999], compute_embeddings_fn=emb_fn, return_latent=True)
    arrs["rev_B3_out"] = lat
    arrs["rev_B3_t"] = np.array([c["t"] for c in p.unet.calls])
    arrs["rev_B3_x"] = torch.stack([c["x"] for c in p.unet.calls])
    arrs["rev_B3_cond"] = torch.stack([c["cond"] for c in p.unet.calls])
    # dynamic guidance works for B == 1 only (utils/generation_sdxl.py:439-440)
    p = StubPipe()
    img, lat = rxl.sample_deterministic(p, ["aa"], num_inference_steps=3, generator=torch.Generator().manual_seed(1),
                                        guidance_scale=19.0, is_sdxl=True, timesteps=[339, 699, 999],
                                        compute_embeddings_fn=emb_fn, return_latent=True, use_dynamic_guidance=True,
                                        tau1=0.7, tau2=0.7, amplify_prompt=["zzzz"])
    arrs["dyn_B1_out"] = lat
    arrs["dyn_B1_t"] = np.array([c["t"] for c in p.unet.calls])
    arrs["dyn_B1_cond"] = torch.stack([c["cond"] for c in p.unet.calls])
    err = ""
    try:
        rxl.sample_deterministic(StubPipe(), ["aa", "bb"], num_inference_steps=3,
                                 generator=torch.Generator().manual_seed(1), guidance_scale=19.0, is_sdxl=True,
                                 timesteps=[339, 699, 999], compute_embeddings_fn=emb_fn, use_dynamic_guidance=True,
                                 tau1=0.7, tau2=0.7)
    except Exception as e:  # the reference raises for B > 1
        err = type(e).__name__
    arrs["dyn_B2_error"] = np.array(err)
    # default timesteps (timesteps=None): DDIMSolver endpoints
    p = StubPipe()
    img, lat = rxl.sample_deterministic(p, ["aa", "bb"], num_inference_steps=4,
                                        generator=torch.Generator().manual_seed(2), guidance_scale=7.0, is_sdxl=True,
                                        timesteps=None, compute_embeddings_fn=emb_fn, return_latent=True)
    arrs["revdef_out"] = lat
    arrs["revdef_t"] = np.array([c["t"] for c in p.unet.calls])
    # forward (inversion)
    p = StubPipe()
    im = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(5))
    lat, start = rxl.inverse_sample_deterministic(p, im, ["aa", "bb"], num_inference_steps=4,
                                                  timesteps=[19, 249, 499, 699], guidance_scale=0.0, is_sdxl=True,
                                                  compute_embeddings_fn=emb_fn, seed=3, return_start_latent=True)
    arrs["fwd_in"], arrs["fwd_out"], arrs["fwd_start"] = im, lat, start
    arrs["fwd_t"] = np.array([c["t"] for c in p.unet.calls])
    arrs["fwd_cond"] = torch.stack([c["cond"] for c in p.unet.calls])
    npz("sdxl_loops.npz", **arrs)


def _probs(gen, bh, n, m):
    return torch.softmax(torch.randn(bh, n, m, generator=gen) * 2.0, dim=-1)


# SD1.5 layer walk (place, is_cross, N) in module-execution order, reduced to small N for fixtures:
# real N = 4096,1024,256,64 -> scaled /16 so the 32^2 store threshold is mimicked by N <= 64 in tests
def layer_walk(scale=1):
    walk = []
    for res in (4096, 4096, 1024, 1024, 256, 256):
        walk += [("down", False, res), ("down", True, res)]
    walk += [("mid", False, 64), ("mid", True, 64)]
    for res in (256, 256, 256, 1024, 1024, 1024, 4096, 4096, 4096):
        walk += [("up", False, res), ("up", True, res)]
    return walk


def gold_p2p():
    tok = StubTokenizer()
    rp2p.tokenizer = tok
    rp2p.device = "cpu"
    rp2p.NUM_DDIM_STEPS = 4
    rp2p.LOW_RESOURCE = False
    heads = 2

    # The walk uses genuine query counts for the <=32^2 threshold but only small ones are instantiated:
    # N in {1024 -> stored, 256 -> stored, 64 -> stored, 4096 -> not stored}.  To keep fixtures small we
    # run a reduced walk: two 'down' (N=1296 > 1024 not stored; N=256), one 'mid' (N=64), two 'up' (256, 1296)
    walk = [("down", False, 1296), ("down", True, 1296), ("down", False, 256), ("down", True, 256),
            ("down", False, 256), ("down", True, 256), ("down", False, 256), ("down", True, 256),
            ("mid", False, 64), ("mid", True, 64),
            ("up", False, 256), ("up", True, 256), ("up", False, 256), ("up", True, 256),
            ("up", False, 256), ("up", True, 256), ("up", False, 1296), ("up", True, 1296)]

    def run(controller, n_prompts, seed, latents=None, steps=4):
        controller.num_att_layers = len(walk)
        gen = torch.Generator().manual_seed(seed)
        ins, outs, lat_out = [], [], []
        for step in range(steps):
            for (place, is_cross, n) in walk:
                m = 77 if is_cross else n
                P = _probs(gen, 2 * n_prompts * heads, n, m)
                ins.append(P.clone())
                R = controller(P, is_cross, place)
                assert R is P
                outs.append(R.clone())
            if latents is not None:
                latents = controller.step_callback(latents)
                lat_out.append(latents.clone())
        return ins, outs, lat_out

    def compact(arrs, name, t):
        """Big tensors are stored as a strided subsample + float64 (sum, sum of squares) of the FULL tensor."""
        t = t.float()
        arrs[name + "__stats"] = np.array([t.double().sum().item(), (t.double() ** 2).sum().item()])
        sr = max(1, t.shape[1] // 24)
        sc = max(1, t.shape[2] // 24)
        arrs[name + "__stride"] = np.array([sr, sc])
        arrs[name] = t[:, ::sr, ::sc].contiguous()

    def pack(prefix, arrs, ins, outs, controller, lat_out=None):
        # inputs are regenerated from the seed by the test; only (compacted) outputs + store are saved
        for i, o in enumerate(outs):
            if not torch.equal(o, ins[i]):
                compact(arrs, f"{prefix}_out{i}", o)
        arrs[f"{prefix}_changed"] = np.array([int(not torch.equal(o, ins[i])) for i, o in enumerate(outs)])
        for key, lst in controller.attention_store.items():
            arrs[f"{prefix}_storelen_{key}"] = np.array(len(lst))
            for j, t in enumerate(lst):
                compact(arrs, f"{prefix}_store_{key}_{j}", t)
        arrs[f"{prefix}_cur_step"] = np.array(controller.cur_step)
        if lat_out:
            arrs[f"{prefix}_latents"] = torch.stack(lat_out)

    arrs = {}
    # 1. AttentionStore
    c = rp2p.AttentionStore()
    ins, outs, _ = run(c, 1, 100)
    pack("store", arrs, ins, outs, c)
    avg = c.get_average_attention()
    compact(arrs, "store_avg_down_cross_0", avg["down_cross"][0])

    prompts = ["a cat sitting on a bench", "a dog sitting on a bench"]
    # 2. AttentionReplace, cross 0.5 / self 0.5
    c = rp2p.make_controller(prompts, True, 0.5, 0.5)
    arrs["replace_alpha"] = c.cross_replace_alpha
    arrs["replace_mapper"] = c.mapper
    arrs["replace_num_self"] = np.array(c.num_self_replace)
    ins, outs, _ = run(c, 2, 101)
    pack("replace", arrs, ins, outs, c)
    # 3. AttentionRefine with per-word cross_replace dict + LocalBlend
    prompts_r = ["a cat sitting on a bench", "a fluffy cat sitting on a red bench"]
    c = rp2p.make_controller(prompts_r, False, {"default_": 0.8, "fluffy": (0.0, 0.4)}, 0.4,
                             blend_words=(("cat",), ("cat",)))
    arrs["refine_alpha"] = c.cross_replace_alpha
    arrs["refine_mapper"] = c.mapper
    arrs["refine_alphas"] = c.alphas
    arrs["refine_lb_alpha_layers"] = c.local_blend.alpha_layers
    lat = torch.randn(2, 4, 64, 64, generator=torch.Generator().manual_seed(9))
    arrs["refine_lat_in"] = lat
    # LocalBlend needs 16x16 (=256) cross maps at down_cross[2:4] and up_cross[:3]: the walk provides
    # down_cross = [256,256,256] (1296 skipped) -> indices 2:4 -> one map; reference semantics kept as is.
    ins, outs, lat_out = run(c, 2, 102, latents=lat)
    pack("refine", arrs, ins, outs, c, lat_out)
    # 4. AttentionReweight chained on Replace
    c = rp2p.make_controller(prompts, True, 0.6, 0.2, equilizer_params={"words": ("dog",), "values": (3.0,)})
    arrs["reweight_equalizer"] = c.equalizer
    ins, outs, _ = run(c, 2, 103)
    pack("reweight", arrs, ins, outs, c)
    # 5. 3 prompts replace (batch of edits sharing the base prompt)
    prompts3 = ["a cat sitting on a bench", "a dog sitting on a bench", "a cat sitting on a sofa"]
    c = rp2p.make_controller(prompts3, True, 0.5, 0.25)
    ins, outs, _ = run(c, 3, 104, steps=2)
    pack("replace3", arrs, ins, outs, c)
    arrs["walk_n"] = np.array([w[2] for w in walk])
    arrs["walk_cross"] = np.array([int(w[1]) for w in walk])
    arrs["walk_place"] = np.array([w[0] for w in walk])
    # helper functions
    arrs["word_inds_cat"] = rp2p.get_word_inds(prompts[0], "cat", tok)
    arrs["word_inds_2"] = rp2p.get_word_inds(prompts[0], 2, tok)
    arrs["equalizer_multi"] = rp2p.get_equalizer(prompts_r[1], ("fluffy", "red"), (2.0, 0.5))
    arrs["time_words_alpha"] = rp2p.get_time_words_attention_alpha(prompts_r, 4, {"default_": (0.1, 0.9), "red": 0.3}, tok)
    # SpatialReplace
    sr = rp2p.SpatialReplace(0.5)
    sr.cur_step = 0
    x = torch.randn(3, 4, 4, 4, generator=torch.Generator().manual_seed(1))
    arrs["spatial_in"], arrs["spatial_out0"] = x, sr.step_callback(x).contiguous()
    arrs["spatial_stop"] = np.array(sr.stop_inject)
    npz("p2p.npz", **arrs)
    jdump("p2p_vocab.json", {"note": "whitespace stub tokenizer: bos=1 eos=2 ids assigned from 10 in first-seen order"})


def gold_seq_aligner():
    tok = StubTokenizer()
    arrs = {}
    pairs = [("a cat sitting on a bench", "a fluffy cat sitting on a red bench"),
             ("a photo of a house on a hill", "a photo of a wooden house on a snowy hill at night"),
             ("the quick brown fox", "the fox")]
    for i, (a, b) in enumerate(pairs):
        m, al = rsa.get_refinement_mapper([a, b], tok)
        arrs[f"refine_mapper_{i}"], arrs[f"refine_alphas_{i}"] = m, al
    rep = [("a cat sitting on a bench", "a dog sitting on a bench"),
           ("a red car in the city", "a blue bus in the city")]
    for i, (a, b) in enumerate(rep):
        arrs[f"replace_mapper_{i}"] = rsa.get_replacement_mapper([a, b], tok)
    m3 = rsa.get_replacement_mapper(["a cat on a bench", "a dog on a bench", "a cat on a sofa"], tok)
    arrs["replace_mapper_3prompts"] = m3
    err = ""
    try:
        rsa.get_replacement_mapper(["a cat", "a big cat"], tok)
    except ValueError as e:
        err = "ValueError"
    arrs["replace_unequal_error"] = np.array(err)
    npz("seq_aligner.npz", **arrs)
    jdump("seq_aligner_prompts.json", {"refine": pairs, "replace": rep})


def gold_sharding():
    """running/sd1.5/generate.py:29-39 prepare_val_prompts, executed with a stub torch.distributed."""
    src = open(os.path.join(REF, "running/sd1.5/generate.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_val_prompts"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    res = []
    for (N, bs, W) in ((128, 8, 1), (64, 8, 8), (128, 16, 8), (100, 8, 4), (5, 8, 2)):
        for rank in range(W):
            dist = types.SimpleNamespace(get_world_size=lambda W=W: W, get_rank=lambda rank=rank: rank)
            ns = {"np": np, "dist": dist}
            exec(compile(mod, "<prepare_val_prompts>", "exec"), ns)
            texts = [f"p{i}" for i in range(N)]
            rb, rbi, allt = ns["prepare_val_prompts"](texts, bs=bs, max_cnt=5000)
            res.append(dict(N=N, bs=bs, W=W, rank=rank, batches=[list(map(str, b)) for b in rb],
                            index=[list(map(int, b)) for b in rbi]))
    jdump("sharding.json", res)


if __name__ == "__main__":
    torch.manual_seed(0)
    gold_timesteps()
    gold_wembed()
    gold_schedules()
    gold_predicted_origin()
    gold_sd15_loops()
    gold_sdxl_loops()
    gold_p2p()
    gold_seq_aligner()
    gold_sharding()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total fixture bytes", total)
