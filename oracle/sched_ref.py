"""ORACLE (test infrastructure, not product code) - CPU restatement of the iCD sampler arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

numpy restatement of the repo-local arithmetic of the reference hot path; every function cites the
reference lines it follows (paths relative to /root/reference).  Pinned against golden vectors captured by
importing the reference itself (tests/golden/make_golden.py -> tests/golden/*.npz|json), see
tests/test_oracle_golden.py.
"""
import math

import numpy as np


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """DDIMScheduler(scaled_linear) constants used by utils/loading.py:39-40 (SD1.5) and :113 (SDXL).

    betas = linspace(sqrt(bs), sqrt(be), T, fp32)**2 ; abar = cumprod(1 - betas)  (all in fp32, like torch).
    """
    betas = np.linspace(np.float32(beta_start) ** np.float32(0.5), np.float32(beta_end) ** np.float32(0.5),
                        num_train_timesteps, dtype=np.float32) ** 2
    return np.cumprod((np.float32(1.0) - betas).astype(np.float32), dtype=np.float32)


def guidance_scale_embedding(w, embedding_dim=512):
    """utils/generation.py:96-122 (dup utils/generation_sdxl.py:84-110).

    w*1000; f_i = exp(-ln(1e4) * i / (half-1)); [sin(w f) || cos(w f)]; zero pad when the dim is odd.
    """
    w = np.asarray(w, dtype=np.float32).reshape(-1) * np.float32(1000.0)
    half = embedding_dim // 2
    step = np.float32(math.log(10000.0)) / np.float32(half - 1)
    freq = np.exp(np.arange(half, dtype=np.float32) * -step).astype(np.float32)
    ang = w[:, None] * freq[None, :]
    emb = np.concatenate([np.sin(ang), np.cos(ang)], axis=1).astype(np.float32)
    if embedding_dim % 2 == 1:
        emb = np.pad(emb, ((0, 0), (0, 1)))
    return emb


def linear_schedule_old(t, guidance_scale, tau1, tau2):
    """utils/generation.py:74-82 == utils/generation_sdxl.py:313-321 (used on the w-embedding path)."""
    t = t / 1000
    if t <= tau1:
        gamma = 1.0
    elif t >= tau2:
        gamma = 0.0
    else:
        gamma = (tau2 - t) / (tau2 - tau1)
    return gamma * guidance_scale


def linear_schedule(t, guidance_scale, tau1=0.4, tau2=0.8):
    """utils/generation.py:85-93 (classic-CFG branch, guided_step)."""
    t = t / 1000
    if t <= tau1:
        return guidance_scale
    if t >= tau2:
        return 1.0
    return (tau2 - t) / (tau2 - tau1) * (guidance_scale - 1.0) + 1.0


def predicted_origin(model_output, timesteps, boundary_timesteps, sample, alphas, sigmas,
                     prediction_type="epsilon"):
    """utils/generation.py:136-155 (dup utils/generation_sdxl.py:112-132).

    eps-pred: x0 = (x - sigma_t eps) / alpha_t ; out = alpha_s x0 + sigma_s eps with (alpha_s, sigma_s) forced to
    (1, 0) where s == 0.  alphas = sqrt(abar), sigmas = sqrt(1 - abar) (fp32 tables, generation.py:385-386).
    """
    t = np.asarray(timesteps).reshape(-1)
    s = np.asarray(boundary_timesteps).reshape(-1)
    shape = (-1,) + (1,) * (sample.ndim - 1)
    a_t, s_t = alphas[t].reshape(shape), sigmas[t].reshape(shape)
    a_s, s_s = alphas[s].copy(), sigmas[s].copy()
    a_s[s == 0] = 1.0
    s_s[s == 0] = 0.0
    a_s, s_s = a_s.reshape(shape), s_s.reshape(shape)
    if prediction_type == "epsilon":
        x0 = (sample - s_t * model_output) / a_t
        return a_s * x0 + s_s * model_output
    if prediction_type == "v_prediction":
        assert np.all(s == 0)
        return a_t * sample - s_t * model_output
    raise ValueError(f"Prediction type {prediction_type} currently not supported.")


def ddim_timesteps(n_steps=50, num_train=1000):
    """utils/generation.py:490-492 / utils/generation_sdxl.py:143-145: (arange(1..n) * (T // n)) - 1."""
    return (np.arange(1, n_steps + 1) * (num_train // n_steps)).round().astype(np.int64) - 1


def default_endpoints(num_endpoints, n_steps=50, max_inverse_timestep_index=49):
    """utils/generation.py:453-465 (_create_forward_inverse_timesteps) == DDIMSolver utils/generation_sdxl.py:160-174."""
    dt = ddim_timesteps(n_steps)
    interval = n_steps // num_endpoints + int(n_steps % num_endpoints > 0)
    idx = np.arange(interval, n_steps, interval) - 1
    endpoints = np.array([0] + dt[idx].tolist(), dtype=np.int64)
    inverse_endpoints = dt[np.array(idx.tolist() + [max_inverse_timestep_index], dtype=np.int64)]
    return endpoints, inverse_endpoints


def generator_tables(reverse_timesteps=None, forward_timesteps=None, num_endpoints=1, num_forward_endpoints=1,
                     n_steps=50, max_forward_timestep_index=49, start_timestep=19):
    """utils/generation.py:495-518 - returns (rev_t, rev_boundary, fwd_t, fwd_boundary) as int64 arrays."""
    if reverse_timesteps is None or forward_timesteps is None:
        ep, iep = default_endpoints(num_endpoints, n_steps, max_forward_timestep_index)
        rev_t, rev_b = iep[::-1].copy(), ep[::-1].copy()
        ep, iep = default_endpoints(num_forward_endpoints, n_steps, max_forward_timestep_index)
        fwd_t, fwd_b = ep.copy(), iep.copy()
        fwd_t[0] = start_timestep
        return rev_t, rev_b, fwd_t, fwd_b
    rev = list(reverse_timesteps)[::-1]                     # generation.py:508 (in-place reverse of the caller's list)
    rev_b = rev[1:] + [0]                                   # :509-510
    fwd = list(forward_timesteps)
    fwd_b = fwd[1:] + [999]                                 # :515-516
    return (np.array(rev, dtype=np.int64), np.array(rev_b, dtype=np.int64),
            np.array(fwd, dtype=np.int64), np.array(fwd_b, dtype=np.int64))


def sdxl_reverse_tables(timesteps):
    """utils/generation_sdxl.py:397-402: reversed list; boundaries shifted by one, last = 0."""
    ts = list(timesteps)[::-1]
    return np.array(ts, dtype=np.int64), np.array(ts[1:] + [0], dtype=np.int64)


def sdxl_forward_tables(timesteps):
    """utils/generation_sdxl.py:263-266: boundaries shifted by one, last = 999."""
    ts = list(timesteps)
    return np.array(ts, dtype=np.int64), np.array(ts[1:] + [999], dtype=np.int64)


def w_vector_sd15(n_unet_batch, guidance_scale):
    """utils/generation.py:232-235: [0,0,0,gs] iff the CFG-doubled batch is 4, else [gs]*2B."""
    if n_unet_batch == 4:
        return np.array([0.0, 0.0, 0.0, guidance_scale], dtype=np.float32)
    return np.full((n_unet_batch,), guidance_scale, dtype=np.float32)


def prepare_val_prompts(n_items, bs, world_size, rank):
    """running/sd1.5/generate.py:29-39 index partition: array_split into ((N-1)//(bs*W)+1)*W batches, rank::W."""
    num_batches = ((n_items - 1) // (bs * world_size) + 1) * world_size
    parts = np.array_split(np.arange(n_items), num_batches)
    return [p.tolist() for p in parts[rank::world_size]]
