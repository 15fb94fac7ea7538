"""Image inversion entry point (API mirror of the reference's utils/inversion.py).

`invert(...)` keeps the reference signature and return value ((image_gt, image_rec), latent, uncond_embeddings).
The consistency branch (`is_cons_inversion=True` -> Generator.cons_inversion) is the hot path and runs on the native
UNet.  The DDIM-inversion branch is kept (it reuses the same UNet).  Null-text optimisation (`do_nti=True`,
utils/inversion.py:11-48) needs autograd THROUGH the UNet, which the inference-only MI355X executor does not provide:
it raises NotImplementedError (out of scope, baseline only).
"""
from .generation import load_512
from .p2p import register_attention_control


def null_optimization(solver, latents, guidance_scale, num_inner_steps, epsilon):
    raise NotImplementedError("null-text optimisation needs autograd through the UNet; the MI355X-native path is "
                              "inference-only (use is_cons_inversion=True, the iCD forward model)")


def invert(solver, stop_step, is_cons_inversion=False, inv_guidance_scale=1, nti_guidance_scale=8, dynamic_guidance=False,
           tau1=0.4, tau2=0.6, w_embed_dim=0, image_path=None, prompt='', offsets=(0, 0, 0, 0), do_nti=False, do_npi=False,
           num_inner_steps=10, early_stop_epsilon=1e-5, seed=0):
    solver.init_prompt(prompt)
    uncond_embeddings, cond_embeddings = solver.context.chunk(2)
    register_attention_control(solver.model, None)
    if isinstance(image_path, list):
        image_gt = [load_512(path, *offsets) for path in image_path]
    else:
        image_gt = load_512(image_path, *offsets)
    if is_cons_inversion:
        image_rec, latents = solver.cons_inversion(image_gt, w_embed_dim=w_embed_dim, guidance_scale=inv_guidance_scale, seed=seed)
    else:
        image_rec, latents = solver.ddim_inversion(image_gt, n_steps=stop_step, guidance_scale=inv_guidance_scale,
                                                   dynamic_guidance=dynamic_guidance, tau1=tau1, tau2=tau2, w_embed_dim=w_embed_dim)
    if do_nti:
        print("Null-text optimization...")
        uncond_embeddings = null_optimization(solver, latents, nti_guidance_scale, num_inner_steps, early_stop_epsilon)
    elif do_npi:
        uncond_embeddings = [cond_embeddings] * solver.n_steps
    else:
        uncond_embeddings = None
    return (image_gt, image_rec), latents[-1], uncond_embeddings
