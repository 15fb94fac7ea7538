"""A/B of the probability kernel's epilogue (icd_probs_epilogue) on the loops that use it: the 4-step reverse pass of one edit pair
(source + target prompt, dynamic guidance) under each shipped controller, with the edit / self-replacement / store accumulation in the
kernel's epilogue and as passes of their own (`controller.fused_epilogue = False`).  Run on the GPU box:
    python tools/ab_fused_epilogue.py > gpurun_out/ab_fused_epilogue.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from invertible_cd_amd import p2p, synthetic
    dev = torch.device("cuda:0")
    wl = bench.SD15Workload(dev)
    wl.net.set_precision("auto")
    p2p.tokenizer = synthetic.SyntheticTokenizer()
    p2p.NUM_DDIM_STEPS = 4
    p2p.device = "cuda"
    prompts = ["a cat sitting on a bench", "a dog sitting on a bench"]
    makers = {
        "AttentionStore": lambda: p2p.AttentionStore(),
        "AttentionReplace": lambda: p2p.make_controller(prompts, True, 0.5, 0.5),
        "AttentionRefine + reweight + blend": lambda: p2p.make_controller(
            prompts, False, {"default_": 0.6, "dog": (0.0, 0.3)}, 0.4, blend_words=(("cat",), ("dog",)),
            equilizer_params={"words": ("dog",), "values": (2.0,)}),
    }
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, 4, 64, 64, generator=g).to(dev).expand(2, -1, -1, -1).contiguous()
    ctx = torch.randn(4, 77, 768, generator=g).to(device=dev, dtype=torch.float16)
    print("4-step reverse pass of one edit pair (B = 2, 64x64 latents, CFG 19 with dynamic guidance), ms per pass, median of 9")
    print(f"{'controller':40s} {'epilogue':>10s} {'own passes':>11s} {'delta':>8s}")
    for name, make in makers.items():
        ms = {}
        for fused in (True, False):
            ts = []
            for it in range(12):
                ctrl = make()
                ctrl.fused_epilogue = fused
                p2p.register_attention_control(wl.model, ctrl)
                wl.net.reset_context_cache()
                wl.solver.context = ctx
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                wl.solver.cons_generation(lat, guidance_scale=19.0, w_embed_dim=512, dynamic_guidance=True, tau1=0.8, tau2=0.8, controller=ctrl)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                p2p.register_attention_control(wl.model, None)
            ms[fused] = sorted(ts[3:])[4]
        print(f"{name:40s} {ms[True]:10.2f} {ms[False]:11.2f} {100 * (ms[True] / ms[False] - 1):+7.1f}%")


if __name__ == "__main__":
    main()
