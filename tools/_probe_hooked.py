import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from invertible_cd_amd import p2p, synthetic, _lib
dev = torch.device("cuda:0")
wl = bench.SD15Workload(dev)
wl.net.set_precision("auto")
p2p.tokenizer = synthetic.SyntheticTokenizer(); p2p.NUM_DDIM_STEPS = 4; p2p.device = "cuda"
g = torch.Generator().manual_seed(0)
B = 8
lat = torch.randn(B, 4, 64, 64, generator=g).to(dev)
ctx = torch.randn(2 * B, 77, 768, generator=g).to(device=dev, dtype=torch.float16)
for name in ("none", "store"):
    for it in range(3):
        ctrl = None if name == "none" else p2p.AttentionStore()
        p2p.register_attention_control(wl.model, ctrl)
        wl.net.reset_context_cache(); wl.solver.context = ctx
        if it == 2: _lib.profile_enable(True)
        wl.solver.cons_generation(lat, guidance_scale=19.0, w_embed_dim=512, dynamic_guidance=True, tau1=0.8, tau2=0.8, controller=ctrl)
        torch.cuda.synchronize()
        if it == 2:
            fam = _lib.profile_read(); recs = _lib.profile_dump(); _lib.profile_enable(False)
        p2p.register_attention_control(wl.model, None)
    print(name, {k: (v["launches"], round(v["ms"], 2)) for k, v in fam.items()}, "total", round(sum(v["ms"] for v in fam.values()), 2))
    if name == "store":
        agg = {}
        for (f, M, N, K, aux, ms, fl) in recs:
            if f in ("softmax", "gemm_batched", "attn_fused"):
                a = agg.setdefault((f, M, N, K, aux), [0, 0.0]); a[0] += 1; a[1] += ms
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            print("   ", k, v[0], round(v[1], 3), "ms", round(v[1] / v[0] * 1e3, 1), "us/call")
