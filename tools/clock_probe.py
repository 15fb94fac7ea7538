#!/usr/bin/env python3
"""Shader clock and package power (rocm-smi) while ONE GEMM shape runs back to back for 4 s - icd_gemm, then the vendor library
(torch.matmul) on the same operands.  Evidence for DESIGN.md section 10: both sit at the ~1.4 kW package limit; what differs is
how many useful MFMA cycles each buys with it.

    python tools/clock_probe.py            # on the GPU box
"""
import sys, os, subprocess, threading, time, json
import torch
from invertible_cd_amd import ops
def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(o)
        c = d[next(iter(d))]
        return {k: v for k, v in c.items() if "sclk" in k.lower() or "ower" in k or "mclk" in k.lower() or "fclk" in k.lower()}
    except Exception as e:
        return {"err": str(e)}
print("idle", smi(), flush=True)
for (M, N, K) in [(8192, 1280, 11520), (8192, 1280, 5120), (32768, 2560, 640)]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16); wt = w.t()
    for name, fn in [("icd", lambda: ops.gemm(a, w, out=out)), ("lib", lambda: torch.matmul(a, wt, out=out))]:
        stop = False; samples = []
        def sampler():
            time.sleep(0.7)
            while not stop:
                samples.append(smi())
        th = threading.Thread(target=sampler); th.start()
        t0 = time.time(); n = 0
        while time.time() - t0 < 4.0:
            for _ in range(200): fn()
            torch.cuda.synchronize(); n += 200
        dt = time.time() - t0
        stop = True; th.join()
        print(f"{M}x{N}x{K} {name}: {dt / n * 1e6:.1f} us/launch  {2.0 * M * N * K * n / dt / 1e12:.0f} TF", flush=True)
        for s in samples[:6]: print("   ", s, flush=True)
