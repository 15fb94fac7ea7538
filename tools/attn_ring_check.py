"""Bitwise check of the flash-attention ring against a fully fenced build.

The shipped kernels synchronise a tile hand-over with a counted `s_waitcnt vmcnt(N)` + a bare `s_barrier` (attention.hip, `wait_landed`):
correct only while every wave issues exactly the counted number of LDS-DMA loads per tile.  This tool builds attention.hip a second
time with -DICD_ATTN_DEBUG_SYNC (a full `__syncthreads()` fence at every tile), links it into ab/libicd_attn_sync.so, and runs the
ragged / causal / wide-head / every-head-dim cases through both libraries in two processes: the outputs must be identical bit for bit.

    python tools/attn_ring_check.py            # builds the fenced library (CPU box is enough), then runs both on cuda:0
    python tools/attn_ring_check.py --build    # build only (no GPU needed)
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SYNC_LIB = os.path.join(ROOT, "ab", "libicd_attn_sync.so")

CASES = [  # B, H, Nq, Nk, d, flags (1 causal, 2 prescaled)
    (2, 8, 4096, 4096, 40, 2), (2, 8, 1024, 1024, 80, 2), (2, 8, 256, 256, 160, 2), (3, 8, 64, 64, 160, 0),
    (2, 10, 4096, 4096, 64, 2), (2, 20, 1024, 1024, 64, 2), (2, 5, 1000, 1000, 64, 0), (1, 8, 1024, 77, 40, 0),
    (2, 12, 77, 77, 64, 1), (2, 20, 77, 77, 64, 1), (2, 8, 333, 517, 80, 0), (1, 4, 4096, 4100, 40, 2), (1, 2, 130, 191, 48, 0),
]


def build_sync():
    from invertible_cd_amd import build as B
    B.build(verbose=False)
    os.makedirs(os.path.dirname(SYNC_LIB), exist_ok=True)
    obj = os.path.join(ROOT, "ab", "attention_sync.o")
    deps = [os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC)] + [B.LIB]
    if os.path.exists(SYNC_LIB) and os.path.getmtime(SYNC_LIB) > max(os.path.getmtime(d) for d in deps):
        return                                    # up to date (built by __graft_entry__.build() beside the product library)
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get("attention.hip", []) + ["-DICD_ATTN_DEBUG_SYNC", "-x", "hip", "-c",
                                                                          os.path.join(B.CSRC, "attention.hip"), "-o", obj]
    subprocess.run(cmd, check=True)
    objs = [os.path.join(B.LIBDIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != "attention.hip"] + [obj]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SYNC_LIB] + objs, check=True)
    print(f"built {SYNC_LIB}")


def worker():
    import ctypes as C
    import torch
    from invertible_cd_amd import _lib
    lib = _lib.load()
    for (B_, H, Nq, Nk, d, flags) in CASES:
        g = torch.Generator(device="cuda").manual_seed(Nq * 31 + Nk * 7 + d)
        q = torch.randn(B_ * Nq, H * d, device="cuda", generator=g).half()
        k = torch.randn(B_ * Nk, H * d, device="cuda", generator=g).half()
        v = torch.randn(B_ * Nk, H * d, device="cuda", generator=g).half()
        if flags & 1 and Nq != Nk:
            continue
        ldv = (Nk + 7) // 8 * 8
        vt = torch.zeros(B_, H * d, ldv, device="cuda", dtype=torch.float16)
        vt[:, :, :Nk] = v.reshape(B_, Nk, H * d).transpose(1, 2)
        out = torch.empty_like(q)
        _lib.check(lib.icd_attention_fused_ex(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B_, H, Nq, Nk, d, q.stride(0),
                                              k.stride(0), vt.stride(1), out.stride(0), vt.stride(0), d ** -0.5, flags,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        print("CASE", B_, H, Nq, Nk, d, flags, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest())


def main():
    if "--worker" in sys.argv:
        return worker()
    build_sync()
    if "--build" in sys.argv:
        return
    outs = []
    for lib in (None, SYNC_LIB):
        env = dict(os.environ)
        env.pop("ICD_AMD_LIB", None)
        if lib:
            env["ICD_AMD_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True, check=True)
        outs.append([l for l in r.stdout.splitlines() if l.startswith("CASE")])
    assert len(outs[0]) == len(outs[1]) > 8
    bad = [a for a, b in zip(*outs) if a != b]
    for a, b in zip(*outs):
        print(("OK   " if a == b else "DIFF ") + a)
    print(f"{len(outs[0]) - len(bad)} / {len(outs[0])} cases bit-identical between the counted ring and the fully fenced build")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
