"""Build libicd_amd.so (HIP kernels + native runtime + C ABI) in-tree for gfx950.

    python -m invertible_cd_amd.build          # incremental
    python -m invertible_cd_amd.build --force

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libicd_amd.so")
SOURCES = ["gemm.hip", "gemm_big.hip", "gemm_pp.hip", "gemm_pp320.hip", "norm.hip", "attention.hip", "elementwise.hip", "p2p.hip", "runtime.hip", "error.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# attention.hip: softmax max-chains need no NaN canonicalisation (infinities are still honoured for the key mask)
EXTRA_FLAGS = {"attention.hip": ["-fno-honor-nans"]}


def source_sha():
    """sha1 (12 hex digits) over the kernel sources: the digest stamped into the library (icd_build_sha) and compared at load time."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "icd_amd.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, sha):
    obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    extra = list(EXTRA_FLAGS.get(src, []))
    if src == "error.cpp":                       # carries the digest of ALL sources: rebuilt whenever it changes
        stamp = os.path.join(LIBDIR, "build_sha.txt")
        stamped = open(stamp).read().strip() if os.path.exists(stamp) else None
        force = force or stamped != sha
        extra.append(f'-DICD_BUILD_SHA="{sha}"')
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _deps()):
        return obj, False
    cmd = [HIPCC] + FLAGS + extra + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    sha = source_sha()
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(lambda s: _compile(s, force, sha), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        # hipcc's host pass drops a kernel template it cannot type-check WITHOUT a diagnostic (seen with a device-only builtin type in a
        # kernel body): the object then lacks the fatbin and the kernels' stubs stay undefined until the first launch.  Fail here instead.
        nm = subprocess.run(["nm", "-D", "--undefined-only", LIB], capture_output=True, text=True)
        lost = [l.split()[-1] for l in nm.stdout.splitlines() if "__device_stub__" in l]
        if lost:
            os.remove(LIB)
            raise RuntimeError(f"{len(lost)} kernels have no host stub (device code dropped by the host pass), e.g. {lost[0]}")
        with open(os.path.join(LIBDIR, "build_sha.txt"), "w") as f:
            f.write(sha + "\n")
        if verbose:
            print(f"[icd-amd] built {LIB} ({os.path.getsize(LIB) >> 10} KiB)")
    elif verbose:
        print(f"[icd-amd] {LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
