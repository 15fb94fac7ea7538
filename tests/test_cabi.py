"""The C-ABI shared library loads on a CPU-only box and exports exactly the entry points include/icd_amd.h declares
(no compute calls here - those are the -m gpu tests)."""
import ctypes
import os
import re

import pytest

from invertible_cd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "icd_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(icd_[a-z0-9_]+)\s*\(", hdr)) - {"icd_attn_hook"})


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m invertible_cd_amd.build` (or __graft_entry__.build())"
    lib = _lib.load()
    assert lib.icd_version() >= 1
    assert lib.icd_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/icd_amd.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes prototypes out of sync with the header"


def test_struct_layouts_match_header_field_counts():
    hdr = open(os.path.join(ROOT, "include", "icd_amd.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} icd_gemm_desc;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    n_fields = sum(len(decl.split(",")) for decl in body.split(";") if decl.strip())
    assert n_fields == len(_lib.GemmDesc._fields_)
    assert ctypes.sizeof(_lib.GemmDesc) % 8 == 0


def test_argument_validation_without_gpu():
    """Error paths return a status + message before any HIP call (safe on a CPU box)."""
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.icd_gemm(ctypes.byref(d), None) == -1
    assert b"non-null" in lib.icd_last_error()
    with pytest.raises(RuntimeError, match="non-null"):
        _lib.check(lib.icd_gemm(ctypes.byref(d), None), "icd_gemm")
    cfg = _lib.UNetConfig()
    h = ctypes.c_void_p()
    assert lib.icd_unet_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert lib.icd_groupnorm_ws_floats(2, 4096, 32) == 2 * 64 * 32 * 2


def test_unet_plan_on_cpu_dry_run():
    """create / bind / finalize / workspace sizing are host-only: exercise them with host pointers (never dereferenced)."""
    import torch
    from invertible_cd_amd.unet_config import SD15
    lib = _lib.load()
    cfg = SD15.scaled((32, 32, 64, 64), cross_dim=16)
    c = _lib.UNetConfig()
    c.in_channels = c.out_channels = 4
    c.num_levels = 4
    for i in range(4):
        c.block_out_channels[i] = cfg.block_out_channels[i]
        c.down_has_attn[i], c.up_has_attn[i] = int(cfg.down_has_attn[i]), int(cfg.up_has_attn[i])
        c.transformer_layers[i], c.num_heads[i] = 1, 4
    c.layers_per_block, c.cross_dim, c.time_cond_proj_dim, c.norm_groups = 2, 16, 512, 32
    h = ctypes.c_void_p()
    assert lib.icd_unet_create(ctypes.byref(c), ctypes.byref(h)) == 0
    assert lib.icd_unet_num_attention_layers(h) == 32
    assert lib.icd_unet_finalize(h) == -4 and b"not bound" in lib.icd_last_error()
    ws = lib.icd_unet_workspace_bytes(h, 2, 16, 16, 77)
    ws2 = lib.icd_unet_workspace_bytes(h, 4, 16, 16, 77)
    assert 0 < ws < ws2
    lib.icd_unet_destroy(h)


def test_header_is_plain_c_and_the_ctypes_mirrors_have_its_layout(tmp_path):
    """include/icd_amd.h compiles as C99 with nothing but <stdint.h> (the boundary a cgo / JNI / ctypes binding sees), and the
    structs the Python host passes have exactly the size and field offsets the C compiler gives them."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler on this box")
    pairs = {"icd_gemm_desc": _lib.GemmDesc, "icd_unet_config": _lib.UNetConfig, "icd_unet_io": _lib.UNetIO}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "icd_amd.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {(a, b): int(c) for a, b, c in (ln.split() for ln in out.splitlines())}
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, *_ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_the_library_carries_the_digest_of_its_sources_and_a_stale_one_is_refused(monkeypatch):
    """icd_build_sha(): the sha of csrc/ is compiled INTO the .so by build.py; _lib.load() compares it with the sources next to it
    (bench.py reports it as roofline.kernels_sha) - an edited-but-not-rebuilt tree cannot run, let alone report a digest it did not
    execute.  A library selected with ICD_AMD_LIB (another build, for A/B) is exempt."""
    from invertible_cd_amd import build
    lib = _lib.load()
    sha = lib.icd_build_sha().decode()
    assert re.fullmatch(r"[0-9a-f]{12}", sha) and sha == build.source_sha() == _lib.build_sha()
    monkeypatch.setattr(_lib, "_lib", None)                     # force a fresh load against "edited" sources
    monkeypatch.setattr(build, "source_sha", lambda: "0" * 12)
    with pytest.raises(RuntimeError, match="was built from kernel sources"):
        _lib.load()
    monkeypatch.setenv("ICD_AMD_LIB", _lib.LIB_PATH)
    assert _lib.load() is not None
    monkeypatch.setattr(_lib, "_lib", lib)


def test_context_cache_key_survives_inference_mode():
    """Inference tensors have no version counter (`_version` raises): such a context never hits the K / V cache instead of crashing
    every forward under torch.inference_mode()."""
    import torch
    from invertible_cd_amd.unet import UNet2DConditionModel as U
    t = torch.zeros(2, 3)
    assert U._ctx_version(t) == t._version
    t.add_(1)
    assert U._ctx_version(t) == 1
    with torch.inference_mode():
        ti = torch.zeros(2, 3)
    assert U._ctx_version(ti) is None
