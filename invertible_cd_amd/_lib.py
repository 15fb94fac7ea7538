"""ctypes binding of libicd_amd.so (the C ABI declared in include/icd_amd.h).

The product path fails loudly when the HIP extension is missing: there is NO CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ICD_AMD_LIB selects another build of the same library (A/B kernel tuning on one GPU box); never a fallback.
LIB_PATH = os.environ.get("ICD_AMD_LIB") or os.path.join(_HERE, "lib", "libicd_amd.so")

ICD_GEMM_GEGLU = 1
ICD_GEMM_OUT_F32 = 2
ICD_GEMM_OUT_TRANS = 4
ICD_GEMM_PAD_HI = 8
ICD_GEMM_RESID_F32 = 16
ICD_GEMM_LN_COMPUTE = 32
ICD_UNET_OPT_XATTN_FUSION = 1
ICD_UNET_OPT_LN_INLINE_STATS = 2
ICD_UNET_OPT_XATTN_TILE = 3
ICD_UNET_OPT_ATTN_VALU_SCALE = 4
ICD_UNET_OPT_RESIDUAL_MODE = 5
ICD_UNET_OPT_RESIDUAL_F32 = 5           # round-3 name
ICD_UNET_OPT_SPLIT_MASK = 6
ICD_UNET_OPT_UPSAMPLE_PHASES = 7
ICD_UNET_OPT_GEMM_TUNE = 8
ICD_GEMM_TUNE_NO_PP = 0x20000000
ICD_SPLIT_GN, ICD_SPLIT_CONV1, ICD_SPLIT_SHORTCUT, ICD_SPLIT_PROJ_OUT, ICD_SPLIT_DOWN, ICD_SPLIT_SAMPLER_OUT, ICD_SPLIT_UP, ICD_SPLIT_TEMB = 1, 2, 4, 8, 16, 32, 64, 128
ICD_SPLIT_QK = 256
ICD_SPLIT_UP_ALL = 512
ICD_SPLIT_DEFAULT, ICD_SPLIT_ACCURATE, ICD_SPLIT_ALL = 447, 1023, 1023
ICD_RESIDUAL_FP16, ICD_RESIDUAL_F32, ICD_RESIDUAL_CARRY, ICD_RESIDUAL_SPLIT = 0, 1, 2, 3
ICD_ATTN_CAUSAL = 1
ICD_ATTN_Q_PRESCALED = 2
ICD_ATTN_TUNE_MODE0 = 4
ICD_HOOK_QUERY = 0
ICD_HOOK_PROBS = 1


class GemmDesc(C.Structure):
    _fields_ = [
        ("a0", C.c_void_p), ("a1", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("rowbias", C.c_void_p),
        ("resid", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("Nw", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32), ("ld_rowbias", C.c_int32),
        ("rows_per_sample", C.c_int32), ("mode", C.c_int32), ("C0", C.c_int32), ("C1", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32), ("ksize", C.c_int32),
        ("stride", C.c_int32), ("upsample", C.c_int32), ("batch", C.c_int32), ("zdiv", C.c_int32),
        ("a_bs0", C.c_int64), ("a_bs1", C.c_int64), ("w_bs0", C.c_int64), ("w_bs1", C.c_int64),
        ("o_bs0", C.c_int64), ("o_bs1", C.c_int64), ("alpha", C.c_float), ("flags", C.c_int32),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p),
        ("xattn_k", C.c_void_p), ("xattn_vt", C.c_void_p), ("xattn_nk", C.c_int32), ("xattn_ldk", C.c_int32),
        ("xattn_ldvt", C.c_int32), ("xattn_vt_bs", C.c_int64), ("xattn_scale", C.c_float),
        ("ln_eps", C.c_float),
        ("tune_group_m", C.c_int32), ("tune_xattn_tile", C.c_int32), ("debug_timeline", C.c_void_p), ("out_f32", C.c_void_p),
        ("resid_carry", C.c_void_p), ("out_carry", C.c_void_p),
        ("conv_tap_base", C.c_int32), ("conv_ktaps", C.c_int32), ("out_remap_w", C.c_int32), ("out_remap_c", C.c_int32),
    ]


class ProbsEpilogue(C.Structure):
    """icd_probs_epilogue: what the shipped controllers do to P, done in the probability kernel's epilogue."""
    _fields_ = [("acc", C.c_void_p), ("edit_At", C.c_void_p), ("edit_D", C.c_void_p), ("first_cond_sample", C.c_int32),
                ("self_from_base", C.c_int32), ("first_cond_row", C.c_int64), ("edit_count", C.c_int32), ("reserved0", C.c_int32)]


class GemmPlanInfo(C.Structure):
    _fields_ = [("kernel", C.c_int32), ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("ksplit", C.c_int32),
                ("ln_inline", C.c_int32), ("xattn", C.c_int32)]


class UNetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("num_levels", C.c_int32),
        ("block_out_channels", C.c_int32 * 4), ("down_has_attn", C.c_int32 * 4), ("up_has_attn", C.c_int32 * 4),
        ("transformer_layers", C.c_int32 * 4), ("num_heads", C.c_int32 * 4), ("layers_per_block", C.c_int32),
        ("cross_dim", C.c_int32), ("use_linear_projection", C.c_int32), ("time_cond_proj_dim", C.c_int32),
        ("addition_time_embed_dim", C.c_int32), ("add_in_dim", C.c_int32), ("norm_groups", C.c_int32),
    ]


ATTN_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                        C.c_int64, C.c_int64, C.POINTER(C.c_void_p))


class ProfileRow(C.Structure):
    _fields_ = [("kind", C.c_int32), ("launches", C.c_int32), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double),
                ("flops_executed", C.c_double)]


class ProfileRecord(C.Structure):
    _fields_ = [("kind", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("aux", C.c_int32), ("ms", C.c_float),
                ("flops", C.c_double), ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("plan_flags", C.c_int32), ("ksplit", C.c_int32)]


PROF_KINDS = ("gemm_conv", "gemm_dense", "gemm_batched", "attn_fused", "groupnorm", "layernorm", "softmax", "misc", "xattn_fused")


class UNetIO(C.Structure):
    _fields_ = [
        ("sample", C.c_void_p), ("timesteps", C.c_void_p), ("context", C.c_void_p), ("timestep_cond", C.c_void_p),
        ("text_embeds", C.c_void_p), ("time_ids", C.c_void_p), ("eps", C.c_void_p), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64), ("batch", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("n_ctx", C.c_int32),
        ("sample_is_f32", C.c_int32), ("hook", ATTN_HOOK), ("hook_user", C.c_void_p),
        ("kv_cache", C.c_void_p), ("kv_cache_bytes", C.c_int64), ("kv_cache_valid", C.c_int32),
    ]


# name -> (restype, argtypes); mirrors include/icd_amd.h one to one (tests/test_cabi.py checks the export list)
SIGNATURES = {
    "icd_last_error": (C.c_char_p, []),
    "icd_version": (C.c_int, []),
    "icd_build_sha": (C.c_char_p, []),
    "icd_gemm": (C.c_int, [C.POINTER(GemmDesc), C.c_void_p]),
    "icd_gemm_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "icd_groupnorm": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "icd_groupnorm_ws_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "icd_groupnorm_carry": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_carry_expand": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "icd_sinusoid_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_split2_act": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_carry_expand2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_layernorm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                C.c_void_p]),
    "icd_groupnorm_f32_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "icd_absmax": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "icd_split_cast": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "icd_layernorm_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "icd_softmax_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int32,
                                   C.c_void_p]),
    "icd_attention_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                      C.c_float, C.c_void_p]),
    "icd_attention_probs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 8 + [C.c_float, C.c_void_p]),
    "icd_attention_probs_split": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 8 + [C.c_float, C.c_void_p]),
    "icd_attention_probs_ex": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 8 + [C.c_float, C.POINTER(ProbsEpilogue), C.c_void_p]),
    "icd_sinusoid": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_silu": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "icd_conv_in": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_pack_latent": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_conv_out": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_int32, C.c_void_p]),
    "icd_x0_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
                              C.c_void_p]),
    "icd_unet_create": (C.c_int, [C.POINTER(UNetConfig), C.POINTER(C.c_void_p)]),
    "icd_unet_destroy": (None, [C.c_void_p]),
    "icd_unet_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_int64]),
    "icd_unet_finalize": (C.c_int, [C.c_void_p]),
    "icd_unet_num_attention_layers": (C.c_int32, [C.c_void_p]),
    "icd_unet_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "icd_unet_workspace_bytes_ex": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "icd_unet_forward": (C.c_int, [C.c_void_p, C.POINTER(UNetIO), C.c_void_p]),
    "icd_unet_kv_cache_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "icd_attention_fused_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 9
                               + [C.c_int64, C.c_float, C.c_int32, C.c_void_p]),
    "icd_activation": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_embed_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p]),
    "icd_local_blend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_accumulate_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "icd_p2p_cross_edit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "icd_pack_nchw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "icd_conv_out_n": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                 C.c_void_p, C.c_int32, C.c_void_p]),
    "icd_gemm_plan": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmPlanInfo)]),
    "icd_unet_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "icd_profile_enable": (C.c_int, [C.c_int32]),
    "icd_profile_read": (C.c_int, [C.POINTER(ProfileRow), C.c_int32]),
    "icd_profile_dump": (C.c_int, [C.POINTER(ProfileRecord), C.c_int32]),
}

_lib = None


def load():
    """Load libicd_amd.so and set the prototypes.  Raises (never falls back) if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X HIP extension is not built.  Run `python -m invertible_cd_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the product path.")
    # torch FIRST.  The torch wheel bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7 - the SONAME this
    # library needs from /opt/rocm): whichever copy is loaded first serves both.  Loaded the other way round, torch later brings in the
    # rest of its bundled ROCm stack beside /opt/rocm's runtime and every launch of this library fails with "no ROCm-capable device is
    # detected" (seen with __graft_entry__.build() followed by smoke() in one process).  Every caller in this package hands torch tensors
    # to the library anyway.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    # the digest of the sources the binary was built from travels inside it: a tree whose csrc/ was edited but not rebuilt must not
    # run (and must not report a kernels_sha it did not execute).  ICD_AMD_LIB (another build, for A/B) is exempt by design.
    if not os.environ.get("ICD_AMD_LIB"):
        from . import build as _build
        if os.path.isdir(_build.CSRC):
            built, now = lib.icd_build_sha().decode(), _build.source_sha()
            if built != now:
                raise RuntimeError(f"{LIB_PATH} was built from kernel sources {built}, csrc/ is now {now}: run `python -m invertible_cd_amd.build`")
    _lib = lib
    return lib


def build_sha():
    """Digest of the kernel sources the LOADED library was built from (icd_build_sha)."""
    return load().icd_build_sha().decode()


def check(status, what=""):
    if status != 0:
        msg = load().icd_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libicd_amd {what} failed (status {status}): {msg}")


def profile_enable(on=True, only=None):
    """on: record every family; only=[family names]: record just those (less event overhead in a timed region)."""
    arg = int(bool(on))
    if on and only:
        arg = -sum(1 << PROF_KINDS.index(k) for k in only)
    check(load().icd_profile_enable(arg), "icd_profile_enable")


def profile_read():
    """{family: dict(launches, ms, flops, bytes)} - call after synchronising the stream."""
    rows = (ProfileRow * len(PROF_KINDS))()
    n = load().icd_profile_read(rows, len(PROF_KINDS))
    if n < 0:
        check(n, "icd_profile_read")
    return {PROF_KINDS[r.kind]: dict(launches=r.launches, ms=r.ms, flops=r.flops, bytes=r.bytes, flops_executed=r.flops_executed)
            for r in rows[:n]}


def profile_dump():
    """Per-launch records [(family, M, N, K, aux, ms, flops)] - call after synchronising the stream."""
    lib = load()
    n = lib.icd_profile_dump(None, 0)
    recs = (ProfileRecord * max(n, 1))()
    n = lib.icd_profile_dump(recs, n)
    if n < 0:
        check(n, "icd_profile_dump")
    return [(PROF_KINDS[r.kind], r.M, r.N, r.K, r.aux, r.ms, r.flops) for r in recs[:n]]


def profile_plans():
    """Per-launch planner records of the GEMM families: dicts with family, M, N, K, aux, tile (m, n), big (gemm_big.hip tile),
    ln_inline (LayerNorm statistics in the main loop), xattn (fused cross-attention epilogue), ksplit - what a test asserts when
    it claims a forward took a given code path.  Call after synchronising the stream."""
    lib = load()
    n = lib.icd_profile_dump(None, 0)
    recs = (ProfileRecord * max(n, 1))()
    n = lib.icd_profile_dump(recs, n)
    if n < 0:
        check(n, "icd_profile_dump")
    return [dict(family=PROF_KINDS[r.kind], M=r.M, N=r.N, K=r.K, aux=r.aux, ms=r.ms, tile=(r.tile_m, r.tile_n),
                 big=bool(r.plan_flags & 1), ln_inline=bool(r.plan_flags & 2), xattn=bool(r.plan_flags & 4), ksplit=r.ksplit)
            for r in recs[:n] if r.tile_m > 0]
