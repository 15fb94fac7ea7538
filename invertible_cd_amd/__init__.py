"""invertible_cd_amd - MI355X-native (gfx950) implementation of the iCD few-step U-Net hot path.

Module names mirror the reference's `utils/` package (generation, generation_sdxl, p2p, seq_aligner, inversion, loading,
dist_utils); `unet` holds the native UNet2DConditionModel duck type, `ops` the block-level operators, `_lib` the ctypes
binding of libicd_amd.so (include/icd_amd.h).  See DESIGN.md / INTEGRATION.md.
"""
import importlib

__all__ = ["generation", "generation_sdxl", "p2p", "seq_aligner", "inversion", "loading", "dist_utils", "unet", "ops",
           "synthetic", "schedulers", "pipelines", "unet_config"]


def __getattr__(name):          # lazy submodule import: `import invertible_cd_amd` stays cheap and GPU-free
    if name in __all__:
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
