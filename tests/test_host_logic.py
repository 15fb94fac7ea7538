"""Host-side logic that needs no GPU: the attention-hook adapter's materialisation rule, the SDXL image processor, the
bench launcher's rank spawning command."""
import numpy as np
import torch


def test_hook_adapter_trusts_needs_probs_only_when_it_describes_forward():
    from invertible_cd_amd import p2p

    class ReadsEverything(p2p.AttentionStore):                 # reference-style subclass: overrides forward only
        def forward(self, attn, is_cross, place_in_unet):
            self.seen = getattr(self, "seen", 0) + 1
            return attn

    class SaysSo(p2p.AttentionStore):                          # overrides both: its own rule is trusted
        def forward(self, attn, is_cross, place_in_unet):
            return attn

        def needs_probs(self, is_cross, n_queries, place_in_unet):
            return False

    class Direct(p2p.AttentionControl):                        # direct subclass: default needs_probs is "always"
        def forward(self, attn, is_cross, place_in_unet):
            return attn

    H = p2p.HookAdapter
    assert H(p2p.AttentionStore(), False, "cpu").probs_mode == 1
    assert H(p2p.EmptyControl(), False, "cpu").probs_mode == 1
    a = H(ReadsEverything(), False, "cpu")
    assert a.probs_mode == 2 and not a.native
    buf = a.query(0, False, "down", 8, 4096, 4096, 4096)        # a 64x64 self-attention layer is materialised for it
    assert buf is not None and buf.shape == (8, 4096, 4096)
    assert H(SaysSo(), False, "cpu").probs_mode == 1
    d = H(Direct(), False, "cpu")
    assert d.probs_mode == 2 and d.query(0, False, "down", 2, 64, 64, 64) is not None      # every layer, as the reference gives it
    s = p2p.AttentionStore(); s.num_att_layers = 4
    assert H(s, False, "cpu").query(0, False, "down", 8, 4096, 4096, 4096) is None and s.cur_att_layer == 1


def test_image_processor_preprocess_matches_vae_image_processor_rules():
    from PIL import Image
    from invertible_cd_amd.pipelines import _ImageProcessor
    ip = _ImageProcessor()
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 255, (64, 72, 3), dtype=np.uint8)
    x = ip.preprocess(Image.fromarray(arr))
    assert x.shape == (1, 3, 64, 72) and x.dtype == torch.float32
    assert torch.equal(x, torch.from_numpy(arr.astype(np.float32) / 255.0).permute(2, 0, 1)[None] * 2 - 1)
    y = ip.preprocess([Image.fromarray(arr), Image.fromarray(arr[::-1].copy())])
    assert y.shape == (2, 3, 64, 72) and torch.equal(y[0], x[0])
    z = ip.preprocess(Image.fromarray(rng.integers(0, 255, (70, 75, 3), dtype=np.uint8)))      # rounded down to multiples of 8
    assert z.shape == (1, 3, 64, 72)
    t01 = torch.rand(2, 3, 16, 16)
    assert torch.equal(ip.preprocess(t01), 2 * t01 - 1)          # [0,1] tensors are normalised
    tneg = t01 * 2 - 1
    assert torch.equal(ip.preprocess(tneg), tneg)                # already in [-1,1]: passed through
    lat = torch.randn(1, 4, 8, 8)
    assert ip.preprocess(lat) is lat                             # 4 channels = latents
    back = ip.postprocess(x, output_type="pil")
    assert np.abs(np.asarray(back[0]).astype(int) - arr.astype(int)).max() <= 1


def test_bench_spawn_command(monkeypatch):
    import subprocess
    import bench
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: (seen.update(cmd=cmd, env=env), 0)[1])
    a = type("A", (), {"gpus": 4})()
    try:
        bench.spawn_ranks(a, ["--gpus", "4", "--steps", "2"])
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert "127.0.0.1" in cmd and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    try:
        bench.spawn_ranks(a, [])
        assert False
    except SystemExit as e:
        assert "GPU(s) visible" in str(e.code)


def test_upsample_phase_weights_reproduce_the_upsampled_conv():
    """unet.upsample_phase_weights: nearest-2x upsampling followed by conv3x3 (pad 1) equals, at output pixel (2y + py, 2x + px), a 2 x 2
    conv over input pixels (y + py - 1 + dy, x + px - 1 + dx) with the 3 x 3 taps that land on the same input pixel summed - the identity the
    executor's phase-form upsampler (icd_gemm_desc.conv_ktaps) and the packer rely on.  Checked in fp64 on the CPU, odd sizes included."""
    import torch.nn.functional as F
    from invertible_cd_amd.unet import upsample_phase_weights
    g = torch.Generator().manual_seed(7)
    for (B, C, O, H, W) in ((2, 5, 3, 4, 6), (1, 8, 8, 7, 5), (1, 3, 4, 1, 2)):
        x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(O, C, 3, 3, generator=g, dtype=torch.float64)
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
        out = torch.zeros_like(ref)
        xp = F.pad(x, (1, 1, 1, 1))                                  # zero border: input pixel -1 / H reads as zero, like the conv's padding
        for ph, wp in enumerate(upsample_phase_weights(w)):
            py, px = ph >> 1, ph & 1
            assert wp.shape == (O, 4, C)
            k = wp.double().reshape(O, 2, 2, C).permute(0, 3, 1, 2)  # taps (dy, dx) row-major -> [O, C, 2, 2]
            # output (y, x) of the phase reads padded pixels (y + py + dy, x + px + dx): a valid 2 x 2 conv of the padded map, shifted by (py, px)
            full = F.conv2d(xp, k)                                   # [B, O, H + 1, W + 1]
            out[:, :, py::2, px::2] = full[:, :, py:py + H, px:px + W]
        assert torch.allclose(out, ref, rtol=1e-12, atol=1e-12)


def test_error_carry_encoding_saturates():
    """ops.carry_encode / carry_decode (the host mirror of gemm_common.h carry_of8 / carry_add8): value = fp16 + 2^-14 * bf8(e5m2); beyond
    abs(v) = 2^13 the carry saturates at the largest finite e5m2 value instead of overflowing, and never makes the value worse than fp16."""
    from invertible_cd_amd import ops
    v = torch.cat([torch.randn(4096) * 3, torch.tensor([8191.7, -20000.3, 60000.9, 65000.0, 1e-7, 0.0])])
    hi, c = ops.carry_encode(v)
    full = ops.carry_decode(hi, c)
    assert torch.isfinite(full).all()
    assert ((full - v).abs() <= (hi.float() - v).abs() + 1e-12).all()
    small = v.abs() < 8192
    assert ((full - v).abs()[small] <= 0.126 * (hi.float() - v).abs()[small] + 1e-9).all()      # 2 mantissa bits of the error: <= 1/8 of it left
