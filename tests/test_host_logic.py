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
