"""Kernel-level parity: every HIP operator against the same op in plain torch fp32 on CPU (the oracle's building
blocks, oracle/unet_ref.py uses exactly these torch.nn.functional calls).

Inputs are fp16-representable (so both sides see identical values); the HIP kernels accumulate in fp32 and round
once to fp16, hence the tolerance: rel-L2 <= 1e-3 (fp16 has an 11-bit significand: 4.9e-4 relative rounding).
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _ops():
    from invertible_cd_amd import ops
    return ops


def r16(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def to_nhwc(x):   # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def from_nhwc(y, B, H, W):
    return y.reshape(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (300, 64, 72), (77 * 3, 640, 768), (1024, 1280, 2560), (64, 1280, 320),
                                   (130, 8, 8)])
def test_gemm_dense(M, N, K):
    ops = _ops()
    a, w = r16(M, K, seed=1), r16(N, K, seed=2, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    res = r16(M, N, seed=4)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=res.cuda())
    ref = a.float() @ w.float().t() + bias + res.float()
    assert rel_l2(out, ref) < TOL
    # asymmetric check without epilogue extras (catches transposed fragments)
    out2 = ops.gemm(a.cuda(), w.cuda())
    assert rel_l2(out2, a.float() @ w.float().t()) < TOL


@pytest.mark.parametrize("M,N,K", [(512, 256, 4096), (300, 1280, 2048), (64, 128, 1024), (2048, 1280, 2304)])
def test_gemm_split_k_shapes(M, N, K):
    """Small-M / deep-K problems take the split-K path (fp32 partial tiles + fused reduce epilogue)."""
    ops = _ops()
    from invertible_cd_amd import _lib
    a, w = r16(M, K, seed=51), r16(N, K, seed=52, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(53))
    res = r16(M, N, seed=54)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=res.cuda())
    assert rel_l2(out, a.float() @ w.float().t() + bias + res.float()) < TOL
    assert _lib.load().icd_gemm_workspace_bytes(512, 256, 4096) > 0        # this shape really is split


def test_conv3x3_split_k_8x8_level():
    """The 8x8 / 1280-channel resnet conv of the UNet (M = B*64, K = 11520): split-K + 256-row tiles."""
    ops = _ops()
    B, H, W, Cc = 4, 8, 8, 1280
    x = r16(B, Cc, H, W, seed=55)
    w = r16(Cc, Cc, 3, 3, seed=56, scale=(9 * Cc) ** -0.5)
    bias = torch.randn(Cc, generator=torch.Generator().manual_seed(57)) * 0.1
    rb = r16(B, Cc, seed=58)
    out = ops.conv3x3(to_nhwc(x).cuda(), B, H, W, ops.pack_conv_weight(w).cuda(), bias.cuda(), rowbias=rb.cuda())
    ref = to_nhwc(F.conv2d(x.float(), w.float(), bias, padding=1) + rb.float()[:, :, None, None])
    assert rel_l2(out, ref) < TOL


@pytest.mark.parametrize("M,N,K", [(512, 256, 320), (300, 512, 64), (1024, 1280, 1280), (256, 256, 4096)])
def test_gemm_big_tile_dense(M, N, K):
    """256x256 / 8-wave tile family (gemm_big.hip), forced so that small shapes exercise it too (incl. ragged M, split-K)."""
    ops = _ops()
    a, w = r16(M, K, seed=61), r16(N, K, seed=62, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(63))
    res = r16(M, N, seed=64)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=res.cuda(), debug_flags=ops.FORCE_BIG_TILE)
    assert rel_l2(out, a.float() @ w.float().t() + bias + res.float()) < TOL
    perm = ops.geglu_perm(N // 2)
    outg = ops.gemm(a.cuda(), w[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(), geglu=True,
                    debug_flags=ops.FORCE_BIG_TILE)
    g = a.float() @ w.float().t() + bias
    val, gate = g.chunk(2, dim=-1)
    assert rel_l2(outg, val * F.gelu(gate)) < TOL


PP_TILE_FLAG = 5 << 24                               # ICD_GEMM_TUNE_BIG_CFG(4): the ping-pong 256 x 256 tile (gemm_pp.hip)
PP320_TILE_FLAG = 6 << 24                            # ICD_GEMM_TUNE_BIG_CFG(5): the ping-pong 256 x 320 tile (gemm_pp320.hip)


@pytest.mark.parametrize("tile", ["256x256", "256x320", "192x256"])
@pytest.mark.parametrize("M,nn,K", [(512, 1, 64), (300, 2, 128), (1000, 4, 1344), (1024, 4, 320), (256, 1, 4096), (2048, 8, 1280), (777, 1, 192)])
def test_gemm_ping_pong_tile(M, nn, K, tile):
    """The ping-pong ("8-phase") tiles forced on small and ragged shapes: 1, 2, 3, odd and even k-tile counts (prologue / tail of the
    counted-vmcnt pipelines), ragged M, split-K where the planner splits, bias + residual, GEGLU (256-wide tiles), plain."""
    ops = _ops()
    bm, bn = (int(x) for x in tile.split("x"))
    N = nn * bn
    flag, lockstep = {"256x256": (PP_TILE_FLAG, 1 << 24), "256x320": (PP320_TILE_FLAG, 2 << 24), "192x256": (7 << 24, 3 << 24)}[tile]
    a, w = r16(M, K, seed=661), r16(N, K, seed=662, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(663))
    res = r16(M, N, seed=664)
    ref = a.float() @ w.float().t()
    info = _lib_plan(ops, M, N, K, flag)
    assert (info.tile_m, info.tile_n) == (bm, bn)
    for _ in range(3):                                # a race in the staging pipeline would come and go between runs
        out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=res.cuda(), debug_flags=flag)
        assert rel_l2(out, ref + bias + res.float()) < TOL
    plain = ops.gemm(a.cuda(), w.cuda(), debug_flags=flag)
    assert rel_l2(plain, ref) < TOL
    # bit-identical to the lockstep tile of the same shape: same k order, same accumulator layout, same epilogue
    # (K <= 1344 or one tile column... the two plans split K alike: same tile grid)
    assert torch.equal(plain, ops.gemm(a.cuda(), w.cuda(), debug_flags=lockstep))
    if bn == 256:
        perm = ops.geglu_perm(N // 2)
        outg = ops.gemm(a.cuda(), w[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(), geglu=True, debug_flags=flag)
        val, gate = (ref + bias).chunk(2, dim=-1)
        assert rel_l2(outg, val * F.gelu(gate)) < TOL


@pytest.mark.parametrize("cfg_i", [0, 1, 2, 3, 4, 5, 6])
def test_gemm_every_big_tile_configuration(cfg_i):
    """Each gemm_big.hip configuration forced in turn (ICD_GEMM_TUNE_BIG_CFG): dense with bias + residual on a ragged M, the
    transposed (V^T) epilogue and a 3x3 conv with time bias."""
    ops = _ops()
    flag = (cfg_i + 1) << 24
    M, N, K = 1000, 1280, 1344                        # N divisible by 256 and 320; 21 k-tiles (odd)
    a, w = r16(M, K, seed=361), r16(N, K, seed=362, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(363))
    res = r16(M, N, seed=364)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=res.cuda(), debug_flags=flag)
    assert rel_l2(out, a.float() @ w.float().t() + bias + res.float()) < TOL
    plain = ops.gemm(a.cuda(), w.cuda(), debug_flags=flag)
    assert rel_l2(plain, a.float() @ w.float().t()) < TOL
    # V^T: 4 samples of 256 keys
    a2 = r16(1024, 320, seed=365)
    w2 = r16(1280, 320, seed=366, scale=320 ** -0.5)
    vt = ops.project_vt(a2.cuda(), w2.cuda(), 4, 256, 256, debug_flags=flag)
    assert rel_l2(vt, (a2.float() @ w2.float().t()).view(4, 256, 1280).transpose(1, 2)) < TOL
    # conv 3x3, B=2, 16x16, 128 -> 1280 channels
    x = r16(2, 128, 16, 16, seed=367)
    wc = r16(1280, 128, 3, 3, seed=368, scale=(9 * 128) ** -0.5)
    rb = r16(2, 1280, seed=369)
    oc = ops.conv3x3(to_nhwc(x).cuda(), 2, 16, 16, ops.pack_conv_weight(wc).cuda(), bias.cuda(), rowbias=rb.cuda(), debug_flags=flag)
    refc = to_nhwc(F.conv2d(x.float(), wc.float(), bias, padding=1) + rb.float()[:, :, None, None])
    assert rel_l2(oc, refc) < TOL


@pytest.mark.parametrize("M,C,N,flags,geglu", [(300, 320, 640, 0, False), (1024, 640, 1280, 0, False), (512, 1280, 1280, 3 << 24, False),
                                               (1000, 1280, 2560, 2 << 24, False), (1000, 1280, 2560, 1 << 24, True),
                                               (1000, 1280, 1280, 4 << 24, False), (2048, 320, 320, 0x100000, False),
                                               (1000, 1280, 2560, (2 << 24) | 0x800000, False), (1000, 1280, 2560, 5 << 24, False),
                                               (1000, 1280, 2560, 5 << 24, True), (700, 320, 1280, 5 << 24, False),
                                               (1000, 1280, 2560, 6 << 24, False), (700, 320, 1280, 6 << 24, False), (2048, 320, 320, 6 << 24, False),
                                               (1000, 1280, 2560, 7 << 24, False), (1000, 1280, 2560, 7 << 24, True), (700, 320, 1280, 7 << 24, False)])
def test_gemm_computes_the_layernorm_statistics_it_applies(M, C, N, flags, geglu):
    """ICD_GEMM_LN_COMPUTE: the GEMM behind a LayerNorm computes (mean, rstd) of its A rows itself - the big tiles from the MFMA
    operand fragments of their main loop (v_dot2 sums in the waves of tile column 0, an LDS table feeds the epilogue, n-tile 0
    stores them for later launches), every other path with a statistics launch first - and applies them.  Checked against
    torch LayerNorm -> Linear, and the stored statistics against the two-pass kernel; rows carry a common offset of ~5 sigma
    (the one-pass variance E[x^2] - mean^2 loses ~(mean / sigma)^2 x 1e-6 of relative accuracy: 2.5e-5 here)."""
    ops = _ops()
    gen = torch.Generator().manual_seed(381)
    x = (torch.randn(M, C, generator=gen) * 1.5 + torch.randn(M, 1, generator=gen) * 7.5).half()
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=gen), 0.3 * torch.randn(C, generator=gen)
    w = torch.randn(N, C, generator=gen) * C ** -0.5
    b = torch.randn(N, generator=gen)
    w16, s, t = ops.fold_layernorm(w, gamma, beta, b)
    st = torch.full((M, 2), float("nan"), device="cuda")
    if geglu:
        perm = ops.geglu_perm(N // 2)
        out = ops.gemm(x.cuda(), w16[perm].contiguous().cuda(), bias=t[perm].contiguous().cuda(), geglu=True, ln_stats=st,
                       ln_colsum=s[perm].contiguous().cuda(), ln_compute=True, debug_flags=flags)
        g = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.t() + b
        val, gate = g.chunk(2, dim=-1)
        want = val * F.gelu(gate)
    else:
        out = ops.gemm(x.cuda(), w16.cuda(), bias=t.cuda(), ln_stats=st, ln_colsum=s.cuda(), ln_compute=True, debug_flags=flags)
        want = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.t() + b
    assert rel_l2(out, want) < 1.5e-3
    st_ref = ops.layernorm_stats(x.cuda())
    assert not torch.isnan(st).any()
    assert torch.allclose(st[:, 0], st_ref[:, 0], rtol=1e-5, atol=1e-4)
    assert ((st[:, 1] - st_ref[:, 1]).abs() / st_ref[:, 1]).max() < 1e-4
    # a second GEMM normalised by the same LayerNorm reads the stored statistics (to_v after to_qk)
    out2 = ops.gemm(x.cuda(), w16.cuda(), bias=t.cuda(), ln_stats=st, ln_colsum=s.cuda(), debug_flags=flags & ~0x800000) if not geglu else None
    if out2 is not None:
        assert rel_l2(out2, want) < 1.5e-3


@pytest.mark.parametrize("ratio", [50.0, 3.0, 0.0])
@pytest.mark.parametrize("M,C,N,flags", [(1024, 640, 1280, 0x100000), (2048, 320, 320, 0x100000), (512, 1280, 1280, 3 << 24), (512, 1280, 1280, 5 << 24),
                                         (512, 1280, 1280, 6 << 24), (512, 1280, 1280, 7 << 24)])
def test_in_loop_layernorm_statistics_survive_a_large_row_offset(M, C, N, flags, ratio):
    """ICD_GEMM_LN_COMPUTE on the big tiles sums x and x^2 of each row from the MFMA operand fragments (one pass).  E[x^2] - mean^2
    cancels when a row's offset dominates its spread: at |mean| / sigma = 50 the one-pass variance alone is off by ~2.5e-3.  Rows with
    |mean| > 4 sigma take an exact second pass in the kernel; asserted here: rstd within 1e-5 (relative) of an fp64 reference at
    |mean| / sigma = 50, 3 (one-pass side of the switch) and 0, and the normalised product against torch."""
    ops = _ops()
    gen = torch.Generator().manual_seed(391)
    sigma = 0.5
    x = (torch.randn(M, C, generator=gen) * sigma + ratio * sigma * (1 + 0.1 * torch.randn(M, 1, generator=gen))).half()
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=gen), 0.3 * torch.randn(C, generator=gen)
    w = torch.randn(N, C, generator=gen) * C ** -0.5
    b = torch.randn(N, generator=gen)
    w16, s, t = ops.fold_layernorm(w, gamma, beta, b)
    st = torch.full((M, 2), float("nan"), device="cuda")
    d = _lib_plan(ops, M, N, C, flags)
    assert d.ln_inline == 1 and d.kernel == 1, "this case must take the in-loop statistics of a big tile"
    out = ops.gemm(x.cuda(), w16.cuda(), bias=t.cuda(), ln_stats=st, ln_colsum=s.cuda(), ln_compute=True, debug_flags=flags)
    xd = x.double()
    mean = xd.mean(1)
    rstd = 1.0 / torch.sqrt(xd.var(1, unbiased=False) + 1e-5)
    e_mean = float(((st[:, 0].cpu().double() - mean).abs() / mean.abs().clamp_min(sigma)).max())
    e_rstd = float(((st[:, 1].cpu().double() - rstd).abs() / rstd).max())
    print(f"[LN statistics in the main loop, |mean|/sigma = {ratio}: {M}x{C}] max rel error mean {e_mean:.2e} rstd {e_rstd:.2e}")
    assert e_mean < 1e-5 and e_rstd < 1e-5
    want = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5) @ w.t() + b
    # (the rank-1 mean correction of the folded form subtracts two numbers ~ ratio times the result: its fp32 rounding grows with the ratio)
    assert rel_l2(out, want) < (1.5e-3 if ratio <= 3 else 5e-3)


def _lib_plan(ops, M, N, K, flags):
    """icd_gemm_plan of a dense LN_COMPUTE launch of this shape."""
    import ctypes as C
    from invertible_cd_amd import _lib
    d = _lib.GemmDesc()
    buf = torch.empty(8, device="cuda")
    d.a0 = d.w = d.out = buf.data_ptr()
    d.ln_stats = d.ln_colsum = buf.data_ptr()
    d.M, d.N, d.K, d.Nw, d.lda, d.ldw, d.ldo = M, N, K, N, K, K, N
    d.mode, d.batch, d.zdiv, d.alpha, d.flags = 0, 1, 1, 1.0, flags | _lib.ICD_GEMM_LN_COMPUTE
    info = _lib.GemmPlanInfo()
    _lib.check(_lib.load().icd_gemm_plan(C.byref(d), C.byref(info)), "icd_gemm_plan")
    return info


@pytest.mark.parametrize("cfg", [
    dict(B=1, H=16, W=16, C0=128, C1=0, Co=256, stride=1, up=False),
    dict(B=2, H=16, W=16, C0=64, C1=64, Co=256, stride=1, up=False),
    dict(B=2, H=16, W=16, C0=128, C1=0, Co=256, stride=2, up=False),
    dict(B=1, H=8, W=8, C0=128, C1=0, Co=512, stride=1, up=True),
    dict(B=4, H=8, W=8, C0=1280, C1=0, Co=1280, stride=1, up=False),
])
def test_conv_big_tile(cfg):
    ops = _ops()
    B, H, W, C0, C1, Co = cfg["B"], cfg["H"], cfg["W"], cfg["C0"], cfg["C1"], cfg["Co"]
    x = r16(B, C0, H, W, seed=65)
    x2 = r16(B, C1, H, W, seed=66) if C1 else None
    Cin = C0 + C1
    w = r16(Co, Cin, 3, 3, seed=67, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(68)) * 0.1
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
    if cfg["up"]:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float(), bias, stride=cfg["stride"], padding=1)
    Ho, Wo = ref.shape[2:]
    rb = r16(B, Co, seed=69)
    res = r16(B * Ho * Wo, Co, seed=70)
    out = ops.conv3x3(to_nhwc(x).cuda(), B, H, W, ops.pack_conv_weight(w).cuda(), bias.cuda(),
                      x2=None if x2 is None else to_nhwc(x2).cuda(), stride=cfg["stride"], upsample=cfg["up"],
                      resid=res.cuda(), rowbias=rb.cuda(), debug_flags=ops.FORCE_BIG_TILE)
    assert rel_l2(out, to_nhwc(ref + rb.float()[:, :, None, None]) + res.float()) < TOL


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=15, W=9, C0=64, C1=0, Co=128, stride=1, up=False, pad_hi=False, big=False),     # odd, non-square
    dict(B=1, H=13, W=21, C0=128, C1=64, Co=256, stride=1, up=False, pad_hi=False, big=True),   # concat, ragged M in big tiles
    dict(B=2, H=14, W=10, C0=64, C1=0, Co=64, stride=2, up=False, pad_hi=False, big=False),     # stride 2, even sizes
    dict(B=2, H=15, W=11, C0=64, C1=0, Co=64, stride=2, up=False, pad_hi=False, big=False),     # stride 2, odd sizes
    dict(B=1, H=7, W=5, C0=64, C1=0, Co=256, stride=1, up=True, pad_hi=False, big=True),        # upsample of an odd map
    dict(B=2, H=12, W=20, C0=64, C1=0, Co=64, stride=2, up=False, pad_hi=True, big=False),      # VAE Downsample2D padding
    dict(B=1, H=16, W=24, C0=128, C1=0, Co=256, stride=2, up=False, pad_hi=True, big=True),
    dict(B=2, H=9, W=9, C0=24, C1=8, Co=40, stride=1, up=False, pad_hi=False, big=False),       # ragged channels (general loader)
])
def test_conv_odd_and_non_square_maps(cfg):
    """Spatial sizes the UNets never use (they are powers of two) but the VAE accepts (any multiple of 8 in pixels)."""
    ops = _ops()
    B, H, W, C0, C1, Co = cfg["B"], cfg["H"], cfg["W"], cfg["C0"], cfg["C1"], cfg["Co"]
    x = r16(B, C0, H, W, seed=165)
    x2 = r16(B, C1, H, W, seed=166) if C1 else None
    Cin = C0 + C1
    w = r16(Co, Cin, 3, 3, seed=167, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(168)) * 0.1
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
    if cfg["up"]:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    if cfg["pad_hi"]:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w.float(), bias, stride=2)
    else:
        ref = F.conv2d(xin, w.float(), bias, stride=cfg["stride"], padding=1)
    out = ops.conv3x3(to_nhwc(x).cuda(), B, H, W, ops.pack_conv_weight(w).cuda(), bias.cuda(),
                      x2=None if x2 is None else to_nhwc(x2).cuda(), stride=cfg["stride"], upsample=cfg["up"],
                      pad_hi=cfg["pad_hi"], debug_flags=ops.FORCE_BIG_TILE if cfg["big"] else 0)
    assert out.shape[0] == B * ref.shape[2] * ref.shape[3]
    assert rel_l2(out, to_nhwc(ref)) < TOL


@pytest.mark.parametrize("M,N,K,flags", [(8192, 1280, 1280, 0),                                    # planner: 192 x 256 tile, fast variant R32 + O32
                                         (4096, 512, 512, 0x100000 | (1 << 24)),                     # forced 256 x 256
                                         (4096, 640, 640, 0x100000 | (2 << 24)),                     # forced 256 x 320 (general epilogue)
                                         (4096, 640, 640, 0x100000 | (4 << 24)),                     # forced 128 x 320
                                         (200, 320, 320, 0),                                         # 128-wide kernel, ragged M
                                         (512, 1280, 5120, 0)])                                      # split-K + reduce kernel
def test_gemm_fp32_residual_stream_outputs(M, N, K, flags):
    """icd_gemm_desc.out_f32 + ICD_GEMM_RESID_F32 (UNet option residual_f32): out32 = resid32 + a w^T + bias in fp32, out = fp16(out32),
    in place on the fp32 stream like the executor's h <- h + f(h); also the start of a chain (no residual, both outputs)."""
    ops = _ops()
    a, w = r16(M, K, seed=140), (r16(N, K, seed=141).float() * K ** -0.5).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(142))
    h32 = torch.randn(M, N, generator=torch.Generator().manual_seed(143)) * 3.0
    ref = h32.double() + a.double() @ w.double().t() + bias.double()
    stream = h32.clone().cuda()
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=stream, out32=stream, debug_flags=flags)       # in place on the fp32 stream
    assert rel_l2(stream, ref) < 1e-5                                                                # fp32 accumulation and storage
    assert torch.equal(out.cpu(), stream.cpu().half())                                              # the fp16 copy is rounded from the sum
    o32 = torch.empty(M, N, device="cuda")
    out2 = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), out32=o32, debug_flags=flags)              # chain start: no residual
    assert rel_l2(o32, a.double() @ w.double().t() + bias.double()) < 1e-5 and torch.equal(out2.cpu(), o32.cpu().half())


@pytest.mark.parametrize("M,N,K,flags", [(8192, 1280, 1280, 0),                                    # planner: 192 x 256 tile, fast variants
                                         (4096, 512, 512, 0x100000 | (1 << 24)),                     # forced 256 x 256
                                         (4096, 640, 640, 0x100000 | (2 << 24)),                     # forced 256 x 320 (160 accumulators)
                                         (4096, 640, 640, 0x100000 | (4 << 24)),                     # forced 128 x 320
                                         (200, 320, 320, 0),                                         # 128-wide kernel, ragged M
                                         (512, 1280, 5120, 0)])                                      # split-K + reduce kernel
def test_gemm_error_carry_of_the_residual_stream(M, N, K, flags):
    """icd_gemm_desc.resid_carry / out_carry (UNet option residual = 2 and 3, every level of the precision policy): a stream tensor is an fp16 value plus one bf8 byte
    holding what the rounding lost.  h <- h + a w^T + bias in place on (hi, carry) like the executor; also the start of a chain."""
    ops = _ops()
    a, w = r16(M, K, seed=140), (r16(N, K, seed=141).float() * K ** -0.5).half()
    bias = torch.randn(N, generator=torch.Generator().manual_seed(142))
    h32 = torch.randn(M, N, generator=torch.Generator().manual_seed(143)) * 3.0
    hi, lo = ops.carry_encode(h32)
    assert rel_l2(ops.carry_decode(hi, lo), h32) < 0.1 * rel_l2(hi, h32)                             # (the encoding itself: > 10 x closer)
    ref = ops.carry_decode(hi, lo).double() + a.double() @ w.double().t() + bias.double()
    hi_d, lo_d = hi.cuda(), lo.cuda()
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=hi_d, resid_carry=lo_d, out=hi_d, out_carry=lo_d, debug_flags=flags)
    assert out.data_ptr() == hi_d.data_ptr()
    e_hi, e_c = rel_l2(hi_d, ref), rel_l2(ops.carry_decode(hi_d, lo_d), ref)
    print(f"[carry {M}x{N}x{K} flags={flags:#x}] fp16 alone {e_hi:.3e}, with the carry {e_c:.3e}")
    assert e_hi < 3e-4 and e_c < 0.1 * e_hi                                                          # what fp16 lost is back (x >= 10)
    assert (hi_d.cpu() != ref.half()).float().mean() < 2e-3                                          # hi is the fp16 rounding of the sum
    # the same launch without the carries reproduces the plain fp16 result of hi + f
    plain = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=hi.cuda(), debug_flags=flags)
    assert rel_l2(plain, hi.double() + a.double() @ w.double().t() + bias.double()) < 3e-4
    o_hi = torch.empty(M, N, device="cuda", dtype=torch.float16)
    o_lo = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), out=o_hi, out_carry=o_lo, debug_flags=flags)       # chain start: no residual
    ref0 = a.double() @ w.double().t() + bias.double()
    assert rel_l2(ops.carry_decode(o_hi, o_lo), ref0) < 0.1 * rel_l2(o_hi, ref0)
    # a residual WITHOUT a carry beside an output WITH one (general path)
    o2_lo = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    o2 = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), resid=hi.cuda(), out_carry=o2_lo, debug_flags=flags)
    ref2 = hi.double() + ref0
    assert rel_l2(ops.carry_decode(o2, o2_lo), ref2) < 0.1 * rel_l2(o2, ref2)


def test_conv_error_carry():
    """The resnet form: conv2 (3 x 3) + carried residual -> carried output, on the conv tiles (256 x 320 at C = 320)."""
    ops = _ops()
    B, H, W, C = 4, 32, 32, 320
    x = r16(B * H * W, C, seed=150)
    wt = (r16(C, 9 * C, seed=151).float() * (9 * C) ** -0.5).half()
    bias = torch.randn(C, generator=torch.Generator().manual_seed(152))
    h32 = torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(153)) * 2.0
    hi, lo = ops.carry_encode(h32)
    o_lo = torch.empty(B * H * W, C, device="cuda", dtype=torch.uint8)
    out = ops.conv3x3(x.cuda(), B, H, W, wt.cuda(), bias=bias.cuda(), resid=hi.cuda(), resid_carry=lo.cuda(), out_carry=o_lo)
    plain = ops.conv3x3(x.cuda(), B, H, W, wt.cuda(), bias=bias.cuda())
    ref = ops.carry_decode(hi, lo).double() + (plain.cpu().double())          # (plain is within fp16 rounding of the conv; compare sums)
    got = ops.carry_decode(out, o_lo).cpu().double()
    # the conv term itself is only known to fp16 precision from `plain`: the carried sum is checked against hi + lo + conv to that
    # precision, and a second carried launch against the first (the pair must not depend on anything but the values)
    assert rel_l2(got, ref) < 2e-4
    o_lo2 = torch.empty_like(o_lo)
    out2 = ops.conv3x3(x.cuda(), B, H, W, wt.cuda(), bias=bias.cuda(), resid=hi.cuda(), resid_carry=lo.cuda(), out_carry=o_lo2)
    assert torch.equal(out2, out) and torch.equal(o_lo2, o_lo)


def test_gemm_rowbias_alpha_f32():
    ops = _ops()
    B, HW, K, N = 3, 64, 128, 192
    a, w = r16(B * HW, K, seed=5), r16(N, K, seed=6, scale=K ** -0.5)
    rb = r16(B, 2 * N, seed=7)[:, N:]           # strided rowbias (slice of a wider buffer)
    out = ops.gemm(a.cuda(), w.cuda(), rowbias=rb.cuda(), rows_per_sample=HW, alpha=0.5, out_f32=True)
    ref = 0.5 * (a.float() @ w.float().t()) + rb.float().repeat_interleave(HW, 0)
    assert out.dtype == torch.float32
    assert rel_l2(out, ref) < 1e-5


@pytest.mark.parametrize("C", [320, 64])
def test_gemm_geglu(C):
    ops = _ops()
    M = 200
    a = r16(M, C, seed=8)
    w = r16(8 * C, C, seed=9, scale=C ** -0.5)
    bias = torch.randn(8 * C, generator=torch.Generator().manual_seed(10)) * 0.1
    perm = ops.geglu_perm(4 * C)
    out = ops.gemm(a.cuda(), w[perm].contiguous().cuda(), bias=bias[perm].contiguous().cuda(), geglu=True)
    g = a.float() @ w.float().t() + bias
    val, gate = g.chunk(2, dim=-1)
    assert out.shape == (M, 4 * C)
    assert rel_l2(out, val * F.gelu(gate)) < TOL


@pytest.mark.parametrize("cfg", [
    dict(B=2, H=16, W=16, C0=64, C1=0, Co=128, stride=1, up=False),
    dict(B=1, H=32, W=32, C0=320, C1=0, Co=320, stride=1, up=False),
    dict(B=2, H=16, W=16, C0=64, C1=0, Co=64, stride=2, up=False),
    dict(B=2, H=8, W=8, C0=128, C1=0, Co=128, stride=1, up=True),
    dict(B=2, H=8, W=8, C0=128, C1=64, Co=96, stride=1, up=False),
    dict(B=1, H=8, W=8, C0=1280, C1=640, Co=1280, stride=1, up=False),
    dict(B=3, H=6, W=10, C0=32, C1=0, Co=40, stride=1, up=False),
])
def test_conv3x3(cfg):
    ops = _ops()
    B, H, W, C0, C1, Co = cfg["B"], cfg["H"], cfg["W"], cfg["C0"], cfg["C1"], cfg["Co"]
    x = r16(B, C0, H, W, seed=11)
    x2 = r16(B, C1, H, W, seed=12) if C1 else None
    Cin = C0 + C1
    w = r16(Co, Cin, 3, 3, seed=13, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(14)) * 0.1
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], 1)
    if cfg["up"]:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float(), bias, stride=cfg["stride"], padding=1)
    Ho, Wo = ref.shape[2:]
    rb = r16(B, Co, seed=15)
    res = r16(B * Ho * Wo, Co, seed=16)
    out = ops.conv3x3(to_nhwc(x).cuda(), B, H, W, ops.pack_conv_weight(w).cuda(), bias.cuda(),
                      x2=None if x2 is None else to_nhwc(x2).cuda(), stride=cfg["stride"], upsample=cfg["up"],
                      resid=res.cuda(), rowbias=rb.cuda())
    ref = to_nhwc(ref + rb.float()[:, :, None, None]) + res.float()
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < TOL


def test_conv1x1_concat():
    ops = _ops()
    B, H, W, C0, C1, Co = 2, 8, 8, 64, 32, 64
    x, x2 = r16(B, C0, H, W, seed=17), r16(B, C1, H, W, seed=18)
    w = r16(Co, C0 + C1, 1, 1, seed=19, scale=(C0 + C1) ** -0.5)
    out = ops.conv3x3(to_nhwc(x).cuda(), B, H, W, ops.pack_conv_weight(w).cuda(), None, x2=to_nhwc(x2).cuda(), ksize=1)
    ref = to_nhwc(F.conv2d(torch.cat([x, x2], 1).float(), w.float()))
    assert rel_l2(out, ref) < TOL


@pytest.mark.parametrize("B,HW,C0,C1,silu,eps", [(2, 1024, 320, 0, True, 1e-5), (3, 64, 1280, 1280, True, 1e-5),
                                                  (2, 256, 1280, 640, True, 1e-5), (1, 4096, 320, 0, False, 1e-6),
                                                  (2, 100, 32, 0, True, 1e-5), (1, 256, 640, 320, False, 1e-6)])
def test_groupnorm(B, HW, C0, C1, silu, eps):
    ops = _ops()
    Cc = C0 + C1
    x = (r16(B, HW, C0, seed=20).float() * 1.5 + 0.3).half()
    x2 = r16(B, HW, C1, seed=21) if C1 else None
    g = 1 + 0.1 * torch.randn(Cc, generator=torch.Generator().manual_seed(22))
    b = 0.1 * torch.randn(Cc, generator=torch.Generator().manual_seed(23))
    out = ops.groupnorm(x.reshape(B * HW, C0).cuda(), B, HW, g.cuda(), b.cuda(), eps, silu,
                        x2=None if x2 is None else x2.reshape(B * HW, C1).cuda())
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    ref = F.group_norm(xin.permute(0, 2, 1), 32, g, b, eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    assert rel_l2(out, ref.reshape(B * HW, Cc)) < TOL


@pytest.mark.parametrize("B,HW,C0,C1,silu", [(2, 4096, 320, 0, True), (2, 1024, 640, 320, True), (3, 64, 1280, 1280, True),
                                             (1, 4096, 320, 320, False), (2, 100, 32, 0, True)])
def test_groupnorm_reads_the_error_carry_and_writes_the_split_operand(B, HW, C0, C1, silu):
    """icd_groupnorm_carry (UNet option residual = 3, the default): the inputs are tensors of the residual stream, fp16 + one bf8 byte of
    what the rounding lost.  The normalisation reads both (streaming kernels and the one-launch small-map kernel), and the optional
    aux output is the second source of the split shortcut conv: [x2 | lo | lo2] with lo = fp16(2^-14 carry), bit-exact."""
    ops = _ops()
    g32 = torch.Generator().manual_seed(30)
    v0 = torch.randn(B * HW, C0, generator=g32) * 1.5 + 0.3
    v1 = torch.randn(B * HW, C1, generator=g32) if C1 else None
    h0, c0 = ops.carry_encode(v0)
    h1, c1 = ops.carry_encode(v1) if C1 else (None, None)
    Cc = C0 + C1
    g = 1 + 0.1 * torch.randn(Cc, generator=torch.Generator().manual_seed(22))
    b = 0.1 * torch.randn(Cc, generator=torch.Generator().manual_seed(23))
    cu = lambda t: None if t is None else t.cuda()
    out, aux = ops.groupnorm_carry(cu(h0), B, HW, g.cuda(), b.cuda(), 1e-5, silu, carry=cu(c0), x2=cu(h1), carry2=cu(c1), with_aux=True)
    plain = ops.groupnorm(cu(h0), B, HW, g.cuda(), b.cuda(), 1e-5, silu, x2=cu(h1))

    def gn(t0, t1):
        xin = t0 if t1 is None else torch.cat([t0, t1], -1)
        r = F.group_norm(xin.double().reshape(B, HW, Cc).permute(0, 2, 1), 32, g.double(), b.double(), 1e-5).permute(0, 2, 1)
        return (F.silu(r) if silu else r).reshape(B * HW, Cc)
    full0, full1 = ops.carry_decode(h0, c0), (ops.carry_decode(h1, c1) if C1 else None)
    ref = gn(full0, full1)                                 # of the values the stream holds (fp16 + carry)
    # the fp16 output rounding (~1.7e-4) is common to both; what the carry removes is the input rounding, seen before that rounding:
    e_c, e_p = rel_l2(out, ref), rel_l2(plain, ref)
    print(f"[groupnorm carry B={B} HW={HW} C={C0}+{C1}] with carry {e_c:.3e}, fp16 inputs only {e_p:.3e}")
    assert e_c < 2.2e-4 and e_c < 0.85 * e_p
    assert rel_l2(out, ref.half()) < 0.5 * rel_l2(plain, ref.half())           # close to the correctly rounded result of the full value
    # aux: bit-exact layout [x2 | lo0 | lo1]
    lo0 = (c0.view(torch.float8_e5m2).float() / 16384.0).half()
    if C1:
        lo1 = (c1.view(torch.float8_e5m2).float() / 16384.0).half()
        assert torch.equal(aux.cpu(), torch.cat([h1, lo0, lo1], 1))
    else:
        assert torch.equal(aux.cpu(), lo0)
    # a source without a carry beside one with (skip tensors of other modes): same as a zero carry
    if C1:
        o2 = ops.groupnorm_carry(cu(h0), B, HW, g.cuda(), b.cuda(), 1e-5, silu, carry=cu(c0), x2=cu(h1), carry2=None)
        assert rel_l2(o2, gn(full0, h1.float())) < 2.2e-4
    assert torch.equal(ops.carry_expand(cu(c0)).cpu(), lo0)


def test_split_operand_gemm_sees_the_full_value_of_the_stream():
    """The split consumers of residual mode 3: a two-source 1x1 conv over [x | lo] against [W | W] (shortcut conv with a concatenated
    skip: [x0 | x1 | lo0 | lo1] through the GroupNorm's aux output), and a 3x3 stride-2 conv over per-tap [x | lo] (the downsampler).
    Against fp64 on the values the stream holds, the split result must be much closer than the fp16-operand one - also for
    activations small enough that lo is subnormal in fp16."""
    ops = _ops()
    B, H, W, C0, C1, Co = 2, 32, 32, 320, 320, 320
    M = B * H * W
    gg = torch.Generator().manual_seed(40)
    for scale in (1.5, 0.02):
        v0, v1 = torch.randn(M, C0, generator=gg) * scale, torch.randn(M, C1, generator=gg) * scale
        h0, c0 = ops.carry_encode(v0)
        h1, c1 = ops.carry_encode(v1)
        w = (torch.randn(Co, C0 + C1, generator=gg) * (C0 + C1) ** -0.5).half()
        g, b = torch.ones(C0 + C1), torch.zeros(C0 + C1)
        _, aux = ops.groupnorm_carry(h0.cuda(), B, H * W, g.cuda(), b.cuda(), 1e-5, True, carry=c0.cuda(), x2=h1.cuda(), carry2=c1.cuda(),
                                     with_aux=True)
        w2 = torch.cat([w, w], 1).contiguous()
        out = ops.conv3x3(h0.cuda(), B, H, W, w2.cuda(), None, x2=aux, ksize=1, out_f32=True)
        plain = ops.conv3x3(h0.cuda(), B, H, W, w.cuda(), None, x2=h1.cuda(), ksize=1, out_f32=True)
        full = torch.cat([ops.carry_decode(h0, c0), ops.carry_decode(h1, c1)], 1).double()
        ref = full @ w.double().t()
        e_s, e_p = rel_l2(out, ref), rel_l2(plain, ref)
        print(f"[split shortcut, |x| ~ {scale}] split {e_s:.3e}, fp16 operand {e_p:.3e}")
        assert e_p > 1e-4 and e_s < 0.2 * e_p
    # downsampler: 3x3 stride 2 over [x | lo] per tap
    C = 320
    v = torch.randn(M, C, generator=gg)
    h, c = ops.carry_encode(v)
    wt = (torch.randn(C, 9, C, generator=gg) * (9 * C) ** -0.5).half()                         # [O, tap, I]
    w1 = wt.reshape(C, 9 * C).contiguous()
    w2 = torch.cat([wt, wt], 2).reshape(C, 18 * C).contiguous()
    lo = ops.carry_expand(c.cuda())
    out = ops.conv3x3(h.cuda(), B, H, W, w2.cuda(), None, x2=lo, stride=2, out_f32=True)
    plain = ops.conv3x3(h.cuda(), B, H, W, w1.cuda(), None, stride=2, out_f32=True)
    full = ops.carry_decode(h, c).reshape(B, H, W, C).permute(0, 3, 1, 2).double()
    ref = to_nhwc(F.conv2d(full, wt.reshape(C, 3, 3, C).permute(0, 3, 1, 2).double(), stride=2, padding=1))
    e_s, e_p = rel_l2(out, ref), rel_l2(plain, ref)
    print(f"[split downsampler] split {e_s:.3e}, fp16 operand {e_p:.3e}")
    assert e_p > 1e-4 and e_s < 0.2 * e_p


@pytest.mark.parametrize("B,H,W,C,Co", [(2, 16, 16, 1280, 1280), (3, 8, 12, 64, 128), (2, 32, 32, 640, 640), (1, 8, 8, 1280, 1280), (2, 6, 10, 40, 24)])
def test_upsampling_conv_in_phase_form(B, H, W, C, Co):
    """Upsample2D (nearest 2x + conv3x3 pad 1) as four 2 x 2 convs on the input grid with tap-summed weights (icd_gemm_desc.conv_ktaps,
    out_remap_w; unet.upsample_phase_weights): against torch's interpolate + conv2d in fp64 on the fp16 operands, and against the 3 x 3
    form with the upsampling in its loader.  Covers the big conv tiles, split-K + reduce (small maps), the 128-wide kernels and the
    general loader (channels not a multiple of 64), and a carried output."""
    ops = _ops()
    from invertible_cd_amd.unet import upsample_phase_weights
    x = r16(B, C, H, W, seed=60)
    w = r16(Co, C, 3, 3, seed=61, scale=(9 * C) ** -0.5)
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(62))
    ref = to_nhwc(F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), bias.double(), padding=1))
    xn = to_nhwc(x).cuda()
    out = torch.empty(B * 4 * H * W, Co, device="cuda", dtype=torch.float16)
    oc = torch.zeros(B * 4 * H * W, Co, device="cuda", dtype=torch.uint8)
    for ph, wp in enumerate(upsample_phase_weights(w)):
        ops.conv3x3(xn, B, H, W, wp.reshape(Co, -1).half().contiguous().cuda(), bias.cuda(), phase=ph, out=out, out_carry=oc)
    old = ops.conv3x3(xn, B, H, W, ops.pack_conv_weight(w).cuda(), bias.cuda(), upsample=True)
    e_new, e_old = rel_l2(out, ref), rel_l2(old, ref)
    print(f"[upsample phases B={B} {H}x{W} C={C}->{Co}] phase form {e_new:.3e}, 3x3 form {e_old:.3e}")
    assert e_new < 3e-4 and e_old < 3e-4                    # (fp16 output rounding 1.7e-4 + the fp16 rounding of the summed taps)
    assert rel_l2(ops.carry_decode(out, oc), ref) < 0.6 * e_new


def test_phase_form_conv_with_fp32_output_lands_on_the_mapped_rows():
    """the row map of the phase form (icd_gemm_desc.out_remap_w) also applies to ICD_GEMM_OUT_F32 outputs on the big tiles
    (their general epilogue wrote GEMM row m instead of the mapped row)."""
    ops = _ops()
    from invertible_cd_amd.unet import upsample_phase_weights
    B, H, W, C, Co = 2, 16, 16, 1280, 1280
    x = r16(B, C, H, W, seed=70)
    w = r16(Co, C, 3, 3, seed=71, scale=(9 * C) ** -0.5)
    xn = to_nhwc(x).cuda()
    o16 = torch.empty(B * 4 * H * W, Co, device="cuda", dtype=torch.float16)
    o32 = torch.full((B * 4 * H * W, Co), float("nan"), device="cuda", dtype=torch.float32)
    for ph, wp in enumerate(upsample_phase_weights(w)):
        wq = wp.reshape(Co, -1).half().contiguous().cuda()
        ops.conv3x3(xn, B, H, W, wq, None, phase=ph, out=o16)
        ops.conv3x3(xn, B, H, W, wq, None, phase=ph, out=o32, out_f32=True)
    assert torch.isfinite(o32).all()                         # every row of the fp32 output was written ...
    assert torch.equal(o32.half(), o16)                      # ... with the value the fp16 output rounds


def test_error_carry_saturates_instead_of_overflowing():
    """|v| >= 2^13: half an ulp of the fp16 value times the carry's 2^14 scale leaves the e5m2 range; the carry is clamped to the largest
    finite value (57344 = 3.5 in value units) instead of becoming inf / NaN, so the stream stays finite up to the fp16 maximum and is
    never worse than the plain fp16 stream."""
    ops = _ops()
    M, N, K = 256, 256, 64
    gg = torch.Generator().manual_seed(50)
    a, w = r16(M, K, seed=51), (r16(N, K, seed=52).float() * K ** -0.5).half()
    mag = torch.empty(M, N).uniform_(8192.0, 60000.0, generator=gg) * (torch.randint(0, 2, (M, N), generator=gg) * 2 - 1)
    hi, lo = ops.carry_encode(mag)
    assert torch.isfinite(ops.carry_decode(hi, lo)).all()
    o_lo = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    out = ops.gemm(a.cuda(), w.cuda(), resid=hi.cuda(), resid_carry=lo.cuda(), out_carry=o_lo)
    got = ops.carry_decode(out, o_lo).cpu()
    ref = ops.carry_decode(hi, lo).double() + a.double() @ w.double().t()
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) <= rel_l2(out, ref) * 1.0001


@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 1280), (513, 640), (5, 32)])
def test_layernorm(rows, C):
    ops = _ops()
    x = (r16(rows, C, seed=24).float() * 2 + 0.5).half()
    g = 1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(25))
    b = 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(26))
    out = ops.layernorm(x.cuda(), g.cuda(), b.cuda())
    assert rel_l2(out, F.layer_norm(x.float(), (C,), g, b, 1e-5)) < TOL


@pytest.mark.parametrize("rows,cols,ld,scale", [(37, 77, 80, 1.0), (130, 256, 256, 0.158), (65, 1024, 1024, 0.158),
                                                 (19, 1000, 1024, 0.5), (9, 4096, 4096, 0.158), (5, 77, 78, 1.0)])
def test_softmax_rows(rows, cols, ld, scale):
    """fp32 scores -> fp16 probabilities; every register-resident width, the ragged tail and the generic fallback (ld % 4)."""
    ops = _ops()
    s = torch.randn(rows, ld, generator=torch.Generator().manual_seed(27)) * 3
    p = ops.softmax_rows(s.cuda(), cols, ld, scale=scale)
    ref = torch.softmax(s[:, :cols] * scale, -1)
    assert rel_l2(p[:, :cols], ref) < TOL
    if ld > cols:
        assert float(p[:, cols:].abs().max()) == 0.0


def _attn_ref(q, k, v, B, H, Nq, Nk, d):
    qh = q.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3)
    kh = k.float().reshape(B, Nk, H, d).permute(0, 2, 1, 3)
    vh = v.float().reshape(B, Nk, H, d).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B * Nq, H * d), p.reshape(B * H, Nq, Nk)


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(2, 8, 256, 256, 40), (1, 8, 1024, 1024, 40), (2, 8, 256, 77, 40),
                                          (1, 5, 320, 77, 64), (2, 10, 200, 200, 64), (1, 8, 64, 64, 160),
                                          (1, 8, 256, 256, 80), (1, 2, 100, 50, 8), (1, 4, 128, 192, 16),
                                          (1, 2, 64, 77, 32)])
def test_attention_fused(B, H, Nq, Nk, d):
    ops = _ops()
    Cc = H * d
    q, k, v = r16(B * Nq, Cc, seed=28), r16(B * Nk, Cc, seed=29), r16(B * Nk, Cc, seed=30)
    ld = (Nk + 7) // 8 * 8
    w_id = torch.eye(Cc).half()
    vt = ops.project_vt(v.cuda(), w_id.cuda(), B, Nk, ld)                      # V^T through the GEMM epilogue
    ref_vt = v.float().reshape(B, Nk, Cc).permute(0, 2, 1)
    assert torch.equal(vt[:, :, :Nk].float().cpu(), ref_vt)
    assert float(vt[:, :, Nk:].abs().max()) == 0.0 if ld > Nk else True
    out = ops.attention_fused(q.cuda(), k.cuda(), vt, B, H, Nq, Nk, d, d ** -0.5)
    ref, _ = _attn_ref(q, k, v, B, H, Nq, Nk, d)
    assert rel_l2(out, ref) < 2e-3      # P is rounded to fp16 before P.V (as in the fp16 reference path)


def test_attention_fused_spiked_rescale():
    """Force the online-softmax rescale branch: one key dominates a late tile."""
    ops = _ops()
    B, H, Nq, Nk, d = 1, 2, 128, 512, 64
    q, k, v = r16(B * Nq, H * d, seed=31), r16(B * Nk, H * d, seed=32), r16(B * Nk, H * d, seed=33)
    k[300] = q[5] * 4.0            # key 300 spikes against query 5 (tile 4)
    k[17] = q[77] * 3.0
    vt = ops.project_vt(v.cuda(), torch.eye(H * d).half().cuda(), B, Nk, Nk)
    out = ops.attention_fused(q.cuda(), k.cuda(), vt, B, H, Nq, Nk, d, d ** -0.5)
    ref, _ = _attn_ref(q, k, v, B, H, Nq, Nk, d)
    assert rel_l2(out, ref) < 2e-3


def _attn_ref_exp2(qp, k, v, B, H, Nq, Nk, d, causal=False):
    """softmax with a PRE-SCALED query: p = 2^(q'.k) / sum (what ICD_ATTN_Q_PRESCALED promises), fp64 accumulation."""
    qh = qp.double().reshape(B, Nq, H, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, Nk, H, d).permute(0, 2, 1, 3)
    vh = v.double().reshape(B, Nk, H, d).permute(0, 2, 1, 3)
    e = qh @ kh.transpose(-1, -2) * 0.6931471805599453
    if causal:
        e = e.masked_fill(torch.ones(Nq, Nk, dtype=torch.bool).triu(1), float("-inf"))
    return (torch.softmax(e, -1) @ vh).permute(0, 2, 1, 3).reshape(B * Nq, H * d)


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(4, 8, 4096, 512, 40),       # MODE 1, two query tiles per wave, P.V on 16x16x32 MFMAs
                                          (4, 8, 4000, 333, 40),      # ... ragged last key tile, ragged query tile
                                          (1, 8, 1024, 1024, 40),     # MODE 1, one query tile per wave (too few blocks for the wide kernel)
                                          (2, 8, 256, 256, 40),
                                          (2, 8, 200, 333, 40),       # ... ragged last key tile, ragged query tile
                                          (2, 10, 256, 320, 64),      # MODE 2 (accumulators start from -M)
                                          (1, 5, 100, 77, 64), (1, 8, 256, 256, 80),
                                          (1, 8, 64, 64, 160)])       # no fast instantiation: VALU form with scale_log2 = 1
@pytest.mark.parametrize("valu", [False, True])
def test_attention_fused_with_a_prescaled_query(B, H, Nq, Nk, d, valu):
    """ICD_ATTN_Q_PRESCALED: q carries d^-1/2 * log2(e) (the executor folds it into the query projection), the kernels use q.k as
    the base-2 exponent and let the MFMA subtract the running offset (head dim 40: a ones column in the first padded head-dim
    slot of K against -M in the same slot of q; 64 / 80: accumulators initialised to -M).  Checked against an fp64 softmax of the
    SAME prescaled fp16 q; `valu` = ICD_ATTN_TUNE_MODE0, the scale / offset FMA on the VALU (must agree with the fast form)."""
    ops = _ops()
    Cc = H * d
    q, k, v = r16(B * Nq, Cc, seed=128), r16(B * Nk, Cc, seed=129), r16(B * Nk, Cc, seed=130)
    qp = (q.float() * (d ** -0.5 * 1.4426950408889634)).half()
    ld = (Nk + 7) // 8 * 8
    vt = ops.project_vt(v.cuda(), torch.eye(Cc).half().cuda(), B, Nk, ld)
    out = ops.attention_fused(qp.cuda(), k.cuda(), vt, B, H, Nq, Nk, d, 123.0, prescaled=True, valu_scale=valu)     # scale is ignored
    ref = _attn_ref_exp2(qp, k, v, B, H, Nq, Nk, d)
    e = rel_l2(out, ref)
    assert torch.isfinite(out).all() and e < 1.5e-3, e
    base, _ = _attn_ref(q, k, v, B, H, Nq, Nk, d)                  # and the un-prescaled definition, up to the rounding of q'
    assert rel_l2(out, base) < 2e-3


@pytest.mark.parametrize("d,Nq", [(40, 256), (40, 1024), (40, 16384), (64, 128), (80, 128)])      # 16384 queries: the wide d = 40 kernel
def test_attention_fused_prescaled_rescale_branch_and_extreme_offsets(d, Nq):
    """The lazy rescale of MODE 1 / 2 under stress: (a) a key that dominates a LATE tile (M moves after O has accumulated: O, l,
    this tile's scores and the offset the MFMA carries must all move exactly once), (b) rows whose exponents sit far from zero in
    both directions (exponents of about +-100: the fp16 offset operand of MODE 1 must stay exact), (c) M moving
    several times in a row (keys of growing magnitude).  fp64 reference on the same prescaled q."""
    ops = _ops()
    B, H, Nk = 1, 8 if d == 40 else 2, 512
    Cc = H * d
    q, k, v = r16(B * Nq, Cc, seed=131), r16(B * Nk, Cc, seed=132), r16(B * Nk, Cc, seed=133)
    qp = (q.float() * (d ** -0.5 * 1.4426950408889634)).half()
    k = k.clone()
    k[300] = q[5] * 4.0                                  # (a) spikes in tile 4 against query 5 (head-wise: every head of row 5)
    k[17] = q[77] * 3.0
    for j, f in enumerate((1.5, 2.5, 3.5, 4.5, 6.0)):    # (c) the maximum of query 9 grows tile after tile
        k[70 + 64 * j] = q[9] * f
    qp[11] = qp[11] * 24.0                               # (b) large |exponents|: the scores of row 11 spread over about +-100
    vt = ops.project_vt(v.cuda(), torch.eye(Cc).half().cuda(), B, Nk, Nk)
    ref = _attn_ref_exp2(qp, k, v, B, H, Nq, Nk, d)
    for valu in (False, True):
        out = ops.attention_fused(qp.cuda(), k.cuda(), vt, B, H, Nq, Nk, d, 1.0, prescaled=True, valu_scale=valu)
        assert torch.isfinite(out).all()
        err_rows = (out.float().cpu() - ref.float()).norm(dim=1) / ref.float().norm(dim=1).clamp_min(1e-6)
        assert rel_l2(out, ref) < 2e-3 and float(err_rows.max()) < 2e-2, (valu, rel_l2(out, ref), float(err_rows.max()), int(err_rows.argmax()))


def test_attention_fused_prescaled_causal():
    """Causal mask with the MFMA-carried offset (head dim 64, the CLIP text-encoder shape): masked scores are -inf before the
    maximum and stay -inf under the offset subtraction."""
    ops = _ops()
    B, H, N, d = 2, 4, 77, 64
    q, k, v = r16(B * N, H * d, seed=134), r16(B * N, H * d, seed=135), r16(B * N, H * d, seed=136)
    qp = (q.float() * (d ** -0.5 * 1.4426950408889634)).half()
    vt = ops.project_vt(v.cuda(), torch.eye(H * d).half().cuda(), B, N, 80)
    out = ops.attention_fused(qp.cuda(), k.cuda(), vt, B, H, N, N, d, 1.0, causal=True, prescaled=True)
    assert rel_l2(out, _attn_ref_exp2(qp, k, v, B, H, N, N, d, causal=True)) < 1.5e-3


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(2, 8, 256, 77, 40), (1, 8, 256, 256, 40), (2, 4, 64, 64, 160), (1, 5, 128, 77, 64)])
def test_attention_materialised(B, H, Nq, Nk, d):
    ops = _ops()
    Cc = H * d
    q, k, v = r16(B * Nq, Cc, seed=34), r16(B * Nk, Cc, seed=35), r16(B * Nk, Cc, seed=36)
    ld = (Nk + 7) // 8 * 8
    vt = ops.project_vt(v.cuda(), torch.eye(Cc).half().cuda(), B, Nk, ld)
    s = ops.attention_scores(q.cuda(), k.cuda(), B, H, Nq, Nk, d, d ** -0.5, ld)
    p = ops.softmax_rows(s.reshape(B * H * Nq, ld), Nk, ld).reshape(B * H, Nq, ld)
    out = ops.attention_apply(p, vt, B, H, Nq, d)
    ref, pref = _attn_ref(q, k, v, B, H, Nq, Nk, d)
    assert rel_l2(p[:, :, :Nk], pref) < TOL
    assert rel_l2(out, ref) < 2e-3


def test_sinusoid_silu():
    ops = _ops()
    t = torch.tensor([999.0, 779.0, 19.0, 0.0])
    out = ops.sinusoid(t.cuda(), 320, 0)
    half = 160
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ref = torch.cat([torch.cos(t[:, None] * f), torch.sin(t[:, None] * f)], -1)
    assert float((out.float().cpu() - ref).abs().max()) < 2e-3      # fp16 output of values in [-1, 1]
    w = torch.tensor([0.0, 7.0, 19.0])
    out = ops.sinusoid(w.cuda(), 512, 1)
    import numpy as np
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "wembed.npz"))
    assert float((out.float().cpu() - torch.from_numpy(g["emb512"][:3])).abs().max()) < 4e-3
    x = r16(64, 1280, seed=37)
    assert rel_l2(ops.silu(x.cuda()), F.silu(x.float())) < TOL


@pytest.mark.parametrize("f32", [False, True])
def test_conv_in_out(f32):
    ops = _ops()
    B, H, W, Cc = 2, 16, 24, 320
    x = r16(B, 4, H, W, seed=38)
    w = r16(Cc, 4, 3, 3, seed=39, scale=1 / 6)
    b = torch.randn(Cc, generator=torch.Generator().manual_seed(40)) * 0.1
    xin = x.float() if f32 else x
    out = ops.conv_in(xin.cuda(), ops.pack_conv_weight(w).cuda(), b.cuda())
    assert rel_l2(out, to_nhwc(F.conv2d(x.float(), w.float(), b, padding=1))) < TOL
    h = r16(B, Cc, H, W, seed=41)
    wo = r16(4, Cc, 3, 3, seed=42, scale=(9 * Cc) ** -0.5)
    bo = torch.randn(4, generator=torch.Generator().manual_seed(43)) * 0.1
    eps = ops.conv_out(to_nhwc(h).cuda(), B, H, W, ops.pack_conv_weight(wo).cuda(), bo.cuda(),
                       torch.float32 if f32 else torch.float16)
    assert eps.shape == (B, 4, H, W)
    assert rel_l2(eps, F.conv2d(h.float(), wo.float(), bo, padding=1)) < (1e-5 if f32 else TOL)


@pytest.mark.parametrize("Cc,H,W", [(512, 9, 13), (640, 8, 8), (128, 20, 12), (320, 64, 64)])
def test_conv_out_channel_widths(Cc, H, W):
    """conv_out across channel widths: one wave per pixel striding over the 8-channel chunks (320, 512, 640 channels), lane groups per
    pixel below (128: the VAE decoder); ragged pixel counts, 3 and 4 output channels, fp16 and fp32 results."""
    ops = _ops()
    B = 2
    h = r16(B, Cc, H, W, seed=44)
    wo = r16(4, Cc, 3, 3, seed=45, scale=(9 * Cc) ** -0.5)
    bo = torch.randn(4, generator=torch.Generator().manual_seed(46)) * 0.1
    ref = F.conv2d(h.float(), wo.float(), bo, padding=1)
    for dt, tol in ((torch.float32, 1e-5), (torch.float16, TOL)):
        eps = ops.conv_out(to_nhwc(h).cuda(), B, H, W, ops.pack_conv_weight(wo).cuda(), bo.cuda(), dt)
        assert rel_l2(eps, ref) < tol, (Cc, dt)
    eps3 = ops.conv_out(to_nhwc(h).cuda(), B, H, W, ops.pack_conv_weight(wo).cuda(), bo.cuda(), torch.float32, cout=3)
    assert rel_l2(eps3, ref[:, :3]) < 1e-5


def test_x0_step_bit_exact_vs_golden(golden_dir):
    """predicted_origin: fp32 arithmetic in the reference's evaluation order -> bit-exact against the vectors
    captured from utils/generation.py:136-155."""
    import numpy as np
    import os
    ops = _ops()
    g = np.load(os.path.join(golden_dir, "predicted_origin.npz"))
    tab = np.load(os.path.join(golden_dir, "alphas_cumprod.npz"))
    # the sqrt tables are part of the fixture: torch's CPU sqrt is not correctly rounded and differs between hosts
    alpha, sigma = torch.from_numpy(tab["alpha_table"]), torch.from_numpy(tab["sigma_table"])
    x, eps = torch.from_numpy(g["x"]), torch.from_numpy(g["eps"])
    for (t, s), ref in zip(g["pairs"].tolist(), g["out"]):
        a_s, s_s = (1.0, 0.0) if s == 0 else (float(alpha[s]), float(sigma[s]))
        coef = torch.tensor([[float(alpha[t]), float(sigma[t]), a_s, s_s]] * 2)
        out = ops.x0_step(x.cuda(), eps.cuda(), coef)
        assert torch.equal(out.cpu(), torch.from_numpy(ref)), (t, s)
    # fp16 latents / fp16 eps -> fp32 result (what torch type promotion gives the reference's fp16 pipelines)
    coef = torch.tensor([[float(alpha[779]), float(sigma[779]), float(alpha[519]), float(sigma[519])]] * 2)
    out = ops.x0_step(x.half().cuda(), eps.half().cuda(), coef, out_dtype=torch.float32)
    assert torch.equal(out.cpu(), torch.from_numpy(g["out_fp16in"]))


def test_errors_raise():
    ops = _ops()
    a, w = r16(16, 12, seed=1), r16(8, 12, seed=2)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.gemm(a.cuda(), w.cuda())


# ------------------------------------------------------------------------------ LayerNorm fused into the consuming GEMM
@pytest.mark.parametrize("M,C,N,flags", [(300, 320, 640, 0), (1024, 640, 1280, 0), (512, 1280, 1280, 3 << 24), (2048, 320, 320, 0x100000),
                                         (131, 64, 128, 0), (512, 96, 256, 0)])
def test_layernorm_fused_into_gemm(M, C, N, flags):
    """out = LayerNorm(x; gamma, beta) @ W^T + b without materialising LayerNorm(x): statistics kernel + gamma folded into W +
    rank-1 epilogue correction (icd_gemm_desc.ln_stats / ln_colsum), every tile family; plain, GEGLU and transposed."""
    ops = _ops()
    g = torch.Generator().manual_seed(95)
    x = ((torch.randn(M, C, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 2.0)).half()      # rows with a large mean
    gamma = (1 + 0.2 * torch.randn(C, generator=g))
    beta = 0.3 * torch.randn(C, generator=g)
    w = r16(N, C, seed=96, scale=C ** -0.5)
    b = 0.1 * torch.randn(N, generator=g)
    res = r16(M, N, seed=97)
    st = ops.layernorm_stats(x.cuda())
    mu, var = x.float().mean(1), x.float().var(1, unbiased=False)
    assert torch.allclose(st[:, 0].cpu(), mu, atol=1e-5, rtol=1e-5) and torch.allclose(st[:, 1].cpu(), (var + 1e-5).rsqrt(), rtol=1e-4)
    w16, s, t = ops.fold_layernorm(w, gamma, beta, b)
    ln = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    ref = ln @ w.float().t() + b
    out = ops.gemm(x.cuda(), w16.cuda(), bias=t.cuda(), resid=res.cuda(), ln_stats=st, ln_colsum=s.cuda(), debug_flags=flags)
    e = rel_l2(out, ref + res.float())
    assert e < TOL, e
    if N % 128 == 0:                                            # GEGLU epilogue
        perm = ops.geglu_perm(N // 2)
        outg = ops.gemm(x.cuda(), w16[perm].contiguous().cuda(), bias=t[perm].contiguous().cuda(), geglu=True, ln_stats=st,
                        ln_colsum=s[perm].contiguous().cuda(), debug_flags=flags)
        val, gate = ref.chunk(2, dim=-1)
        assert rel_l2(outg, val * F.gelu(gate)) < TOL
    if M % 64 == 0:                                             # transposed (V^T) epilogue: no bias term (W beta rides elsewhere)
        Bs = 2 if M % 128 == 0 else 1
        n_tok = M // Bs
        w16v, sv, _ = ops.fold_layernorm(w, gamma, torch.zeros(C))
        vt = ops.project_vt(x.cuda(), w16v.cuda(), Bs, n_tok, n_tok, ln_stats=st, ln_colsum=sv.cuda())
        refv = (F.layer_norm(x.float(), (C,), gamma, None, 1e-5) @ w.float().t()).reshape(Bs, n_tok, N).transpose(1, 2)
        assert rel_l2(vt, refv) < TOL


# ------------------------------------------------------------------------------ query projection + cross-attention, one launch
@pytest.mark.parametrize("B,n_tok,C,X,nk,ln", [(2, 256, 128, 64, 77, False), (1, 1024, 1280, 128, 77, True), (3, 256, 256, 64, 13, True),
                                               (2, 512, 640, 96, 96, False), (2, 1024, 1280, 128, 77, True), (3, 768, 1280, 64, 77, False),
                                               (2, 2048, 768, 64, 96, True)])
def test_query_projection_with_fused_cross_attention(B, n_tok, C, X, nk, ln):
    """The north-star kernel: q = LN(h) W_q^T never leaves the accumulators; S^T = K q^T, softmax over <= 96 keys and
    O^T = V^T P^T run in the GEMM's epilogue (heads of 64).  Reference: torch fp32, the op sequence of utils/p2p.py:321-342."""
    ops = _ops()
    g = torch.Generator().manual_seed(123)
    H = C // 64
    h = (torch.randn(B * n_tok, C, generator=g) * 1.2 + torch.randn(B * n_tok, 1, generator=g)).half()
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    wq = r16(C, C, seed=124, scale=C ** -0.5 * 3.0)                  # peaky softmax
    ctx = r16(B * nk, X, seed=125)
    wk, wv = r16(C, X, seed=126, scale=X ** -0.5 * 2.0), r16(C, X, seed=127, scale=X ** -0.5)
    scale = 64 ** -0.5
    x = F.layer_norm(h.float(), (C,), gamma, beta, 1e-5) if ln else h.float()
    q = (x @ wq.float().t()).reshape(B, n_tok, H, 64).permute(0, 2, 1, 3)
    kk = (ctx.float() @ wk.float().t()).half().float().reshape(B, nk, H, 64).permute(0, 2, 1, 3)
    vv = (ctx.float() @ wv.float().t()).half().float().reshape(B, nk, H, 64).permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ kk.transpose(-1, -2) * scale, -1) @ vv).permute(0, 2, 1, 3).reshape(B * n_tok, C)
    k_dev = ops.gemm(ctx.cuda(), wk.cuda())                             # [B*nk, C]
    ld = (nk + 7) // 8 * 8
    vt_dev = ops.project_vt(ctx.cuda(), wv.cuda(), B, nk, ld)           # [B, C, ld], pad keys zero
    if ln:
        st = ops.layernorm_stats(h.cuda())
        w16, s, t = ops.fold_layernorm(wq, gamma, beta)
        out = ops.query_cross_attention(h.cuda(), w16.cuda(), k_dev, vt_dev, B, n_tok, nk, scale, bias=t.cuda(), ln_stats=st,
                                        ln_colsum=s.cuda())
    else:
        out = ops.query_cross_attention(h.cuda(), wq.cuda(), k_dev, vt_dev, B, n_tok, nk, scale)
    e = rel_l2(out, ref)
    print(f"[xattn fused B={B} n={n_tok} C={C} nk={nk} ln={ln}] rel-L2 = {e:.3e}")
    assert e < 2e-3                                                    # P is rounded to fp16 before P.V, like attention_fused
    if ln:                                                             # the launch computes the LayerNorm statistics itself
        st2 = torch.full_like(st, float("nan"))
        out2 = ops.query_cross_attention(h.cuda(), w16.cuda(), k_dev, vt_dev, B, n_tok, nk, scale, bias=t.cuda(), ln_stats=st2,
                                         ln_colsum=s.cuda(), ln_compute=True)
        assert rel_l2(out2, ref) < 2e-3 and torch.allclose(st2, st, rtol=1e-4, atol=1e-4)
    # and against the unfused kernels of this library on the same operands (q rounded to fp16 in between)
    qd = ops.gemm(h.cuda(), w16.cuda(), bias=t.cuda(), ln_stats=st, ln_colsum=s.cuda()) if ln else ops.gemm(h.cuda(), wq.cuda())
    un = ops.attention_fused(qd, k_dev, vt_dev, B, H, n_tok, nk, 64, scale)
    assert rel_l2(out, un) < 2e-3


@pytest.mark.parametrize("B,n_tok,C,nk,ln", [(8, 1024, 1280, 77, True), (2, 1024, 1280, 77, False), (3, 768, 1280, 96, True), (2, 2048, 768, 77, True),
                                             (5, 256, 1280, 13, True)])
def test_fused_cross_attention_on_the_192_row_tile_gives_the_bits_of_the_256_row_tile(B, n_tok, C, nk, ln):
    """Round 5: the fused launch on 192 x 256 blocks laid out per sample - ceil(n_tok / 192) m-tiles, the last one partial (1024 = 5 x 192 +
    64, 2048 = 10 x 192 + 128, 256 = 192 + 64; 768 = 4 x 192 exactly) - so that SDXL's 1024-token layers at 8 images per GPU make 240 blocks
    instead of 160.  Same k order, same MFMAs, same epilogue per query row: the output and the LayerNorm statistics the launch stores are
    bit-identical to the 256 x 256 host tile; nothing outside a sample's rows is read into a block or written by it (NaN canaries behind
    the output and in the statistics buffer of the 256-row run are irrelevant here: the outputs are fresh allocations of exactly M rows,
    and the op test above covers the values)."""
    ops = _ops()
    g = torch.Generator().manual_seed(321)
    h = (torch.randn(B * n_tok, C, generator=g) * 1.1 + 0.5 * torch.randn(B * n_tok, 1, generator=g)).half().cuda()
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    wq = r16(C, C, seed=322, scale=C ** -0.5 * 3.0)
    ctx = r16(B * nk, 64, seed=323).cuda()
    k_dev = ops.gemm(ctx, r16(C, 64, seed=324, scale=0.25).cuda())
    vt_dev = ops.project_vt(ctx, r16(C, 64, seed=325, scale=0.125).cuda(), B, nk, (nk + 7) // 8 * 8)
    outs, stats = {}, {}
    for tile in (5, 6, 0):
        if ln:
            w16, s, t = ops.fold_layernorm(wq, gamma, beta)
            st = torch.full((B * n_tok, 2), float("nan"), device="cuda")
            outs[tile] = ops.query_cross_attention(h, w16.cuda(), k_dev, vt_dev, B, n_tok, nk, 0.125, bias=t.cuda(), ln_stats=st, ln_colsum=s.cuda(),
                                                   ln_compute=True, xattn_tile=tile)
            stats[tile] = st
        else:
            outs[tile] = ops.query_cross_attention(h, wq.cuda(), k_dev, vt_dev, B, n_tok, nk, 0.125, xattn_tile=tile)
    assert torch.isfinite(outs[6].float()).all()
    assert torch.equal(outs[5], outs[6]) and torch.equal(outs[0], outs[6])
    if ln:
        assert torch.isfinite(stats[6]).all() and torch.equal(stats[5], stats[6])


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(2, 8, 256, 77, 40), (1, 8, 1024, 77, 80), (2, 4, 64, 13, 160), (2, 8, 1024, 1024, 80),
                                         (1, 8, 256, 256, 160), (3, 2, 100, 90, 64), (1, 2, 200, 333, 48)])
def test_attention_probs_one_pass(B, H, Nq, Nk, d):
    """Materialised probabilities without the fp32 score tensor (icd_attention_probs): exact softmax for <= 96 keys, two
    sweeps (running max / sum, then P) for longer rows; vs torch fp32 softmax(q k^T * scale), pad columns zero."""
    ops = _ops()
    q, k = r16(B * Nq, H * d, seed=131, scale=1.5), r16(B * Nk, H * d, seed=132, scale=1.5)
    scale = d ** -0.5
    ld = (Nk + 7) // 8 * 8
    P = ops.attention_probs(q.cuda(), k.cuda(), B, H, Nq, Nk, d, scale, ld)
    qq = q.float().reshape(B, Nq, H, d).permute(0, 2, 1, 3)
    kk = k.float().reshape(B, Nk, H, d).permute(0, 2, 1, 3)
    ref = torch.softmax(qq @ kk.transpose(-1, -2) * scale, -1).reshape(B * H, Nq, Nk)
    assert P.shape == (B * H, Nq, ld)
    got = P.float().cpu()
    assert float(got[:, :, Nk:].abs().max()) == 0.0 if ld > Nk else True
    assert rel_l2(got[:, :, :Nk], ref) < TOL
    assert float((got[:, :, :Nk].sum(-1) - 1).abs().max()) < 2e-3
    # same probabilities as the two-kernel path it replaces (fp32 scores -> row softmax), up to the fp16 rounding of P
    S = ops.attention_scores(q.cuda(), k.cuda(), B, H, Nq, Nk, d, scale, ld)
    P2 = ops.softmax_rows(S.reshape(B * H * Nq, ld), Nk, ld).reshape(B * H, Nq, ld)
    assert float((P.float() - P2.float()).abs().max()) < 2e-3


@pytest.mark.parametrize("B,H,Nq,Nk,d", [(2, 8, 1024, 1024, 40), (2, 4, 256, 77, 64), (1, 8, 300, 256, 80), (2, 8, 64, 64, 160)])
def test_attention_probs_from_split_operands(B, H, Nq, Nk, d):
    """icd_attention_probs_split (UNet split bit ICD_SPLIT_QK, layers whose probabilities a controller keeps): q and k with the error
    carry of their fp16 rounding; scores = qh.kh + 2^-14 (ql.kh + qh.kl).  Against the fp64 softmax of the values q and k hold (hi +
    carry): the split maps sit at the fp16 rounding of P itself, the fp16-operand maps visibly above it (sharp rows: the score error is
    the map's relative error)."""
    ops = _ops()
    g = torch.Generator().manual_seed(170)
    q32, k32 = torch.randn(B * Nq, H * d, generator=g) * 2.5, torch.randn(B * Nk, H * d, generator=g) * 2.5
    qh, qc = ops.carry_encode(q32)
    kh, kc = ops.carry_encode(k32)
    scale = d ** -0.5
    P = ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk, d, scale, q_carry=qc.cuda(), k_carry=kc.cuda())
    P0 = ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk, d, scale)
    qq = ops.carry_decode(qh, qc).double().reshape(B, Nq, H, d).permute(0, 2, 1, 3)
    kk = ops.carry_decode(kh, kc).double().reshape(B, Nk, H, d).permute(0, 2, 1, 3)
    ref = torch.softmax(qq @ kk.transpose(-1, -2) * scale, -1).reshape(B * H, Nq, Nk)
    e_s, e_p = rel_l2(P[:, :, :Nk], ref), rel_l2(P0[:, :, :Nk], ref)
    print(f"[probs split B={B} H={H} {Nq}x{Nk} d={d}] split {e_s:.3e}, fp16 operands {e_p:.3e}")
    assert e_s < 2.6e-4 and e_s < 0.6 * e_p
    # one-sided carries (cross-attention whose K comes without one) are accepted
    P1 = ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk, d, scale, q_carry=qc.cuda())
    assert rel_l2(P1[:, :, :Nk], ref) < e_p


@pytest.mark.parametrize("split", [False, True])
def test_probability_kernel_epilogue_does_the_controllers_work_bit_for_bit(split):
    """icd_attention_probs_ex (icd_probs_epilogue): the store accumulation, the self-attention replacement and the cross-attention edit of the
    reference's controllers in the probability kernel's epilogue - one pass over P - against the separate passes they replace: the same
    kernels' P, icd_p2p_cross_edit, torch's fp16 in-place add, a row copy.  Same bits, with and without split q / k, with a CFG-doubled
    batch (the unconditional half untouched)."""
    ops = _ops()
    P, H, d, Nk = 3, 8, 40, 77                         # prompts (base + 2 edits), heads
    B, first = 2 * P, P                                # CFG-doubled: [uncond x 3 | cond x 3]
    g = torch.Generator().manual_seed(180)
    for Nq in (1024, 200):
        q32, k32 = torch.randn(B * Nq, H * d, generator=g) * 2, torch.randn(B * Nk, H * d, generator=g) * 2
        qh, qc = ops.carry_encode(q32); kh, kc = ops.carry_encode(k32)
        kw = dict(q_carry=qc.cuda(), k_carry=kc.cuda()) if split else {}
        scale, ld = d ** -0.5, 80
        T = Nk
        A = torch.rand(P - 1, T, T, generator=g) * (torch.rand(P - 1, T, T, generator=g) < 0.05)
        D = torch.rand(P - 1, T, generator=g)
        At, Dp = ops.p2p_pack_operator(A.cuda(), D.cuda())
        acc0 = (torch.rand(P * H, Nq, ld, generator=g) * 0.1).half()
        # reference: separate passes
        ref = ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk, d, scale, ld, **kw)
        cond = ref[first * H:]
        ops.p2p_cross_edit(cond[:, :, :Nk], P, At, Dp)
        acc_ref = acc0.cuda().clone()
        acc_ref += cond
        # fused
        acc = acc0.cuda().clone()
        got = ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk, d, scale, ld, acc=acc, edit=(At, Dp), first_cond_sample=first, **kw)
        assert torch.equal(got, ref) and torch.equal(acc, acc_ref), (Nq, split)
        assert not torch.equal(got[first * H + H:], ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk, d, scale, ld, **kw)[first * H + H:])
    # self-attention (long rows, two sweeps over K staged through LDS): replacement by the base prompt's rows + accumulation; ragged
    # query / key counts (waves without queries still stage and synchronise; the last stage is partial) and every head-dim class
    for Nq, Nk2, dd in ((256, 256, 40), (200, 333, 80), (96, 104, 160), (1024, 1024, 80), (130, 100, 128)):
        q32, k32 = torch.randn(B * Nq, H * dd, generator=g) * 2, torch.randn(B * Nk2, H * dd, generator=g) * 2
        qh, qc = ops.carry_encode(q32); kh, kc = ops.carry_encode(k32)
        kw = dict(q_carry=qc.cuda(), k_carry=kc.cuda()) if split else {}
        ld2 = (Nk2 + 7) // 8 * 8
        ref = ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk2, dd, dd ** -0.5, **kw)
        base = ref[first * H:(first + 1) * H]
        for e in range(1, P):
            ref[(first + e) * H:(first + e + 1) * H] = base
        acc0 = (torch.rand(P * H, Nq, ld2, generator=g) * 0.1).half().cuda()          # padded rows, as the executor's P
        acc_ref = acc0.clone(); acc_ref += ref[first * H:]                             # (pad columns of P are zero)
        acc = acc0.clone()
        got = ops.attention_probs(qh.cuda(), kh.cuda(), B, H, Nq, Nk2, dd, dd ** -0.5, acc=acc[:, :, :Nk2], self_from_base=True,
                                  first_cond_sample=first, **kw)
        assert torch.equal(got, ref) and torch.equal(acc, acc_ref), (Nq, Nk2, dd, split)


def test_flash_attention_ring_is_bit_identical_to_a_fully_fenced_build():
    """The flash kernels hand tiles over with a counted s_waitcnt vmcnt(N) + a bare s_barrier (attention.hip wait_landed); a miscounted
    ring would read LDS rows that have not landed.  tools/attn_ring_check.py builds attention.hip again with a full fence at every tile
    (-DICD_ATTN_DEBUG_SYNC) and runs ragged / causal / wide-head cases of every head dim on both libraries: same bits."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_ring_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "cases bit-identical" in r.stdout and "DIFF" not in r.stdout


def test_ping_pong_tiles_race_screen():
    """tools/pp_stress.py: every ping-pong tile (dense and conv) launched repeatedly beside a second stream's launches and compared bit for bit
    with the lockstep tile - a fragment read before its DMA landed or a slot restaged under a reader would show up as rare mismatches."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "pp_stress.py"), "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RACE SCREEN CLEAN" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
