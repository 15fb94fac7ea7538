"""AutoencoderKL on the HIP kernels vs the CPU oracle (oracle/vae_ref.py) - SURVEY.md section 8f rank 1.

Reduced widths for the oracle-checked cases (the oracle finishes in seconds), full SD-VAE width for size-independent
properties.  Tolerance: rel-L2 < 1e-3 (round 6; 3e-3 before the residual stream carried its rounding error) against the fp32 oracle on identical fp16-representable weights (the decoder chains
~30 fp16-storage operators; measured values are printed).
"""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _env():
    from invertible_cd_amd import synthetic, vae
    from oracle import vae_ref
    return synthetic, vae, vae_ref


def _setup(widths, seed):
    synthetic, vae, vae_ref = _env()
    cfg = vae.SD_VAE.scaled(widths)
    sd = {k: v.half().float() for k, v in synthetic.synthetic_vae_state_dict(cfg, seed=seed).items()}
    ocfg = dict(vae_ref.SD_VAE, block_out_channels=tuple(widths))
    return cfg, sd, ocfg, vae, vae_ref


@pytest.mark.parametrize("widths,fused", [((32, 64, 128, 128), True), ((32, 64, 128, 128), False), ((64, 64, 192, 192), None)])
def test_decode_matches_oracle(widths, fused):
    cfg, sd, ocfg, vae, vae_ref = _setup(widths, seed=3)
    m = vae.AutoencoderKL(cfg, sd, fused_attention=fused)
    z = (torch.randn(2, 4, 16, 24, generator=torch.Generator().manual_seed(5)) * 1.5).half().float()
    got = m.decode(z.cuda())["sample"].float().cpu()
    ref = vae_ref.decode(sd, ocfg, z)
    assert got.shape == ref.shape == (2, 3, 128, 192)
    e = rel_l2(got, ref)
    print(f"[vae decode {widths} fused={fused}] rel-L2 = {e:.3e}")
    assert e < 1e-3                                       # round 6: error-carried residual stream (1.3e-3 on the plain fp16 stream)
    # round 5: the decoder's three Upsample2D convs run in phase form (four 2 x 2 convs on the input grid); the 3 x 3 form stays selectable
    m.upsample_phases = False
    old = m.decode(z.cuda())["sample"].float().cpu()
    e_old = rel_l2(old, ref)
    print(f"[vae decode {widths} fused={fused}] 3x3 upsampler form rel-L2 = {e_old:.3e}, distance between the forms {rel_l2(got, old):.3e}")
    assert e_old < 1e-3 and abs(e - e_old) < 3e-4 and not torch.equal(got, old)


@pytest.mark.parametrize("widths,fused", [((32, 64, 128, 128), True), ((32, 64, 128, 128), False)])
def test_encode_mean_matches_oracle(widths, fused):
    cfg, sd, ocfg, vae, vae_ref = _setup(widths, seed=4)
    m = vae.AutoencoderKL(cfg, sd, fused_attention=fused)
    x = (torch.rand(2, 3, 128, 192, generator=torch.Generator().manual_seed(6)) * 2 - 1).half().float()
    out = m.encode(x.cuda())
    got = out["latent_dist"].mean.float().cpu()
    assert out.latent_dist.mode() is out["latent_dist"].mean
    ref = vae_ref.encode_mean(sd, ocfg, x)
    assert got.shape == ref.shape == (2, 4, 16, 24)
    e = rel_l2(got, ref)
    print(f"[vae encode {widths} fused={fused}] rel-L2 = {e:.3e}")
    assert e < 1e-3


def test_odd_latent_sizes_round_trip_shapes_and_parity():
    """Latents with odd height / width (72 x 104 pixel images): every level of the decoder and encoder sees odd maps."""
    cfg, sd, ocfg, vae, vae_ref = _setup((32, 64, 64, 64), seed=9)
    m = vae.AutoencoderKL(cfg, sd)
    z = (torch.randn(2, 4, 9, 13, generator=torch.Generator().manual_seed(10)) * 1.5).half().float()
    got = m.decode(z.cuda())["sample"].float().cpu()
    ref = vae_ref.decode(sd, ocfg, z)
    assert got.shape == ref.shape == (2, 3, 72, 104) and rel_l2(got, ref) < 1e-3
    x = ref.clamp(-1, 1).half().float()
    e = rel_l2(m.encode(x.cuda())["latent_dist"].mean.float().cpu(), vae_ref.encode_mean(sd, ocfg, x))
    assert e < 1e-3


def test_chunking_and_interface():
    """max_chunk splits the batch without changing results; return_dict=False / fp32 output / error behaviour."""
    cfg, sd, ocfg, vae, vae_ref = _setup((32, 64, 64, 64), seed=7)
    z = torch.randn(5, 4, 8, 8, generator=torch.Generator().manual_seed(8)).cuda()
    a = vae.AutoencoderKL(cfg, sd, max_chunk=8).decode(z)["sample"]
    b = vae.AutoencoderKL(cfg, sd, max_chunk=2).decode(z, return_dict=False)[0]
    assert torch.equal(a, b) and a.dtype == torch.float16
    m = vae.AutoencoderKL(cfg, sd).to(torch.float32)
    assert m.dtype == torch.float32 and m.decode(z)["sample"].dtype == torch.float32
    assert m.config.scaling_factor == 0.18215
    with pytest.raises(ValueError):
        m.decode(torch.zeros(1, 3, 8, 8))
    with pytest.raises(ValueError):
        m.encode(torch.zeros(1, 3, 20, 16))
    smp = m.encode(torch.zeros(1, 3, 16, 16))["latent_dist"].sample(torch.Generator().manual_seed(0))
    assert smp.shape == (1, 4, 2, 2) and torch.isfinite(smp).all()
    bad = dict(sd); bad.pop("decoder.conv_out.bias")
    with pytest.raises(KeyError):
        vae.AutoencoderKL(cfg, bad)


def test_full_width_properties():
    """SD-VAE width, 64x64 latents (512x512 images): finite, deterministic, batch-independent; encode(decode) shape."""
    synthetic, vae, vae_ref = _env()
    sd = synthetic.synthetic_vae_state_dict(vae.SD_VAE, seed=0, device="cuda", dtype=torch.float16)
    m = vae.AutoencoderKL(vae.SD_VAE, sd, max_chunk=4)
    del sd
    z = torch.randn(3, 4, 64, 64, generator=torch.Generator().manual_seed(9)).cuda() * 2.0
    a, b = m.decode(z)["sample"], m.decode(z)["sample"]
    assert a.shape == (3, 3, 512, 512) and torch.isfinite(a).all() and torch.equal(a, b)
    one = m.decode(z[1:2])["sample"]
    assert rel_l2(one.float(), a[1:2].float()) < 2e-3                # other samples in the batch do not matter
    lat = m.encode(a.clamp(-1, 1))["latent_dist"].mean
    assert lat.shape == (3, 4, 64, 64) and torch.isfinite(lat).all()


def test_encode_moments_and_sample_match_oracle():
    """`latent_dist` carries all 8 moment channels (utils/generation_sdxl.py:273-276 draws `.sample(generator)`):
    mean / logvar vs the oracle's quant_conv(encoder(x)), std = exp(0.5 * clamp(logvar, -30, 20)), and the sample drawn
    from a CPU generator like diffusers' randn_tensor."""
    cfg, sd, ocfg, vae, vae_ref = _setup((32, 64, 128, 128), seed=11)
    m = vae.AutoencoderKL(cfg, sd, dtype=torch.float32)
    x = (torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(8)) * 2 - 1).half().float()
    d = m.encode(x.cuda()).latent_dist
    mom = vae_ref.encode_moments(sd, ocfg, x)
    e_mean, e_lv = rel_l2(d.mean.cpu(), mom[:, :4]), rel_l2(d.logvar.cpu(), mom[:, 4:].clamp(-30, 20))
    print(f"[vae moments] mean rel-L2 = {e_mean:.3e}  logvar rel-L2 = {e_lv:.3e}")
    assert e_mean < 1e-3 and e_lv < 1e-3
    assert torch.allclose(d.std, torch.exp(0.5 * d.logvar)) and torch.allclose(d.var, torch.exp(d.logvar))
    s1 = d.sample(torch.Generator().manual_seed(21))
    noise = torch.randn(d.mean.shape, generator=torch.Generator().manual_seed(21), dtype=d.mean.dtype)
    assert torch.equal(s1.cpu(), (d.mean + d.std * noise.cuda()).cpu())
    assert s1.shape == (2, 4, 8, 12) and not torch.equal(s1, d.mean)
    s2 = d.sample(torch.Generator(device="cuda").manual_seed(21))            # a device generator draws on the device
    assert s2.is_cuda and torch.isfinite(s2).all() and not torch.equal(s1, s2)


# ------------------------------------------------------------------------------ fp32-fidelity path (`vae.to(torch.float32)`)
def test_fp32_fidelity_decode_and_encode_match_oracle_tightly():
    """`.to(torch.float32)` (utils/generation_sdxl.py:465-466, diffusers force_upcast) selects fp32 storage + split3 fp16
    operands: the result must sit far below the fp16 path's ~1.3e-3, on fp32 weights that are NOT fp16-representable."""
    synthetic, vae, vae_ref = _env()
    cfg = vae.SDXL_VAE.scaled((32, 64, 128, 128))
    sd = synthetic.synthetic_vae_state_dict(cfg, seed=13)                           # genuine fp32 weights
    ocfg = dict(vae_ref.SDXL_VAE, block_out_channels=(32, 64, 128, 128))
    m = vae.AutoencoderKL(cfg, sd).to(torch.float32)
    z = torch.randn(2, 4, 16, 24, generator=torch.Generator().manual_seed(5)) * 1.5
    got = m.decode(z.cuda())["sample"]
    assert got.dtype == torch.float32
    e_dec = rel_l2(got.cpu(), vae_ref.decode(sd, ocfg, z))
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(6)) * 2 - 1
    d = m.encode(x.cuda()).latent_dist
    mom = vae_ref.encode_moments(sd, ocfg, x)
    e_mean, e_lv = rel_l2(d.mean.cpu(), mom[:, :4]), rel_l2(d.logvar.cpu(), mom[:, 4:].clamp(-30, 20))
    print(f"[vae fp32-fidelity] decode rel-L2 = {e_dec:.3e}  encode mean = {e_mean:.3e}  logvar = {e_lv:.3e}")
    assert e_dec < 3e-4 and e_mean < 3e-4 and e_lv < 3e-4
    m.to(torch.float16)                                                              # and back: the fp16 path is untouched
    e16 = rel_l2(m.decode(z.cuda())["sample"].float().cpu(), vae_ref.decode(sd, ocfg, z))
    assert 3e-4 < e16 < 4e-3


def test_fp32_fidelity_survives_activations_beyond_the_fp16_range():
    """Real SDXL-VAE weights drive the decoder's residual stream past 65504 (why the reference upcasts it).  Emulated by
    scaling the residual-branch output convs: the fp16 path overflows to inf / NaN, the fp32-fidelity path stays finite and
    within 1e-3 of the fp32 oracle."""
    synthetic, vae, vae_ref = _env()
    cfg = vae.SDXL_VAE.scaled((32, 64, 64, 64))
    sd = synthetic.synthetic_vae_state_dict(cfg, seed=14)
    for k in ("decoder.conv_in.weight", "decoder.conv_in.bias", "post_quant_conv.bias"):
        sd[k] = sd[k] * 100.0                                   # (the weights themselves stay far inside the fp16 range)
    ocfg = dict(vae_ref.SDXL_VAE, block_out_channels=(32, 64, 64, 64))
    z = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(7)) * 2.0e3     # residual stream ~1e5 - 1e6 from the first conv on
    ref = vae_ref.decode(sd, ocfg, z)
    m = vae.AutoencoderKL(cfg, sd)
    bad = m.decode(z.cuda())["sample"]
    assert not torch.isfinite(bad).all()                                             # the fp16-storage path cannot hold these
    good = m.to(torch.float32).decode(z.cuda())["sample"]
    e = rel_l2(good.cpu(), ref)
    print(f"[vae fp32-fidelity, overflowing activations] rel-L2 = {e:.3e}  max|ref| = {float(ref.abs().max()):.3e}")
    assert torch.isfinite(good).all() and e < 1e-3


def test_fp32_groupnorm_keeps_its_digits_under_a_large_dc_offset_and_the_guard_sees_nan():
    """icd_groupnorm_f32_split on groups whose |mean| is ~1e3 standard deviations (real SDXL-VAE activations do this - the reason the
    fp32 path exists): raw fp32 sums of x and x^2 would lose ~6 of 7 digits of the variance to cancellation; the kernel sums
    (x - pivot) with the group's first element as pivot, like torch's Welford stays accurate.  Also: a NaN activation must trip the
    finiteness guard in front of icd_split_cast (icd_absmax maps NaN to +inf; fmaxf alone drops it)."""
    import pytest
    from invertible_cd_amd import ops
    B, HW, Cc = 2, 4096, 128
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * HW, Cc, generator=g, dtype=torch.float64)
    x = x + 1000.0 * (1.0 + torch.arange(Cc, dtype=torch.float64) // 4 % 7)[None, :]      # per-group DC offsets of 1e3..7e3 sigma
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    ref = torch.nn.functional.group_norm(x.reshape(B, HW, Cc).permute(0, 2, 1), 32, gamma.double(), beta.double(), 1e-6).permute(0, 2, 1).reshape(B * HW, Cc)
    x32 = x.float().cuda()
    out = ops.groupnorm_f32_split(x32, B, HW, gamma.cuda(), beta.cuda(), 1e-6, False).float()
    got = (out[:, :Cc] + out[:, Cc:2 * Cc]).cpu()                                           # hi + lo
    ref32 = torch.nn.functional.group_norm(x.float().reshape(B, HW, Cc).permute(0, 2, 1).double(), 32, gamma.double(), beta.double(), 1e-6)
    ref32 = ref32.permute(0, 2, 1).reshape(B * HW, Cc)                                      # what fp32-rounded inputs allow at best
    e, e_in = rel_l2(got, ref), rel_l2(ref32, ref)
    print(f"[fp32 GroupNorm, mean/std ~ 1e3] rel-L2 vs fp64 = {e:.3e} (input rounding alone: {e_in:.3e})")
    assert e < 3e-4 and e < 4 * e_in + 1e-5             # x itself carries 2^-24 * 1e3 sigma ~ 6e-5 sigma of input rounding
    bad = x32.clone()
    bad[5, 7] = float("nan")
    assert ops.absmax(bad) == float("inf")
    with pytest.raises(FloatingPointError):
        ops.split_cast_guarded(bad)
