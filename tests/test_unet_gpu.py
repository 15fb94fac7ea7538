"""UNet forward parity: native MI355X executor (through the C ABI) vs the CPU fp32 oracle on identical seeded weights
and inputs.

Tolerance: the product stores activations in fp16 (fp32 accumulate); the oracle is fp32 end to end.  Weights and inputs
are rounded to fp16 on both sides so the comparison measures the kernels, not the weight quantisation.  The north star's
bar is 1e-3 rel-L2 "of the reference".  Bars (round 4):
  * absolute: EVERY case, reduced width and full width, on the benchmarked tiles too, < 1e-3 against the fp32 oracle in the default
    (and benchmarked) mode - the residual stream with its error carry (measured 0.70 - 0.85e-3; the plain fp16 stream of rounds 1 - 3
    sat at 1.0 - 1.2e-3 and is kept as option residual = 0, asserted < 2e-3 where a test switches to it);
  * relative: every case also runs the oracle's own graph in fp16 with stock torch ops on the GPU (the "fp16-torch noise floor":
    how far an fp16 diffusers-style pipeline - what the reference's configs 2 - 5 run - sits from fp32 on these weights, 1.9 - 2.6e-3)
    and the product must be <= 1.5 x that floor (it is at 0.3 - 0.4 x).
`pytest -s` output of the GPU suite is kept as profiles/r04_parity.txt.
(Round 4 dropped three full-width single-forward cases that larger ones subsume - SD1.5 B=2 at 32x32 / 64x64 and SDXL B=1 at 32x32 are
covered by the carried-stream case at 32x32, the B=8 64x64 benchmarked-tile case, the full-width loop test and the SDXL B=2 128x128 case -
to keep the whole GPU suite, most of which is CPU oracle time, well inside the driver's 20-minute step on a slow host.)
"""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _mods():
    from invertible_cd_amd import synthetic, unet, unet_config
    from oracle import unet_ref
    return synthetic, unet, unet_config, unet_ref


def _oracle_cfg(unet_ref, cfg):
    base = unet_ref.SD15 if cfg.addition_time_embed_dim == 0 else unet_ref.SDXL
    o = dict(base)
    o["block_out_channels"] = cfg.block_out_channels
    o["cross_dim"] = cfg.cross_dim
    o["num_heads"] = cfg.num_heads
    if cfg.addition_time_embed_dim:
        o["add_in_dim"] = cfg.add_in_dim
    return o


def _run_case(cfg, B, H, W, t, seed, tol, with_cond=True, n_ctx=77, f32_io=False, variants=None, check_plans=None, variant_tol=None,
              cache_tag=None):
    """Product vs fp32 oracle vs the fp16-torch floor.  `variants`: {name: {option: value}} - further runs of the SAME handle under
    other per-handle options (UNet2DConditionModel.set_option), each compared with the oracle too; `check_plans(name, plans)` gets
    the planner records of every run (name None = defaults) so a case can assert the code path it claims to pin.  Returns
    {name: (error, eps)}."""
    from invertible_cd_amd import _lib
    synthetic, unet, unet_config, unet_ref = _mods()
    sd = {k: v.half().float() for k, v in synthetic.synthetic_state_dict(cfg, seed=seed).items()}
    inp = synthetic.synthetic_inputs(cfg, B, H, W, seed=seed, n_ctx=n_ctx)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    cond = None
    if with_cond:
        cond = torch.randn(B, cfg.time_cond_proj_dim, generator=torch.Generator().manual_seed(seed + 5)).half().float()
    added = None
    if cfg.addition_time_embed_dim:
        added = {"text_embeds": inp["text_embeds"].half().float(), "time_ids": inp["time_ids"]}
    oracle = lambda: unet_ref.unet_forward(sd, _oracle_cfg(unet_ref, cfg), lat, t, ctx, timestep_cond=cond, added_cond=added)
    if cache_tag is None:
        ref = oracle()
    else:                                            # full-width cases: a committed fixture keyed by everything the oracle's result depends on
        import oracle_cache
        ref = oracle_cache.lookup(cache_tag, sd, [cfg.name, lat, t, ctx, cond, added], oracle)
    model = unet.UNet2DConditionModel(cfg, sd)
    x = lat.cuda() if f32_io else lat.half().cuda()
    kw = dict(encoder_hidden_states=ctx.cuda(), timestep_cond=None if cond is None else cond.cuda(),
              added_cond_kwargs=None if added is None else {k: v.cuda() for k, v in added.items()})

    def run(name, opts):
        for k, v in opts.items():
            model.set_option(k, v)
        _lib.profile_enable(True)
        try:
            eps = model(x, torch.tensor(t), **kw).sample
            torch.cuda.synchronize()
            plans = _lib.profile_plans()
        finally:
            _lib.profile_enable(False)
        if check_plans is not None:
            check_plans(name, plans)
        return eps

    eps = run(None, {})
    assert eps.shape == lat.shape and eps.dtype == x.dtype
    assert torch.isfinite(eps).all()
    err = rel_l2(eps, ref)
    # fp16-torch noise floor: the oracle's graph, fp16 weights / activations, stock torch kernels on the GPU
    sd16 = {k: v.cuda().half() for k, v in sd.items()}
    flo = unet_ref.unet_forward(sd16, _oracle_cfg(unet_ref, cfg), lat, t, ctx, timestep_cond=cond, added_cond=added)
    floor = rel_l2(flo.float().cpu(), ref)
    del sd16, flo
    print(f"[{cfg.name} B={B} {H}x{W} t={t}] rel-L2(eps) = {err:.3e}  fp16-torch floor = {floor:.3e}  ratio = {err / floor:.2f}  "
          f"|ref| rms = {ref.pow(2).mean().sqrt():.3f}")
    assert err < tol
    assert err <= 1.5 * floor + 1e-4, f"product error {err:.3e} is more than 1.5x the fp16-torch floor {floor:.3e}"
    # second call on the same handle (workspace reuse) must be bit-identical
    eps2 = model(x, torch.tensor(t), return_dict=False, **kw)[0]
    assert torch.equal(eps, eps2)
    results = {None: (err, eps)}
    for name, opts in (variants or {}).items():
        e2 = run(name, opts)
        ev = rel_l2(e2, ref)
        print(f"[{cfg.name} B={B} {H}x{W} t={t}] variant {name} {opts}: rel-L2(eps) = {ev:.3e}  ratio to the floor = {ev / floor:.2f}  "
              f"distance to the default run = {rel_l2(e2, eps):.3e}")
        assert torch.isfinite(e2).all() and ev < (variant_tol or tol) and ev <= 1.5 * floor + 1e-4
        results[name] = (ev, e2)
    return results


def test_unet_tiny_sd15_topology():
    _, _, uc, _ = _mods()
    cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
    _run_case(cfg, B=2, H=32, W=32, t=779, seed=1, tol=1e-3)


def test_unet_tiny_sd15_ragged_and_f32_io():
    _, _, uc, _ = _mods()
    cfg = uc.SD15.scaled((32, 64, 64, 128), cross_dim=40, heads=(4, 4, 4, 4))
    # ... and the accurate level on the same ragged shapes: split GEMMs through the general loaders (32 / 64 channels), the phase-form
    # upsampler over [h | lo | h] at odd widths, GroupNorm's carry read in the one-launch kernel
    r = _run_case(cfg, B=3, H=16, W=24, t=19, seed=2, tol=1e-3, with_cond=False, n_ctx=13, f32_io=True,
                  variants={"accurate": {"residual": 3, "split_mask": 1023}, "phases_off": {"upsample_phases": 0}})
    assert r["accurate"][0] < 0.6e-3 and r["phases_off"][0] < 0.6e-3


def test_unet_tiny_sdxl_topology():
    _, _, uc, _ = _mods()
    cfg = uc.SDXL.scaled((64, 128, 256), cross_dim=128, heads=(2, 4, 8))
    _run_case(cfg, B=2, H=32, W=32, t=999, seed=3, tol=1e-3)




@pytest.mark.parametrize("which", ["sd15_tiny", "sdxl_tiny", "sd15_full_32"])
def test_carried_residual_stream_meets_the_north_star_tolerance(which):
    """UNet option residual (icd_unet_set_option ICD_UNET_OPT_RESIDUAL_MODE).  With fp16 activation storage the residual stream's own
    roundings (one per x <- x + f(x), 100-300 adds deep) put a single forward at 1.0 - 1.2e-3 from an fp32 evaluation; the north star's
    bar is 1e-3.  Mode 2 (the fast level of the precision policy) keeps what each rounding lost in one bf8 byte per element (icd_gemm_desc.resid_carry /
    out_carry), mode 1 (round 3) accumulated the chain in an fp32 twin.  Asserted: the default is < 1e-3 against the fp32 oracle and
    clearly better than the plain fp16 stream of the SAME handle; the carry is as good as the fp32 twin (within 5 %); switching back
    restores the default result bit for bit.  (A plain single evaluation runs at the 'fast' level of the precision policy - the error carry;
    the 'accurate' level, residual = 3 with every ICD_SPLIT_* bit, is what inversion loops and controller passes run.)"""
    _, _, uc, _ = _mods()
    if which == "sd15_tiny":
        cfg, B, H = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64), 2, 32
    elif which == "sdxl_tiny":
        cfg, B, H = uc.SDXL.scaled((64, 128, 256), cross_dim=128, heads=(2, 4, 8)), 2, 32
    else:
        cfg, B, H = uc.SD15, 2, 32
    r = _run_case(cfg, B=B, H=H, W=H, t=779, seed=31, tol=1e-3,
                  variants={"fp16": {"residual": 0}, "twin": {"residual": 1}, "split": {"residual": 3}, "accurate": {"residual": 3, "split_mask": 1023},
                            "back": {"residual": 2, "split_mask": 447}}, variant_tol=2e-3)
    e_c, e16, e32, e_s, e_a = r[None][0], r["fp16"][0], r["twin"][0], r["split"][0], r["accurate"][0]
    print(f"[{which}] fp16 residual stream {e16:.3e} -> error carry {e_c:.3e} (fp32 twin {e32:.3e}) -> carry + split consumers {e_s:.3e} "
          f"-> + upsampler convs (the 'accurate' level of the precision policy) {e_a:.3e}")
    assert e_c < 1.0e-3 and e_c < 0.85 * e16 and e_c < 1.05 * e32
    # round 5: GroupNorm reads fp16 + carry, the shortcut / proj_out / sampler GEMMs take hi + lo - the simulated budget
    # (tests/error_budget_sim.py) predicts 0.58 - 0.65 x the carry-only error; asserted: the accurate level is < 0.5e-3
    assert e_s < 0.70 * e_c and e_a < 1.02 * e_s and e_a < 0.5e-3
    assert torch.equal(r["back"][1], r[None][1])






@pytest.mark.slow
def test_unet_full_sd15_b8_64x64_on_the_benchmarked_tiles_and_layernorm_statistics_ab():
    """The code path bench.py times, against the oracle: full-width SD1.5, B = 8 at 64x64 latents (6.4 TFLOP of CPU oracle).  At this
    size the planner puts the UNet on the 256-wide tiles of gemm_big.hip (convs and Linears) and the first GEMM behind every
    LayerNorm takes the statistics from its own MFMA operand fragments (ICD_GEMM_LN_COMPUTE in the main loop) - both asserted from
    the executor's planner records, not assumed.  The same handle then runs with option ln_inline_stats = 0 (a separate
    icd_layernorm_stats pass over the residual stream): no launch may report in-loop statistics, the two results must DIFFER (they
    are two evaluation orders - a zero distance would mean the A/B never switched anything, which is what this test's predecessor
    measured at B = 2, 32x32) and both must sit equally far from the fp32 oracle."""
    _, _, uc, _ = _mods()
    seen = {}

    def check(name, plans):
        if name in ("accurate", "fast"):          # (the split-operand launches of the accurate level: checked against the oracle only)
            return
        dense = [p for p in plans if p["family"] == "gemm_dense"]
        conv = [p for p in plans if p["family"] == "gemm_conv"]
        big_flops = sum(2.0 * p["M"] * p["N"] * p["K"] for p in dense + conv if p["big"])
        all_flops = sum(2.0 * p["M"] * p["N"] * p["K"] for p in dense + conv)
        inline = [p for p in dense if p["ln_inline"]]
        seen[name] = (big_flops / all_flops, len(inline))
        print(f"[plans {name}] {len(dense)} dense + {len(conv)} conv launches, {100 * big_flops / all_flops:.1f} % of the GEMM flops on gemm_big "
              f"tiles {sorted({p['tile'] for p in dense + conv if p['big']})}, {len(inline)} launches with in-loop LayerNorm statistics")
        assert big_flops / all_flops > 0.8, "the benchmarked tiles did not run"
        if name is None:
            # planner at this size (icd_gemm_plan): to_qk at 64^2 / 32^2 / 16^2, attn2.to_q at 64^2 / 32^2, the C = 320 GEGLU projections
            assert len(inline) >= 25 and all(p["big"] and p["ksplit"] == 1 for p in inline)
            assert {(256, 320), (256, 256), (128, 320)} <= {p["tile"] for p in dense + conv if p["big"]}
        else:
            assert len(inline) == 0

    r = _run_case(uc.SD15, B=8, H=64, W=64, t=779, seed=8, tol=1e-3,
                  variants={"ln_pass": {"ln_inline_stats": 0}, "accurate": {"ln_inline_stats": 1, "residual": 3, "split_mask": 1023},
                            "fast": {"residual": 2}}, check_plans=check, cache_tag="sd15_full_64x64_b8_eps")
    # (round 6: the default run follows the 'auto' policy, whose probe picks the level for these weights; the two levels are pinned explicitly)
    print(f"[sd15 B=8 64x64] fast level {r['fast'][0]:.3e} -> accurate level {r['accurate'][0]:.3e}; default (auto policy) {r[None][0]:.3e}")
    assert r["accurate"][0] < 0.6e-3 and r["accurate"][0] < 0.7 * r["fast"][0]   # what the inversion / edit loops run, on the benchmarked tiles
    d = rel_l2(r["ln_pass"][1], r[None][1])
    print(f"[LN statistics in the consuming GEMM vs a pass over the stream] vs oracle {r[None][0]:.3e} / {r['ln_pass'][0]:.3e}; "
          f"the two differ by {d:.3e}")
    assert 0.0 < d < 2e-3 and abs(r[None][0] - r["ln_pass"][0]) < 2e-4


def test_context_projection_cache_is_exact_and_notices_a_changed_context():
    """Steps 2..n of a sampling loop pass the SAME context tensor: the executor then skips the two context-only projections
    (cross-attention K / V^T of every layer, icd_unet_io.kv_cache).  The cache must (a) never change a bit of the output, (b) refill
    when the context tensor is updated in place (version counter) or replaced by another tensor - also one that the allocator
    places at the address of a freed one."""
    from invertible_cd_amd import _lib
    synthetic, unet, uc, _ = _mods()
    cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
    model = unet.UNet2DConditionModel(cfg, synthetic.synthetic_state_dict(cfg, seed=21))
    inp = synthetic.synthetic_inputs(cfg, 2, 16, 16, seed=21)
    x = inp["latents"].half().cuda()
    ctx = inp["context"].half().cuda()
    run = lambda c: model(x, 519, encoder_hidden_states=c).sample.clone()

    def dense_launches(c):
        _lib.profile_enable(True)
        try:
            out = run(c)
            torch.cuda.synchronize()
            return out, _lib.profile_read()["gemm_dense"]["launches"]
        finally:
            _lib.profile_enable(False)

    model.kv_cache_enabled = False
    run(ctx)                                          # (the first plain call of a model also probes the precision levels: two evaluations)
    ref, n_off = dense_launches(ctx)
    model.kv_cache_enabled = True
    a, n_fill = dense_launches(ctx)                   # fills the cache
    b, n_hit = dense_launches(ctx)                    # reads it
    assert torch.equal(a, ref) and torch.equal(b, ref)
    assert n_fill == n_off and n_hit == n_off - 2     # exactly the two context projections are skipped
    ctx2 = ctx * 1.5
    ref2 = run(ctx2)                                  # another tensor -> refill
    model.kv_cache_enabled = False
    assert torch.equal(ref2, run(ctx2)) and not torch.equal(ref2, ref)
    model.kv_cache_enabled = True
    run(ctx2)
    ctx2.mul_(0.5)                                    # in-place update of the cached tensor -> the version counter moves -> refill
    got = run(ctx2)
    model.kv_cache_enabled = False
    assert torch.equal(got, run(ctx2))
    model.kv_cache_enabled = True
    for i in range(3):                                # fresh tensors of one shape: the allocator hands out recycled addresses
        c = (ctx * (1.0 + 0.25 * i)).contiguous()
        got = run(c)
        model.kv_cache_enabled = False
        want = run(c)
        model.kv_cache_enabled = True
        assert torch.equal(got, want)
        del c


def test_forward_under_inference_mode_and_after_a_failing_hook():
    """(a) torch.inference_mode(): context tensors created there have no version counter - the K / V cache treats them as a miss instead
    of raising, results equal the no_grad ones bit for bit.  (b) a controller that raises aborts the forward: the exception surfaces and
    the context cache is dropped (a half-written cache must not be trusted by the next forward)."""
    synthetic, unet, uc, _ = _mods()
    cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
    model = unet.UNet2DConditionModel(cfg, synthetic.synthetic_state_dict(cfg, seed=23))
    inp = synthetic.synthetic_inputs(cfg, 2, 16, 16, seed=23)
    x, ctx = inp["latents"].half().cuda(), inp["context"].half().cuda()
    ref = model(x, 519, encoder_hidden_states=ctx).sample.clone()
    with torch.inference_mode():
        xi, ci = x.clone(), ctx.clone()
        a = model(xi, 519, encoder_hidden_states=ci).sample.clone()
        b = model(xi, 519, encoder_hidden_states=ci).sample.clone()
    assert torch.equal(a, ref) and torch.equal(b, ref)

    class Boom:
        num_att_layers = 0

        def __call__(self, *a, **k):
            raise ValueError("controller failed")
    model(x, 519, encoder_hidden_states=ctx)
    assert model._kv is not None
    model.attn_controller = Boom()
    with pytest.raises(Exception):
        model(x, 519, encoder_hidden_states=ctx)
    assert model._kv is None
    model.attn_controller = None
    assert torch.equal(model(x, 519, encoder_hidden_states=ctx).sample, ref)


def test_executor_replicas_run_two_batches_in_flight_with_sequential_results():
    """UNet2DConditionModel.replica(): a second native handle over the SAME packed weights (own arena, caches, plugin slot, the options
    of the original).  Two host threads on two HIP streams drive one replica each through different batches at the same time (what
    bench.py's InFlight does): every output equals the sequential single-handle result bit for bit - the library keeps no process-wide
    execution state, and concurrent launches on the idle CUs change the clock, not the arithmetic."""
    import threading
    synthetic, unet, uc, _ = _mods()
    cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
    model = unet.UNet2DConditionModel(cfg, synthetic.synthetic_state_dict(cfg, seed=41))
    model.set_option("xattn_fusion", 0)
    rep = model.replica()
    assert rep._packed is model._packed and rep._h.value != model._h.value and rep._options == model._options
    ins = [synthetic.synthetic_inputs(cfg, 3, 32, 32, seed=50 + i) for i in range(4)]
    xs = [i["latents"].half().cuda() for i in ins]
    cs = [i["context"].half().cuda() for i in ins]
    want = [model(x, 519, encoder_hidden_states=c).sample.clone() for x, c in zip(xs, cs)]
    assert torch.equal(rep(xs[0], 519, encoder_hidden_states=cs[0]).sample, want[0])
    got, errs = [None] * 4, []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def work(net, stream, idx):
        try:
            with torch.cuda.stream(stream):
                for _ in range(3):                                   # several rounds: arenas and caches are reused while the other runs
                    for k in idx:
                        got[k] = net(xs[k], 519, encoder_hidden_states=cs[k]).sample.clone()
                stream.synchronize()
        except BaseException as e:                                    # noqa: BLE001
            errs.append(e)
    torch.cuda.synchronize()
    th = [threading.Thread(target=work, args=(model, streams[0], (0, 2))), threading.Thread(target=work, args=(rep, streams[1], (1, 3)))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_precision_policy_selects_the_level_the_samplers_ask_for():
    """UNet2DConditionModel.precision = "auto" (default): the error carry for a plain evaluation, the accurate level (residual 3, every
    ICD_SPLIT_* bit) inside `with unet.editing():` and whenever an attention controller is attached; each level keeps its own arena; an
    explicit set_option('residual') switches the policy off; a replica inherits it; the forced levels do what they say."""
    from invertible_cd_amd import _lib, p2p
    synthetic, unet, uc, _ = _mods()
    cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
    sd = {k: v.half().float() for k, v in synthetic.synthetic_state_dict(cfg, seed=1).items()}
    inp = synthetic.synthetic_inputs(cfg, 2, 16, 16, seed=1)
    m = unet.UNet2DConditionModel(cfg, sd)
    x, kw = inp["latents"].half().cuda(), dict(encoder_hidden_states=inp["context"].half().cuda())
    fast, acc = (_lib.ICD_RESIDUAL_CARRY, _lib.ICD_SPLIT_DEFAULT), (_lib.ICD_RESIDUAL_SPLIT, _lib.ICD_SPLIT_ACCURATE)
    assert m.precision == "auto"
    e_fast = m(x, 500, **kw).sample
    assert m._applied == fast
    with m.editing():
        e_acc = m(x, 500, **kw).sample
        assert m._applied == acc
    assert not torch.equal(e_fast, e_acc) and rel_l2(e_fast, e_acc) < 2e-3
    assert torch.equal(m(x, 500, **kw).sample, e_fast) and m._applied == fast and len(m._ws_pool) == 1      # back; ONE arena serves both levels
    # (round 6) the first plain evaluation probed both levels on its own data: these weights keep the fast level for plain generation
    assert m._auto_plain == "fast" and 0 < m._auto_gap <= m.AUTO_ESCALATE_GAP
    class _M:                                                  # register_attention_control wants a model with .unet
        pass
    holder = _M()
    holder.unet = m
    store = p2p.AttentionStore()
    p2p.register_attention_control(holder, store)
    m(x, 500, **kw)
    assert m._applied == acc and store.cur_step == 1 and len(store.attention_store["down_cross"]) > 0      # a controller is an editing pipeline
    p2p.register_attention_control(holder, None)
    r = m.replica()
    assert r.precision == "auto" and torch.equal(r(x, 500, **kw).sample, e_fast)
    assert torch.equal(m.set_precision("accurate")(x, 500, **kw).sample, e_acc)
    m.set_option("residual", 0)
    assert m.precision is None
    e16 = m(x, 500, **kw).sample
    with m.editing():
        assert torch.equal(m(x, 500, **kw).sample, e16)                                                    # the policy is off
    with pytest.raises(ValueError):
        m.set_precision("exact")


def test_gemm_tune_option_switches_the_tile_family_not_the_result():
    """ICD_UNET_OPT_GEMM_TUNE (round 6): ICD_GEMM_TUNE_NO_PP sends every launch of the handle to the lockstep tiles of rounds 1 - 5 - the
    same-process A/B switch of bench.py --gemm-tune / tools/unet_ab.py.  Full-width SD1.5, B = 4 at 64 x 64: the planner records show the
    ping-pong kernels' shapes moving to other tiles or staying put, and eps moves by what two valid tile plans of one evaluation differ by
    (a tile with another column count changes which launches take their LayerNorm statistics from their own main loop, a split-K plan
    regroups a sum: fp16-level noise, 0.8e-3 here - the distance test_full_size_sd15_properties allows between two plans of a sample)."""
    from invertible_cd_amd import _lib
    synthetic, unet, uc, _ = _mods()
    cfg = uc.SD15
    m = unet.UNet2DConditionModel(cfg, synthetic.synthetic_state_dict(cfg, seed=2, device="cuda", dtype=torch.float16)).set_precision("fast")
    inp = synthetic.synthetic_inputs(cfg, 4, 64, 64, seed=2, device="cuda")
    kw = dict(encoder_hidden_states=inp["context"].half(), timestep_cond=torch.randn(4, 512, device="cuda").half())
    x = inp["latents"].half()

    def run():
        _lib.profile_enable(True)
        try:
            eps = m(x, 519, **kw).sample
            torch.cuda.synchronize()
            return eps, [p for p in _lib.profile_plans() if p["family"] in ("gemm_dense", "gemm_conv")]
        finally:
            _lib.profile_enable(False)

    run()                                                     # (fills the context-projection cache: two launches the later passes skip)
    e_pp, p_pp = run()
    m.set_option("gemm_tune", _lib.ICD_GEMM_TUNE_NO_PP)
    e_ls, p_ls = run()
    m.set_option("gemm_tune", 0)
    e_back, _ = run()
    assert torch.equal(e_back, e_pp) and len(p_pp) == len(p_ls)
    moved = sum(1 for a, b in zip(p_pp, p_ls) if a["tile"] != b["tile"])
    d = rel_l2(e_ls, e_pp)
    print(f"[gemm_tune] {len(p_pp)} GEMM launches, {moved} on another tile shape without the ping-pong family; eps distance {d:.3e}")
    assert torch.isfinite(e_ls).all() and 0 < d < 2e-3 and moved > 0
    with pytest.raises(RuntimeError):
        m.set_option("gemm_tune", 1)                          # only planner bits are accepted


def test_eight_pixel_latents_take_the_3x3_upsampler_at_the_one_pixel_level():
    """latents of 2^(levels - 1) pixels put the lowest level at 1 x 1; the phase-form upsampler (out_remap_w >= 2) cannot map a
    one-pixel-wide level, the executor takes the 3 x 3 form there - every precision level still matches the oracle."""
    synthetic, unet, uc, unet_ref = _mods()
    cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
    sd = {k: v.half().float() for k, v in synthetic.synthetic_state_dict(cfg, seed=43).items()}
    inp = synthetic.synthetic_inputs(cfg, 2, 8, 8, seed=43)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    cond = torch.randn(2, 512, generator=torch.Generator().manual_seed(47)).half().float()
    ref = unet_ref.unet_forward(sd, _oracle_cfg(unet_ref, cfg), lat, 519, ctx, timestep_cond=cond)
    m = unet.UNet2DConditionModel(cfg, sd)
    for level in ("fast", "accurate"):
        eps = m.set_precision(level)(lat.half().cuda(), 519, encoder_hidden_states=ctx.cuda(), timestep_cond=cond.cuda()).sample
        e = rel_l2(eps, ref)
        print(f"[8 x 8 latents, {level}] rel-L2 = {e:.3e}")
        assert e < 1.5e-3


def test_large_magnitude_channels_keep_the_stream_finite_at_every_precision_level():
    """Real checkpoints have residual-stream channels far outside the unit range the synthetic weights produce.  Here a handful of conv_in
    output channels carry a DC offset of +- 3000 .. 24000 (the stream then holds values beyond 2^13, where half an fp16 ulp times the carry's
    2^14 scale leaves the bf8 range, and LayerNorm rows sit hundreds of sigma from zero): every precision level must stay finite, the carried
    levels must not be worse than the plain fp16 stream on the same weights (the carry saturates instead of overflowing, gemm_common.h
    carry_of8), and all of them track the fp32 oracle as well as fp16 storage of such values allows."""
    synthetic, unet, uc, unet_ref = _mods()
    cfg = uc.SD15.scaled((64, 128, 256, 256), cross_dim=64)
    sd = {k: v.half().float() for k, v in synthetic.synthetic_state_dict(cfg, seed=41).items()}
    off = torch.zeros(64)
    off[[3, 17, 40, 41, 63]] = torch.tensor([3000.0, -9000.0, 16000.0, -24000.0, 12000.0])
    sd["conv_in.bias"] = (sd["conv_in.bias"] + off).half().float()
    inp = synthetic.synthetic_inputs(cfg, 2, 32, 32, seed=41)
    lat, ctx = inp["latents"].half().float(), inp["context"].half().float()
    cond = torch.randn(2, 512, generator=torch.Generator().manual_seed(46)).half().float()
    ref = unet_ref.unet_forward(sd, _oracle_cfg(unet_ref, cfg), lat, 519, ctx, timestep_cond=cond)
    assert torch.isfinite(ref).all()
    m = unet.UNet2DConditionModel(cfg, sd)
    errs = {}
    for name, opts in (("fp16", {"residual": 0}), ("carry", {"residual": 2}), ("accurate", {"residual": 3, "split_mask": 1023})):
        for k, v in opts.items():
            m.set_option(k, v)
        eps = m(lat.half().cuda(), torch.tensor(519), encoder_hidden_states=ctx.cuda(), timestep_cond=cond.cuda()).sample
        assert torch.isfinite(eps).all(), name
        errs[name] = rel_l2(eps, ref)
    # the data-aware 'auto' policy (round 6): on THESE weights the probe finds the fast level more than AUTO_ESCALATE_GAP away from the accurate
    # one and plain generation runs at the accurate level - inside the north star's 1e-3 where the fast level is not
    m2 = unet.UNet2DConditionModel(cfg, sd)
    assert m2.precision == "auto"
    eps = m2(lat.half().cuda(), torch.tensor(519), encoder_hidden_states=ctx.cuda(), timestep_cond=cond.cuda()).sample
    errs["auto"] = rel_l2(eps, ref)
    print(f"[large-magnitude channels] auto policy: probe gap {m2._auto_gap:.3e} -> {m2._auto_plain}")
    assert m2._auto_plain == "accurate" and m2._auto_gap > m2.AUTO_ESCALATE_GAP
    assert errs["auto"] < 1e-3 and errs["auto"] == pytest.approx(errs["accurate"], rel=1e-6)
    print("[large-magnitude channels] rel-L2 vs the fp32 oracle: " + ", ".join(f"{k} {v:.3e}" for k, v in errs.items()))
    assert errs["carry"] <= 1.05 * errs["fp16"] and errs["accurate"] <= 1.05 * errs["carry"]
    assert errs["fp16"] < 2e-2                                  # (fp16 storage of 2^14-sized values: ulp 16 on a stream whose signal is O(1))
