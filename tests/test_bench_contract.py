"""The bench line the driver parses, checked on the committed evidence (profiles/r05_bench_default.json is the stdout of
`python bench.py --steps 20 --warmup 5` on an MI355X): every key of the contract is there, with the types and relations the contract states."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    lines = [l for l in open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints exactly ONE line on stdout"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "images/sec" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f16"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = images of one step / time of one step
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    assert "traffic_kernels_sha" in r and "kernels_sha" in r          # a stale PMC summary is visible in the line itself
    # achieved = algorithmic work per launch / average launch duration
    assert abs(r["achieved"] - r["algorithmic_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) / r["achieved"] < 2e-2
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str) and c["unit"] == d["unit"]
    assert 5 <= c["seconds"] <= 120                          # a bounded sample
    s = d["sdxl"]                                             # BASELINE config 4's per-GPU share rides in the same line
    assert s["value"] > 0 and s["roofline"]["kernel"] == "gemm_dense" and "workload" in s["config"]
    # round 5: both traffic figures are measured in the run itself, and the edit legs carry a roofline object too
    assert r["traffic_source"].startswith("live") and s["roofline"]["traffic_source"].startswith("live")
    assert r["traffic_kernels_sha"] == r["kernels_sha"] and s["roofline"]["traffic_kernels_sha"] == s["roofline"]["kernels_sha"]
    for key in ("edit", "sdxl_edit"):
        e = d[key]["roofline"]
        assert e["bound"] == "mfma" and 0 < e["frac"] < 1 and abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-3
        assert abs(e["achieved"] - e["algorithmic_per_launch"] / (e["avg_launch_us"] * 1e-6) / 1e12) / e["achieved"] < 2e-2
    # algorithmic flops (the operator as the reference computes it) and the flops issued to the matrix cores are both on the line where
    # they differ: the phase-form upsampler issues 4/9 of the 3 x 3 conv on the upsampled map
    assert r["kernel"] == "gemm_conv" and 0 < r["executed_frac"] < r["frac"]


def test_every_baseline_config_rides_on_the_default_line():
    """configs[2] (edit), configs[3] (sdxl) and configs[4] (sdxl_edit) are objects of the default line, each with its own workload
    description and the value = batch / time relation; the sequential pass, the plain-fp16-stream rate, the RCCL object of the run
    (a world of one executes the collective too) and the per-rank spread are reported."""
    d = _line()
    for key, batch in (("edit", 8), ("sdxl", 8), ("sdxl_edit", 16)):
        o = d[key]
        assert o["unit"] == "images/sec" and o["value"] > 0 and "workload" in o["config"] and o["config"]["per_gpu_batch"] == batch, key
        assert abs(o["value"] - o["config"]["global_batch"] / (o["ms_per_step"] * 1e-3)) / o["value"] < 1e-3, key
        assert o["ms_per_step_per_rank"]["ranks"] == d["n_gpus"] and o["ms_per_step_per_rank"]["min"] <= o["ms_per_step_per_rank"]["max"], key
    assert d["edit"]["attention_store_tensors_per_pass"] > 0                 # the controller really was in the loop
    assert d["edit"]["config"]["attention_store_accumulate"] == "probability kernel epilogue"      # ... with its work on P inside the kernel
    # round 4: `value` is the pass with `in_flight_batches` independent batches in flight and no events; the sequential pass with
    # events around the dominant family (the roofline leg) rides beside it, and so does the plain-fp16-stream rate
    assert d["config"]["in_flight_batches"] >= 1
    # round 5: the precision policy a user gets - plain generation on the error carry, the edit legs at the accurate level
    assert d["config"]["precision"] == {"policy": "auto", "residual_stream": "fp16 + bf8 error carry"} == d["sdxl"]["config"]["precision"]
    for key in ("edit", "sdxl_edit"):
        pr = d[key]["config"]["precision"]
        assert pr["policy"] == "auto" and pr["residual_stream"].endswith("split consumers") and pr["split_mask"] in (511, 1023), key
    assert 0 < d["accurate_level"]["value"] < d["value"] and d["value_one_batch_at_a_time"] == d["one_batch_at_a_time"]["value"]
    if d["config"]["in_flight_batches"] > 1:
        seq = d["one_batch_at_a_time"]
        assert 0 < seq["value"] <= d["value"] * 1.02 and abs(seq["value"] - d["config"]["global_batch"] / (seq["ms_per_step"] * 1e-3)) / seq["value"] < 1e-3
    else:
        ev = d["event_overhead"]
        assert abs(ev["frac"] - (ev["ms_per_step_with_events"] / ev["ms_per_step_without_events"] - 1)) < 2e-3
    f16 = d["fp16_stream"]
    assert f16["value"] > 0 and abs(f16["value"] - d["config"]["global_batch"] / (f16["ms_per_step"] * 1e-3)) / f16["value"] < 1e-3
    assert d["rccl"]["backend"] == "nccl" and d["rccl"]["rccl_world_size"] == d["n_gpus"] and d["rccl"]["all_gather_executed"] is True
    assert d["config"]["lora_fused"] is True and d["sdxl"]["config"]["lora_fused"] is True
    r = d["ms_per_step_per_rank"]
    assert r["ranks"] == 1 and r["min"] == r["max"]


def test_committed_traffic_summaries_were_measured_on_the_current_kernel_sources():
    """bench.py fills roofline.traffic from profiles/*hbm_traffic*.json; the latest ones must carry the sha of csrc/ as it is now
    (else the next bench line would show traffic_kernels_sha != kernels_sha)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    for arch, batch, fam in (("sd15", 32, "gemm_conv"), ("sdxl", 8, "gemm_dense")):
        traffic, source, sha = bench.hbm_traffic(arch, batch, fam)
        assert traffic and source.startswith("profiles/r"), (arch, source)
        if sha != bench.csrc_sha():              # reported, not red: kernels move during a round, the evidence follows at its end
            pytest.skip(f"{source} was measured on kernel sources {sha}, csrc/ is now {bench.csrc_sha()}: re-run tools/profile_round.sh")
