"""The bench line the driver parses, checked on the committed evidence (profiles/r02_bench_default.json is the stdout of
`python bench.py` on an MI355X): every key of the contract is there, with the types and relations the contract states."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    lines = [l for l in open(os.path.join(ROOT, "profiles", "r02_bench_default.json")).read().splitlines() if l.strip()]
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
