"""The JSON line bench.py prints, checked on the lines committed from the GPU box (profiles/r02_v*_bench_default_*.json -- the driver's own command
for the three workloads): every key the round contract names is there with the right type, the roofline objects carry what they must, and the
numbers are self-consistent (value = env-steps of the timed steps / their wall time; frac = achieved / peak).  No GPU: the lines are data."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[23]_v*_bench_default_*.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_keeps_the_contract(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)]:
        assert isinstance(d[key], typ), key
    assert d["metric"] == "env-steps/sec (rollout+update)" and d["unit"] == "env-steps/s" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    c = d["config"]
    env_steps = d["steps"] * c["num_envs_per_gpu"] * c["num_steps"] * d["n_gpus"]
    assert abs(d["value"] - env_steps / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and r["avg_launch_ms"] > 0 and r["launches"] > 0
    assert "traffic" in r
    if "/r03_" in path.replace(os.sep, "/"):
        # round 3: the roofline is the one of the instruction stream the kernel issues -- it cannot pass 1; the transport fields are top-level
        assert 0.0 < r["frac"] < 1.0
        if "cartpole" in os.path.basename(path) or "mountaincar" in os.path.basename(path):
            assert r["peak"] == 2500.0 and r["achieved_fp32_equiv"] > 0 and r["fp32_mfma_peak"] == 157.3 and "limiter" in r
            assert abs(r["achieved"] * 1e12 * r["avg_launch_ms"] * 1e-3 - r["executed_flops_per_launch"]) <= 1e-6 * r["executed_flops_per_launch"]
        for key in ("transport", "transport_requested", "transport_fallback_reason", "comm_ranks", "transport_ab"):
            assert key in d, key
        assert d["comm_ranks"] == d["n_gpus"] and (d["transport"] == "none") == (d["n_gpus"] == 1)
        bar = d["gae_roofline"]["bar"]
        assert bar["target_frac"] == 0.40 and "size_met_from_envs" in bar and "frac_at_config1" in bar and "floor_us" in bar
    g = d["gae_roofline"]
    assert g["bound"] == "hbm" and g["unit"] == "GB/s" and g["peak"] == 8000.0
    assert abs(g["achieved"] - g["bytes_per_launch"] / (g["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * g["achieved"]
    if "cartpole" in os.path.basename(path):   # the headline workload carries the CPU baseline and the PMC traffic
        b = d["cpu_baseline"]
        assert b["kind"] in ("reference", "port") and b["value"] > 0 and b["cores"] >= 1 and b["unit"] == "env-steps/s" and b["sample"]
        if "/r03_" in path.replace(os.sep, "/") and b["kind"] == "reference":
            assert "oversubscribed" in b["note"]
        assert r["traffic"] > 0 and d["dtype"] == "f32"


def test_there_are_committed_lines():
    assert len(LINES) >= 3
