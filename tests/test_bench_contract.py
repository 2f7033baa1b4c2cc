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


R06 = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r06_v*_bench.json")) if "config4" not in os.path.basename(p))


@pytest.mark.parametrize("path", R06, ids=[os.path.basename(p) for p in R06])
def test_round6_default_line_carries_the_other_workloads_and_the_repeats(path):
    """The driver's default command (`python bench.py --gpus 1 --steps K --warmup W`) times configs[1] as `value` -- and, in the same run, repeats that K-step region
    (`value_runs`: median / min / max) and times BASELINE configs[3] and one GPU's share of configs[4] live for >= 0.5 s each (`other_workloads`), so that those numbers
    are driver-observed and not only builder-run lines under profiles/."""
    d = json.loads(open(path).read().strip().splitlines()[-1])
    assert d["config"]["workload"].endswith("(BASELINE.json configs[1])") and d["n_gpus"] == 1
    vr = d["value_runs"]
    assert vr["runs"] == len(vr["values"]) >= 6 and vr["values"][0] == d["value"] and vr["min"] <= vr["median"] <= vr["max"] and vr["unit"] == "env-steps/s"
    assert vr["min"] == min(vr["values"]) and vr["max"] == max(vr["values"])
    ow = d["other_workloads"]
    assert set(ow) == {"mountaincar", "config4"}
    for name, idx in (("mountaincar", 3), ("config4", 4)):
        w = ow[name]
        assert "failed" not in w, w
        assert "BASELINE.json configs[%d]" % idx in w["config"]["workload"] and "model" not in w["config"]
        assert w["unit"] == "env-steps/s" and w["timed_seconds"] >= 0.45 and w["steps"] >= 10
        env_steps = w["steps"] * w["config"]["num_envs_per_gpu"] * w["config"]["num_steps"]
        assert abs(w["value"] - env_steps / w["timed_seconds"]) <= 1e-6 * w["value"]
        assert abs(w["ms_per_step"] - 1e3 * w["timed_seconds"] / w["steps"]) <= 1e-9 * w["ms_per_step"] + 1e-9
        r = w["roofline"]
        assert r["bound"] == "mfma" and r["peak"] == 2500.0 and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert ow["config4"]["dtype"] == "bf16" and ow["mountaincar"]["config"]["num_envs_per_gpu"] == 8192 and ow["config4"]["config"]["num_envs_per_gpu"] == 2048
    # the profile set the line quotes is tied to the sources it ran on, and names the commit it was collected at
    p = d["profiles"]
    assert p["tied"] is True and p["git_head_at_collection"] and d["roofline"]["traffic"] > 0 and d["roofline"]["rocprof"]["avg_us"] > 0
    assert abs(d["roofline"]["rocprof"]["avg_us"] - 1e3 * d["roofline"]["avg_launch_ms"]) <= 0.15 * d["roofline"]["rocprof"]["avg_us"]   # trace and HIP events agree


def test_there_is_a_round6_line():
    assert len(R06) >= 1
