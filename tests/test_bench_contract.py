"""bench.py's reporting contract on the CPU (no GPU): the algorithmic-byte accounting is SURVEY.md 8(d)'s, and the
committed bench lines under profiles/ carry every field the contract names."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_follow_the_survey_worked_example():
    import bench
    N, D, px = 200000, 3500000, 1024 * 1024
    sb = bench.stage_bytes(N, D, px)
    fwd = sb["preprocess"] + sb["binning"] + sb["render_fwd"]
    bwd = sb["render_bwd"] + sb["gaussian_bwd"]
    # fwd = 64 + 84 + 140 + 21 = 309 MB, bwd = 21 + 154 + 126 + 121 = 422 MB at N = 200k, D = 3.5 M
    assert abs(fwd / 1e6 - 309) < 1.5 and abs(bwd / 1e6 - 422) < 1.5
    assert sb["render_bwd"] == px * 20 + D * 80 and sb["binning"] == D * 24
    assert bench.HBM_PEAK_GBS == 8000.0
    # the workloads are BASELINE.json's configs 2-5
    assert bench.WORKLOADS["config3"][:4] == (200000, 1024, 1024, 3) and bench.WORKLOADS["config3"][5]
    assert bench.WORKLOADS["config2"][:4] == (50000, 512, 512, 3) and not bench.WORKLOADS["config2"][5]
    assert bench.WORKLOADS["config5"][:4] == (500000, 2048, 2048, 3)


def test_committed_bench_lines_carry_the_contract_fields():
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_v*_config3_bench.json")))
    assert lines
    d = json.load(open(lines[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "200k Gaussians" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["n_gpus"]) < 1e-2  # fps x seconds per step = ranks
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
