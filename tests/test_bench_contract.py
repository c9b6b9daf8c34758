"""bench.py's reporting contract on the CPU (no GPU): the algorithmic-byte accounting is SURVEY.md 8(d)'s, and the
committed bench lines under profiles/ carry every field the contract names."""
import glob
import json
import os

import pytest

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
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_v*_config3_bench.json")))
    assert lines
    d = json.load(open(lines[-1]))
    if os.path.basename(lines[-1]).startswith("r02"):
        # round 2: the reference's own binning timed in the same run, measured traffic and the VALU ceiling
        u = d["upstream_rect"]
        assert u["tile_rect"] == 0 and u["num_rendered"] > d["config"]["num_rendered"] and u["value"] > 0
        assert d["config"]["tile_rect"] == 1 and d["config"]["ranks"] == d["n_gpus"]
        r = d["roofline"]
        assert r["traffic"] and r["traffic_over_algorithmic"] == round(r["traffic"] / r["algorithmic_bytes"], 3)
        assert "rocprofv3" in r["traffic_source"]
        for k in ("render_fwd", "render_bwd"):
            assert 0 < r["valu"][k]["issue_slot_util"] < 1 and r["valu"][k]["insts_per_launch"] > 0
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


def test_gpus_flag_starts_the_ranks_itself_and_refuses_a_mismatch(monkeypatch):
    """`python bench.py --gpus N` with no launcher starts N ranks as a CHILD `torch.distributed.run` on the same file
    (round 1 parsed the flag and ran one rank); under a launcher whose WORLD_SIZE differs from --gpus it exits non-zero
    instead of printing a mislabeled line.  No GPU is touched by either path."""
    import subprocess
    import sys

    import bench
    import pytest
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert e.value.code == 7  # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # launched under a launcher with another world size: refuse
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert e.value.code == 2


def test_total_frames_is_config_4_as_stated_300_frames_partitioned_over_the_ranks(monkeypatch):
    """BASELINE config 4: a 300-frame sequence sharded over 8 GPUs (render.py:51-62 is the loop being partitioned).
    `--total-frames 300`: rank r times frames r, r + 8, ... exactly once (38 or 37 frames: the tail SURVEY.md 8(e) names),
    every frame of the sequence is timed by exactly one rank, and the flag reaches the ranks bench.py starts itself."""
    import subprocess

    import bench
    import pytest
    timed = []
    for r in range(8):
        frames, K = bench.frame_plan(r, 8, steps=100, warmup=5, total_frames=300)
        assert K == (38 if r < 4 else 37) and len(frames) == 8 + 5 + K
        assert frames[-K:] == list(range(r, 300, 8))
        assert set(frames[:13]) <= set(frames[-K:])  # the untimed frames are this rank's own, re-used
        timed += frames[-K:]
    assert sorted(timed) == list(range(300))
    # weak scaling (the default): every rank times --steps frames of its own
    frames, K = bench.frame_plan(3, 8, steps=20, warmup=5)
    assert K == 20 and frames == [3 + 8 * i for i in range(33)]
    assert bench.frame_plan(5, 8, steps=20, warmup=5, total_frames=4) == ([], 0)  # more ranks than frames: refused in main()
    a = bench.parse_args(["--gpus", "8", "--workload", "config4", "--total-frames", "300"])
    assert a.total_frames == 300 and a.workload == "config4"
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"] = cmd
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "8", "--workload", "config4", "--total-frames", "300"])
    assert seen["cmd"][-6:] == ["--gpus", "8", "--workload", "config4", "--total-frames", "300"]


def test_ranks_get_disjoint_core_slices():
    """bench.pin_rank_to_cores: rank r of the node takes the r-th contiguous slice of the cores the process may run on
    (set before anything touches the GPU).  Checked in child processes: the test process's own affinity stays as it is."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, os, sys; sys.path.insert(0, %r); import bench; "
            "print(json.dumps([bench.pin_rank_to_cores(int(sys.argv[1]), int(sys.argv[2])), sorted(os.sched_getaffinity(0))]))" % root)
    ncores = len(os.sched_getaffinity(0))
    if ncores < 4:
        pytest.skip("fewer than four cores: pin_rank_to_cores leaves the scheduler alone")
    seen = []
    for rk in range(2):
        r = subprocess.run([sys.executable, "-c", code, str(rk), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        label, cores = json.loads(r.stdout.strip().splitlines()[-1])
        assert label is not None and label.endswith("(%d cores)" % (ncores // 2))
        seen.append(set(cores))
    assert not (seen[0] & seen[1]) and len(seen[0]) == len(seen[1]) == ncores // 2
    # one rank per node, or switched off: nothing is pinned
    r = subprocess.run([sys.executable, "-c", code, "0", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       cwd=root, env=dict(os.environ, GSPLAT_BENCH_NO_AFFINITY="1"))
    assert json.loads(r.stdout.strip().splitlines()[-1]) == [None, sorted(os.sched_getaffinity(0))]
