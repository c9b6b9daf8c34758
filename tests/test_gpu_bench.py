"""bench.py's multi-rank code path on the GPU box: rendezvous, the one broadcast of the Gaussian state, frame sharding,
barriers, max over ranks -- started as a CHILD process (never an exec of the test process) with two ranks on the one GPU
(GSPLAT_BENCH_REHEARSAL=1: gloo instead of RCCL; the line is marked as a rehearsal and is not a reported number)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv):
    env = dict(os.environ, GSPLAT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_started_by_bench_itself_weak_scaling():
    d = _bench("--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-pmc", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert "REHEARSAL" in d["config"]["backend"] and d["config"]["total_frames"] == 6
    cs = d["config"]["state_checksums"]
    assert len(cs) == 2 and cs[0] == cs[1] and cs[0] != 0  # both ranks hold the broadcast state, bit for bit
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 2) < 1e-2
    assert d["config"]["broadcast_s"] > 0


def test_two_ranks_strong_scaling_over_a_sequence_of_fixed_length():
    """BASELINE config 4's statement in miniature: a sequence of 7 frames over 2 ranks (4 + 3: the tail), every frame
    rendered once inside the timed region, value = 7 / the slower rank's time."""
    d = _bench("--gpus", "2", "--workload", "tiny", "--total-frames", "7", "--warmup", "1", "--no-pmc", "--no-cpu-baseline",
               "--no-upstream-leg")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 4
    assert d["config"]["total_frames"] == 7 and d["config"]["frames_per_rank"] == [3, 4]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 2) < 1e-2
    cs = d["config"]["state_checksums"]
    assert cs[0] == cs[1]


def test_one_rank_on_the_rccl_backend_runs_the_whole_multi_rank_code_path():
    """The `nccl` (= RCCL) process group itself, which a one-GPU box cannot run with two ranks on one card: ONE rank,
    GSPLAT_BENCH_FORCE_DIST=1, in a fresh child process under the launcher's environment variables -- librccl is loaded,
    init_process_group("nccl", device_id=...) runs, the packed Gaussian state goes through dist.broadcast on the GPU, the
    checksum through all_gather, the timing through barrier and all_reduce(MAX).  (The 8-GPU node is the driver's: this is
    what makes its first run boring.)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, GSPLAT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("GSPLAT_BENCH_REHEARSAL", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "tiny", "--steps", "3",
                        "--warmup", "1", "--no-pmc", "--no-cpu-baseline", "--no-upstream-leg"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "nccl (RCCL)" and d["n_gpus"] == 1 and d["config"]["ranks"] == 1
    cs = d["config"]["state_checksums"]
    assert len(cs) == 1 and cs[0] != 0 and d["config"]["broadcast_s"] > 0 and d["value"] > 0

