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
