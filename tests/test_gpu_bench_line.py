"""bench.py on a partitioned run (here: ONE rank that still communicates, OCN_FORCE_DISTRIBUTED=1, through a one-rank RCCL world):
the conservative sequence is measured first and reported beside the default one; a default sequence that never returns ends in the
conservative line with exit code 0, not in a hang (bench.py: conservative_leg, emit_fallback)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, timeout=400):
    env = dict(os.environ, OCN_FORCE_DISTRIBUTED="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    env.update(extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-strict"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
    return p, json.loads(lines[0])


def test_partitioned_bench_line_carries_the_conservative_measurement():
    p, d = _run({})
    assert p.returncode == 0, p.stderr[-4000:]
    rccl = d["config"]["rccl"]
    assert rccl["fast_path"] == "ok" and "fallback" not in d
    c = rccl["conservative"]
    assert c["driver"] == "python" and c["all_gather"] == "collective" and c["correct_on_load_model"] is False and c["finite"]
    assert d["driver"] == "c" and d["config"]["finite"]
    # same synthetic initial state, same number of steps: the two sequences agree to rounding (their pressure solves differ in order)
    for x, y in zip(c["sum_of_squares"], d["config"]["state_checksum"]["sum_of_squares"]):
        assert abs(x - y) <= 1e-9 * abs(y)
    assert c["ms_per_step"] > 0 and d["ms_per_step"] > 0 and rccl["legs_relative_difference"] <= 1e-9


def test_default_sequence_that_never_returns_ends_in_the_conservative_line():
    p, d = _run({"OCN_BENCH_INJECT_FAST_HANG": "1", "OCN_BENCH_FAST_DEADLINE_S": "3"})
    assert p.returncode == 0, p.stderr[-4000:]
    assert d["fallback"] is True and d["driver"] == "python" and d["config"]["all_gather"] == "collective"
    assert d["config"]["rccl"]["fast_path"].startswith("did not finish within")
    assert d["ms_per_step"] == d["config"]["rccl"]["conservative"]["ms_per_step"] and d["value"] > 0
    assert "reporting the conservative measurement" in p.stderr


def test_default_sequence_with_a_different_state_ends_in_the_conservative_line():
    """The two legs start from the same state and take the same steps; a default sequence whose global sums differ from the conservative
    one's beyond OCN_BENCH_LEGS_RTOL (here: a tolerance nothing can meet) is not reported."""
    p, d = _run({"OCN_BENCH_LEGS_RTOL": "-1"})
    assert p.returncode == 0, p.stderr[-4000:]
    assert d["fallback"] is True and d["driver"] == "python"
    assert d["config"]["rccl"]["fast_path"].startswith("finished with a different state")
