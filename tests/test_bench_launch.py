"""`python bench.py --gpus N` without a launcher: the script starts its own ranks as CHILD processes -- the parent
never touches torch or HIP -- relays rank 0's ONE JSON line and leaves with the launcher's exit code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines(text):
    return [json.loads(x) for x in text.splitlines() if x.lstrip().startswith("{") and '"metric"' in x]


def test_self_launch_without_a_gpu_still_prints_one_line_and_fails():
    """no device in this container: every rank leaves at once ("needs an MI355X"); the parent must come back non-zero with ONE parsable
    line that says so -- and must not have imported torch itself (FEMBRAIN_BENCH_PARENT_TRACE lists the parent's modules at exit)"""
    try:
        from fembrain_amd import lib
        if lib.lib().fb_device_count() > 0:
            pytest.skip("a HIP device is visible: the GPU variant of this test runs instead")
    except Exception:
        pass
    env = dict(os.environ, FEMBRAIN_BENCH_PARENT_TRACE="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    recs = _lines(out.stdout)
    assert len(recs) == 1 and recs[0]["value"] is None and recs[0]["n_gpus"] == 2 and "error" in recs[0], out.stdout[-2000:]
    assert "torch.distributed.run" in recs[0]["launched"] and "--nproc-per-node 2" in recs[0]["launched"]
    assert "parent modules: torch=False" in out.stderr, out.stderr[-2000:]


@pytest.mark.gpu
def test_bench_gpus_2_runs_from_a_plain_python_command_on_the_one_gpu_box(gpu):
    """exactly the command form the driver uses for N = 1, with --gpus 2: both ranks on device 0 through the host-staged communicator
    (FEMBRAIN_BENCH_LOCAL_COMM=1; RCCL refuses two ranks on one device).  One JSON line, rc 0, the sharded self-check passed, the
    collective-library leg and the 8M-tet `scaling_8m` block present.  Timings of this mode are meaningless; the control flow is the point."""
    env = dict(os.environ, FEMBRAIN_BENCH_LOCAL_COMM="1", FEMBRAIN_BENCH_8M_STEPS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    recs = _lines(out.stdout)
    assert len(recs) == 1, out.stdout[-3000:]
    d = recs[0]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d.get("error") is None
    assert d["config"]["sharded_self_check"]["ok"]
    assert d["config"]["communicator"]["ranks"] == 2 and d["config"]["rccl_ranks"] == 0      # (the rehearsal's transport is not RCCL, and says so)
    leg = d["config"]["collective_library_leg"]
    assert leg["collective"]["cg_iterations"] == leg["as_timed"]["cg_iterations"] > 0
    assert "collective" in d["config"]["exchange_trials_ms_per_step"]
    s8 = d["scaling_8m"]
    assert s8["n_gpus"] == 2 and s8["steps_per_s"] > 0 and len(s8["us_per_cg_iteration_per_rank"]) == 2 and s8["n1_steps_per_s_committed"] > 0
