"""The example scripts run (they are documentation that must not rot)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args,expect", [("blob_to_fem.py", ["", "0.12"], "surface:"), ("cut_and_resync.py", ["16"], "next step:")])
def test_example_runs(gpu, script, args, expect):
    cmd = [sys.executable, os.path.join(ROOT, "examples", script)] + [a for a in args if a]
    if script == "blob_to_fem.py":
        cmd = [sys.executable, os.path.join(ROOT, "examples", script), os.path.join(ROOT, "tests", "golden", "blob", "ventricle.blob"), "0.12"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert expect in out.stdout
