"""
bench.py's launch logic without a GPU: `python bench.py --gpus N` is the form the driver uses; whatever cannot run must
end in ONE JSON line carrying "error" and a non-zero exit code -- never a traceback / assert.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def _no_gpu():
    import torch

    return torch.cuda.device_count() == 0


@pytest.mark.parametrize("n", [1, 2, 8])
def test_no_gpu_prints_an_error_line(n):
    if not _no_gpu():
        pytest.skip("a GPU is visible: the launch itself is covered by tests/test_gpu_bench.py")
    r, line = _bench("--gpus", str(n))
    assert r.returncode != 0 and "Traceback" not in r.stderr
    assert line["error"].startswith("no GPU visible") and line["n_gpus"] == n and line["value"] is None


def test_bad_rank_counts_are_reported_not_asserted():
    r, line = _bench("--gpus", "3")
    assert r.returncode != 0 and "power of two" in line["error"] and "Traceback" not in r.stderr
    r, line = _bench("--gpus", "2", env={"WORLD_SIZE": "4"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in line["error"] and "Traceback" not in r.stderr
