"""a short run of the randomised differential MSM test (tools/stress_msm.py): random sizes, offsets,
forced windows, skewed scalars, mixed batches -- all against the C oracle"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_msm_against_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_msm.py"), "15"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
