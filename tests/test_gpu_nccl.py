"""RCCL path of the party exchanges on the GPU box (world size 1: API / dtype / layout sanity;
the multi-rank semantics are covered by the gloo tests on CPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from zkhip.net import TorchDistNet
net = TorchDistNet(device=torch.device("cuda", 0))
a = np.arange(36, dtype=np.uint64).reshape(2, 18)
out = net.all_gather(a)
assert len(out) == 1 and out[0].dtype == np.uint64 and (out[0] == a).all()
b = net.all_to_all([np.arange(8, dtype=np.uint64)])
assert (b[0] == np.arange(8)).all()
assert (net.upload, net.download) == (0, 0)  # n_parties - 1 = 0 peers
dist.destroy_process_group()
print("ok")
"""


def test_torchdist_net_over_rccl_world1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _SCRIPT, os.path.join(ROOT, "scalable-collaborative-zksnark_amd")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
