"""Multi-GPU checks that need real hardware: skipped on a one-GPU box (the driver's GPU test box has one).  ADVICE r05: the in-graph RCCL
route and the bf16 wire have never run with more than one real rank; this is the test that settles it where two GPUs exist."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")
def test_two_rank_rccl_gradient_exchange_layouts_agree():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29612", os.path.join(ROOT, "tools", "rccl_two_rank_check.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=900, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "all checks passed" in r.stdout


def test_rccl_check_script_runs_on_one_rank():
    """the same script on a 1-rank RCCL group with the trainer told that world = 2 (collectives = identity): all three code paths execute, the
    in-graph route with its watchdog armed"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_two_rank_check.py")], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env={**env, "STEPS": "2", "MASTER_PORT": "29614", "FAKE_WORLD": "2"})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "all checks passed" in r.stdout
