"""The RCCL branch of the multi-GPU path (bench.py: init_process_group("nccl"), the device pose block of vk_voldor_device_block fed
straight into all_gather_into_tensor, barrier + max-over-ranks timing) must have run before the driver's 8-GPU SCALE run does:
with ONE rank on any GPU box (a one-rank RCCL communicator still goes through every call), and with two ranks where the box has two
GPUs.  The gloo twin of the exchange runs on CPU in tests/test_dist_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_nccl_branch_with_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras", "--force-dist"], capture_output=True, text=True,
                       cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["value"] > 10 and j["n_registered"] == 5


def test_nccl_two_ranks_all_gather():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on the box")
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29542",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras"], capture_output=True, text=True, cwd=ROOT,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 20
