"""RCCL (torch.distributed backend "nccl") branches of the multi-GPU code on the one GPU of the test box: a process
group of ONE rank in a child process (tests/rccl_one_rank.py) -- every collective degenerates to a copy, but the calls,
dtypes, split arguments, in-place reduce-scatter / all-gather forms and stream ordering are the ones the 8-GPU runs
issue (the two-rank tests of tests/test_gpu_trainer.py use gloo, whose branches stage differently)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_branches_with_a_one_rank_group(dev):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-1 OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
