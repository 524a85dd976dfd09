"""The two checking tools as tests: a fixed-seed slice of the randomised GEMM parity sweep (all three schedules, every operand
mode) and the two-rank data-parallel training rehearsal on one GPU (gloo standing in for RCCL).  Each runs in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_fuzz_fixed_seed(gpu_device):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_fuzz.py"), "48", "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    assert res.returncode == 0 and "0 mismatches" in res.stdout, res.stdout[-2000:]


def test_gemm_bf16_fuzz_fixed_seed(gpu_device):
    """bf16 operands: stream kernel == ring kernel (bit for bit) == fp64 on the same bf16 values, random shapes / segments / epilogues."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_bf16_fuzz.py"), "40", "4"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "0 mismatches" in res.stdout, res.stdout[-2000:]


def test_ddp_training_rehearsal_two_ranks_gloo(gpu_device):
    env = dict(os.environ, LDM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(ROOT, "tools", "ddp_rehearsal.py")], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "DDP_REHEARSAL_OK" in res.stdout, res.stdout[-3000:]
