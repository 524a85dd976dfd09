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


def test_bench_line_contract(gpu_device):
    """bench.py prints ONE JSON line with the driver's contract: metric / value / unit / n_gpus / steps / warmup / ms_per_step /
    higher_is_better / scaling / vs_baseline (null) / dtype / data / config.workload, plus `roofline` and `cpu_baseline` and the
    secondary legs; run here on a small batch so that it takes seconds."""
    import json
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--batch", "8", "--num-steps", "2",
                          "--train-batch", "8", "--train-latent", "32", "--train-steps", "1", "--train-warmup", "1", "--vae-batch", "1",
                          "--vae-size", "64", "--vae-steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["metric"] == "images_per_sec_256x256_50step_ldm" and d["unit"] == "images/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["outputs_finite"] is True
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"] and c["gpu_vs_cpu_rel_l2"]["ok"] is True
    assert d["slices_check"]["ok"] is True
    for leg in ("train_mode", "split_schedule", "autocast_bf16", "cfg2", "train_step", "vae_train_step"):
        assert leg in d and "error" not in d[leg], (leg, d.get(leg))
    assert d["train_step"]["bf16"]["ms_per_step"] > 0 and d["train_step"]["f32"]["ms_per_step"] > 0 and d["vae_train_step"]["ms_per_step"] > 0
