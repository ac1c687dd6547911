"""bench.py's launch contract, the part that needs no GPU: --gpus and the launcher's world must agree,
and a bare `--gpus N` starts its N ranks itself (as a child process, before torch is imported)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CLEAN = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_gpus_must_match_the_launchers_world():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                       timeout=60, env={**CLEAN, "WORLD_SIZE": "1"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                       timeout=60, env={**CLEAN, "WORLD_SIZE": "4"})
    assert r.returncode == 2 and "--gpus 1" in r.stderr and not r.stdout.strip()


def test_bare_gpus_n_launches_n_ranks_and_relays_their_failure():
    """no GPU here: each of the two child ranks stops with "needs a GPU" -- which proves the ranks were
    started (one message per rank), that no JSON line is invented, and that the child's failure is
    the parent's exit code"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible: tests/test_gpu_bench.py runs the real two-rank launch")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--block", "6", "6", "6"], capture_output=True, text=True, timeout=300,
                       cwd=str(ROOT), env=CLEAN)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert r.stderr.count("bench.py needs a GPU") == 2
