"""bench.py's contract, end to end on the GPU: one JSON line on stdout with the driver's keys,
the roofline and the cpu_baseline objects (small workload, a few steps)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "5", "--warmup", "2", "--block", "24", "24",
                        "24", "--cpu-steps", "2", "--evolved-steps", "60"], capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "particle-substeps/s" and d["data"] == "synthetic" and d["dtype"] == "f32"
    n = 24 ** 3
    assert d["config"]["particles"] == n and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(n * 5 / (d["ms_per_step"] * 5e-3), rel=1e-6)
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"]) and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["unit"] == d["unit"] and "sample" in cb
    assert d["health"]["nonfinite"] == 0 and d["health"]["neighbor_overflow"] == 0
    assert "achieved_hbm_gbs" in rf and "limited_by" in rf and "limiters" in rf
    assert "permutation" in d["config"]["upload_order"]
    ev = d["evolved"]
    assert ev["after_substeps"] >= 60 and ev["ms_per_step"] > 0 and ev["nonfinite"] == 0
    assert ev["neighbor_overflow"] == 0 and ev["mean_neighbors"] > 5 and "lambda" in ev["phases"]


def test_bench_two_ranks_rehearsal_over_gloo():
    """the N>1 code path of bench.py (scene split into slabs, ghost-depth tuning on a shared
    transport, per-rank stepping, max-over-ranks timing, whole-scene statistics) with two ranks
    sharing this one GPU over gloo -- a rehearsal of the launch the driver makes with nccl"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "2", "--block", "24", "20", "20",
                        "--backend", "gloo", "--extras-strong-block", "30", "16", "16", "--extras-weak-block", "16",
                        "16", "16"], capture_output=True, text=True, timeout=400, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["ranks_seen"]["all_reduce_of_ones"] == 2 and len(d["ranks_seen"]["devices"]) == 2
    assert d["config"]["particles"] == 2 * 24 * 20 * 20
    assert "slab2" in d["config"]["parallelism"] and "picked from" in d["config"]["parallelism"]
    # the ghost-depth candidates timed before the run (the overlapped refresh needs the native step, i.e. nccl)
    for cand in ("'L4'", "'L2'", "'L1'"):
        assert cand in d["config"]["parallelism"], d["config"]["parallelism"]
    assert "REHEARSAL" in d["data"]
    # the secondary records of a multi-GPU launch: BASELINE's strong (C4) and viscous weak (C5) scenes,
    # here at rehearsal sizes, each with its one-GPU reference time and an efficiency
    ex = d["extras"]
    assert ex["strong_16m"]["scaling"] == "strong" and ex["weak_8m_visc"]["scaling"] == "weak"
    for rec in ex.values():
        assert rec["ms_per_step"] > 0 and rec["parallel_efficiency"] > 0 and rec["nonfinite"] == 0
        assert rec["neighbor_overflow"] == 0 and len(rec["owned_per_rank"]) == 2
        # every rank's own account of its sub-step: owned + ghost counts, device time per phase
        assert [r["rank"] for r in rec["per_rank"]] == [0, 1]
        for r, owned in zip(rec["per_rank"], rec["owned_per_rank"]):
            assert r["owned"] == owned and r["ghosts"] > 0 and r["wall_ms_per_step"] > 0
            assert {"predict", "sort", "neighbors", "lambda", "delta"} <= set(r["device_ms_per_step"])
    assert sum(ex["strong_16m"]["owned_per_rank"]) == 30 * 16 * 16
    assert sum(ex["weak_8m_visc"]["owned_per_rank"]) == 2 * 16 ** 3
    assert d["health"]["nonfinite"] == 0 and d["health"]["neighbor_overflow"] == 0


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun around it (the way the driver launches N = 1): the
    script starts its two ranks itself, as a child process, and relays rank 0's one JSON line --
    n_gpus is what --gpus asked for, and both ranks answered the all-reduce.  (gloo: the two ranks
    share this box's one GPU, a rehearsal of the nccl launch.)"""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--block", "24", "16", "16", "--backend", "gloo", "--no-extras"],
                       capture_output=True, text=True, timeout=400, cwd=str(ROOT), env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"]["all_reduce_of_ones"] == 2
    assert d["config"]["particles"] == 2 * 24 * 16 * 16 and "slab2" in d["config"]["parallelism"]
    assert [r_["rank"] for r_ in d["per_rank"]] == [0, 1]
    assert sum(r_["owned"] for r_ in d["per_rank"]) == d["config"]["particles"]
    assert "starting 2 ranks" in r.stderr
