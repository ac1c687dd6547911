"""A fixed-seed slice of tools/fuzz_parity.py: random small scenes with random parameters (shape,
iterations, jitter, velocities, masses, gradient kernel, viscosity, wrapping grids, container,
extra collider, coincident particles), HIP against the CPU oracle, bit for bit."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_random_scenes_bit_exact_with_the_oracle():
    spec = importlib.util.spec_from_file_location("fuzz_parity", ROOT / "tools" / "fuzz_parity.py")
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(2024)
    bad = []
    for c in range(80):
        ok, desc = fz.one_case(rng)
        if not ok:
            bad.append((c, desc))
    assert not bad, bad


def test_random_native_slab_runs_bit_exact_with_the_oracle():
    """a fixed-seed slice of tools/fuzz_slabs.py: 2-4 rank threads, 1-4 ghost layers, 1-5
    iterations, migration, viscosity stages, a plane moved mid-run -- pbf_slab_step over the
    loopback transport against the single-domain oracle"""
    spec = importlib.util.spec_from_file_location("fuzz_slabs", ROOT / "tools" / "fuzz_slabs.py")
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(99)
    bad = []
    for c in range(40):
        ok, desc = fz.one_case(rng)
        if not ok:
            bad.append((c, desc))
    assert not bad, bad


def test_random_boundary_dem_and_constraint_runs_bit_exact_with_the_oracle():
    """a fixed-seed slice of tools/fuzz_next_rows.py (boundary-particle walls, DEM,
    distance constraints)"""
    spec = importlib.util.spec_from_file_location("fuzz_next_rows", ROOT / "tools" / "fuzz_next_rows.py")
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(4242)
    bad = []
    for c in range(60):
        ok, desc = fz.one_case(rng)
        if not ok:
            bad.append((c, desc))
    assert not bad, bad
