"""GPU: the C++ host facade (Simulation / ParticleSystem / NeighborSearch / ConstraintSolver,
fluidsimulation_amd/host) driven by the headless pbf_run binary, against the CPU oracle."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from fluidsimulation_amd import scenes
from oracle import pyoracle

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
RUN = ROOT / "fluidsimulation_amd" / "host" / "pbf_run"


def read_dump(path):
    b = Path(path).read_bytes()
    n = struct.unpack_from("<Q", b, 0)[0]
    pos = np.frombuffer(b, np.float32, n * 3, 8).reshape(n, 3)
    vel = np.frombuffer(b, np.float32, n * 3, 8 + n * 12).reshape(n, 3)
    k = struct.unpack_from("<Q", b, 8 + n * 24)[0]
    nb0 = np.frombuffer(b, np.uint64, k, 16 + n * 24)
    return pos, vel, nb0


@pytest.mark.parametrize("dims,steps,iters", [((6, 12, 6), 4, 2), ((10, 100, 10), 3, 2)])
def test_headless_driver_matches_oracle(tmp_path, dims, steps, iters):
    assert RUN.exists(), "pbf_run not built: run __graft_entry__.build()"
    dump = tmp_path / "state.bin"
    r = subprocess.run([str(RUN), "--dims", *map(str, dims), "--steps", str(steps), "--iters", str(iters),
                        "--dump", str(dump)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    pos, vel, nb0 = read_dump(dump)
    p, ipos, mass, cols = scenes.c1_lattice(*dims, iterations=iters)
    o = pyoracle.Oracle(p)
    o.upload(ipos, uniform_mass=mass)
    o.set_colliders(cols)
    for _ in range(steps):
        o.step(0.0005)
    op, ov, _, _ = o.download()
    assert np.array_equal(pos, op) and np.array_equal(vel, ov)
    oc, on = o.neighbors()
    assert nb0.tolist() == on[0, :oc[0]].tolist()   # FetchNeighbors(0) in m_search_cache order
    assert f"rest_density {o.rest_density():.9g}" in r.stdout


def test_headless_driver_resumes_from_a_snapshot(tmp_path):
    """3 + 3 sub-steps through a PBFSNAP1 restart file == 6 sub-steps in one go (first-frame rest
    density travels in the file header)"""
    dims, iters = (6, 12, 6), 2
    snap, d1, d2 = tmp_path / "s.pbf", tmp_path / "a.bin", tmp_path / "b.bin"
    base = [str(RUN), "--dims", *map(str, dims), "--iters", str(iters)]
    r = subprocess.run(base + ["--steps", "3", "--save", str(snap)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base + ["--steps", "3", "--resume", str(snap), "--dump", str(d1)], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base + ["--steps", "6", "--dump", str(d2)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    a, b = read_dump(d1), read_dump(d2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_headless_driver_c2_scene_with_boundary_particles(tmp_path):
    """--scene c2: the README CUDA scene built by the C++ facade as the reference's Initialize builds
    it (Simulation::Initialize on an empty ParticleSystem: SetupSimParams -> GenerateFluidCube ->
    InitializeBoundaryParticles, Simulation.cpp:42-72; ParticleSystem::AllocateBoundaryParticles ->
    pbf_set_boundary) == the Python generators + the oracle's boundary coupling; getSimParams()
    (Simulation.h:76) holds SetupSimParams' values"""
    h, r, _, _, _ = scenes.sim_params_constants()
    assert h == np.float32(0.041528303) and r == np.float32(0.012331058)
    dump = tmp_path / "c2.bin"
    res = subprocess.run([str(RUN), "--scene", "c2", "--steps", "3", "--iters", "4", "--dump", str(dump)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert ("SimParams: effective_radius 0.0415283032 particle_radius 0.0123310583 epsilon 100 grid 64x64x64 "
            "boundary 32000") in res.stdout, res.stdout
    pos, vel, _ = read_dump(dump)
    p, ipos, mass, cols, bpos, bm = scenes.c2_reference_cuda_scene(iterations=4)
    assert pos.shape == ipos.shape
    o = pyoracle.Oracle(p)
    o.upload(ipos, uniform_mass=mass)
    o.set_boundary(bpos, uniform_mass=bm)
    for _ in range(3):
        o.step(0.0005)
    op, ov, _, _ = o.download()
    assert np.array_equal(pos, op) and np.array_equal(vel, ov)


SLAB_RUN = ROOT / "fluidsimulation_amd" / "host" / "pbf_slab_run"


@pytest.mark.parametrize("slabs,layers", [(2, 2), (3, 4)])
def test_cpp_slab_driver_equals_one_ctx(slabs, layers):
    """the multi-slab driver in plain C++ (host/pbf_slab_run.cpp + SlabDriver.h over
    pbf_slab_attach / pbf_slab_step): rank threads over the loopback transport on this one GPU,
    --check compares with ONE ctx stepping the whole scene, bit for bit"""
    assert SLAB_RUN.exists(), "pbf_slab_run not built: run __graft_entry__.build()"
    r = subprocess.run([str(SLAB_RUN), "--dims", "60", "10", "8", "--steps", "6", "--iters", "4", "--slabs", str(slabs),
                        "--ghost-layers", str(layers), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of 4800 particles differ" in r.stdout
