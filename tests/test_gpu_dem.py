"""GPU parity of the PBD granular / DEM solver, SURVEY.md 8(f) rank 3.

PARITY UNPINNED: solve_pbd_dem exists only as CUDA dead code in the reference
(cuda_simulation.cu:1730-2225, call site commented out at Simulation.cpp:202-213).  The HIP path
is held bit-exact to this repo's CPU restatement of it (oracle/pbf_oracle.c, orc_step_dem)."""
import numpy as np
import pytest

from fluidsimulation_amd import PbfContext, PbfError, scenes
from fluidsimulation_amd import binding as B
from oracle import pyoracle

pytestmark = pytest.mark.gpu
DT = 0.0005


def pair(p, pos, mass, cols, bpos=None, bm=None, vel=None, masses=None):
    g, o = PbfContext(p, device=0), pyoracle.Oracle(p)
    for s in (g, o):
        s.upload(pos, vel=vel, mass=masses, uniform_mass=mass)
        s.set_colliders(cols)
        if bpos is not None:
            s.set_boundary(bpos, uniform_mass=bm)
    return g, o


def same(g, o):
    gp, gv, gq = g.download()
    op, ov, oq, _ = o.download()
    return np.array_equal(gp, op) and np.array_equal(gv, ov) and np.array_equal(gq, oq)


@pytest.mark.parametrize("variant", ["planes", "particle_walls", "per_particle_mass"])
def test_dem_steps_bit_exact(variant):
    h, r, _, _, _ = scenes.sim_params_constants()
    rng = np.random.default_rng(17)
    if variant == "planes":
        p, pos, mass, cols = scenes.dam_break(10, 12, 8, iterations=1)
        p.container_enabled = 0
        bpos = bm = None
    else:
        p, pos, mass, cols, bpos, bm = scenes.boundary_tank(10, 12, 8, iterations=1)
    # overlapping, moving grains: jitter of a quarter radius, random velocities
    pos = (pos + rng.uniform(-0.25, 0.25, pos.shape).astype(np.float32) * np.float32(r)).astype(np.float32)
    vel = rng.normal(0, 0.5, pos.shape).astype(np.float32)
    masses = None
    if variant == "per_particle_mass":
        masses = rng.uniform(0.01, 0.03, pos.shape[0]).astype(np.float32)
    g, o = pair(p, pos, mass, cols, bpos, bm, vel=vel, masses=masses)
    dem = B.dem_params(float(r), 1.0, 0.75, 3)
    for s in range(20):
        g.step_dem(DT, dem)
        o.step_dem(DT, dem)
        if s in (0, 4):
            assert same(g, o), f"state differs after sub-step {s + 1}"
    g.sync()
    assert same(g, o)
    q = g.download()[0]
    assert np.isfinite(q).all()
    assert np.abs(q - pos).max() > 1e-4     # the solver moved things


def test_dem_zero_iterations_and_empty_input():
    h, r, _, _, _ = scenes.sim_params_constants()
    p, pos, mass, cols = scenes.dam_break(6, 6, 6, iterations=1)
    g, o = pair(p, pos, mass, cols)
    dem = B.dem_params(float(r), iterations=0)        # friction pass only
    for _ in range(3):
        g.step_dem(DT, dem); o.step_dem(DT, dem)
    assert same(g, o)
    g.upload(np.zeros((0, 3), np.float32), uniform_mass=mass)
    g.step_dem(DT, dem)
    g.sync()
    with pytest.raises(PbfError):
        g.step_dem(DT, B.dem_params(0.0))


def test_dem_pile_on_particle_walls_300_steps():
    """a grain block settles on the boundary-particle floor: the pile holds, matches the oracle"""
    h, r, _, _, _ = scenes.sim_params_constants()
    p, pos, mass, cols, bpos, bm = scenes.boundary_tank(12, 10, 8, iterations=1)
    g, o = pair(p, pos, mass, cols, bpos, bm)
    dem = B.dem_params(float(r), iterations=3)
    for _ in range(300):
        g.step_dem(DT, dem); o.step_dem(DT, dem)
    g.sync()
    assert same(g, o)
    q, v, _ = g.download()
    assert q[:, 1].min() > -0.5 * float(r) and np.abs(v).max() < 1.0
    # a fluid step afterwards still works on the same ctx (the neighbour table is rebuilt)
    g.step(DT); o.step(DT)
    assert same(g, o)
