"""The HIP path against fixtures made by the reference's OWN compiled code (tests/golden/ref_*.npz,
written by tests/golden/make_ref_golden.py from Particle.cpp / Constraints.cpp / ConstraintSolver.cpp
/ CollisionDetection.cpp built where they lie under /root/reference): predict + collision response
and the PBD / XPBD distance-constraint solve, bit for bit, with no oracle in between."""
from pathlib import Path

import numpy as np
import pytest

from fluidsimulation_amd import PbfContext
from fluidsimulation_amd import binding as B

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("name", ["ref_constraints_pbd", "ref_constraints_xpbd"])
def test_hip_constraint_projection_equals_compiled_reference(name):
    z = np.load(GOLDEN / f"{name}.npz")
    n = z["x"].shape[0]
    p = B.default_params()
    p.capacity = n
    p.gravity[:] = [0, 0, 0]
    p.solver_iterations = int(z["iterations"])
    g = PbfContext(p, device=0)
    g.upload(z["x"], mass=z["mass"])                 # zero velocity, zero gravity: predicted = current
    cons = [(int(a), int(b), float(r), float(s), float(c)) for a, b, r, s, c in
            zip(z["p0"], z["p1"], z["rest"], z["stiffness"], z["compliance"])]
    xpbd = int(z["mode"]) == 1
    g.set_distance_constraints(cons, B.PBF_CONSTRAINTS_XPBD if xpbd else B.PBF_CONSTRAINTS_PBD)
    dt = float(z["dt"])
    g.predict(dt)
    g.build_grid()
    assert g.download()[2].tobytes() == z["x"].tobytes()
    g.project_constraints(dt)
    assert g.download()[2].tobytes() == z["x_out"].tobytes()
    if xpbd:
        assert g.constraint_lambdas()[0].tobytes() == z["lambdas"].tobytes()


def test_hip_predict_and_collision_response_equal_compiled_reference():
    z = np.load(GOLDEN / "ref_collision.npz")
    p = B.default_params()
    p.capacity = z["pos"].shape[0]
    p.gravity[:] = [0, 0, 0]
    g = PbfContext(p, device=0)
    g.upload(z["pos"], vel=z["vel"], uniform_mass=1.0)
    cols = [B.plane(tuple(z["collider_a"][j]), float(z["collider_d"][j])) if int(z["collider_type"][j]) == B.PBF_COLLIDER_PLANE
            else B.aabb(tuple(z["collider_a"][j]), tuple(z["collider_b"][j])) for j in range(len(z["collider_type"]))]
    g.set_colliders(cols)
    g.predict(float(z["dt"]))
    pos, vel, pred = g.download()
    assert pos.tobytes() == z["pos"].tobytes()
    assert vel.tobytes() == z["vel_out"].tobytes() and pred.tobytes() == z["pred_out"].tobytes()
    # ParticleSet::Update on the collided state (no solver iterations in between)
    g.finalize(float(z["dt"]))
    pos, vel, _ = g.download()
    assert pos.tobytes() == z["pos_updated"].tobytes() and vel.tobytes() == z["vel_updated"].tobytes()


@pytest.mark.parametrize("name", ["ref_loops_c1_first_substep", "ref_loops_jitter_8x16x8"])
def test_hip_solver_loops_equal_compiled_reference(name):
    """tests/golden/ref_loops_*.npz (tests/golden/make_ref_loops.py: every pair evaluated by the
    reference's compiled SPHKernel / glm, accumulated in the order of Simulation.cpp:273-300,
    607-712 over lists ordered as NeighborSearch.cpp:194-227): the HIP path's cells, neighbour
    table IN ORDER, first-frame rest density, rho_i, C_i, lambda_i and corrected positions of one
    solver iteration, bit for bit, with no oracle in between."""
    from fluidsimulation_amd import scenes
    z = np.load(GOLDEN / f"{name}.npz")
    n = z["pos"].shape[0]
    p, _, _, cols = scenes.c1_lattice(2, 2, 2, iterations=1)
    p.capacity = n
    g = PbfContext(p, device=0)
    g.upload(z["pos"], vel=z["vel"], uniform_mass=float(z["mass"]))
    g.set_colliders(cols)
    dt = float(z["dt"])
    g.predict(dt)
    _, v, q = g.download()
    assert v.tobytes() == z["vel_predicted"].tobytes() and q.tobytes() == z["pred"].tobytes()
    g.build_grid()
    assert np.array_equal(g.cells()[0], z["cell"])
    counts, nb = g.neighbors()
    k = z["neighbors"].shape[1]
    assert np.array_equal(counts, z["neighbor_counts"]) and np.array_equal(nb[:, :k], z["neighbors"])
    assert not nb[:, k:].any()
    g.compute_rest_density()
    assert np.float32(g.rest_density()).tobytes() == z["rest_density"].tobytes()
    g.solve_lambda(dt)
    dens, Cc, lam = g.scalars()
    assert dens.tobytes() == z["density"].tobytes() and Cc.tobytes() == z["C"].tobytes()
    assert lam.tobytes() == z["lambdas"].tobytes()
    g.solve_delta(dt)
    assert g.download()[2].tobytes() == z["new_positions"].tobytes()
