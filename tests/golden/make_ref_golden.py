"""Writes tests/golden/ref_*.npz from the reference's OWN compiled sources (oracle/_ref:
Particle.cpp, Constraints.cpp, ConstraintSolver.cpp, CollisionDetection.cpp, Collider.cpp built
where they lie under /root/reference by oracle/Makefile).  Runs only in the container that has
/root/reference; the fixtures are data (inputs + the reference's outputs) and travel to the GPU
box, where the oracle AND the HIP path must reproduce them bit for bit.

    make -C oracle && python -m tests.golden.make_ref_golden
"""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from oracle import pyoracle  # noqa: E402

HERE = Path(__file__).resolve().parent
DT = np.float32(0.0005)


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def constraints_case(ref, mode, seed):
    """ConstraintSolver::SolveConstraints over DistanceConstraint objects: a random graph with
    shared particles, unequal masses, clamped stiffness, some zero compliances, one zero-length
    pair (the EPSILON guard), three solver iterations"""
    rng = np.random.default_rng(seed)
    n, k, iters = 300, 900, 3
    x = rng.uniform(0.0, 2.0, (n, 3)).astype(np.float32)
    mass = rng.uniform(0.5, 3.0, n).astype(np.float32)
    p0 = rng.integers(0, n, k).astype(np.uint32)
    p1 = ((p0 + rng.integers(1, n, k)) % n).astype(np.uint32)
    rest = rng.uniform(0.05, 1.0, k).astype(np.float32)
    stiff = rng.uniform(-0.2, 1.3, k).astype(np.float32)
    comp = (rng.uniform(0, 1e-6, k) * (rng.random(k) < 0.7)).astype(np.float32)
    x[p1[0]] = x[p0[0]]
    out, lam = x.copy(), np.zeros(k, np.float32)
    u32 = C.POINTER(C.c_uint32)
    rc = ref.ref_solve_distance_constraints(mode, float(DT), iters, n, fp(out), fp(mass), k, p0.ctypes.data_as(u32),
                                            p1.ctypes.data_as(u32), fp(rest), fp(stiff), fp(comp), fp(lam))
    assert rc == 0 and np.isfinite(out).all()
    return dict(x=x, mass=mass, p0=p0, p1=p1, rest=rest, stiffness=stiff, compliance=comp,
                iterations=np.int32(iters), dt=DT, mode=np.int32(mode), x_out=out, lambdas=lam)


def collision_case(ref, seed):
    """the particle loop of Simulation::CollisionDetection over ParticleSet::TestCollision /
    OnCollision (four planes and one AABB, in list order), then ParticleSet::Update.  Gravity is
    zero, so the predicted position the reference would have computed is pos + dt * vel."""
    rng = np.random.default_rng(seed)
    n = 2000
    pos = rng.uniform(-0.7, 1.7, (n, 3)).astype(np.float32)
    pos[:, 1] = rng.uniform(-0.05, 1.0, n).astype(np.float32)
    vel = rng.normal(0, 30.0, (n, 3)).astype(np.float32)
    ctype = np.array([4, 4, 2, 4], np.int32)        # Collider::ColliderTypes: PLANE = 4, AABB = 2
    a = np.array([[0, 1, 0], [1, 0, 0], [0.2, 0.1, 0.2], [-1, 0, 0]], np.float32)   # normal | min
    b = np.array([[0, 0, 0], [0, 0, 0], [0.9, 0.8, 1.1], [0, 0, 0]], np.float32)    # - | max
    d = np.array([0.0, -0.5, 0.0, -1.5], np.float32)
    pred = (pos + DT * vel).astype(np.float32)
    v, q, new = vel.copy(), pred.copy(), pred.copy()
    rc = ref.ref_particleset_collide(n, fp(pos), fp(v), fp(q), fp(new), float(DT), len(ctype),
                                     ctype.ctypes.data_as(C.POINTER(C.c_int32)), fp(a), fp(b), fp(d))
    assert rc == 0 and (q != pred).any(axis=1).sum() > n // 10
    pos_u, vel_u = pos.copy(), np.zeros_like(pos)
    ref.ref_particleset_update(n, fp(pos_u), fp(vel_u), fp(new), float(DT))
    return dict(pos=pos, vel=vel, dt=DT, collider_type=ctype, collider_a=a, collider_b=b, collider_d=d,
                vel_out=v, pred_out=q, new_out=new, pos_updated=pos_u, vel_updated=vel_u)


def main():
    ref = pyoracle.load_ref()
    if ref is None or not ref.ref_has_particles():
        raise SystemExit("oracle/_ref with Particle.cpp / Constraints.cpp is needed: make -C oracle (in the "
                         "container that holds /root/reference)")
    np.savez_compressed(HERE / "ref_constraints_pbd.npz", **constraints_case(ref, 0, 11))
    np.savez_compressed(HERE / "ref_constraints_xpbd.npz", **constraints_case(ref, 1, 12))
    np.savez_compressed(HERE / "ref_collision.npz", **collision_case(ref, 13))
    print("wrote ref_constraints_pbd.npz, ref_constraints_xpbd.npz, ref_collision.npz")


if __name__ == "__main__":
    main()
