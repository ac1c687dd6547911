#!/usr/bin/env python3
"""Randomised differential run: HIP path against the CPU oracle on small random scenes with
random parameters (block shape, iterations, jitter, velocities, per-particle masses, gradient
kernel, viscosity, grid size / origin so that cells wrap, container on / off, extra colliders),
a few sub-steps each, compared bit for bit.    python tools/fuzz_parity.py [cases] [seed] [max block edge]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import PbfContext, scenes
from fluidsimulation_amd import binding as B
from oracle import pyoracle

DT = 0.0005


MAX_EDGE = [14]


def one_case(rng):
    nx, ny, nz = (int(v) for v in rng.integers(3, MAX_EDGE[0], 3))
    kw = {}
    if rng.random() < 0.3:
        kw.update(xsph_c=float(rng.choice([0.0, 0.01, 0.05])), vorticity_eps=float(rng.choice([0.0, 2e-6, 1e-4])))
    p, pos, mass, cols = scenes.dam_break(nx, ny, nz, iterations=int(rng.integers(0, 6)), **kw)
    h, r, _, _, _ = scenes.sim_params_constants()
    pos = (pos + rng.uniform(-0.4, 0.4, pos.shape) * float(r)).astype(np.float32)
    vel = (rng.normal(0, float(rng.choice([0.0, 0.5, 5.0, 40.0])), pos.shape)).astype(np.float32)
    masses = rng.uniform(0.5, 2.0, pos.shape[0]).astype(np.float32) * np.float32(mass) if rng.random() < 0.3 else None
    if rng.random() < 0.3:
        p.gradient_kernel = B.PBF_GRAD_SPIKY
    if rng.random() < 0.3:       # a small grid somewhere else: keys wrap on every axis
        p.grid_dim[:] = [int(v) for v in rng.integers(3, 9, 3)]
        p.grid_origin_cell[:] = [int(v) for v in rng.integers(-50, 50, 3)]
    if rng.random() < 0.3:
        p.container_enabled = 0
    if rng.random() < 0.3:
        cols = list(cols) + [B.aabb((0.05, 0.0, 0.05), (0.15, 0.1, 0.2))]
    if rng.random() < 0.2:       # the CUDA path's hard box (PBF_COLLIDER_HARD_BOX) in front of / behind the others
        lo, hi = pos.min(axis=0), pos.max(axis=0)
        box = B.hard_box(lo - rng.uniform(0.0, 0.02, 3), hi + rng.uniform(0.0, 0.02, 3), float(rng.choice([0.1, 0.5, 1.0])))
        cols = [box] + list(cols) if rng.random() < 0.5 else list(cols) + [box]
    if rng.random() < 0.2:       # duplicates: coincident particles (distance 0)
        k = int(rng.integers(1, 4))
        pos[-k:] = pos[:k]
    steps = int(rng.integers(1, 5))
    g, o = PbfContext(p, device=0), pyoracle.Oracle(p)
    for s in (g, o):
        s.upload(pos, vel=vel, mass=masses, uniform_mass=mass)
        s.set_colliders(cols)
    for _ in range(steps):
        g.step(DT)
        o.step(DT)
    gp, gv, gq = g.download()
    op, ov, oq, _ = o.download()
    ok = gp.tobytes() == op.tobytes() and gv.tobytes() == ov.tobytes()
    nonfinite = not (np.isfinite(op).all() and np.isfinite(ov).all())
    if not ok and nonfinite:
        # a state that already holds NaN / inf (spiky gradient at distance 0: q^2 / 0) is outside the
        # parity claim: an out-of-support pair contributes 0 * diff on the device (NaN for a
        # non-finite diff) where the reference returns a zero vector, so NaN spreads faster there
        # (DESIGN.md section 3).  Both sides must at least agree that the state went non-finite.
        ok = not (np.isfinite(gp).all() and np.isfinite(gv).all())
    desc = dict(nonfinite=nonfinite, dims=(nx, ny, nz), iters=int(p.solver_iterations), steps=steps, spiky=int(p.gradient_kernel),
                grid=list(p.grid_dim), container=int(p.container_enabled), masses=masses is not None, **kw)
    o.close()
    return ok, desc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3:
        MAX_EDGE[0] = int(sys.argv[3])
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        ok, desc = one_case(rng)
        if not ok:
            bad += 1
            print("MISMATCH", c, desc, flush=True)
        if (c + 1) % 50 == 0:
            print(f"{c + 1} cases, {bad} mismatches", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
