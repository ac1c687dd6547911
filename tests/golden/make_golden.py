"""Writes the golden fixtures under tests/golden/ from the CPU oracle.

    python -m tests.golden.make_golden

The reference holds no golden vectors, known-answer tests or fixtures of its own
(SURVEY.md section 4), and its step loop cannot be built here, so these vectors
come from the oracle restatement; they pin it against drift and let the GPU box (which has no
/root/reference) check both the oracle and the HIP path against the same committed data.
A fixture is data only: scene recipe (meta), any random inputs, expected outputs.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from fluidsimulation_amd import aabb, scenes  # noqa: E402

DT = 0.0005
HERE = Path(__file__).resolve().parent

CASES = {
    "c1_6x12x6_5steps": {"scene": "c1", "dims": [6, 12, 6], "iters": 2, "steps": 5},
    "c1_full_10k_1step": {"scene": "c1", "dims": [10, 100, 10], "iters": 2, "steps": 1},
    "c1_full_10k_200steps": {"scene": "c1", "dims": [10, 100, 10], "iters": 2, "steps": 200},
    "dam_12x12x12_3steps": {"scene": "dam", "dims": [12, 12, 12], "iters": 4, "steps": 3},
    "dam_visc_10x10x10_3steps": {"scene": "dam", "dims": [10, 10, 10], "iters": 2, "steps": 3,
                                 "xsph_c": 0.01, "vorticity_eps": 0.05, "random_vel": 0.5},
    "c1_mass_aabb_6x8x6_3steps": {"scene": "c1", "dims": [6, 8, 6], "iters": 3, "steps": 3,
                                  "random_vel": 40.0, "random_mass": True,
                                  "aabb": [[0.6, 1.2, -9.6], [1.9, 2.7, -8.2]]},
    # block held by boundary-particle walls only (no planes, no clamp)
    "btank_10x10x8_6steps": {"scene": "btank", "dims": [10, 10, 8], "iters": 4, "steps": 6},
}


def scene_of(meta):
    if meta["scene"] == "btank":
        return scenes.boundary_tank(*meta["dims"], iterations=meta["iters"])[:4]
    if meta["scene"] == "c1":
        p, pos, mass, cols = scenes.c1_lattice(*meta["dims"], iterations=meta["iters"])
    else:
        p, pos, mass, cols = scenes.dam_break(*meta["dims"], iterations=meta["iters"],
                                              xsph_c=meta.get("xsph_c", 0.0),
                                              vorticity_eps=meta.get("vorticity_eps", 0.0))
    if "aabb" in meta:
        cols = cols + [aabb(*meta["aabb"])]
    return p, pos, mass, cols


def random_inputs(meta, n):
    rng = np.random.default_rng(1234)
    vel = masses = None
    if "random_vel" in meta:
        vel = rng.normal(0, meta["random_vel"], (n, 3)).astype(np.float32)
    if meta.get("random_mass"):
        masses = rng.uniform(0.05, 0.2, n).astype(np.float32)
    return vel, masses


def build_case(meta, vel=None, masses=None, backend="oracle"):
    """Returns (solver, steps) with the scene uploaded; backend 'oracle' or 'hip'."""
    p, pos, mass, cols = scene_of(meta)
    if backend == "oracle":
        from oracle import pyoracle
        s = pyoracle.Oracle(p)
    else:
        from fluidsimulation_amd import PbfContext
        s = PbfContext(p, device=0)
    s.upload(pos, vel=vel, mass=masses, uniform_mass=mass)
    s.set_colliders(cols)
    if meta["scene"] == "btank":
        bpos, bmass = scenes.boundary_tank(*meta["dims"], iterations=meta["iters"])[4:]
        s.set_boundary(bpos, uniform_mass=bmass)
    return s, meta["steps"]


def main():
    for name, meta in CASES.items():
        p, pos, _, _ = scene_of(meta)
        vel, masses = random_inputs(meta, pos.shape[0])
        o, steps = build_case(meta, vel, masses)
        for _ in range(steps):
            o.step(DT)
        opos, ovel, _, _ = o.download()
        extra = {}
        if vel is not None:
            extra["vel0"] = vel
        if masses is not None:
            extra["mass0"] = masses
        if meta["scene"] == "btank":
            extra["bvol"], _, extra["blambda"] = o.boundary()
        np.savez_compressed(HERE / f"{name}.npz", meta=json.dumps(meta), pos=opos, vel=ovel,
                            rest_density=np.float32(o.rest_density()), **extra)
        print(name, pos.shape[0], "particles", steps, "steps ->",
              (HERE / f"{name}.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
