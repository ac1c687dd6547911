#!/usr/bin/env python3
"""Randomised differential run of the native slab sub-step: W rank threads (loopback transport,
one GPU) against the single-domain CPU oracle, bit for bit, over random block shapes, slab
counts, ghost depths, iteration counts, velocities (migration), viscosity, a plane moved
mid-run, the overlapped ghost refresh (pbf_slab_set_overlap) and the granular sub-step
(pbf_slab_step_dem).    python tools/fuzz_slabs.py [cases] [seed]"""
import sys, threading
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from fluidsimulation_amd import scenes
from fluidsimulation_amd import slab as S
from fluidsimulation_amd.binding import RcclComm
from oracle import pyoracle

DT = 0.0005


def one_case(rng):
    world = int(rng.integers(2, 5))
    layers = int(rng.integers(1, 5))
    iters = int(rng.integers(1, 6))
    ny, nz = int(rng.integers(3, 8)), int(rng.integers(3, 8))
    nx = int(rng.integers(world * (2 * layers + 2) * 2, world * (2 * layers + 2) * 2 + 30))
    kw = {}
    if rng.random() < 0.35:
        kw.update(xsph_c=float(rng.choice([0.0, 0.01])), vorticity_eps=float(rng.choice([0.0, 2e-6])))
    p, pos, mass, cols = scenes.dam_break(nx, ny, nz, iterations=iters, **kw)
    vel = rng.normal(0, float(rng.choice([0.0, 2.0, 6.0])), pos.shape).astype(np.float32)
    steps = int(rng.integers(1, 7))
    # a plane moved by one cell hands particles over that are already one plane out: needs two ghost layers
    rebalance_at = int(rng.integers(0, steps)) if rng.random() < 0.3 and layers >= 2 else -1
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    desc0 = dict(world=world, layers=layers, iters=iters, dims=(nx, ny, nz), steps=steps, **kw)
    if rng.random() < 0.3:
        # the Python-orchestrated protocol (slab.py::SlabGroup over the public slab entry points, the
        # fallback transport's code path), all slabs in this process
        backends = []
        ids = np.arange(pos.shape[0], dtype=np.uint32)
        for r in range(world):
            q = p.copy()
            q.capacity = max(int(p.capacity), 4096)
            b = S.HipSlabBackend(q, cols, mass, planes[r], planes[r + 1], device=0, max_records=int(q.capacity),
                                 ghost_layers=layers)
            S.populate(b, pos, mass, ids, planes[r], planes[r + 1], p.grid_spacing, vel=vel)
            backends.append(b)
        group = S.SlabGroup(backends, S.LocalExchange(world), iters)
        for _ in range(steps):
            group.step(DT)
        for b in backends:
            b.sync()
        gids, gpos, gvel = group.export()
        o = pyoracle.Oracle(p)
        o.upload(pos, vel=vel, uniform_mass=mass)
        o.set_colliders(cols)
        for _ in range(steps):
            o.step(DT)
        op, ov, _, _ = o.download()
        o.close()
        ok = np.array_equal(gids, ids) and np.array_equal(gpos, op) and np.array_equal(gvel, ov)
        return ok, dict(desc0, path="python")
    overlap = bool(rng.random() < 0.4)
    dem = None
    if layers >= 2 and rng.random() < 0.25:      # the granular sub-step: overlapping grains, no container clamp
        from fluidsimulation_amd import binding as B
        r_ = float(scenes.sim_params_constants()[1])
        dem = B.dem_params(r_, float(rng.choice([0.5, 1.0])), 0.75, int(rng.integers(0, 5)))
        p.container_enabled = 0
        pos = (pos + rng.uniform(-0.25, 0.25, pos.shape).astype(np.float32) * np.float32(r_)).astype(np.float32)
        rebalance_at = -1
    cons, cmode = None, 0
    if dem is None and rng.random() < 0.25:       # distance constraints across the cuts (global ids)
        idx = lambda i, j, k: (i * ny + j) * nz + k
        d_ = 2.0 * float(scenes.sim_params_constants()[1])
        cons = []
        for i in range(nx):
            for j in range(ny):
                for k in range(nz):
                    if i + 1 < nx and rng.random() < 0.7: cons.append((idx(i, j, k), idx(i + 1, j, k), d_ * float(rng.uniform(0.8, 1.2)), 0.8, 1e-7))
                    if j + 1 < ny and rng.random() < 0.5: cons.append((idx(i, j, k), idx(i, j + 1, k), 0.9 * d_, 0.6, 1e-6))
        cons = [cons[t] for t in rng.permutation(len(cons))]
        cmode = int(rng.integers(0, 2))
        vel = (vel * np.float32(0.3)).astype(np.float32)      # the links must stay shorter than the ghost planes reach
    uid = RcclComm.local_id()
    out, errs = {}, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                q = S.rank_local_params(p, pos, planes, rank, world, headroom=3.0, ghost_layers=layers)
                b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
                if cons is not None:
                    b.ctx.set_distance_constraints(cons, cmode)
                S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                           p.grid_spacing, vel=vel)
                comm = RcclComm(uid, rank, world, 0)
                b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                                  b.max_records)
                if overlap:
                    b.ctx.slab_set_overlap(True)
                mine = list(planes)
                for s in range(steps):
                    if dem is not None:
                        b.ctx.slab_step_dem(DT, dem)
                    else:
                        b.ctx.slab_step(DT)
                    if s == rebalance_at:
                        mine[1] += 1
                        b.set_bounds(mine[rank], mine[rank + 1])
                b.sync()
                out[rank] = b.export()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    desc = dict(world=world, layers=layers, iters=iters, dims=(nx, ny, nz), steps=steps, rebalance_at=rebalance_at,
                overlap=overlap, dem=None if dem is None else int(dem.iterations),
                constraints=None if cons is None else (len(cons), cmode), **kw)
    if errs:
        return False, dict(desc, errors=errs)
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    order = np.argsort(S.record_ids(rec))
    o = pyoracle.Oracle(p)
    o.upload(pos, vel=vel, uniform_mass=mass)
    o.set_colliders(cols)
    if cons is not None:
        o.set_distance_constraints(cons, cmode)
    for _ in range(steps):
        if dem is not None:
            o.step_dem(DT, dem)
        else:
            o.step(DT)
    op, ov, _, _ = o.download()
    o.close()
    ok = rec.shape[0] == pos.shape[0] and np.array_equal(rec[order, 0:3], op) and np.array_equal(rec[order, 4:7], ov)
    return ok, desc


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for c in range(cases):
        ok, desc = one_case(rng)
        if not ok:
            bad += 1
            print("MISMATCH", c, desc, flush=True)
        if (c + 1) % 20 == 0:
            print(f"{c + 1} cases, {bad} mismatches", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
