#!/usr/bin/env python3
"""One-off validation at size: W native slabs (rank threads, loopback transport, one GPU) against
ONE ctx on the whole scene, bit for bit -- optionally with the viscosity stages (C5's settings).

    python tools/check_big_slabs.py nx ny nz W layers steps [viscous]
"""
import sys, threading, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from fluidsimulation_amd import PbfContext, scenes
from fluidsimulation_amd import slab as S
from fluidsimulation_amd.binding import RcclComm

DT = 0.0005
nx, ny, nz, W, layers, steps = [int(a) for a in sys.argv[1:7]]
viscous = len(sys.argv) > 7
kw = dict(xsph_c=0.01, vorticity_eps=2e-6) if viscous else {}
p, pos, mass, cols = scenes.dam_break(nx, ny, nz, iterations=4, **kw)
n = pos.shape[0]
planes = S.split_planes(pos, p.grid_spacing, W, min_width=2 * layers + 1)
uid = RcclComm.local_id()
out, errs, times = {}, [], {}


def rank_main(rank):
    try:
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            q = S.rank_local_params(p, pos, planes, rank, W, headroom=1.5, ghost_layers=layers)
            b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
            S.populate(b, pos, mass, np.arange(n, dtype=np.uint32), planes[rank], planes[rank + 1], p.grid_spacing)
            comm = RcclComm(uid, rank, W, 0)
            b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == W - 1 else rank + 1, b.max_records)
            b.sync(); t0 = time.perf_counter()
            for _ in range(steps):
                b.ctx.slab_step(DT)
            b.sync(); times[rank] = (time.perf_counter() - t0) / steps
            out[rank] = b.export()
    except Exception as e:  # noqa: BLE001
        errs.append((rank, repr(e)))


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
[t.start() for t in ts]
[t.join() for t in ts]
if errs:
    raise SystemExit(f"rank errors: {errs}")
rec = np.concatenate([out[r] for r in range(W)], axis=0)
order = np.argsort(S.record_ids(rec))
g = PbfContext(p, device=0)
g.upload(pos, uniform_mass=mass)
g.set_colliders(cols)
g.sync(); t0 = time.perf_counter()
for _ in range(steps):
    g.step(DT)
g.sync(); single = (time.perf_counter() - t0) / steps
gp, gv, _ = g.download()
bad = int((rec[order, 0:3] != gp).any(axis=1).sum() + (rec[order, 4:7] != gv).any(axis=1).sum())
print(f"{n} particles, {W} slabs, {layers} ghost layers, {steps} sub-steps{', viscous' if viscous else ''}: "
      f"slabs {max(times.values())*1e3:.3f} ms/sub-step (all ranks on one GPU), one ctx {single*1e3:.3f}; "
      f"{bad} particles differ")
sys.exit(1 if bad or rec.shape[0] != n else 0)
