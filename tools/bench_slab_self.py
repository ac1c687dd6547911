#!/usr/bin/env python3
"""Rehearsal of the N>1 transport cost on ONE GPU: a single slab whose boundary-plane payloads
really go through torch.distributed's nccl backend (RCCL) -- sent to this same rank and dropped
on arrival -- so the host and device cost of every exchange of a sub-step (counts, migrants +
ghost planes, halo refreshes) is on the clock, with realistic message sizes, while the physics
stays that of the single slab.  Not a bench result: the wire is a device-local copy."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import torch.distributed as dist
from fluidsimulation_amd import PbfContext, scenes
from fluidsimulation_amd import slab as S

DT = 0.0005
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


class SelfExchange(S.TorchExchange):
    """every payload goes to rank 0 (self) and is received into a scratch buffer of the same
    size; the slab is told that nothing arrived"""

    def __init__(self):
        super().__init__(0, 1)
        self.left = self.right = 0

    def counts(self, backends):
        own, _, _ = super().counts(backends)[0]
        self.sent = own
        return [(own, [0, 0], [0, 0])]

    def run(self, outs, backends, recv_counts):
        to_left, to_right = outs[0]
        sizes = ([t.shape[0] for t in to_left], [t.shape[0] for t in to_right])
        got = super().run(outs, backends, [sizes])[0]
        return [([t[:0] for t in got[0]], [t[:0] for t in got[1]])]


def run(layers, steps=50, warm=10, block=(100, 100, 100)):
    S.GHOST_LAYERS = layers
    p, pos, mass, cols = scenes.dam_break(*block, iterations=4)
    cx = S.cell_x(pos[:, 0], p.grid_spacing)
    lo, hi = int(cx.min()), int(cx.max()) + 1 + 40     # closed ends: the boundary planes are real
    q = S.rank_local_params(p, pos, [S.INT32_MIN, S.INT32_MAX], 0, 1)
    b = S.HipSlabBackend(q, cols, mass, lo, hi, device=0, ghost_layers=layers)
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    S.populate(b, pos, mass, ids, lo, hi, p.grid_spacing)
    grp = S.SlabGroup([b], SelfExchange(), 4)
    for _ in range(warm):
        grp.step(DT)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        grp.step(DT)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    print(f"ghost_layers={layers}: {t*1e3:.3f} ms/step, boundary planes {b.bcount}, sent counts {grp.exchange.sent}", flush=True)
    del grp, b


if __name__ == "__main__":
    p, pos, mass, cols = scenes.dam_break(100, 100, 100, iterations=4)
    g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
    for _ in range(10): g.step(DT)
    g.sync(); t0 = time.perf_counter()
    for _ in range(50): g.step(DT)
    g.sync(); print(f"single ctx: {(time.perf_counter()-t0)/50*1e3:.3f} ms/step", flush=True)
    del g
    for L in [int(a) for a in sys.argv[1:]] or [2]:
        run(L)
    dist.destroy_process_group()
