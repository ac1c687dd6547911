#!/usr/bin/env python3
"""CPU model of the sweeps' vector-L1 cost (one clock per distinct 32-byte sector of a 64-lane
16-byte gather, tools/micro/gather_cost.hip) for two lane layouts, from the ORACLE's neighbour lists
of a dam-break block: (a) one lane per particle, four entries per trip (the shipped walk);
(b) four lanes per particle ("quad"), one entry per lane and trip.  Prints sectors and pair slots
per particle and sweep.     python tools/gather_layout_model.py NX NY NZ STEPS
"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fluidsimulation_amd import scenes  # noqa: E402
from oracle import pyoracle  # noqa: E402

nx, ny, nz, steps = [int(v) for v in sys.argv[1:5]]
p, pos, mass, cols = scenes.dam_break(nx, ny, nz, iterations=4)
o = pyoracle.Oracle(p, threads=8)
o.upload(pos, uniform_mass=mass)
o.set_colliders(cols)
for _ in range(steps):
    o.step(0.0005)
o.predict_positions(0.0005); o.collision_detection(0.0005); o.spatial_search()
xyz, _ = o.cells()
cnt, nb = o.neighbors()
n = pos.shape[0]
c = xyz - np.array(list(p.grid_origin_cell))
key = (c[:, 0].astype(np.int64) * p.grid_dim[1] + c[:, 1]) * p.grid_dim[2] + c[:, 2]
order = np.lexsort((np.arange(n), key))            # slot -> id
slot_of = np.empty(n, np.int64); slot_of[order] = np.arange(n)
K = nb.shape[1]
lists = slot_of[nb[order]]                          # slot-indexed rows, entries = slots
cn = cnt[order].astype(np.int64)
pad = (cn + 3) // 4 * 4
idx = np.arange(K)[None, :]
lists = np.where(idx < cn[:, None], lists, np.arange(n)[:, None])   # padding = the particle itself
print(f"{n} particles after {steps} sub-steps: mean neighbours {cn.mean():.2f}, max {cn.max()}")

def model(L):
    """L lanes per particle; L <= 4: a trip covers 4 entries (4 / L per lane, one gather instruction each);
    L > 4: a trip covers L entries, one per lane"""
    ppw = 64 // L
    per_trip = max(4, L)
    sect = slots = waves = instr = 0
    for w0 in range(0, n - ppw + 1, ppw * 37):     # sample every 37th wave
        sl = slice(w0, w0 + ppw)
        trips = -(-int(pad[sl].max()) // per_trip)
        waves += 1
        for t in range(trips):
            for u in range(max(1, 4 // L)):        # gather instructions of this trip
                cols_ = [per_trip * t + u + q * max(1, 4 // L) for q in range(L)] if L <= 4 else \
                        [per_trip * t + q for q in range(L)]
                cols_ = [c_ for c_ in cols_ if c_ < K]
                blk = lists[sl][:, cols_]
                ok = (np.array(cols_)[None, :] < pad[sl][:, None])
                a_ = blk[ok]
                if a_.size:
                    sect += len(np.unique(a_ // 2)); instr += 1
            slots += 64 * max(1, 4 // L)
    npart = waves * ppw
    print(f"  {L:2d} lane(s) per particle: {sect / npart:6.2f} sectors, {instr / npart:5.2f} gather instr, "
          f"{slots / npart:6.2f} pair slots per particle and sweep ({cn.mean():.1f} real pairs); {sect / max(instr, 1):.1f} sectors per gather")

for l in (1, 2, 4, 8, 16):
    model(l)


def aligned_model(level):
    """one lane per particle, the walk re-synchronised at every change of dx (level 1: three segments) or
    of (dx, dy) (level 2: nine segments): a wave pads each segment to its longest lane"""
    cx = c[order]                                     # cell coordinates per slot
    sect = slots = waves = instr = 0
    for w0 in range(0, n - 64 + 1, 64 * 37):
        sl = np.arange(w0, w0 + 64)
        waves += 1
        segs = []
        for dx in (-1, 0, 1):
            for dy in ((-1, 0, 1) if level == 2 else (None,)):
                per_lane = []
                for i in sl:
                    js = lists[i, :cn[i]]
                    m = (cx[js, 0] - cx[i, 0]) == dx
                    if dy is not None:
                        m &= (cx[js, 1] - cx[i, 1]) == dy
                    per_lane.append(js[m])
                segs.append(per_lane)
        for per_lane in segs:
            longest = max(len(a_) for a_ in per_lane)
            trips = -(-longest // 4)
            for t in range(trips):
                for u in range(4):
                    k = 4 * t + u
                    a_ = np.array([pl[k] if k < len(pl) else i for pl, i in zip(per_lane, sl) if 4 * t < max(len(pl), 1) or True])
                    act = np.array([4 * t < len(pl) for pl in per_lane])
                    if act.any():
                        sect += len(np.unique(a_[act] // 2)); instr += 1
                slots += 64 * 4
    npart = waves * 64
    print(f"  aligned at every {'(dx, dy)' if level == 2 else 'dx'}: {sect / npart:6.2f} sectors, {instr / npart:5.2f} gather instr, "
          f"{slots / npart:6.2f} pair slots per particle and sweep; {sect / max(instr, 1):.1f} sectors per gather")


aligned_model(1)
aligned_model(2)
