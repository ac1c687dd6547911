#!/usr/bin/env python3
"""Rehearsal of the N>1 exchange cost on ONE GPU ("mirror" mode): a single slab with closed ends
whose boundary-plane payloads really travel through the transport -- torch.distributed's nccl
backend, or the C-ABI RCCL path (pbf_comm_*) -- addressed to this same rank, and come back as
this slab's ghosts (ids tagged so they stay distinct).  Every exchange of a sub-step (counts,
migrants + ghost planes, halo refreshes) runs with realistic message sizes and the ghost work
is on the GPU, so host and device overhead of the protocol are on the clock.  The physics is
that of the single slab: a timing rehearsal, not a bench result.

    python tools/bench_slab_self.py [--transport torch|rccl] [layers ...]
"""
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


SHIFT = [0.0]


def tag(t):
    """received records get bit 30 of the id set (a mirrored particle is not its original) and
    move 64 cells down in x, into the unused half of the rank-local grid: they are sorted, swept
    and refreshed like real ghosts but do not touch the owned particles' neighbour rows"""
    if t.shape[0]:
        t[:, 7].view(torch.int32).bitwise_or_(0x40000000)
        t[:, 0] -= SHIFT[0]
        t[:, 8] -= SHIFT[0]
    return t


def mirror(base):
    class Mirror(base):
        def _init_peers(self):
            self.left = self.right = 0

        def counts(self, backends):
            own, _, _ = super().counts(backends)[0]
            self.sent = own
            return [(own, own[2:4], own[4:6])]

        def run(self, outs, backends, recv_counts):
            (fl, fr), = super().run(outs, backends, recv_counts)
            if fl[0].shape[1] == S.REC:
                fl, fr = [tag(t) for t in fl], [tag(t) for t in fr]
            return [(fl, fr)]
    return Mirror


def run_native_open(layers, steps=50, warm=10, block=(100, 100, 100)):
    """pbf_slab_step with a one-rank communicator and open ends: the native pipeline (split, the
    one host synchronisation, grid, sweeps) without any neighbour -- what the C++ sub-step costs
    over the plain single-ctx step before exchanges are added"""
    from fluidsimulation_amd.binding import RcclComm
    p, pos, mass, cols = scenes.dam_break(*block, iterations=4)
    q = S.rank_local_params(p, pos, [S.INT32_MIN, S.INT32_MAX], 0, 1, ghost_layers=layers)
    b = S.HipSlabBackend(q, cols, mass, S.INT32_MIN, S.INT32_MAX, device=0, ghost_layers=layers)
    S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), S.INT32_MIN, S.INT32_MAX, p.grid_spacing)
    b.ctx.slab_attach(RcclComm(RcclComm.unique_id(), 0, 1, 0), -1, -1, b.max_records)
    for _ in range(warm):
        b.ctx.slab_step(DT)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        b.ctx.slab_step(DT)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    print(f"native sub-step, no neighbours: {t*1e3:.3f} ms/step", flush=True)
    del b


def run(layers, transport, steps=50, warm=10, block=(100, 100, 100)):
    p, pos, mass, cols = scenes.dam_break(*block, iterations=4)
    SHIFT[0] = 64.0 * float(p.grid_spacing)
    cx = S.cell_x(pos[:, 0], p.grid_spacing)
    lo, hi = int(cx.min()), int(cx.max()) + 1          # closed ends: both boundary planes are real
    q = S.rank_local_params(p, pos, [S.INT32_MIN, S.INT32_MAX], 0, 1, ghost_layers=layers)
    b = S.HipSlabBackend(q, cols, mass, lo, hi, device=0, ghost_layers=layers)
    ids = np.arange(pos.shape[0], dtype=np.uint32)
    S.populate(b, pos, mass, ids, lo, hi, p.grid_spacing)
    if transport == "rccl":
        from fluidsimulation_amd.binding import RcclComm
        ex = mirror(S.P2PExchange)(RcclComm(RcclComm.unique_id(), 0, 1, 0), 0, 1)
    else:
        ex = mirror(S.TorchExchange)(0, 1)
    ex._init_peers()
    grp = S.SlabGroup([b], ex, 4)
    T = {}
    if PROFILE:
        def wrap(obj, name, key):
            f = getattr(obj, name)
            def g(*a, **k):
                t = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0) + time.perf_counter() - t; return r
            setattr(obj, name, g)
        for nm in ["predict", "split_launch", "split_finish", "append_owned", "append_ghosts", "build",
                   "lambda_pass", "delta_pass", "halo_pack", "halo_unpack", "finalize_stage"]:
            wrap(b, nm, nm)
        for nm in ["counts", "run"] + (["halo_fused"] if hasattr(ex, "halo_fused") else []):
            wrap(ex, nm, "x." + nm)
    for _ in range(warm):
        grp.step(DT)
    torch.cuda.synchronize(); T.clear(); t0 = time.perf_counter()
    for _ in range(steps):
        grp.step(DT)
    host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps
    b.sync()
    print(f"{transport:5s} ghost_layers={layers}: {t*1e3:.3f} ms/step (host {host*1e3:.3f}), boundary planes "
          f"{b.bcount}, ghosts {b.gcount}, n on device {b.ctx.n}", flush=True)
    for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
        print(f"      host {k:14s} {v/steps*1e6:8.1f} us/step")
    del grp, b


PROFILE = "--profile" in sys.argv
if PROFILE:
    sys.argv.remove("--profile")

if __name__ == "__main__":
    args = sys.argv[1:]
    transports = ["torch", "rccl"]
    if args and args[0] == "--transport":
        transports = [args[1]]
        args = args[2:]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    p, pos, mass, cols = scenes.dam_break(100, 100, 100, iterations=4)
    g = PbfContext(p, device=0); g.upload(pos, uniform_mass=mass); g.set_colliders(cols)
    for _ in range(10): g.step(DT)
    g.sync(); t0 = time.perf_counter()
    for _ in range(50): g.step(DT)
    g.sync(); print(f"single ctx: {(time.perf_counter()-t0)/50*1e3:.3f} ms/step", flush=True)
    del g
    run_native_open(2)
    for tr in [t for t in transports if t in ("torch", "rccl")]:
        for L in [int(a) for a in args] or [2]:
            run(L, tr)
    dist.destroy_process_group()
