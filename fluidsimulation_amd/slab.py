"""Slab decomposition of one PBF scene over the GPUs of a node.

One process per GPU; rank r owns the grid cells whose x index lies in [planes[r], planes[r+1]).
Per sub-step (ghost membership and neighbour tables are frozen for the sub-step, as the
reference freezes its neighbour lists across solver iterations, Simulation.cpp:112-118):

    predict+collide -> migrate leavers (point-to-point, left/right) -> exchange the boundary
    cell planes as ghosts -> cell grid / counting sort / neighbour table over owned+ghost ->
    iterations x { lambda ; refresh ghost lambda ; delta-p ; refresh ghost positions } ->
    velocity update (+ ghost refreshes between the viscosity stages)

The exchange is neighbour point-to-point over RCCL (torch.distributed "nccl" backend: one
direct xGMI link per slab pair, no ring) with payloads packed/unpacked by the C-ABI slab entry
points straight into torch-owned device buffers.  The same orchestration runs over gloo on CPU
tensors (tests) and, for single-process tests, over an in-process hand-off between slabs.

PyTorch is plumbing here: device buffers, the current HIP stream and torch.distributed.
"""
from __future__ import annotations

import numpy as np

from .binding import PbfContext, PbfParams

INT32_MIN, INT32_MAX = -(2 ** 31), 2 ** 31 - 1
REC = 12  # floats per particle record: pos.xyz, mass, vel.xyz, id bits, pred.xyz, 0
HALO_PRED_LAMBDA, HALO_VELOCITY, HALO_OMEGA, HALO_POSITION = 0, 1, 2, 3


# --------------------------------------------------------------------------- partitioning

def cell_x(pos_x, spacing):
    """NeighborSearch::Flooring on the x axis, in fp32 like the kernels."""
    return np.floor(np.asarray(pos_x, np.float32) / np.float32(spacing)).astype(np.int64)


def split_planes(pos, spacing, world):
    """Cell planes planes[0..world] (open ends) that split the particles into `world` slabs of
    near-equal count; boundaries snap to cell planes so a cell belongs to exactly one rank."""
    cx = cell_x(pos[:, 0], spacing)
    planes = [INT32_MIN]
    if world > 1:
        lo = int(cx.min())
        hist = np.bincount(cx - lo)
        csum = np.cumsum(hist)
        total = int(csum[-1])
        for r in range(1, world):
            k = int(np.searchsorted(csum, total * r / world, side="left")) + 1
            plane = lo + k
            plane = max(plane, planes[-1] + 1 if planes[-1] != INT32_MIN else plane)
            planes.append(plane)
    planes.append(INT32_MAX)
    return planes


def make_records(pos, vel, pred, ids, mass):
    n = pos.shape[0]
    rec = np.zeros((n, REC), np.float32)
    rec[:, 0:3] = pos
    rec[:, 3] = mass
    rec[:, 4:7] = vel
    rec[:, 7] = np.asarray(ids, np.uint32).view(np.float32)
    rec[:, 8:11] = pred
    return rec


def record_ids(rec):
    return np.ascontiguousarray(rec[:, 7]).view(np.uint32) & np.uint32(0x7FFFFFFF)


# --------------------------------------------------------------------------- HIP backend

class HipSlabBackend:
    """One slab on one GPU through the C ABI.  Buffers are torch CUDA tensors whose data_ptr()
    goes to the pbf_slab_* entry points; kernels run on torch's current stream so that RCCL
    ops issued by torch.distributed are ordered with them without host synchronisation."""

    def __init__(self, params: PbfParams, colliders, uniform_mass, x_lo, x_hi, device=0,
                 max_records=None):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = PbfContext(params, device=device)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.ctx.upload(np.zeros((0, 3), np.float32), uniform_mass=uniform_mass)
        self.ctx.set_colliders(colliders)
        self.ctx.slab_set_bounds(x_lo, x_hi)
        cap = int(params.capacity)
        self.max_records = int(max_records or max(65536, cap // 4))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.send = [torch.empty((self.max_records, REC), **f32) for _ in range(2)]
        self.hsend = [torch.empty((self.max_records, 4), **f32) for _ in range(2)]
        self.bcount = [0, 0]
        self.gcount = [0, 0]
        self.viscous = params.xsph_c != 0.0 or params.vorticity_eps != 0.0
        self.vorticity = params.vorticity_eps != 0.0
        self.xsph = params.xsph_c != 0.0

    # -- tensors in / out
    def to_backend(self, rec_np):
        return self.torch.from_numpy(np.ascontiguousarray(rec_np, np.float32)).to(self.device)

    def empty(self, n, width):
        return self.torch.empty((n, width), dtype=self.torch.float32, device=self.device)

    # -- protocol
    def predict(self, dt):
        self.ctx.predict(dt)

    def partition(self):
        _, nl, nr = self.ctx.slab_partition(self.send[0].data_ptr(), self.send[1].data_ptr(),
                                            self.max_records)
        self.gcount = [0, 0]
        return self.send[0][:nl], self.send[1][:nr]

    def append_owned(self, rec):
        if rec.shape[0]:
            rec = rec.contiguous()
            self.ctx.slab_append(rec.data_ptr(), rec.shape[0], -1)

    def pack_ghosts(self):
        out = []
        for side in (0, 1):
            k = self.ctx.slab_pack_ghosts(side, self.send[side].data_ptr(), self.max_records)
            self.bcount[side] = k
            out.append(self.send[side][:k])
        return out[0], out[1]

    def append_ghosts(self, rec, side):
        rec = rec.contiguous()
        self.gcount[side] = rec.shape[0]
        self.ctx.slab_append(rec.data_ptr() if rec.shape[0] else 0, rec.shape[0], side)

    def build(self):
        self.ctx.build_grid()

    def lambda_pass(self, dt):
        self.ctx.solve_lambda(dt)

    def delta_pass(self, dt):
        self.ctx.solve_delta(dt)

    def halo_pack(self, what):
        out = []
        for side in (0, 1):
            k = self.bcount[side]
            self.ctx.slab_pack_halo(side, what, self.hsend[side].data_ptr())
            out.append(self.hsend[side][:k])
        return out[0], out[1]

    def halo_unpack(self, what, from_left, from_right):
        for side, t in ((0, from_left), (1, from_right)):
            t = t.contiguous()
            self.ctx.slab_unpack_halo(side, what, t.data_ptr() if t.shape[0] else 0, t.shape[0])

    def finalize_stage(self, stage, dt):
        self.ctx.finalize_stage(stage, dt)

    def export(self):
        buf = self.empty(int(self.ctx.params.capacity), REC)
        k = self.ctx.slab_export(buf.data_ptr(), buf.shape[0])
        return buf[:k].cpu().numpy()

    def sync(self):
        self.ctx.sync()


# --------------------------------------------------------------------------- exchange

class LocalExchange:
    """All slabs live in this process: hand the payloads over directly."""

    def __init__(self, world):
        self.world = world

    def run(self, outs, backends, width, recv_counts=None):
        ins = []
        for s in range(self.world):
            fl = outs[s - 1][1] if s > 0 else backends[s].empty(0, width)
            fr = outs[s + 1][0] if s < self.world - 1 else backends[s].empty(0, width)
            ins.append((fl, fr))
        return ins


class TorchExchange:
    """One slab per process; neighbours are ranks r-1 and r+1 of the torch.distributed group.
    nccl backend: payloads stay on the GPU (RCCL send/recv over the direct xGMI link);
    gloo backend: CPU tensors (staged through the host when the slab lives on a GPU)."""

    def __init__(self, rank, world, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.group = rank, world, group
        self.backend = dist.get_backend(group)
        self.on_gpu = self.backend == "nccl"

    def _wire(self, t):
        if self.on_gpu:
            return t.contiguous()
        return t.detach().cpu().contiguous()

    def run(self, outs, backends, width, recv_counts=None):
        torch, dist = self.torch, self.dist
        (to_left, to_right), b = outs[0], backends[0]
        left = self.rank - 1 if self.rank > 0 else None
        right = self.rank + 1 if self.rank < self.world - 1 else None
        dev = to_left.device if self.on_gpu else torch.device("cpu")
        if recv_counts is None:
            snd = torch.tensor([to_left.shape[0], to_right.shape[0]], dtype=torch.int64, device=dev)
            rcv = torch.zeros(2, dtype=torch.int64, device=dev)
            ops = []
            if left is not None:
                ops += [dist.P2POp(dist.isend, snd[0:1], left, self.group),
                        dist.P2POp(dist.irecv, rcv[0:1], left, self.group)]
            if right is not None:
                ops += [dist.P2POp(dist.isend, snd[1:2], right, self.group),
                        dist.P2POp(dist.irecv, rcv[1:2], right, self.group)]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            recv_counts = [int(v) for v in rcv.tolist()]
        bufs = [torch.empty((recv_counts[0], width), dtype=torch.float32, device=dev),
                torch.empty((recv_counts[1], width), dtype=torch.float32, device=dev)]
        ops = []
        keep = []
        for peer, payload, buf in ((left, to_left, bufs[0]), (right, to_right, bufs[1])):
            if peer is None:
                continue
            if payload.shape[0]:
                w = self._wire(payload)
                keep.append(w)
                ops.append(dist.P2POp(dist.isend, w, peer, self.group))
            if buf.shape[0]:
                ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if not self.on_gpu and to_left.device.type != "cpu":
            bufs = [x.to(to_left.device) for x in bufs]
        return [(bufs[0], bufs[1])]


# --------------------------------------------------------------------------- orchestration

class SlabGroup:
    """Steps the local slabs (one per process under torch.distributed, `world` of them in the
    in-process test mode) through one PBF sub-step with the exchanges in between."""

    def __init__(self, backends, exchange, iterations):
        self.backends = backends
        self.exchange = exchange
        self.iterations = iterations

    def _halo(self, what):
        outs = [b.halo_pack(what) for b in self.backends]
        counts = None
        if isinstance(self.exchange, TorchExchange):
            counts = list(self.backends[0].gcount)
        ins = self.exchange.run(outs, self.backends, 4, recv_counts=counts)
        for b, (fl, fr) in zip(self.backends, ins):
            b.halo_unpack(what, fl, fr)

    def step(self, dt):
        B = self.backends
        for b in B:
            b.predict(dt)
        outs = [b.partition() for b in B]
        ins = self.exchange.run(outs, B, REC)
        for b, (fl, fr) in zip(B, ins):
            b.append_owned(fl)
            b.append_owned(fr)
        outs = [b.pack_ghosts() for b in B]
        ins = self.exchange.run(outs, B, REC)
        for b, (fl, fr) in zip(B, ins):
            b.append_ghosts(fl, 0)
            b.append_ghosts(fr, 1)
        for b in B:
            b.build()
        viscous = B[0].viscous
        for it in range(self.iterations):
            for b in B:
                b.lambda_pass(dt)
            self._halo(HALO_PRED_LAMBDA)
            for b in B:
                b.delta_pass(dt)
            if it + 1 < self.iterations or viscous:
                self._halo(HALO_PRED_LAMBDA)
        for b in B:
            b.finalize_stage(0, dt)
        if viscous:
            self._halo(HALO_VELOCITY)
            if B[0].vorticity:
                for b in B:
                    b.finalize_stage(1, dt)
                self._halo(HALO_OMEGA)
                for b in B:
                    b.finalize_stage(2, dt)
                if B[0].xsph:
                    self._halo(HALO_VELOCITY)
            if B[0].xsph:
                for b in B:
                    b.finalize_stage(3, dt)

    def export(self):
        """(ids, pos, vel) of every locally owned particle, sorted by id."""
        recs = [b.export() for b in self.backends]
        rec = np.concatenate(recs, axis=0) if recs else np.zeros((0, REC), np.float32)
        ids = record_ids(rec)
        order = np.argsort(ids, kind="stable")
        return ids[order], rec[order, 0:3], rec[order, 4:7]


def populate(backend, pos, mass, ids, planes_lo, planes_hi, spacing, vel=None):
    """Initial population of one slab: the particles whose cell x index is in [lo, hi)."""
    cx = cell_x(pos[:, 0], spacing)
    m = (cx >= planes_lo) & (cx < planes_hi)
    v = np.zeros_like(pos) if vel is None else vel
    rec = make_records(pos[m], v[m], pos[m], ids[m], np.float32(mass))
    backend.append_owned(backend.to_backend(rec))
    return int(m.sum())


def rank_local_params(params: PbfParams, pos, planes, rank, world, headroom=1.5):
    """This rank's pbf_params: capacity for its share (+ headroom for drift and ghosts) and a
    rank-local cell grid that only spans the slab's x range (+ ghost planes and a migration
    margin).  The wrap-around key makes the x extent of the global scene irrelevant to a rank,
    so the per-step scan cost follows the slab, not the scene."""
    p = params.copy()
    share = int(np.ceil(pos.shape[0] / world))
    p.capacity = int(share * headroom) + 65536
    cx = cell_x(pos[:, 0], params.grid_spacing)
    if params.container_enabled:
        x_min = int(np.floor(params.container_min[0] / params.grid_spacing))
        x_max = int(np.floor(min(params.container_max[0], 1e30) / params.grid_spacing))
    else:
        x_min, x_max = int(cx.min()), int(cx.max())
    lo = max(planes[rank], x_min) if planes[rank] != INT32_MIN else x_min
    hi = min(planes[rank + 1], x_max + 1) if planes[rank + 1] != INT32_MAX else x_max + 1
    width = max(hi - lo, 1) + 8
    gx = 16
    while gx < width:
        gx *= 2
    p.grid_dim[0] = min(gx, int(params.grid_dim[0]))
    p.grid_origin_cell[0] = lo - 4
    return p


class SlabSimulation:
    """What bench.py and multi-GPU callers use: this rank's slab of a scene, over RCCL."""

    def __init__(self, params: PbfParams, pos, mass, colliders, rank, world, device=0, vel=None,
                 headroom=1.5):
        import torch.distributed as dist
        planes = split_planes(pos, params.grid_spacing, world)
        self.planes = planes
        p = rank_local_params(params, pos, planes, rank, world, headroom)
        self.backend = HipSlabBackend(p, colliders, mass, planes[rank], planes[rank + 1], device=device)
        ids = np.arange(pos.shape[0], dtype=np.uint32)
        self.n_initial = populate(self.backend, pos, mass, ids, planes[rank], planes[rank + 1],
                                  params.grid_spacing, vel=vel)
        self.group = SlabGroup([self.backend], TorchExchange(rank, world), int(params.solver_iterations))
        self.ctx = self.backend.ctx
        self.dist = dist

    def step(self, dt):
        self.group.step(dt)

    def sync(self):
        self.backend.sync()

    def export(self):
        return self.group.export()
