"""Slab decomposition of one PBF scene over the GPUs of a node.

One process per GPU; rank r owns the grid cells whose x index lies in [planes[r], planes[r+1]).
Per sub-step (ghost membership and neighbour tables are frozen for the sub-step, as the
reference freezes its neighbour lists across solver iterations, Simulation.cpp:112-118):

    predict+collide -> split (one pass: stayers compacted, leavers packed as migrants, the
    boundary cell planes packed as ghosts) -> counts to the neighbours, ONE host synchronisation
    -> migrants + ghosts to the neighbours in one batch -> cell grid / counting sort / neighbour
    table over owned+ghost -> iterations x { lambda ; delta-p ; refresh ghost positions } ->
    velocity update (+ one ghost velocity refresh between vorticity and XSPH)

A particle that leaves towards a neighbour stays behind as a ghost (it lands in the neighbour's
boundary planes, which is exactly what this slab needs from there), and its new owner lists it
with its boundary-plane particles, so migration and ghost exchange need no second round.

Two cell planes per side travel as ghosts (GHOST_LAYERS): every sweep also runs over the ghosts,
and with the second layer present the inner layer's lambda / vorticity comes out of the local
sweep bit-identical to the owner's, so only the positions moved by delta-p (and the
velocities changed by the vorticity force) are refreshed.  With one layer the lambda and omega
of the ghosts are refreshed from their owners as well.

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
GHOST_LAYERS = 2


# --------------------------------------------------------------------------- partitioning

def cell_x(pos_x, spacing):
    """NeighborSearch::Flooring on the x axis, in fp32 like the kernels."""
    return np.floor(np.asarray(pos_x, np.float32) / np.float32(spacing)).astype(np.int64)


def split_planes(pos, spacing, world, min_width=2 * GHOST_LAYERS):
    """Cell planes planes[0..world] (open ends) that split the particles into `world` slabs of
    near-equal count; boundaries snap to cell planes so a cell belongs to exactly one rank.
    Interior slabs are at least min_width cells wide: the ghost layers of a slab come from its
    direct neighbours only, and a migrant (which lands at most GHOST_LAYERS planes deep) must
    not land in the far boundary planes of its new owner."""
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
            plane = max(plane, planes[-1] + min_width if planes[-1] != INT32_MIN else plane)
            planes.append(plane)
    planes.append(INT32_MAX)
    return planes


def shift_planes(planes, counts, tolerance=0.05, min_width=2 * GHOST_LAYERS):
    """Incremental re-balancing: every interior plane moves by at most ONE cell
    towards the heavier of its two slabs (so that, together with the < 1 cell a particle moves
    per sub-step, a leaver still lands inside the ghost planes: the one-round exchange holds).
    Pure function of data every rank has -> all ranks take the same decision."""
    new = list(planes)
    world = len(planes) - 1
    for r in range(1, world):
        a, b = counts[r - 1], counts[r]
        left_lo = new[r - 1]                       # already final for this sweep
        right_hi = planes[r + 1]
        if a > b * (1.0 + tolerance) and (left_lo == INT32_MIN or planes[r] - 1 - left_lo >= min_width):
            new[r] = planes[r] - 1
        elif b > a * (1.0 + tolerance) and (right_hi == INT32_MAX or right_hi - (planes[r] + 1) >= min_width + 1):
            new[r] = planes[r] + 1
    return new


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
                 max_records=None, ghost_layers=GHOST_LAYERS):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = PbfContext(params, device=device)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.ctx.upload(np.zeros((0, 3), np.float32), uniform_mass=uniform_mass)
        self.ctx.set_colliders(colliders)
        self.ctx.slab_set_bounds(x_lo, x_hi, ghost_layers)
        self.ghost_layers = ghost_layers
        cap = int(params.capacity)
        self.max_records = int(max_records or max(65536, cap // 4))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.mig = [torch.empty((self.max_records, REC), **f32) for _ in range(2)]
        self.gh = [torch.empty((self.max_records, REC), **f32) for _ in range(2)]
        self.hsend = [torch.empty((self.max_records, 4), **f32) for _ in range(2)]
        self.counts = torch.zeros(8, dtype=torch.int32, device=self.device)
        self.bcount = [0, 0]   # boundary-list sizes (what a halo refresh sends)
        self.mcount = [0, 0]   # this step's migrants out (kept as ghosts)
        self.gcount = [0, 0]   # ghosts per side (what a halo refresh receives)
        self.in_step = False
        self.viscous = params.xsph_c != 0.0 or params.vorticity_eps != 0.0
        self.vorticity = params.vorticity_eps != 0.0
        self.xsph = params.xsph_c != 0.0

    # -- tensors in / out
    def to_backend(self, rec_np):
        return self.torch.from_numpy(np.ascontiguousarray(rec_np, np.float32)).to(self.device)

    # -- protocol
    def predict(self, dt):
        self.ctx.predict(dt)

    def split_launch(self):
        self.ctx.slab_split(self.mig[0].data_ptr(), self.mig[1].data_ptr(), self.gh[0].data_ptr(),
                            self.gh[1].data_ptr(), self.max_records, self.counts.data_ptr())

    def count_tensor(self):
        """int32[8] on the slab's device: {stay, 0, mig_left, ghost_left, mig_right, ghost_right, 0, 0}"""
        return self.counts

    def split_finish(self, own):
        self.ctx.slab_split_done(own[:6])
        self.mcount = [own[2], own[4]]
        self.bcount = [own[3], own[5]]
        self.gcount = [0, 0]
        self.in_step = True
        return ([self.mig[0][:own[2]], self.gh[0][:own[3]]], [self.mig[1][:own[4]], self.gh[1][:own[5]]])

    def append_owned(self, from_left, from_right=None):
        a = from_left.contiguous()
        b = from_right.contiguous() if from_right is not None else a[:0]
        if a.shape[0] + b.shape[0]:
            self.ctx.slab_append_owned(a.data_ptr() if a.shape[0] else 0, a.shape[0],
                                       b.data_ptr() if b.shape[0] else 0, b.shape[0])
        if self.in_step:
            self.bcount[0] += a.shape[0]
            self.bcount[1] += b.shape[0]

    def append_ghosts(self, from_left, from_right):
        l, r = from_left.contiguous(), from_right.contiguous()
        self.gcount = [l.shape[0] + self.mcount[0], r.shape[0] + self.mcount[1]]
        self.ctx.slab_append_ghosts(l.data_ptr() if l.shape[0] else 0, l.shape[0],
                                    r.data_ptr() if r.shape[0] else 0, r.shape[0])
        self.in_step = False

    def set_bounds(self, x_lo, x_hi):
        self.ctx.slab_set_bounds(x_lo, x_hi, self.ghost_layers)

    def owned_count(self):
        """owned particles after the last sub-step (ghosts sit behind them until the next split)"""
        return self.ctx.slab_owned_count()

    def build(self):
        self.ctx.build_grid()

    def lambda_pass(self, dt):
        self.ctx.solve_lambda(dt)

    def delta_pass(self, dt):
        self.ctx.solve_delta(dt)

    def halo_pack(self, what):
        kl, kr = self.bcount
        self.ctx.slab_pack_halo(what, self.hsend[0].data_ptr(), self.hsend[1].data_ptr())
        return self.hsend[0][:kl], self.hsend[1][:kr]

    def halo_unpack(self, what, from_left, from_right):
        l, r = from_left.contiguous(), from_right.contiguous()
        self.ctx.slab_unpack_halo(what, l.data_ptr() if l.shape[0] else 0, l.shape[0],
                                  r.data_ptr() if r.shape[0] else 0, r.shape[0])

    def finalize_stage(self, stage, dt):
        self.ctx.finalize_stage(stage, dt)

    def export(self):
        buf = self.torch.empty((int(self.ctx.params.capacity), REC), dtype=self.torch.float32,
                               device=self.device)
        k = self.ctx.slab_export(buf.data_ptr(), buf.shape[0])
        return buf[:k].cpu().numpy()

    def sync(self):
        self.ctx.sync()


# --------------------------------------------------------------------------- exchange

class LocalExchange:
    """All slabs live in this process: hand the payloads over directly."""

    def __init__(self, world):
        self.world = world

    def counts(self, backends):
        own = [[int(v) for v in b.count_tensor().cpu().tolist()] for b in backends]
        res = []
        for s in range(self.world):
            fl = own[s - 1][4:6] if s > 0 else [0, 0]
            fr = own[s + 1][2:4] if s < self.world - 1 else [0, 0]
            res.append((own[s], fl, fr))
        return res

    def run(self, outs, backends, recv_counts=None):
        ins = []
        for s in range(self.world):
            fl = outs[s - 1][1] if s > 0 else [t[:0] for t in outs[s][0]]
            fr = outs[s + 1][0] if s < self.world - 1 else [t[:0] for t in outs[s][1]]
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
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None

    def _batch(self, ops):
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()

    def counts(self, backends):
        """The split counters go to the neighbours device-to-device (nccl) before the one
        read-back of the sub-step: own six counters + the two pairs received."""
        torch, dist = self.torch, self.dist
        c = backends[0].count_tensor()
        if not self.on_gpu:
            c = c.cpu()
        r = torch.zeros(4, dtype=c.dtype, device=c.device)
        ops = []
        if self.left is not None:
            ops += [dist.P2POp(dist.isend, c[2:4], self.left, self.group),
                    dist.P2POp(dist.irecv, r[0:2], self.left, self.group)]
        if self.right is not None:
            ops += [dist.P2POp(dist.isend, c[4:6], self.right, self.group),
                    dist.P2POp(dist.irecv, r[2:4], self.right, self.group)]
        self._batch(ops)
        allc = [int(v) for v in torch.cat([c[:6], r]).cpu().tolist()]
        return [(allc[:6], allc[6:8], allc[8:10])]

    def run(self, outs, backends, recv_counts):
        """outs[0] = (payloads to the left, payloads to the right); recv_counts[0] = (row counts
        from the left, from the right), position by position; one batch of sends/receives."""
        torch, dist = self.torch, self.dist
        to_left, to_right = outs[0]
        n_left, n_right = recv_counts[0]
        home = (to_left + to_right)[0].device
        dev = home if self.on_gpu else torch.device("cpu")
        ops, keep, got = [], [], []
        for peer, payloads, sizes in ((self.left, to_left, n_left), (self.right, to_right, n_right)):
            bufs = []
            for t, k in zip(payloads, sizes):
                buf = torch.empty((k if peer is not None else 0, t.shape[1]), dtype=t.dtype, device=dev)
                bufs.append(buf)
                if peer is None:
                    continue
                if t.shape[0]:
                    w = t.contiguous() if self.on_gpu else t.detach().cpu().contiguous()
                    keep.append(w)
                    ops.append(dist.P2POp(dist.isend, w, peer, self.group))
                if k:
                    ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
            got.append(bufs)
        self._batch(ops)
        if not self.on_gpu and home.type != "cpu":
            got = [[x.to(home) for x in bufs] for bufs in got]
        return [(got[0], got[1])]


class P2PExchange:
    """One slab per process, neighbours r-1 and r+1, over a transport that offers
    batch(ops, stream) with ops = (send, peer, pointer, bytes) and NCCL's matching rule (the
    k-th send to a peer pairs with that peer's k-th receive from here): binding.RcclComm, i.e.
    grouped ncclSend / ncclRecv issued straight from the C-ABI library on the ctx stream.  Host
    cost per exchange is a few microseconds (torch.distributed's batch_isend_irecv costs ~90),
    which is what keeps a rank GPU-bound at 1 M particles.  Same protocol as TorchExchange."""

    def __init__(self, comm, rank, world, stream_of=None):
        import torch
        self.torch = torch
        self.comm, self.rank, self.world = comm, rank, world
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self.on_gpu = True
        self.stream_of = stream_of or (lambda t: torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0)
        self._r = None
        self._hrecv = None

    def counts(self, backends):
        torch = self.torch
        c = backends[0].count_tensor()
        if self._r is None or self._r.device != c.device:
            self._r = torch.zeros(4, dtype=c.dtype, device=c.device)   # open ends stay 0 for good
        r = self._r
        item = c.element_size()
        ops = []
        if self.left is not None:
            ops += [(True, self.left, c.data_ptr() + 2 * item, 2 * item), (False, self.left, r.data_ptr(), 2 * item)]
        if self.right is not None:
            ops += [(True, self.right, c.data_ptr() + 4 * item, 2 * item),
                    (False, self.right, r.data_ptr() + 2 * item, 2 * item)]
        self.comm.batch(ops, self.stream_of(c))
        allc = [int(v) for v in torch.cat([c[:6], r]).cpu().tolist()]   # the one host synchronisation
        return [(allc[:6], allc[6:8], allc[8:10])]

    def run(self, outs, backends, recv_counts):
        torch = self.torch
        to_left, to_right = outs[0]
        n_left, n_right = recv_counts[0]
        home = (to_left + to_right)[0]
        ops, keep, got = [], [], []
        for peer, payloads, sizes in ((self.left, to_left, n_left), (self.right, to_right, n_right)):
            bufs = []
            for t, k in zip(payloads, sizes):
                buf = torch.empty((k if peer is not None else 0, t.shape[1]), dtype=t.dtype, device=t.device)
                bufs.append(buf)
                if peer is None:
                    continue
                w = t.contiguous()
                keep.append(w)
                ops.append((True, peer, w.data_ptr(), w.numel() * w.element_size()))
                ops.append((False, peer, buf.data_ptr(), buf.numel() * buf.element_size()))
            got.append(bufs)
        self.comm.batch(ops, self.stream_of(home))
        self._keep = keep   # alive until the stream has consumed them (next exchange at the latest)
        return [(got[0], got[1])]

    def halo_fused(self, backend, what):
        """one C call: pack -> grouped send/recv -> unpack (HIP backend only)"""
        ctx = getattr(backend, "ctx", None)
        if ctx is None or not hasattr(self.comm, "_h"):
            return False
        torch = self.torch
        if self._hrecv is None:
            self._hrecv = [torch.empty_like(backend.hsend[0]), torch.empty_like(backend.hsend[1])]
        if max(backend.gcount) > self._hrecv[0].shape[0]:
            raise RuntimeError("halo buffer too small for the ghost count")
        ctx.slab_halo_exchange(self.comm, what, -1 if self.left is None else self.left,
                               -1 if self.right is None else self.right, backend.hsend[0].data_ptr(),
                               backend.hsend[1].data_ptr(), self._hrecv[0].data_ptr(), self._hrecv[1].data_ptr())
        return True


# --------------------------------------------------------------------------- orchestration

class SlabGroup:
    """Steps the local slabs (one per process under torch.distributed, `world` of them in the
    in-process test mode) through one PBF sub-step with the exchanges in between."""

    def __init__(self, backends, exchange, iterations):
        self.backends = backends
        self.exchange = exchange
        self.iterations = iterations

    def _halo(self, what):
        fused = getattr(self.exchange, "halo_fused", None)
        if fused is not None and len(self.backends) == 1 and fused(self.backends[0], what):
            return
        outs = [tuple([t] for t in b.halo_pack(what)) for b in self.backends]
        counts = [([b.gcount[0]], [b.gcount[1]]) for b in self.backends]
        ins = self.exchange.run(outs, self.backends, counts)
        for b, (fl, fr) in zip(self.backends, ins):
            b.halo_unpack(what, fl[0], fr[0])

    def step(self, dt):
        B = self.backends
        for b in B:
            b.predict(dt)
            b.split_launch()
        cn = self.exchange.counts(B)            # the one host synchronisation of the sub-step
        outs = [b.split_finish(own) for b, (own, _, _) in zip(B, cn)]
        ins = self.exchange.run(outs, B, [(fl, fr) for (_, fl, fr) in cn])
        for b, (fl, fr) in zip(B, ins):
            b.append_owned(fl[0], fr[0])        # migrants in
            b.append_ghosts(fl[1], fr[1])       # neighbour planes (+ own migrants out)
        for b in B:
            b.build()
        viscous = B[0].viscous
        # with a second ghost layer the local sweeps give the inner layer's lambda / omega
        L = B[0].ghost_layers
        local_ghost_sweeps = L >= 2
        # Communication-avoiding halo: a solver iteration on stale ghosts is still exact for the
        # ghost layers that have all their neighbours' data -- lambda needs positions one cell
        # further out, delta-p needs lambda one cell further out again -- so every iteration
        # uses up two of the `valid` layers, and the ghost positions are refreshed from their
        # owners only when fewer than two valid layers are left (L = 2: every iteration;
        # L = 4: every other; L >= 2 * iterations: never).
        valid = L
        for it in range(self.iterations):
            if local_ghost_sweeps and valid < 2:
                self._halo(HALO_PRED_LAMBDA)
                valid = L
            for b in B:
                b.lambda_pass(dt)
            if not local_ghost_sweeps:
                self._halo(HALO_PRED_LAMBDA)
            for b in B:
                b.delta_pass(dt)
            if local_ghost_sweeps:
                valid -= 2
            elif it + 1 < self.iterations:
                self._halo(HALO_PRED_LAMBDA)
        if viscous:
            self._halo(HALO_PRED_LAMBDA)
        # the velocity update also runs over the ghosts: their old and (refreshed) new positions
        # are both here, so v = (new - old) / dt needs no exchange
        for b in B:
            b.finalize_stage(0, dt)
        if viscous:
            if B[0].vorticity:
                for b in B:
                    b.finalize_stage(1, dt)
                if not local_ghost_sweeps:
                    self._halo(HALO_OMEGA)
                for b in B:
                    b.finalize_stage(2, dt)
                if B[0].xsph:
                    self._halo(HALO_VELOCITY)
            if B[0].xsph:
                for b in B:
                    b.finalize_stage(3, dt)

    def rebalance(self, planes, counts, tolerance=0.05):
        """Moves the slab planes one cell towards balance (call between sub-steps, with the
        owned counts of ALL slabs); the next split migrates the particles of the moved plane.
        Returns the new planes."""
        new = shift_planes(planes, counts, tolerance, min_width=2 * self.backends[0].ghost_layers)
        if new != list(planes):
            first = getattr(self.exchange, "rank", 0)
            for k, b in enumerate(self.backends):
                b.set_bounds(new[first + k], new[first + k + 1])
        return new

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


def rank_local_params(params: PbfParams, pos, planes, rank, world, headroom=1.5,
                      ghost_layers=GHOST_LAYERS):
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
    margin = ghost_layers + 2          # ghost planes + migration / re-balancing drift
    width = max(hi - lo, 1) + 2 * margin
    gx = 16
    while gx < width:
        gx *= 2
    p.grid_dim[0] = min(gx, int(params.grid_dim[0]))
    p.grid_origin_cell[0] = lo - margin
    return p


def make_rccl_comm(rank, world, device):
    """RCCL communicator of the C-ABI library, bootstrapped over the existing torch.distributed
    group: rank 0's unique id is broadcast, every rank joins, and an all-reduce of the outcome
    makes the ranks agree on success (a rank-local failure must not leave the others in RCCL)."""
    import torch
    import torch.distributed as dist
    from .binding import PBF_COMM_ID_BYTES, RcclComm
    dev = torch.device("cuda", device)
    box = torch.zeros(PBF_COMM_ID_BYTES + 1, dtype=torch.uint8, device=dev)
    if rank == 0:
        try:
            uid = RcclComm.unique_id()
            box[:PBF_COMM_ID_BYTES] = torch.frombuffer(bytearray(uid), dtype=torch.uint8).to(dev)
            box[PBF_COMM_ID_BYTES] = 1
        except Exception:  # noqa: BLE001
            pass
    dist.broadcast(box, 0)
    host = box.cpu()
    if int(host[PBF_COMM_ID_BYTES]) != 1:
        raise RuntimeError("rank 0 could not create an RCCL unique id")
    comm, err = None, ""
    try:
        comm = RcclComm(bytes(host[:PBF_COMM_ID_BYTES].tolist()), rank, world, device)
    except Exception as e:  # noqa: BLE001
        err = str(e)
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) != 1:
        if comm is not None:
            comm.close()
        raise RuntimeError(err or "another rank failed to join the RCCL communicator")
    return comm


class SlabSimulation:
    """What bench.py and multi-GPU callers use: this rank's slab of a scene, over RCCL."""

    def __init__(self, params: PbfParams, pos, mass, colliders, rank, world, device=0, vel=None,
                 headroom=1.5, rebalance_every=0, ghost_layers=GHOST_LAYERS, transport="auto", native_step=True,
                 comm=None, overlap=False):
        """comm: an RcclComm of this rank to share with other SlabSimulations of the same group (it
        stays the caller's to close); None = create one when the transport asks for it"""
        import torch.distributed as dist
        planes = split_planes(pos, params.grid_spacing, world, min_width=2 * ghost_layers)
        self.planes = planes
        self.ghost_layers = ghost_layers
        p = rank_local_params(params, pos, planes, rank, world, headroom, ghost_layers)
        self.backend = HipSlabBackend(p, colliders, mass, planes[rank], planes[rank + 1], device=device,
                                      ghost_layers=ghost_layers)
        ids = np.arange(pos.shape[0], dtype=np.uint32)
        self.n_initial = populate(self.backend, pos, mass, ids, planes[rank], planes[rank + 1],
                                  params.grid_spacing, vel=vel)
        self.transport = "torch.distributed"
        exchange = None
        self.owns_comm = comm is None
        if comm is not None:
            exchange = P2PExchange(comm, rank, world)
            self.transport = "rccl (C ABI)"
        elif transport in ("rccl", "auto") and dist.get_backend() == "nccl":
            try:
                exchange = P2PExchange(make_rccl_comm(rank, world, device), rank, world)
                self.transport = "rccl (C ABI)"
            except Exception as e:  # noqa: BLE001 -- every rank takes the same decision (see make_rccl_comm)
                if transport == "rccl":
                    raise
                import sys
                print(f"[slab] RCCL transport unavailable ({e}); using torch.distributed", file=sys.stderr)
        if exchange is None:
            exchange = TorchExchange(rank, world)
        self.group = SlabGroup([self.backend], exchange, int(params.solver_iterations))
        # with the C-ABI transport the whole sub-step runs natively (pbf_slab_step): same
        # protocol as SlabGroup.step, issued from C++
        self.native = isinstance(exchange, P2PExchange) and native_step
        if self.native:
            self.backend.ctx.slab_attach(exchange.comm, -1 if rank == 0 else rank - 1,
                                         -1 if rank == world - 1 else rank + 1, self.backend.max_records)
            self.transport = "rccl (C ABI, native sub-step)"
            if overlap:
                self.backend.ctx.slab_set_overlap(True)
                self.transport += ", ghost refresh overlapped with the interior lambda sweep"
        self.overlap = bool(overlap and self.native)
        self.ctx = self.backend.ctx
        # global cell planes the re-balancing histograms cover: the tank, or the initial block +- 64
        if params.container_enabled:
            lo = int(np.floor(params.container_min[0] / params.grid_spacing))
            hi = int(np.floor(min(params.container_max[0], 1e30) / params.grid_spacing)) + 1
        else:
            cx = cell_x(pos[:, 0], params.grid_spacing)
            lo, hi = int(cx.min()) - 64, int(cx.max()) + 65
        self.x_first, self.n_planes = lo, max(1, min(hi - lo, 4096))
        self.dist = dist
        self.rank, self.world = rank, world
        self.steps = 0
        self.rebalance_every = rebalance_every

    def step(self, dt):
        if self.native:
            self.backend.ctx.slab_step(dt)
        else:
            self.group.step(dt)
        self.steps += 1
        if self.rebalance_every and self.steps % self.rebalance_every == 0:
            self.rebalance()

    def owned_counts(self):
        """owned particles of every rank (all-gather of one integer)"""
        import torch
        dev = self.backend.device if self.group.exchange.on_gpu else torch.device("cpu")
        mine = torch.tensor([self.backend.owned_count()], dtype=torch.int64, device=dev)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [int(t.item()) for t in out]

    def rebalance(self, tolerance=0.05):
        """native path: pbf_slab_rebalance (plane histograms gathered over the C-ABI transport, every
        interior plane moved up to ghost_layers - 1 cells towards the equal-count split); otherwise
        one cell per call from the owned counts (SlabGroup.rebalance)"""
        if self.native and self.ghost_layers >= 2:
            self.planes, moved = self.ctx.slab_rebalance(self.planes, self.x_first, self.n_planes,
                                                         self.ghost_layers - 1)
            return moved
        new = self.group.rebalance(self.planes, self.owned_counts(), tolerance)
        moved = sum(1 for a, b in zip(new, self.planes) if a != b)
        self.planes = new
        return moved

    def stats(self):
        """Whole-scene health figures: the per-rank pbf_get_stats (owned particles only) combined
        with two small all-reduces (sum and max), SURVEY.md 8(e)."""
        import torch
        st = self.ctx.stats()
        dev = self.backend.device if self.group.exchange.on_gpu else torch.device("cpu")
        n = float(st.n)
        sums = torch.tensor([n, st.nonfinite, st.neighbor_overflow, n * st.mean_neighbor_count,
                             n * st.mean_abs_C, n * st.mean_C, n * st.com[0], n * st.com[1],
                             n * st.com[2]], dtype=torch.float64, device=dev)
        maxs = torch.tensor([st.max_neighbor_count, st.max_abs_C, st.max_speed],
                            dtype=torch.float64, device=dev)
        self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(maxs, op=self.dist.ReduceOp.MAX)
        sums, maxs = sums.tolist(), maxs.tolist()
        tot = max(sums[0], 1.0)
        return {"n": int(sums[0]), "nonfinite": int(sums[1]), "neighbor_overflow": int(sums[2]),
                "mean_neighbor_count": sums[3] / tot, "mean_abs_C": sums[4] / tot,
                "mean_C": sums[5] / tot, "com": [sums[6] / tot, sums[7] / tot, sums[8] / tot],
                "max_neighbor_count": int(maxs[0]), "max_abs_C": maxs[1], "max_speed": maxs[2],
                "rest_density": st.rest_density}

    def sync(self):
        self.backend.sync()

    def close(self):
        """drains the stream and drops the library's RCCL communicator; call before
        torch.distributed.destroy_process_group() (every rank)"""
        try:
            self.backend.sync()
        finally:
            comm = getattr(self.group.exchange, "comm", None)
            if self.owns_comm and comm is not None and hasattr(comm, "close"):
                comm.close()

    def export(self):
        return self.group.export()
