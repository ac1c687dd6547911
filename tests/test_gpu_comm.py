"""GPU: the C-ABI RCCL transport (pbf_comm_*) on the one GPU of the test box.  A communicator of
one rank can still send to itself inside a group, which exercises the real ncclSend / ncclRecv
path (binding, stream ordering, byte counts); the left/right protocol logic above it is covered
on CPU over a transport with NCCL's matching rule (tests/test_slab_cpu.py) and the multi-rank
run itself by bench.py --gpus N on the multi-GPU node."""
import numpy as np
import pytest

from fluidsimulation_amd import PbfContext, scenes
from fluidsimulation_amd.binding import PbfError, RcclComm

pytestmark = pytest.mark.gpu
DT = 0.0005


def test_self_send_recv_moves_the_bytes_in_order():
    import torch
    comm = RcclComm(RcclComm.unique_id(), 0, 1, 0)
    s = torch.cuda.current_stream().cuda_stream
    a = torch.arange(1000, dtype=torch.float32, device="cuda")
    b = torch.arange(5000, 5300, dtype=torch.float32, device="cuda")
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    empty = torch.zeros(0, device="cuda")
    comm.batch([(True, 0, a.data_ptr(), a.numel() * 4), (False, 0, ra.data_ptr(), ra.numel() * 4),
                (True, 0, empty.data_ptr(), 0), (False, 0, 0, 0),
                (True, 0, b.data_ptr(), b.numel() * 4), (False, 0, rb.data_ptr(), rb.numel() * 4)], s)
    torch.cuda.synchronize()
    assert torch.equal(a, ra) and torch.equal(b, rb)
    with pytest.raises(PbfError):
        comm.batch([(True, 3, a.data_ptr(), 16)], s)    # peer out of range
    comm.close()


def test_world1_slab_over_rccl_equals_single_ctx():
    """the slab pipeline with the RCCL exchange object and no neighbours"""
    import torch
    from fluidsimulation_amd import slab as S
    p, pos, mass, cols = scenes.dam_break(16, 12, 10, iterations=3)
    comm = RcclComm(RcclComm.unique_id(), 0, 1, 0)
    q = S.rank_local_params(p, pos, [S.INT32_MIN, S.INT32_MAX], 0, 1)
    b = S.HipSlabBackend(q, cols, mass, S.INT32_MIN, S.INT32_MAX, device=0)
    S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), S.INT32_MIN, S.INT32_MAX, p.grid_spacing)
    grp = S.SlabGroup([b], S.P2PExchange(comm, 0, 1), 3)
    g = PbfContext(p, device=0)
    g.upload(pos, uniform_mass=mass)
    g.set_colliders(cols)
    for _ in range(5):
        grp.step(DT)
        g.step(DT)
    ids, gpos, gvel = grp.export()
    pos1, vel1, _ = g.download()
    assert np.array_equal(gpos, pos1) and np.array_equal(gvel, vel1)


def test_fused_halo_exchange_mirrors_the_boundary_planes():
    """pbf_slab_halo_exchange = pack -> grouped send/recv -> unpack, against the three separate
    calls; the slab's own boundary planes come back as its ghosts (rank 0 is its own neighbour)"""
    import torch
    from fluidsimulation_amd import slab as S
    p, pos, mass, cols = scenes.dam_break(24, 8, 8, iterations=2)
    p.max_neighbors = 128
    cx = S.cell_x(pos[:, 0], p.grid_spacing)
    lo, hi = int(cx.min()), int(cx.max()) + 1
    comm = RcclComm(RcclComm.unique_id(), 0, 1, 0)

    def slab():
        q = S.rank_local_params(p, pos, [S.INT32_MIN, S.INT32_MAX], 0, 1)
        b = S.HipSlabBackend(q, cols, mass, lo, hi, device=0)
        S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), lo, hi, p.grid_spacing)
        b.predict(DT)
        b.split_launch()
        own = [int(v) for v in b.count_tensor().cpu().tolist()]
        (ml, gl), (mr, gr) = b.split_finish(own)
        assert own[3] > 0 and own[5] > 0 and own[2] == 0 and own[4] == 0
        tagged = []
        for t in (gl, gr):
            t = t.clone()
            t[:, 7].view(torch.int32).bitwise_or_(0x40000000)
            tagged.append(t)
        b.append_owned(ml, mr)
        b.append_ghosts(tagged[0], tagged[1])
        b.build()
        b.lambda_pass(DT)
        b.delta_pass(DT)
        return b

    a, b = slab(), slab()
    # separate calls, in-process hand-off
    sl, sr = a.halo_pack(0)
    a.halo_unpack(0, sl.clone(), sr.clone())
    # one fused call over RCCL
    ex = S.P2PExchange(comm, 0, 1)
    ex.left = ex.right = 0
    assert ex.halo_fused(b, 0)
    for s in (a, b):
        s.lambda_pass(DT); s.delta_pass(DT); s.finalize_stage(0, DT)
    ra, rb = a.export(), b.export()
    oa, ob = np.argsort(S.record_ids(ra)), np.argsort(S.record_ids(rb))
    assert np.array_equal(ra[oa], rb[ob])
    assert np.isfinite(ra).all()


def _native_rank(rank, world, uid, p, pos, mass, cols, vel, planes, layers, steps, out, errs, rebalance,
                 headroom=3.0, overlap=False, dem=None):
    try:
        import torch
        from fluidsimulation_amd import slab as S
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            q = S.rank_local_params(p, pos, planes, rank, world, headroom=headroom, ghost_layers=layers)
            b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
            S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                       p.grid_spacing, vel=vel)
            comm = RcclComm(uid, rank, world, 0)
            b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                              b.max_records)
            if overlap:
                b.ctx.slab_set_overlap(True)
            my_planes = list(planes)
            for s in range(steps):
                if dem is not None:
                    b.ctx.slab_step_dem(DT, dem)
                else:
                    b.ctx.slab_step(DT)
                if rebalance and s == 2:      # every rank moves the same plane by one cell
                    my_planes[1] += 1
                    b.set_bounds(my_planes[rank], my_planes[rank + 1])
            b.sync()
            out[rank] = b.export()
    except Exception as e:  # noqa: BLE001
        errs.append((rank, repr(e)))


@pytest.mark.parametrize("world,layers,viscous,rebalance,overlap",
                         [(2, 2, False, False, False), (3, 2, False, True, False), (3, 4, False, False, False),
                          (2, 4, True, False, False), (2, 1, True, False, False),
                          # the ghost refresh overlapped with the interior lambda sweep (pbf_slab_set_overlap)
                          (3, 2, False, True, True), (3, 4, False, False, True), (2, 2, True, False, True)])
def test_native_slab_step_over_loopback_equals_single_domain(world, layers, viscous, rebalance, overlap):
    """pbf_slab_step (the native sub-step that runs over RCCL on the multi-GPU node) with `world`
    rank threads on this one GPU over the library's loopback transport (same matching rule and
    stream semantics as the RCCL path): bit-identical with the single-domain oracle"""
    import threading
    from fluidsimulation_amd import slab as S
    from oracle import pyoracle
    from tests.slab_common import slab_scene
    kw = dict(xsph_c=0.01, vorticity_eps=2e-6, vel_scale=2.0) if viscous else {}
    p, pos, mass, cols, vel = slab_scene(nx=60 if layers == 4 else 30, ny=6, nz=6, iterations=3, **kw)
    steps = 6
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    uid = RcclComm.local_id()
    out, errs = {}, []
    ts = [threading.Thread(target=_native_rank, args=(r, world, uid, p, pos, mass, cols, vel, planes, layers,
                                                      steps, out, errs, rebalance, 3.0, overlap)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    order = np.argsort(S.record_ids(rec))
    o = pyoracle.Oracle(p)
    o.upload(pos, vel=vel, uniform_mass=mass)
    o.set_colliders(cols)
    for _ in range(steps):
        o.step(DT)
    op, ov, _, _ = o.download()
    assert rec.shape[0] == pos.shape[0]
    assert np.array_equal(rec[order, 0:3], op) and np.array_equal(rec[order, 4:7], ov)
    assert all(out[r].shape[0] > 0 for r in range(world))


@pytest.mark.parametrize("world,layers,walls", [(3, 2, False), (2, 3, False), (2, 2, True)])
def test_native_granular_slab_step_equals_single_domain(world, layers, walls):
    """pbf_slab_step_dem: the PBD granular / DEM sub-step under the slab
    decomposition -- rank threads over the loopback transport, jittered overlapping grains with random
    velocities (so grains migrate and contacts cross the cuts), optionally on boundary-particle walls --
    bit-identical with the single-domain oracle's step_dem.  PARITY UNPINNED like pbf_step_dem itself."""
    import threading
    from fluidsimulation_amd import binding as B
    from fluidsimulation_amd import slab as S
    from oracle import pyoracle
    h, r, _, _, _ = scenes.sim_params_constants()
    rng = np.random.default_rng(23)
    if walls:
        p, pos, mass, cols, bpos, bm = scenes.boundary_tank(40, 8, 6, iterations=1)
    else:
        p, pos, mass, cols = scenes.dam_break(40, 8, 6, iterations=1)
        p.container_enabled = 0
        bpos = bm = None
    pos = (pos + rng.uniform(-0.25, 0.25, pos.shape).astype(np.float32) * np.float32(r)).astype(np.float32)
    vel = rng.normal(0, 0.8, pos.shape).astype(np.float32)
    dem = B.dem_params(float(r), 1.0, 0.75, 3)
    steps = 8
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    uid = RcclComm.local_id()
    out, errs = {}, []

    def rank_main(rank):
        try:
            import torch
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                q = S.rank_local_params(p, pos, planes, rank, world, headroom=3.0, ghost_layers=layers)
                b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
                if bpos is not None:
                    b.ctx.set_boundary(bpos, uniform_mass=bm)
                S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                           p.grid_spacing, vel=vel)
                comm = RcclComm(uid, rank, world, 0)
                b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                                  b.max_records)
                with pytest.raises(PbfError):          # the single-ctx entry point says where to go
                    b.ctx.step_dem(DT, dem)
                for _ in range(steps):
                    b.ctx.slab_step_dem(DT, dem)
                b.sync()
                out[rank] = b.export()
        except BaseException as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r_,)) for r_ in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    rec = np.concatenate([out[r_] for r_ in range(world)], axis=0)
    order = np.argsort(S.record_ids(rec))
    o = pyoracle.Oracle(p)
    o.upload(pos, vel=vel, uniform_mass=mass)
    o.set_colliders(cols)
    if bpos is not None:
        o.set_boundary(bpos, uniform_mass=bm)
    for _ in range(steps):
        o.step_dem(DT, dem)
    op, ov, _, _ = o.download()
    assert rec.shape[0] == pos.shape[0] and all(out[r_].shape[0] > 0 for r_ in range(world))
    assert np.array_equal(rec[order, 0:3], op) and np.array_equal(rec[order, 4:7], ov)
    assert np.abs(op - pos).max() > 1e-4


@pytest.mark.parametrize("world,layers,mode,viscous", [(3, 2, 0, False), (2, 2, 1, False), (2, 3, 1, True), (3, 1, 1, False)])
def test_native_slab_step_with_distance_constraints_equals_single_domain(world, layers, mode, viscous):
    """distance constraints under the slab decomposition: every rank holds the
    whole list (global particle ids), pbf_slab_step projects it level by level between the solver
    iterations and the velocity update, with a refresh of the ghost positions after every level -- a
    shuffled lattice of links, many of them across the slab cuts, particles migrating meanwhile --
    bit-identical with the single-domain oracle's Gauss-Seidel sweep in list order, PBD and XPBD."""
    import threading
    from fluidsimulation_amd import slab as S
    from oracle import pyoracle
    from tests.slab_common import slab_scene
    from tests.test_gpu_constraints import lattice_links
    nx, ny, nz = 36, 5, 4
    kw = dict(xsph_c=0.01, vorticity_eps=2e-6) if viscous else {}
    p, pos, mass, cols, vel = slab_scene(nx=nx, ny=ny, nz=nz, iterations=2, vel_scale=1.5, **kw)
    h, r, _, _, _ = scenes.sim_params_constants()
    cons = lattice_links(nx, ny, nz, 2.0 * float(r), np.random.default_rng(9))
    steps = 5
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    crossing = sum(1 for c_ in cons if np.searchsorted(planes[1:-1], S.cell_x(pos[c_[0]:c_[0] + 1, 0], p.grid_spacing)[0], side="right")
                   != np.searchsorted(planes[1:-1], S.cell_x(pos[c_[1]:c_[1] + 1, 0], p.grid_spacing)[0], side="right"))
    assert crossing >= 10 * (world - 1)                  # links really straddle the cuts
    uid = RcclComm.local_id()
    out, errs = {}, []

    def rank_main(rank):
        try:
            import torch
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                q = S.rank_local_params(p, pos, planes, rank, world, headroom=3.0, ghost_layers=layers)
                b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
                b.ctx.set_distance_constraints(cons, mode)        # global ids; after pbf_slab_set_bounds
                S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                           p.grid_spacing, vel=vel)
                comm = RcclComm(uid, rank, world, 0)
                b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                                  b.max_records)
                with pytest.raises(PbfError):          # outside the native step the ghost copies cannot be refreshed
                    b.ctx.project_constraints(DT)
                for _ in range(steps):
                    b.ctx.slab_step(DT)
                b.sync()
                out[rank] = b.export()
        except BaseException as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r_,)) for r_ in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    rec = np.concatenate([out[r_] for r_ in range(world)], axis=0)
    order = np.argsort(S.record_ids(rec))
    o = pyoracle.Oracle(p)
    o.upload(pos, vel=vel, uniform_mass=mass)
    o.set_colliders(cols)
    o.set_distance_constraints(cons, mode)
    for _ in range(steps):
        o.step(DT)
    op, ov, _, _ = o.download()
    assert rec.shape[0] == pos.shape[0] and all(out[r_].shape[0] > 0 for r_ in range(world))
    assert np.array_equal(rec[order, 0:3], op) and np.array_equal(rec[order, 4:7], ov)


def test_native_slab_constraint_beyond_the_ghost_planes_fails_on_every_rank():
    """a constraint between particles many cell planes apart cannot be projected by a rank that holds only
    one endpoint: counted on the device, carried in the sub-step's counter block, and the NEXT sub-step fails
    on every rank with the same message -- never a silently missing constraint"""
    import threading
    from fluidsimulation_amd import slab as S
    from tests.slab_common import slab_scene
    world, layers = 2, 2
    p, pos, mass, cols, vel = slab_scene(nx=40, ny=4, nz=4, iterations=1, vel_scale=0.0)
    far = (0, pos.shape[0] - 1, 0.5, 0.5, 1e-7)          # first and last particle: one in each end slab
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    uid = RcclComm.local_id()
    msgs, errs = {}, []

    def rank_main(rank):
        try:
            import torch
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                q = S.rank_local_params(p, pos, planes, rank, world, headroom=3.0, ghost_layers=layers)
                b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
                b.ctx.set_distance_constraints([far], 1)
                S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                           p.grid_spacing, vel=vel)
                comm = RcclComm(uid, rank, world, 0)
                b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                                  b.max_records)
                b.ctx.slab_step(DT)                     # the projection is skipped and counted
                with pytest.raises(PbfError) as e:
                    b.ctx.slab_step(DT)                 # ... and every rank learns of it here
                msgs[rank] = str(e.value)
        except BaseException as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r_,)) for r_ in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errs, errs
    assert msgs[0] == msgs[1] and "beyond the slab's ghost planes" in msgs[0] and "rank 0" in msgs[0]


def test_native_slab_sweep_windows_fall_back_when_cells_wrap():
    """pbf_slab_step sweeps only the owned cell planes plus the ghost planes an iteration still
    needs -- a run of slots of the cell-sorted arrays, as long as no cell wraps around the
    rank-local grid in x.  Two stray owned particles placed exactly one grid period away from each
    rank's OUTERMOST ghost plane (never inside a sweep window) wrap onto it: the windows must
    switch themselves off for those sub-steps, and the result stays bit-identical with the
    single-domain oracle."""
    import threading
    from fluidsimulation_amd import slab as S
    from oracle import pyoracle
    from tests.slab_common import slab_scene
    world, layers, steps = 2, 4, 3
    p, pos, mass, cols, vel = slab_scene(nx=60, ny=6, nz=6, iterations=3)
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    q0 = S.rank_local_params(p, pos, planes, 0, world, headroom=3.0, ghost_layers=layers)
    q1 = S.rank_local_params(p, pos, planes, 1, world, headroom=3.0, ghost_layers=layers)
    sp = float(p.grid_spacing)
    y, z = float(pos[:, 1].mean()), float(pos[:, 2].mean())
    stray = np.array([[(planes[1] + layers - 1 - q0.grid_dim[0] + 0.5) * sp, y, z],      # owned by rank 0
                      [(planes[1] - layers + q1.grid_dim[0] + 0.5) * sp, y, z]], np.float32)  # by rank 1
    assert S.cell_x(stray[0:1, 0], sp)[0] < planes[1] <= S.cell_x(stray[1:2, 0], sp)[0]
    pos2 = np.concatenate([pos, stray]).astype(np.float32)
    vel2 = np.concatenate([vel, np.zeros((2, 3), np.float32)])
    uid = RcclComm.local_id()
    out, errs = {}, []

    def rank_main(rank):
        try:
            import torch
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                q = q0 if rank == 0 else q1         # grids sized WITHOUT the strays: they wrap
                b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
                S.populate(b, pos2, mass, np.arange(pos2.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                           p.grid_spacing, vel=vel2)
                comm = RcclComm(uid, rank, world, 0)
                b.ctx.slab_attach(comm, -1 if rank == 0 else 0, -1 if rank == 1 else 1, b.max_records)
                for _ in range(steps):
                    b.ctx.slab_step(DT)
                b.sync()
                out[rank] = b.export()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    order = np.argsort(S.record_ids(rec))
    o = pyoracle.Oracle(p)
    o.upload(pos2, vel=vel2, uniform_mass=mass)
    o.set_colliders(cols)
    for _ in range(steps):
        o.step(DT)
    op, ov, _, _ = o.download()
    assert rec.shape[0] == pos2.shape[0]
    assert np.array_equal(rec[order, 0:3], op) and np.array_equal(rec[order, 4:7], ov)


def test_c4_16m_particles_four_native_slabs_equal_one_ctx():
    """C4 at full size (256x250x250 = 16 000 000 particles, 4 iterations): four slabs stepped by
    pbf_slab_step (rank threads over the loopback transport, all on this one GPU) against ONE
    ctx stepping the whole scene -- bit-identical after two sub-steps; plus the size-independent
    structure checks on the single ctx (sortedness of the cell order, scene statistics)"""
    import threading
    import torch
    from fluidsimulation_amd import slab as S
    from fluidsimulation_amd import scenes
    p, pos, mass, cols = scenes.dam_break(256, 250, 250, iterations=4)
    n = pos.shape[0]
    assert n == 16_000_000
    steps, world, layers = 2, 4, 4
    # -- one ctx
    g = PbfContext(p, device=0)
    g.upload(pos, uniform_mass=mass)
    g.set_colliders(cols)
    g.predict(DT); g.build_grid()
    xyz, key, sid, start = g.cells()
    ks = key[sid]
    assert np.all(ks[1:] >= ks[:-1])
    same = ks[1:] == ks[:-1]
    assert np.all(sid[1:][same] > sid[:-1][same])
    assert int(start[-1]) == n
    del xyz, key, sid, start, ks, same
    g.solve(DT); g.finalize(DT)
    for _ in range(steps - 1):
        g.step(DT)
    g.sync()
    st = g.stats()
    assert st.n == n and st.nonfinite == 0 and st.neighbor_overflow == 0
    assert 15.0 < st.mean_neighbor_count < 30.0
    p1, v1, _ = g.download()
    del g
    torch.cuda.empty_cache()
    # -- four native slabs
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    uid = RcclComm.local_id()
    out, errs = {}, []
    vel = np.zeros_like(pos)
    ts = [threading.Thread(target=_native_rank, args=(r, world, uid, p, pos, mass, cols, vel, planes, layers,
                                                      steps, out, errs, False)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errs, errs
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    assert rec.shape[0] == n
    order = np.argsort(S.record_ids(rec))
    assert np.array_equal(rec[order, 0:3], p1) and np.array_equal(rec[order, 4:7], v1)


def test_c5_two_native_slabs_of_8m_particles_with_xsph_and_vorticity_equal_one_ctx():
    """C5 at its full PER-GPU size (BASELINE config 5: 8 M particles per GPU, XSPH c = 0.01 +
    vorticity confinement): two slabs of 8 000 000 particles (a 400x200x200 block) stepped by
    pbf_slab_step over the loopback transport, four ghost layers, against ONE ctx on the whole
    16 M scene -- bit-identical after two sub-steps, viscosity stages included; plus the
    size-independent health checks (no non-finite state, no neighbour-row overflow, lattice
    neighbour counts)."""
    import threading
    import torch
    from fluidsimulation_amd import slab as S
    p, pos, mass, cols = scenes.dam_break(400, 200, 200, iterations=4, xsph_c=0.01, vorticity_eps=2e-6)
    n = pos.shape[0]
    assert n == 16_000_000
    steps, world, layers = 2, 2, 4
    g = PbfContext(p, device=0)
    g.upload(pos, uniform_mass=mass)
    g.set_colliders(cols)
    for _ in range(steps):
        g.step(DT)
    g.sync()
    st = g.stats()
    assert st.n == n and st.nonfinite == 0 and st.neighbor_overflow == 0
    assert 15.0 < st.mean_neighbor_count < 30.0 and st.max_neighbor_count <= 64
    p1, v1, _ = g.download()
    assert np.isfinite(v1).all() and float(np.abs(v1).max()) > 0.0
    del g
    torch.cuda.empty_cache()
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    owned = np.bincount(np.searchsorted(planes[1:-1], S.cell_x(pos[:, 0], p.grid_spacing), side="right"),
                        minlength=world)
    assert owned.min() > 7_500_000, owned          # 8 M per slab, up to the plane snapping
    uid = RcclComm.local_id()
    out, errs = {}, []
    vel = np.zeros_like(pos)
    ts = [threading.Thread(target=_native_rank, args=(r, world, uid, p, pos, mass, cols, vel, planes, layers,
                                                      steps, out, errs, False)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errs, errs
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    assert rec.shape[0] == n
    order = np.argsort(S.record_ids(rec))
    assert np.array_equal(rec[order, 0:3], p1) and np.array_equal(rec[order, 4:7], v1)


def test_c5_64m_particles_eight_native_slabs_of_8m_with_xsph_and_vorticity_equal_one_ctx():
    """C5 in its REAL shape (BASELINE config 5: 64 M particles, 8 M per GPU, XSPH c = 0.01 + vorticity
    confinement): the 1600x200x200 block as eight slabs of 8 000 000 particles, eight rank threads
    stepping pbf_slab_step over the loopback transport on this one GPU (288 GB holds all of it),
    against ONE ctx on the whole 64 M scene -- bit-identical after two sub-steps, viscosity stages
    included; plus the size-independent health checks."""
    import threading
    import torch
    from fluidsimulation_amd import slab as S
    p, pos, mass, cols = scenes.dam_break(1600, 200, 200, iterations=4, xsph_c=0.01, vorticity_eps=2e-6)
    n = pos.shape[0]
    assert n == 64_000_000
    steps, world, layers = 2, 8, 2
    free0 = torch.cuda.mem_get_info(0)[0]
    g = PbfContext(p, device=0)
    used_one_ctx = free0 - torch.cuda.mem_get_info(0)[0]
    g.upload(pos, uniform_mass=mass)
    g.set_colliders(cols)
    for _ in range(steps):
        g.step(DT)
    g.sync()
    st = g.stats()
    assert st.n == n and st.nonfinite == 0 and st.neighbor_overflow == 0
    assert 15.0 < st.mean_neighbor_count < 30.0 and st.max_neighbor_count <= 64
    p1, v1, _ = g.download()
    assert float(np.abs(v1).max()) > 0.0
    g.close()
    del g
    torch.cuda.empty_cache()
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    owned = np.bincount(np.searchsorted(planes[1:-1], S.cell_x(pos[:, 0], p.grid_spacing), side="right"),
                        minlength=world)
    assert owned.min() > 7_500_000 and owned.max() < 8_500_000, owned     # 8 M per slab, up to the plane snapping
    uid = RcclComm.local_id()
    out, errs = {}, []
    vel = np.zeros_like(pos)
    ts = [threading.Thread(target=_native_rank, args=(r, world, uid, p, pos, mass, cols, vel, planes, layers,
                                                      steps, out, errs, False, 1.25)) for r in range(world)]
    import time
    for t in ts:
        t.start()
    low, t_end = free0, time.time() + 900
    while any(t.is_alive() for t in ts) and time.time() < t_end:    # high-water mark of the eight slab ctxs
        low = min(low, torch.cuda.mem_get_info(0)[0])
        time.sleep(0.02)
    for t in ts:
        t.join(timeout=10)
    assert not errs, errs
    print(f"C5 64 M: one ctx holds {used_one_ctx / 2**30:.1f} GiB of HBM; high-water mark of the eight slab ctxs + "
          f"exchange buffers {(free0 - low) / 2**30:.1f} GiB")
    ids = np.concatenate([S.record_ids(out[r]) for r in range(world)])
    assert ids.shape[0] == n
    order = np.argsort(ids)
    del ids
    for col, want in ((slice(0, 3), p1), (slice(4, 7), v1)):
        got = np.concatenate([out[r][:, col] for r in range(world)], axis=0)[order]
        assert np.array_equal(got, want)
        del got


def test_c4_eight_native_slabs_of_2m_particles_equal_one_ctx():
    """C4 at its full size (BASELINE config 4: 16 M particles, strong scaling over 8 GPUs): the
    256x250x250 block cut into eight slabs of 2 M particles, eight rank threads stepping
    pbf_slab_step over the loopback transport with two ghost layers, against ONE ctx on the whole
    scene -- bit-identical after two sub-steps.  A 250x250 cell plane holds 62 500 particles, so
    every slab's neighbour rows are in the window-coded 16-bit format (pbf_device.h)."""
    import threading
    import torch
    from fluidsimulation_amd import slab as S
    p, pos, mass, cols = scenes.dam_break(256, 250, 250, iterations=4)
    n = pos.shape[0]
    assert n == 16_000_000
    steps, world, layers = 2, 8, 2
    g = PbfContext(p, device=0)
    g.upload(pos, uniform_mass=mass)
    g.set_colliders(cols)
    for _ in range(steps):
        g.step(DT)
    g.sync()
    st = g.stats()
    assert st.n == n and st.nonfinite == 0 and st.neighbor_overflow == 0
    assert st.row_format_particles[2] > 0.9 * n, list(st.row_format_particles)
    p1, v1, _ = g.download()
    del g
    torch.cuda.empty_cache()
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    owned = np.bincount(np.searchsorted(planes[1:-1], S.cell_x(pos[:, 0], p.grid_spacing), side="right"),
                        minlength=world)
    assert owned.min() > 1_800_000 and owned.max() < 2_200_000, owned
    uid = RcclComm.local_id()
    out, errs = {}, []
    vel = np.zeros_like(pos)
    ts = [threading.Thread(target=_native_rank, args=(r, world, uid, p, pos, mass, cols, vel, planes, layers,
                                                      steps, out, errs, False)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errs, errs
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    assert rec.shape[0] == n
    order = np.argsort(S.record_ids(rec))
    assert np.array_equal(rec[order, 0:3], p1) and np.array_equal(rec[order, 4:7], v1)


def test_slab_simulation_bootstraps_the_native_path_on_an_nccl_group():
    """what bench.py --gpus N does on every rank, here with a world of one: the RCCL unique id
    travels over the torch.distributed (nccl) group, the communicator is created through the C
    ABI, the ctx attaches it and steps natively; equal to the plain ctx"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from fluidsimulation_amd.slab import SlabSimulation
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        p, pos, mass, cols = scenes.dam_break(20, 16, 12, iterations=4)
        sim = SlabSimulation(p, pos, mass, cols, rank=0, world=1, device=0, ghost_layers=4)
        assert sim.native and "native" in sim.transport
        g = PbfContext(p, device=0)
        g.upload(pos, uniform_mass=mass)
        g.set_colliders(cols)
        for _ in range(5):
            sim.step(DT)
            g.step(DT)
        sim.sync()
        st = sim.stats()
        assert st["n"] == pos.shape[0] and st["nonfinite"] == 0
        ids, gpos, gvel = sim.export()
        p1, v1, _ = g.download()
        assert np.array_equal(gpos, p1) and np.array_equal(gvel, v1)
        assert sim.owned_counts() == [pos.shape[0]]
        # bench.py's ghost-depth tuning: a second simulation of the same scene on the SAME
        # communicator; the one that is dropped must leave the communicator to the survivor
        other = SlabSimulation(p, pos, mass, cols, rank=0, world=1, device=0, ghost_layers=2,
                               comm=sim.group.exchange.comm)
        assert other.native and not other.owns_comm
        for _ in range(5):
            other.step(DT)
        other.sync()
        assert np.array_equal(other.export()[1], p1)
        sim.sync()
        sim.owns_comm, other.owns_comm = False, True
        sim.close()          # does not destroy the shared communicator
        del sim
        other.step(DT)
        g.step(DT)
        other.sync()
        assert np.array_equal(other.export()[1], g.download()[0])
        other.close()
    finally:
        dist.destroy_process_group()


def _native_rank_with_walls(rank, world, uid, p, pos, mass, bpos, bm, planes, layers, steps, out, errs):
    try:
        import torch
        from fluidsimulation_amd import slab as S
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            q = S.rank_local_params(p, pos, planes, rank, world, headroom=3.0, ghost_layers=layers)
            b = S.HipSlabBackend(q, [], mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
            b.ctx.set_boundary(bpos, uniform_mass=bm)      # every rank holds the whole (static) wall set
            S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                       p.grid_spacing)
            comm = RcclComm(uid, rank, world, 0)
            b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                              b.max_records)
            for _ in range(steps):
                b.ctx.slab_step(DT)
            b.sync()
            out[rank] = b.export()
    except Exception as e:  # noqa: BLE001
        errs.append((rank, repr(e)))


@pytest.mark.parametrize("world,layers", [(2, 2), (3, 4)])
def test_native_slabs_with_boundary_particle_walls(world, layers):
    """8(f) rank 1 under the slab decomposition: every rank holds the static wall set; the fluid
    is held by the particle walls only (no planes, no clamp); equal to the single-domain oracle"""
    import threading
    from fluidsimulation_amd import slab as S
    from oracle import pyoracle
    p, pos, mass, cols, bpos, bm = scenes.boundary_tank(56, 8, 6, iterations=4, tank_x_factor=1.3)
    assert cols == []
    steps = 12
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    uid = RcclComm.local_id()
    out, errs = {}, []
    ts = [threading.Thread(target=_native_rank_with_walls, args=(r, world, uid, p, pos, mass, bpos, bm, planes,
                                                                layers, steps, out, errs)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    order = np.argsort(S.record_ids(rec))
    o = pyoracle.Oracle(p)
    o.upload(pos, uniform_mass=mass)
    o.set_boundary(bpos, uniform_mass=bm)
    for _ in range(steps):
        o.step(DT)
    op, ov, _, _ = o.download()
    assert rec.shape[0] == pos.shape[0]
    assert np.array_equal(rec[order, 0:3], op) and np.array_equal(rec[order, 4:7], ov)


def test_native_slab_step_failure_is_collective_not_a_hang():
    """One rank of three is attached with exchange buffers far too small for its ghost planes.
    Every rank's counters reach every rank ahead of the sub-step's read-back, so ALL THREE return
    PBF_ERR_CAPACITY naming rank 1 from the same pbf_slab_step call -- nobody enters the payload
    exchange, nobody is left waiting in a receive (without the all-gather rank 1 alone returned
    and its neighbours blocked until the transport's time-out)."""
    import threading
    import time
    from fluidsimulation_amd import slab as S
    from fluidsimulation_amd import binding as B
    from tests.slab_common import slab_scene
    world, layers = 3, 2
    p, pos, mass, cols, vel = slab_scene(nx=36, ny=6, nz=6, iterations=2)
    planes = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers + 1)
    uid = RcclComm.local_id()
    res, errs = {}, []

    def rank_main(rank):
        try:
            import torch
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                q = S.rank_local_params(p, pos, planes, rank, world, headroom=3.0, ghost_layers=layers)
                b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
                S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                           p.grid_spacing, vel=vel)
                comm = RcclComm(uid, rank, world, 0)
                b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                                  8 if rank == 1 else b.max_records)
                try:
                    b.ctx.slab_step(DT)
                    res[rank] = (B.PBF_OK, "")
                except PbfError as e:
                    res[rank] = (e.code, str(e))
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    t0 = time.perf_counter()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=100)
    assert not errs, errs
    assert time.perf_counter() - t0 < 45.0, "ranks waited for a time-out: the failure was not collective"
    assert sorted(res) == [0, 1, 2]
    for r in range(world):
        assert res[r][0] == B.PBF_ERR_CAPACITY and "rank 1:" in res[r][1], (r, res[r])
    assert len({res[r][1] for r in range(world)}) == 1, "every rank reports the same failure"


def test_slab_bounds_reject_an_interior_slab_narrower_than_two_ghost_depths():
    from fluidsimulation_amd import binding as B
    p, pos, mass, cols = scenes.dam_break(8, 6, 6, iterations=2)
    g = PbfContext(p, device=0)
    with pytest.raises(PbfError) as e:
        g.slab_set_bounds(10, 13, 2)          # 3 planes < 2 * 2
    assert e.value.code == B.PBF_ERR_INVALID and "2 * ghost_layers" in str(e.value)
    g.slab_set_bounds(10, 14, 2)              # exactly 2 * ghost_layers: fine
    g.slab_set_bounds(-2**31, 12, 4)          # an open end lifts the rule
    g.slab_set_bounds(10, 2**31 - 1, 4)


def test_native_rebalance_moves_planes_towards_equal_counts_and_keeps_the_result():
    """pbf_slab_rebalance between sub-steps of a collapsing dam break cut deliberately unevenly
    (rank 0 owns two thirds): the plane histograms of all ranks are gathered on every rank, every
    rank moves the planes identically (up to ghost_layers - 1 cells per call), the owned counts
    approach balance, and the state stays bit-identical with the single-domain oracle."""
    import threading
    from fluidsimulation_amd import slab as S
    from oracle import pyoracle
    from tests.slab_common import slab_scene
    world, layers, steps = 3, 3, 12
    p, pos, mass, cols, vel = slab_scene(nx=60, ny=6, nz=6, iterations=2)
    even = S.split_planes(pos, p.grid_spacing, world, min_width=2 * layers)
    planes = [even[0], even[1] + 6, even[2] + 5, even[3]]           # off balance
    lo = int(np.floor(p.container_min[0] / p.grid_spacing))
    n_planes = int(np.floor(p.container_max[0] / p.grid_spacing)) + 1 - lo
    uid = RcclComm.local_id()
    out, errs, hist = {}, [], {}

    def rank_main(rank):
        try:
            import torch
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                q = S.rank_local_params(p, pos, planes, rank, world, headroom=3.0, ghost_layers=layers)
                q.grid_dim[0] = p.grid_dim[0]          # room for the planes to travel
                q.grid_origin_cell[0] = p.grid_origin_cell[0]
                b = S.HipSlabBackend(q, cols, mass, planes[rank], planes[rank + 1], device=0, ghost_layers=layers)
                S.populate(b, pos, mass, np.arange(pos.shape[0], dtype=np.uint32), planes[rank], planes[rank + 1],
                           p.grid_spacing, vel=vel)
                comm = RcclComm(uid, rank, world, 0)
                b.ctx.slab_attach(comm, -1 if rank == 0 else rank - 1, -1 if rank == world - 1 else rank + 1,
                                  b.max_records)
                mine, track = list(planes), []
                for s in range(steps):
                    b.ctx.slab_step(DT)
                    if s % 2 == 1:
                        mine, moved = b.ctx.slab_rebalance(mine, lo, n_planes, layers - 1)
                        track.append((list(mine), moved, b.ctx.slab_owned_count()))
                b.sync()
                hist[rank] = track
                out[rank] = b.export()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs, errs
    # every rank took the same decisions; planes moved, never more than layers - 1 cells per call
    assert [h[0] for h in hist[0]] == [h[0] for h in hist[1]] == [h[0] for h in hist[2]]
    seq = [planes] + [h[0] for h in hist[0]]
    assert seq[-1] != planes and sum(h[1] for h in hist[0]) > 0
    for a, b in zip(seq, seq[1:]):
        assert all(abs(x - y) <= layers - 1 for x, y in zip(a[1:-1], b[1:-1]))
        assert b[2] - b[1] >= 2 * layers
    first = [len(out[r]) for r in range(world)]
    start = np.bincount(np.searchsorted(planes[1:-1], S.cell_x(pos[:, 0], p.grid_spacing), side="right"),
                        minlength=world)
    assert max(first) - min(first) < max(start) - min(start), (start, first)
    rec = np.concatenate([out[r] for r in range(world)], axis=0)
    order = np.argsort(S.record_ids(rec))
    o = pyoracle.Oracle(p)
    o.upload(pos, vel=vel, uniform_mass=mass)
    o.set_colliders(cols)
    for _ in range(steps):
        o.step(DT)
    op, ov, _, _ = o.download()
    assert rec.shape[0] == pos.shape[0]
    assert np.array_equal(rec[order, 0:3], op) and np.array_equal(rec[order, 4:7], ov)


def test_native_rebalance_rejects_moves_the_migration_cannot_carry():
    from fluidsimulation_amd import binding as B
    p, pos, mass, cols = scenes.dam_break(8, 6, 6, iterations=2)
    g = PbfContext(p, device=0)
    g.slab_set_bounds(-2**31, 2**31 - 1, 2)
    with pytest.raises(PbfError) as e:
        g.slab_rebalance([-2**31, 2**31 - 1], 0, 16, 1)      # no communicator attached
    assert e.value.code == B.PBF_ERR_INVALID
