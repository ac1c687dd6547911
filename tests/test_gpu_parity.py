"""GPU parity: the HIP path through the C ABI against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): cell indices, sorted order and neighbour sets bit-exact;
positions/velocities within 1e-5 relative in fp32 -- the kernels restate the reference's
arithmetic op for op, so the tests hold them to bit-exact; 200-step density error and
centre of mass within 1%.
"""
import numpy as np
import pytest

from fluidsimulation_amd import PbfContext, PbfError, aabb, scenes
from fluidsimulation_amd import binding as B
from oracle import pyoracle

pytestmark = pytest.mark.gpu

DT = 0.0005  # GLFWApp.cpp:393


def make_pair(p, pos, mass, cols, vel=None, masses=None):
    g = PbfContext(p, device=0)
    o = pyoracle.Oracle(p)
    for s in (g, o):
        s.upload(pos, vel=vel, mass=masses, uniform_mass=mass)
        s.set_colliders(cols)
    return g, o


def rel_inf(a, b):
    den = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) / den


def jitter(pos, amp, seed=12345):
    """xorshift32 jitter -- deterministic, no numpy RNG state involved."""
    n = pos.size
    x = np.uint32(seed)
    out = np.empty(n, np.float32)
    s = int(x)
    for i in range(n):
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        out[i] = (s / 4294967296.0 - 0.5) * 2.0 * amp
    return (pos + out.reshape(pos.shape)).astype(np.float32)


# ------------------------------------------------------------------ sub-step by sub-step

def test_predict_and_collide_bit_exact():
    p, pos, mass, cols = scenes.c1_lattice(6, 10, 6)
    rng = np.random.default_rng(7)
    vel = rng.normal(0, 400.0, pos.shape).astype(np.float32)  # fast enough to cross planes
    cols = cols + [aabb((0.6, 1.2, -9.6), (1.9, 2.7, -8.2))]
    g, o = make_pair(p, pos, mass, cols, vel=vel)
    g.predict(DT)
    o.predict_positions(DT)
    o.collision_detection(DT)
    gp, gv, gq = g.download()
    op, ov, oq, onew = o.download()
    assert np.array_equal(gp, op)
    assert np.array_equal(gv, ov)
    assert np.array_equal(gq, oq) and np.array_equal(gq, onew)
    # the collider branches were really taken
    free = pos + np.float32(DT) * (vel + np.float32(DT) * np.float32(-9.8) * np.array([0, 1, 0], np.float32))
    assert (np.abs(oq - free).max(axis=1) > 1e-3).sum() > 10


def test_hard_box_collider_the_cuda_paths_wall_response_bit_exact():
    """PBF_COLLIDER_HARD_BOX: the box the reference's CUDA path bakes into integrate_pbd_d
    (cuda_simulation.cu:473-511: x, z in [-1, 1], y >= 0, whole-vector x boundary_damping 0.1) as a
    selectable collider -- clamp, no re-prediction.  PARITY UNPINNED (CUDA-only code): held to this repo's
    oracle bit for bit, first on one predict pass with fast particles, then over 40 sub-steps of the README
    scene's fluid cube dropped into that box."""
    from fluidsimulation_amd.binding import hard_box
    p, pos, mass, _ = scenes.c1_lattice(6, 10, 6)
    rng = np.random.default_rng(11)
    vel = rng.normal(0, 900.0, pos.shape).astype(np.float32)
    lo, hi = pos.min(axis=0) - 0.2, pos.max(axis=0) + 0.2
    cols = [hard_box(lo, hi, 0.1)]
    g, o = make_pair(p, pos, mass, cols, vel=vel)
    g.predict(DT)
    o.predict_positions(DT)
    o.collision_detection(DT)
    gp, gv, gq = g.download()
    op, ov, oq, onew = o.download()
    assert np.array_equal(gv, ov) and np.array_equal(gq, oq) and np.array_equal(gq, onew)
    assert (oq[:, 0] >= lo[0]).all() and (oq[:, 0] <= hi[0]).all() and (oq[:, 1] >= lo[1]).all()
    assert (oq[:, 2] >= lo[2]).all() and (oq[:, 2] <= hi[2]).all()
    assert (oq[:, 1] > hi[1]).any()                       # the box is open at the top, as the reference's
    hit = (oq[:, 0] == lo[0]) | (oq[:, 0] == hi[0]) | (oq[:, 1] == lo[1]) | (oq[:, 2] == lo[2]) | (oq[:, 2] == hi[2])
    assert hit.sum() > 20
    free_v = vel + np.float32(DT) * np.float32(-9.8) * np.array([0, 1, 0], np.float32)
    assert (np.abs(ov[hit]).max(axis=1) <= 0.1 * np.abs(free_v[hit]).max(axis=1) + 1e-3).all()   # whole vector damped
    # the README scene's cube (64 000 particles, Simulation.cpp:528-576) in the CUDA path's box
    p2, pos2, m2, _cols, _b, _bm = scenes.c2_reference_cuda_scene(iterations=3)   # iterations = 3: Simulation.cpp:148
    box = [hard_box((-1.0, 0.0, -1.0), (1.0, 1.0e9, 1.0), 0.1)]
    g, o = make_pair(p2, pos2, m2, box)
    for _ in range(40):
        g.step(DT)
        o.step(DT)
    g.sync()
    q, v, _ = g.download()
    oq2, ov2, _, _ = o.download()
    assert np.array_equal(q, oq2) and np.array_equal(v, ov2)
    # the box acts on the PREDICTED positions: the solver's corrections may carry a particle a hair outside
    assert np.isfinite(q).all() and q[:, 1].min() > -0.01 and np.abs(q[:, [0, 2]]).max() < 1.01


def test_cells_sorted_order_neighbor_sets_bit_exact():
    p, pos, mass, cols = scenes.c1_lattice(8, 16, 8)
    pos = jitter(pos, 0.2)
    g, o = make_pair(p, pos, mass, cols)
    g.predict(DT)
    g.build_grid()
    o.predict_positions(DT)
    o.collision_detection(DT)
    o.spatial_search()
    xyz, key, sid, start = g.cells()
    oxyz, ohash = o.cells()
    # cell indices == NeighborSearch::Flooring
    assert np.array_equal(xyz, oxyz)
    gd = np.array(list(p.grid_dim), np.int64)
    w = np.mod(oxyz.astype(np.int64), gd)
    okey = ((w[:, 0] * gd[1] + w[:, 1]) * gd[2] + w[:, 2]).astype(np.uint32)
    assert np.array_equal(key, okey)
    # sorted order == stable sort of the original indices by cell key
    assert np.array_equal(sid, np.argsort(okey, kind="stable").astype(np.uint32))
    # cell_start == exclusive prefix sum of the occupancy
    occ = np.bincount(okey, minlength=start.size - 1)
    assert np.array_equal(start, np.concatenate([[0], np.cumsum(occ)]).astype(np.uint32))
    # neighbour lists: same members in the same order as m_search_cache
    gc, gn = g.neighbors()
    oc, on = o.neighbors(stride=int(p.max_neighbors))
    assert np.array_equal(gc, oc)
    assert np.array_equal(gn, on)
    assert oc.max() > 20


def test_neighbors_equal_brute_force():
    """the reference's own cross-check idea: SpatialSearch == NaiveSearch as sets"""
    p, pos, mass, cols = scenes.c1_lattice(5, 9, 5)
    pos = jitter(pos, 0.24, seed=99)
    g, o = make_pair(p, pos, mass, [])
    g.build_grid()
    o.naive_search()  # uses m_new_positions == positions here
    gc, gn = g.neighbors()
    oc, on = o.neighbors(stride=int(p.max_neighbors))
    assert np.array_equal(gc, oc)
    for i in range(pos.shape[0]):
        assert set(gn[i, :gc[i]]) == set(on[i, :oc[i]])


def test_rest_density_first_frame_bit_exact():
    p, pos, mass, cols = scenes.c1_lattice(7, 11, 6)
    g, o = make_pair(p, pos, mass, cols)
    g.predict(DT); g.build_grid(); g.compute_rest_density()
    o.predict_positions(DT); o.collision_detection(DT); o.spatial_search(); o.compute_rest_density()
    assert g.rest_density() == o.rest_density()
    assert 0.3 < g.rest_density() < 0.9


def test_density_lambda_correction_bit_exact():
    p, pos, mass, cols = scenes.c1_lattice(7, 12, 7)
    pos = jitter(pos, 0.05, seed=3)
    g, o = make_pair(p, pos, mass, cols)
    g.predict(DT); g.build_grid(); g.compute_rest_density()
    o.predict_positions(DT); o.collision_detection(DT); o.spatial_search(); o.compute_rest_density()
    for it in range(2):
        g.solve(DT, iterations=1)
        o.compute_density(); o.compute_lambdas(); o.compute_correction(DT); o.update_predict_position()
        gd, gC, gl = g.scalars()
        od, oC, ol = o.scalars()
        assert np.array_equal(gd, od), f"density differs at iteration {it}"
        assert np.array_equal(gC, oC)
        assert np.array_equal(gl, ol)
        _, _, gq = g.download()
        _, _, oq, onew = o.download()
        assert np.array_equal(gq, oq) and np.array_equal(gq, onew)
    g.finalize(DT)
    o.apply_solver_results(DT)
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


def test_fast_sqrt_is_correctly_rounded_exhaustively():
    """sqrt_rn (csrc/pbf_device.h) == sqrtf for every non-negative float bit pattern"""
    p, pos, mass, cols = scenes.c1_lattice(2, 2, 2)
    g = PbfContext(p, device=0)
    bad, first = g.selftest_sqrt()
    assert bad == 0, f"{bad} mismatches, first at bit pattern {first:#x}"


# ------------------------------------------------------------------ whole steps

@pytest.mark.parametrize("dims", [(10, 100, 10), (4, 4, 4), (1, 1, 1)])
def test_c1_steps_bit_exact(dims):
    """C1 (the README OpenMP scene, 10x100x10) and tiny/degenerate lattices: 5 sub-steps."""
    p, pos, mass, cols = scenes.c1_lattice(*dims, iterations=2)
    g, o = make_pair(p, pos, mass, cols)
    for _ in range(5):
        g.step(DT)
        o.step(DT)
    g.sync()
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op)
    assert np.array_equal(gv, ov)
    assert g.rest_density() == o.rest_density()


def test_empty_input():
    p, pos, mass, cols = scenes.c1_lattice(2, 2, 2)
    g = PbfContext(p, device=0)
    g.upload(np.zeros((0, 3), np.float32), uniform_mass=mass)
    g.set_colliders(cols)
    g.step(DT)
    g.sync()
    assert g.download()[0].shape == (0, 3)


def test_per_particle_mass_and_velocity():
    p, pos, mass, cols = scenes.c1_lattice(6, 8, 6, iterations=3)
    rng = np.random.default_rng(11)
    masses = rng.uniform(0.05, 0.2, pos.shape[0]).astype(np.float32)
    vel = rng.normal(0, 1.0, pos.shape).astype(np.float32)
    g, o = make_pair(p, pos, mass, cols, vel=vel, masses=masses)
    for _ in range(3):
        g.step(DT); o.step(DT)
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


def test_c2_style_block_bit_exact():
    """h = 0.0415 dam-break block (C2 family, reduced to 20^3), 4 iterations, container
    clamp, fixed rho0, s_corr ~ dt^3: bit-exact with the oracle after 3 sub-steps; a fortiori
    within the 1e-5 relative bar of north_star."""
    p, pos, mass, cols = scenes.dam_break(20, 20, 20, iterations=4)
    g, o = make_pair(p, pos, mass, cols)
    for _ in range(3):
        g.step(DT); o.step(DT)
    g.sync()
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)
    assert rel_inf(gp, op) <= 1e-5 and rel_inf(gv, ov) <= 1e-5


def test_c2_64k_single_step():
    """C2 at full size (40^3 = 64 000 particles, 4 iterations): one sub-step bit-exact."""
    p, pos, mass, cols = scenes.c2_cube()
    g, o = make_pair(p, pos, mass, cols)
    g.step(DT); o.step(DT)
    g.sync()
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


def test_200_steps_statistics_within_1pct():
    """200 sub-steps of the C1 scene (10 000 particles, 2 iterations): mean/max
    density-constraint error and centre of mass within 1 % of the oracle (north_star bar) --
    and, because the arithmetic is restated op for op, the state itself is still bit-exact."""
    p, pos, mass, cols = scenes.c1_lattice(10, 100, 10, iterations=2)
    g, o = make_pair(p, pos, mass, cols)
    for _ in range(200):
        g.step(DT); o.step(DT)
    g.sync()
    gs, os_ = g.stats(), o.stats()
    assert gs.nonfinite == 0 and os_.nonfinite == 0
    assert abs(gs.mean_abs_C - os_.mean_abs_C) <= 0.01 * os_.mean_abs_C
    assert abs(gs.max_abs_C - os_.max_abs_C) <= 0.01 * os_.max_abs_C
    for a in range(3):
        assert abs(gs.com[a] - os_.com[a]) <= 0.01 * max(abs(os_.com[a]), 1.0)
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


def test_xsph_vorticity_bit_exact_vs_own_oracle():
    """a20: no reference implementation (parity unpinned) -- checked against this repo's CPU
    restatement of Macklin & Mueller 2013 eqs. 15-17."""
    p, pos, mass, cols = scenes.dam_break(12, 12, 12, iterations=2, xsph_c=0.01, vorticity_eps=0.05)
    rng = np.random.default_rng(5)
    vel = rng.normal(0, 0.5, pos.shape).astype(np.float32)
    g, o = make_pair(p, pos, mass, cols, vel=vel)
    for _ in range(3):
        g.step(DT); o.step(DT)
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


@pytest.mark.parametrize("xsph,vort", [(0.01, 0.05), (0.0, 0.05), (0.02, 0.0)])
def test_xsph_vorticity_with_per_particle_masses(xsph, vort):
    """the un-packed confinement sweep (per-particle masses: the fourth slot of the gathered
    position carries m_j, so |omega| cannot ride there) and the m_j weighting of vorticity / XSPH"""
    p, pos, mass, cols = scenes.dam_break(11, 12, 10, iterations=2, xsph_c=xsph, vorticity_eps=vort)
    rng = np.random.default_rng(23)
    masses = (mass * rng.uniform(0.6, 1.5, pos.shape[0])).astype(np.float32)
    vel = rng.normal(0, 0.5, pos.shape).astype(np.float32)
    g, o = make_pair(p, pos, mass, cols, vel=vel, masses=masses)
    for _ in range(3):
        g.step(DT); o.step(DT)
    g.sync()
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.isfinite(gp).all() and np.isfinite(gv).all()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


@pytest.mark.parametrize("ny,formats", [(120, {1}), (140, {1, 2}), (160, {2})])
def test_neighbour_rows_switch_format_where_a_cell_plane_outgrows_16_bit_offsets(ny, formats):
    """The neighbour table stores, per 256-slot group, 16-bit slot offsets relative to the particle
    where every candidate of the group lies within +-32767 slots (format 1), 16-bit codes into the
    group's nine candidate windows where not (format 2), and 32-bit slots for groups that touch the
    grid edge (format 0).  A thin, wide block makes one x-plane of cells hold about 24 k / 33 k / 43 k
    particles: all format 1, a mix of 1 and 2, mostly format 2.  Tables IN ORDER and two
    sub-steps must equal the oracle."""
    p, pos, mass, cols = scenes.dam_break(3, ny, ny, iterations=2)
    g, o = make_pair(p, pos, mass, cols)
    g.predict(DT); g.build_grid()
    o.predict_positions(DT); o.collision_detection(DT); o.spatial_search()
    gc, gn = g.neighbors()
    oc, on = o.neighbors(stride=int(p.max_neighbors))
    assert np.array_equal(gc, oc) and np.array_equal(gn, on)
    census = list(g.stats().row_format_particles)
    assert sum(census) == pos.shape[0]
    # format 0: grid-edge groups, if any; the slots at either end of the block see one plane less
    assert all(census[f] for f in formats) and (ny != 120 or not census[2]) and (ny != 160 or census[2] > 2 * census[1]), census
    g, o = make_pair(p, pos, mass, cols)
    for _ in range(2):
        g.step(DT); o.step(DT)
    g.sync()
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


@pytest.mark.parametrize("mask,used", [(B.PBF_ROWS_ABS32, {0}), (B.PBF_ROWS_ABS32 | B.PBF_ROWS_WIN16, {0, 2}),
                                       (B.PBF_ROWS_ALL, {0, 1}),
                                       (B.PBF_ROWS_ALL | B.PBF_ROWS_UNSTAGED, {0, 1}),
                                       (B.PBF_ROWS_ABS32 | B.PBF_ROWS_WIN16 | B.PBF_ROWS_UNSTAGED, {0, 2})])
def test_results_do_not_depend_on_the_neighbour_row_format(mask, used):
    """pbf_set_row_formats: a small scene through 32-bit rows only, through the window-coded 16-bit
    rows only large scenes would pick, through the default choice, and the 16-bit formats again with
    every group built on the neighbour kernel's global path (where groups go whose candidate windows
    exceed the LDS budget) -- the table read back in order, three sub-steps with XSPH and vorticity (every sweep kernel decodes the rows)"""
    p, pos, mass, cols = scenes.dam_break(14, 13, 12, iterations=2, xsph_c=0.01, vorticity_eps=0.05)
    g, o = make_pair(p, pos, mass, cols)
    g.set_row_formats(mask)
    for _ in range(3):
        g.step(DT); o.step(DT)
    g.sync()
    census = list(g.stats().row_format_particles)
    assert sum(census) == pos.shape[0] and {f for f in range(3) if census[f]} <= used and census[max(used)] > 0, census
    gc, gn = g.neighbors()
    oc, on = o.neighbors(stride=int(p.max_neighbors))
    assert np.array_equal(gc, oc) and np.array_equal(gn, on)
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)
    with pytest.raises(PbfError):
        g.set_row_formats(16)


def test_spiky_gradient_option():
    p, pos, mass, cols = scenes.dam_break(10, 10, 10, iterations=2)
    p.gradient_kernel = B.PBF_GRAD_SPIKY
    g, o = make_pair(p, pos, mass, cols)
    for _ in range(2):
        g.step(DT); o.step(DT)
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


def test_neighbor_overflow_in_an_earlier_substep_is_still_reported():
    """rows are cut in the first sub-steps only (an expanding lattice: 6 neighbours at spacing
    0.75, none from spacing 1.25 on, max_neighbors = 4); the run is not synchronised in between --
    pbf_sync and the statistics after the LAST, clean sub-step must still say so"""
    p, pos, mass, cols = scenes.c1_lattice(6, 6, 6, iterations=0)
    p.max_neighbors = 4
    p.gravity[:] = [0, 0, 0]
    g = PbfContext(p, device=0)
    centre = pos.mean(axis=0)
    vel = ((pos - centre) * np.float32(1000.0)).astype(np.float32)      # spacing grows 0.25 per sub-step
    g.upload(pos, vel=vel, uniform_mass=mass)
    for _ in range(6):
        g.step(DT)
    st = g.stats()
    assert st.max_neighbor_count == 0          # the last search was clean ...
    assert st.neighbor_overflow > 0            # ... but rows were cut before
    with pytest.raises(PbfError) as e:
        g.sync()
    assert e.value.code == B.PBF_ERR_NEIGHBOR_OVERFLOW and "earlier sub-steps" in str(e.value)
    g.upload(pos, uniform_mass=mass)           # a new upload starts a clean record
    p2 = g.stats()
    assert p2.neighbor_overflow == 0


def test_a_pile_of_particles_in_one_cell_is_refused_quickly():
    """300 000 particles inside one cell (user error: positions never initialised, a collapsed
    emitter): the rank-by-id pass and the neighbour build cost O(occupancy) per particle, 9e10 steps
    each here.  A cell key beyond max(8 x (max_neighbors + 1), 2^34 / n) particles is refused by the
    scan: both passes are skipped, pbf_sync says why; the ctx works again after the next upload.
    (Small wrap-around grids whose keys alias many cells stay legal: tools/fuzz_parity.py.)"""
    import time
    p, pos, mass, cols = scenes.c1_lattice(6, 6, 6, iterations=2)
    p.capacity = 300000
    g = PbfContext(p, device=0)
    rng = np.random.default_rng(3)
    pile = (np.float32(10.2) + rng.uniform(0, 0.5, (300000, 3))).astype(np.float32)
    g.upload(pile, uniform_mass=mass)
    t0 = time.perf_counter()
    for _ in range(3):
        g.step(DT)
    with pytest.raises(PbfError) as e:
        g.sync()
    assert time.perf_counter() - t0 < 5.0
    assert e.value.code == B.PBF_ERR_NEIGHBOR_OVERFLOW and "a cell key held 300000 particles" in str(e.value)
    assert "limit 57266" in str(e.value)
    # 20 000 in one cell: milliseconds of work, not refused -- the ordinary overflow report
    g.upload(pile[:20000], uniform_mass=mass)
    g.step(DT)
    with pytest.raises(PbfError) as e:
        g.sync()
    assert "exceed max_neighbors" in str(e.value)
    # a dense but legal cell (below the limit) takes the ordinary path: rows overflow, reported as such
    g.upload(pile[:400], uniform_mass=mass)
    g.step(DT)
    with pytest.raises(PbfError) as e:
        g.sync()
    assert "exceed max_neighbors" in str(e.value)
    # and a sane scene runs on the same ctx afterwards, equal to the oracle
    o = pyoracle.Oracle(p)
    g.upload(pos, uniform_mass=mass); o.upload(pos, uniform_mass=mass)
    g.set_colliders(cols); o.set_colliders(cols)
    for _ in range(2):
        g.step(DT); o.step(DT)
    g.sync()
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


def test_neighbor_overflow_is_loud():
    p, pos, mass, cols = scenes.c1_lattice(6, 6, 6)
    p.max_neighbors = 8
    g = PbfContext(p, device=0)
    g.upload(pos, uniform_mass=mass)
    g.step(DT)
    with pytest.raises(PbfError) as e:
        g.sync()
    assert e.value.code == B.PBF_ERR_NEIGHBOR_OVERFLOW


def test_stats_match_oracle():
    p, pos, mass, cols = scenes.c1_lattice(8, 20, 8, iterations=2)
    g, o = make_pair(p, pos, mass, cols)
    for _ in range(4):
        g.step(DT); o.step(DT)
    gs, os_ = g.stats(), o.stats()
    assert gs.n == os_.n and gs.max_neighbor_count == os_.max_neighbor_count
    assert gs.mean_neighbor_count == pytest.approx(os_.mean_neighbor_count, rel=1e-12)
    assert gs.mean_abs_C == pytest.approx(os_.mean_abs_C, rel=1e-9)
    assert gs.max_abs_C == pytest.approx(os_.max_abs_C, rel=1e-12)
    assert list(gs.com) == pytest.approx(list(os_.com), rel=1e-9)
    assert gs.max_speed == pytest.approx(os_.max_speed, rel=1e-9)


def test_wraparound_grid_large_domain():
    """particles spread over more cells than the grid holds: the wrapped grid aliases cells,
    the distance test must still give the exact neighbour sets"""
    p, pos, mass, cols = scenes.c1_lattice(6, 6, 6)
    p.grid_dim[:] = [4, 5, 3]
    pos = jitter(pos * np.float32(1.7), 0.3, seed=21)
    g, o = make_pair(p, pos, mass, [])
    g.build_grid()
    o.spatial_search()
    gc, gn = g.neighbors()
    oc, on = o.neighbors(stride=int(p.max_neighbors))
    assert np.array_equal(gc, oc)
    for i in range(pos.shape[0]):
        assert sorted(gn[i, :gc[i]]) == sorted(on[i, :oc[i]])


# ------------------------------------------------------------------ committed golden fixtures

import json
from pathlib import Path

_GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("name", sorted(p.stem for p in _GOLDEN.glob("*.npz") if not p.stem.startswith("ref_")))
def test_hip_reproduces_golden(name):
    from tests.golden.make_golden import build_case
    z = np.load(_GOLDEN / f"{name}.npz")
    meta = json.loads(str(z["meta"]))
    g, steps = build_case(meta, z["vel0"] if "vel0" in z else None,
                          z["mass0"] if "mass0" in z else None, backend="hip")
    for _ in range(steps):
        g.step(DT)
    g.sync()
    pos, vel, _ = g.download()
    assert np.array_equal(pos, z["pos"]) and np.array_equal(vel, z["vel"])
    assert np.float32(g.rest_density()) == z["rest_density"]


# ------------------------------------------------------------------ BASELINE full size (C3)

def test_c3_1m_particles_two_steps_and_structure():
    """C3 at full size (100^3 = 1 000 000 particles, 4 iterations): cell order, neighbour
    symmetry and the state after two sub-steps, against the oracle on the same inputs."""
    p, pos, mass, cols = scenes.c3_block()
    g, o = make_pair(p, pos, mass, cols)
    g.predict(DT); g.build_grid()
    xyz, key, sid, start = g.cells()
    # sortedness: keys ascend along the sorted order, ids ascend inside a cell
    ks = key[sid]
    assert np.all(np.diff(ks.astype(np.int64)) >= 0)
    same = ks[1:] == ks[:-1]
    assert np.all(sid[1:][same] > sid[:-1][same])
    assert start[-1] == pos.shape[0]
    # neighbour relation is symmetric: every (i, j) has its (j, i)
    cnt, nb = g.neighbors()
    i_idx = np.repeat(np.arange(pos.shape[0], dtype=np.int64), cnt)
    j_idx = nb[np.arange(nb.shape[1])[None, :] < cnt[:, None]].astype(np.int64)
    fwd = np.sort(i_idx * pos.shape[0] + j_idx)
    bwd = np.sort(j_idx * pos.shape[0] + i_idx)
    assert np.array_equal(fwd, bwd)
    # finish this sub-step and run one more; compare with the oracle
    g.solve(DT); g.finalize(DT)
    g.step(DT)
    g.sync()
    o.step(DT); o.step(DT)
    gp, gv, _ = g.download()
    op, ov, _, _ = o.download()
    assert np.array_equal(gp, op) and np.array_equal(gv, ov)


# ------------------------------------------------------------------ 8(f) rank 2: export hand-off

def test_export_positions_is_the_reference_vbo_layout():
    """N packed float3 in particle-index order, written on the device (what the reference
    uploads to its VBO, ParticleSystem.cpp:372-389)"""
    import torch
    p, pos, mass, cols = scenes.dam_break(12, 10, 8, iterations=2)
    g = PbfContext(p)
    g.upload(pos, uniform_mass=mass)
    g.set_colliders(cols)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        g.step(DT)
    vbo = torch.full((pos.shape[0], 3), float("nan"), dtype=torch.float32, device="cuda")
    g.export_positions(vbo.data_ptr())
    torch.cuda.synchronize()
    dpos, _, _ = g.download()
    assert np.array_equal(vbo.cpu().numpy(), dpos)


@pytest.mark.parametrize("scene", ["c1_first_frame_mean", "dam_break_per_particle_mass"])
def test_snapshot_restart_continues_bit_identically(tmp_path, scene):
    if scene == "c1_first_frame_mean":
        p, pos, mass, cols = scenes.c1_lattice(6, 12, 6)
        masses = None
    else:
        p, pos, mass, cols = scenes.dam_break(10, 10, 10, iterations=3)
        masses = (np.float32(mass) * (1 + 0.1 * np.sin(np.arange(pos.shape[0])))).astype(np.float32)

    def fresh():
        g = PbfContext(p)
        g.upload(pos, uniform_mass=mass, mass=masses)
        g.set_colliders(cols)
        return g

    a = fresh()
    for _ in range(4):
        a.step(DT)
    a.snapshot_save(tmp_path / "s.pbf")
    for _ in range(4):
        a.step(DT)
    b = PbfContext(p)
    b.set_colliders(cols)
    b.snapshot_load(tmp_path / "s.pbf")
    assert b.n == pos.shape[0]
    for _ in range(4):
        b.step(DT)
    pa, va, _ = a.download()
    pb, vb, _ = b.download()
    assert np.array_equal(pa, pb) and np.array_equal(va, vb)
    assert a.rest_density() == b.rest_density()
    # a snapshot taken before the first frame keeps "first frame pending"
    c = fresh()
    c.snapshot_save(tmp_path / "t.pbf")
    d = PbfContext(p)
    d.set_colliders(cols)
    d.snapshot_load(tmp_path / "t.pbf")
    c.step(DT)
    d.step(DT)
    assert c.rest_density() == d.rest_density()
    assert np.array_equal(c.download()[0], d.download()[0])


def test_snapshot_load_rejects_garbage(tmp_path):
    from fluidsimulation_amd import PbfError
    p, pos, mass, cols = scenes.dam_break(4, 4, 4)
    g = PbfContext(p)
    (tmp_path / "bad.pbf").write_bytes(b"not a snapshot" * 10)
    with pytest.raises(PbfError):
        g.snapshot_load(tmp_path / "bad.pbf")
    with pytest.raises(PbfError):
        g.snapshot_load(tmp_path / "missing.pbf")


@pytest.mark.gpu
def test_addressing_at_the_capacity_limit():
    """A ctx of 2^28 slots (the largest pbf_create accepts, ~95 GiB) holding a small scene: neighbour rows are
    1 GiB apart, so the sweeps' buffer loads run with scalar row offsets of up to 3 GiB and the builder's
    stores reach row k at k GiB.  Bit-identical with the oracle and with a ctx of the scene's own size
    (tools/check_big_capacity.py is the stand-alone form)."""
    import torch
    if torch.cuda.mem_get_info(0)[0] < 120 * 2**30:
        pytest.skip("needs 120 GiB of free HBM")
    p, pos, mass, cols = scenes.dam_break(24, 20, 16, iterations=4, max_neighbors=48)
    pos = np.ascontiguousarray(pos[np.random.default_rng(5).permutation(pos.shape[0])])
    o = pyoracle.Oracle(p)
    small = PbfContext(p, device=0)
    p.capacity = 1 << 28
    big = PbfContext(p, device=0)
    for c in (o, small, big):
        c.upload(pos, uniform_mass=mass)
        c.set_colliders(cols)
        for _ in range(4):
            c.step(DT)
    bp, bv, _ = big.download()
    sp, sv, _ = small.download()
    op, ov = o.download()[:2]
    assert bp.tobytes() == op.tobytes() and bv.tobytes() == ov.tobytes()
    assert sp.tobytes() == bp.tobytes() and sv.tobytes() == bv.tobytes()
    p.capacity = (1 << 28) + 1
    with pytest.raises(PbfError):
        PbfContext(p, device=0)
