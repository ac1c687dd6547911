"""Synthetic scenes of BASELINE.json's configs (all deterministic, fp32).

Each generator restates the reference's own scene code and returns
(params, pos[n,3] float32, uniform_mass, colliders).  No RNG: the jitter in
GenerateFluidCube is dead code (reference Simulation.cpp:558-565).
"""
from __future__ import annotations

import numpy as np

from .binding import PBF_RHO0_FIXED, default_params, plane

F = np.float32
REF_PI = F(3.1415926)  # reference common.h:8


def sim_params_constants():
    """Simulation::SetupSimParams derived constants (Simulation.cpp:315-326), in fp32:
    h = 0.041528303, r = 0.012331058."""
    mass = F(0.015)
    n_kernel = F(20.0)
    rho0 = F(1000.0)
    volume = F(n_kernel * mass) / rho0
    h = np.power(F(F(3.0) * volume) / F(F(4.0) * REF_PI), F(1.0) / F(3.0), dtype=F)
    r = F(np.power(REF_PI / F(F(6.0) * n_kernel), F(1.0) / F(3.0), dtype=F) * h)
    return F(h), F(r), F(volume), mass, rho0


def reference_planes():
    """The five plane colliders of the shipped scene (GLFWApp.cpp:177-191, Plane.cpp:49)."""
    return [plane((0, 1, 0), 0.0), plane((1, 0, 0), -10.0), plane((-1, 0, 0), -10.0),
            plane((0, 0, 1), -15.0), plane((0, 0, -1), 1.0)]


def c1_lattice(nx=10, ny=100, nz=10, iterations=2):
    """C1: GLFWApp::GenerateFluidParticles (GLFWApp.cpp:437-465): nx*ny*nz lattice at spacing
    0.5, index k + nz*j + nz*ny*i, mass 0.1; h = 1, spacing-1 hash grid, five planes,
    first-frame rest density, eps 1e-6, s_corr ~ dt^1.  The full scene is 10x100x10."""
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    x = F(0.0) + F(0.5) * i.astype(F)
    y = F(0.5) + F(0.5) * j.astype(F)
    z = F(-10.0) + F(0.5) * k.astype(F)
    pos = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(F)
    p = default_params()
    p.solver_iterations = iterations  # SetSolverIteration(2), GLFWApp.cpp:216
    p.capacity = pos.shape[0]
    return p, pos, 0.1, reference_planes()


def tank_planes(lo, hi):
    """Open-top tank as reference-style planes: hit iff dot(n, p) <= d."""
    return [plane((0, 1, 0), lo[1]), plane((1, 0, 0), lo[0]), plane((-1, 0, 0), -hi[0]),
            plane((0, 0, 1), lo[2]), plane((0, 0, -1), -hi[2])]


def _grid_dim_for(extent_cells):
    """cells along one axis: the extent plus one spare cell on either side (no fluid cell on a
    grid face) and one for the partial last cell, rounded up to a multiple of 8.  The key wraps
    with %, so no power of two is needed and the per-step scan only covers what the tank spans."""
    return max(16, (int(extent_cells) + 3 + 7) // 8 * 8)


def dam_break(nx, ny, nz, iterations=4, tank_x_factor=2.5, xsph_c=0.0, vorticity_eps=0.0,
              origin=(0.0, 0.0, 0.0), max_neighbors=96):
    """C2-C5 family: nx*ny*nz block at the SetupSimParams spacing d = 2r filling the low-x end
    of an open-top tank tank_x_factor times longer in x.  h = 0.041528303,
    mass 0.015, rho0 = 1000 fixed, cell = h, s_corr ~ dt^3 as on the reference's GPU path
    (cuda_simulation.cu:973); container clamp + planes keep the block in the tank.
    These are build-side choices layered on the oracle (parity vs the oracle only)."""
    h, r, _vol, mass, rho0 = sim_params_constants()
    d = F(F(2.0) * r)
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    half = F(0.5) * d
    x = F(origin[0]) + half + d * i.astype(F)
    y = F(origin[1]) + half + d * j.astype(F)
    z = F(origin[2]) + half + d * k.astype(F)
    pos = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(F)
    lo = np.array(origin, dtype=F)
    hi = lo + np.array([F(tank_x_factor) * d * F(nx), F(4.0) * d * F(ny), d * F(nz)], dtype=F)
    p = default_params()
    p.capacity = pos.shape[0]
    p.max_neighbors = max_neighbors
    p.effective_radius = float(h)
    p.grid_spacing = float(h)
    ext = (hi - lo) / h
    p.grid_dim[:] = [_grid_dim_for(int(ext[0])), _grid_dim_for(int(min(ext[1], 2.0 * ny * d / h))),
                     _grid_dim_for(int(ext[2]))]
    # one spare cell below the tank on every axis: no fluid cell sits on a grid face, where
    # neighbour keys wrap and the neighbour build leaves its LDS-staged path
    c0 = np.floor(lo / h).astype(int)
    p.grid_origin_cell[:] = [int(c0[0]) - 1, int(c0[1]) - 1, int(c0[2]) - 1]
    p.epsilon = 100.0                 # SimParams.epsilon, Simulation.cpp:337
    p.scorr_dt_power = 3              # cuda_simulation.cu:973
    p.rest_density_mode = PBF_RHO0_FIXED
    p.rest_density = float(rho0)      # Simulation.cpp:318
    p.solver_iterations = iterations
    p.container_enabled = 1
    p.container_min[:] = [float(v) for v in lo]
    p.container_max[:] = [float(hi[0]), 1.0e9, float(hi[2])]
    p.xsph_c = xsph_c
    p.vorticity_eps = vorticity_eps
    return p, pos, float(mass), tank_planes(lo, hi)


def c2_cube(iterations=4, **kw):
    """C2: the README CUDA scene's particle count (40^3 = 64 000, Simulation.cpp:528-576)."""
    return dam_break(40, 40, 40, iterations=iterations, **kw)


def c3_block(iterations=4, **kw):
    """C3: 100^3 = 1 000 000 particles."""
    return dam_break(100, 100, 100, iterations=iterations, **kw)


def fluid_cube_reference():
    """Simulation::GenerateFluidCube (Simulation.cpp:528-576) exactly: 8*20^3 = 64 000 particles,
    lattice i,j,k in [-n,n), centre (0,0.75,0), k fastest."""
    _h, r, _v, mass, _rho = sim_params_constants()
    d = F(F(2.0) * r)
    n = int(F(0.5) / d)
    rng = np.arange(-n, n)
    i, j, k = np.meshgrid(rng, rng, rng, indexing="ij")
    x = F(0.0) + d * i.astype(F)
    y = F(0.75) + d * j.astype(F)
    z = F(0.0) + d * k.astype(F)
    return np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(F), float(mass)


def boundary_particles_reference():
    """Simulation::InitializeBoundaryParticles (Simulation.cpp:356-490) exactly: one-layer particle
    walls -- floor 80x80 at y = 0, left/right walls one diameter outside the floor's x extent,
    back/front walls one diameter outside the side walls' z extent -- 32 000 particles.  The
    reference's CPU step never reads them (only its CUDA kernels do), so the solver does not
    either; the generator exists so the scene data of row a5 is reproducible."""
    _h, r, _v, mass, _rho = sim_params_constants()
    d = F(F(2.0) * r)
    thickness = 1
    out = []
    # bottom: half_extend1 = (1, d*thickness, 1)
    nx, nz = int(F(1.0) / d), int(F(1.0) / d)
    i, k = np.meshgrid(np.arange(-nx, nx), np.arange(-nz, nz), indexing="ij")
    x = F(0.0) + d * i.astype(F)
    z = F(0.0) + d * k.astype(F)
    y = np.full_like(x, F(0.0) - d * F(0))
    out.append(np.stack([x, y, z], -1).reshape(-1, 3))
    left_margin = F(x.min() - d)
    right_margin = F(x.max() + d)
    # left & right: half_extend2 = (d*thickness, 1, 1); origins (margin, ny*d, 0); interleaved
    ny, nz = int(F(1.0) / d), int(F(1.0) / d)
    j, k = np.meshgrid(np.arange(-ny, ny), np.arange(-nz, nz), indexing="ij")
    yy = d * j.astype(F) + F(F(ny) * d)
    zz = d * k.astype(F) + F(0.0)
    xl = np.full_like(yy, F(d * F(0)) + left_margin)
    xr = np.full_like(yy, F(d * F(0)) + right_margin)
    lr = np.stack([np.stack([xl, yy, zz], -1), np.stack([xr, yy, zz], -1)], axis=2)  # left, right, ...
    out.append(lr.reshape(-1, 3))
    back_margin = F(zz.min() - d)
    front_margin = F(zz.max() + d)
    # back & front: half_extend3 = (1, 1, d*thickness); origins (0, ny*d, margin); interleaved
    nx, ny = int(F(1.0) / d), int(F(1.0) / d)
    i, j = np.meshgrid(np.arange(-nx, nx), np.arange(-ny, ny), indexing="ij")
    xx = d * i.astype(F) + F(0.0)
    yy = d * j.astype(F) + F(F(ny) * d)
    zb = np.full_like(xx, F(d * F(0)) + back_margin)
    zf = np.full_like(xx, F(d * F(0)) + front_margin)
    bf = np.stack([np.stack([xx, yy, zb], -1), np.stack([xx, yy, zf], -1)], axis=2)
    out.append(bf.reshape(-1, 3))
    assert thickness == 1
    return np.concatenate(out, axis=0).astype(F), float(mass)


def box_boundary_particles(lo, hi, d, top=False):
    """One-layer particle walls around the box [lo, hi]: floor, four side
    walls (and optionally a lid) on planes half a spacing outside the box, lattice spacing d.
    Edge/corner columns belong to one wall only (no duplicates)."""
    lo = np.asarray(lo, F)
    hi = np.asarray(hi, F)
    d = F(d)
    half = F(0.5) * d

    def axis(a, b):
        n = int(np.ceil(float((b - a) / d))) + 2          # one extra column on each side
        return (a - half) + d * np.arange(n, dtype=F)

    xs, ys, zs = axis(lo[0], hi[0]), axis(lo[1], hi[1]), axis(lo[2], hi[2])
    out = []
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    out.append(np.stack([X, np.full_like(X, lo[1] - half), Z], -1).reshape(-1, 3))      # floor
    if top:
        out.append(np.stack([X, np.full_like(X, ys[-1]), Z], -1).reshape(-1, 3))
    yy = ys[1:-1] if top else ys[1:]
    Y, Z = np.meshgrid(yy, zs, indexing="ij")
    out.append(np.stack([np.full_like(Y, xs[0]), Y, Z], -1).reshape(-1, 3))             # x-
    out.append(np.stack([np.full_like(Y, xs[-1]), Y, Z], -1).reshape(-1, 3))            # x+
    X, Y = np.meshgrid(xs[1:-1], yy, indexing="ij")
    out.append(np.stack([X, Y, np.full_like(X, zs[0])], -1).reshape(-1, 3))             # z-
    out.append(np.stack([X, Y, np.full_like(X, zs[-1])], -1).reshape(-1, 3))            # z+
    return np.concatenate(out, 0).astype(F)


def boundary_tank(nx, ny, nz, iterations=4, tank_x_factor=2.0, wall_height_factor=1.5, **kw):
    """dam_break's fluid block held by PARTICLE walls instead of planes/clamp: returns
    (params, pos, mass, colliders=[], boundary_pos, boundary_mass).  Parameters as the
    reference's CUDA scene (h = 0.0415, rho0 = 1000, eps = 100, s_corr ~ dt^3)."""
    p, pos, mass, _cols = dam_break(nx, ny, nz, iterations=iterations, tank_x_factor=tank_x_factor, **kw)
    h, r, _v, _m, _rho = sim_params_constants()
    d = F(F(2.0) * r)
    lo = np.zeros(3, F)
    hi = np.array([F(tank_x_factor) * d * F(nx), F(wall_height_factor) * d * F(ny), d * F(nz)], F)
    bpos = box_boundary_particles(lo, hi, d)
    p.container_enabled = 0
    # the walls sit half a spacing outside the box: two spare cells below
    c0 = np.floor((lo - d) / h).astype(int)
    p.grid_origin_cell[:] = [int(c0[0]) - 1, int(c0[1]) - 1, int(c0[2]) - 1]
    return p, pos, float(mass), [], bpos, float(mass)


def c2_reference_cuda_scene(iterations=4):
    """The reference's README CUDA scene (C2): GenerateFluidCube's 64 000 particles above
    InitializeBoundaryParticles' 32 000-particle open box (Simulation.cpp:356-576), SimParams of
    Simulation.cpp:312-354; no plane colliders."""
    h, r, _v, mass, rho0 = sim_params_constants()
    pos, m = fluid_cube_reference()
    bpos, bm = boundary_particles_reference()
    p = default_params()
    p.capacity = pos.shape[0]
    p.max_neighbors = 96
    p.effective_radius = float(h)
    p.grid_spacing = float(h)
    p.grid_dim[:] = [64, 64, 64]                       # Simulation.cpp:46
    c0 = np.floor(bpos.min(axis=0) / h).astype(int)
    p.grid_origin_cell[:] = [int(c0[0]) - 1, int(c0[1]) - 1, int(c0[2]) - 1]
    p.epsilon = 100.0
    p.scorr_dt_power = 3
    p.rest_density_mode = PBF_RHO0_FIXED
    p.rest_density = float(rho0)
    p.solver_iterations = iterations
    return p, pos, float(m), [], bpos, float(bm)
