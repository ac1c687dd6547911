"""Writes tests/golden/ref_loops_*.npz: the hot loops of one solver iteration -- neighbour lists,
first-frame rest density, rho_i, C_i, lambda_i and the corrected positions -- recomputed pair by
pair from calls into the reference's OWN compiled code (oracle/_ref: SPHKernel.h, the vendored glm,
Particle.cpp / CollisionDetection.cpp / Collider.cpp, built where they lie under /root/reference by
oracle/Makefile).  Neither the oracle nor the HIP path takes part: the fixtures pin both.

What comes from the compiled reference (one call per element):
    glm::distance2, glm::distance, vec3 - vec3, float * vec3, glm::dot, glm::pow(double, int),
    SPHKernel::Poly6_W, SPHKernel::Poly6_W_Gradient, ParticleSet::TestCollision / OnCollision.
What this script restates (the "glue"; everything else of the loops) -- scalar fp32 / fp64
expressions evaluated with numpy scalars of the same width, each operation rounded once:
    PredictPositions                       Simulation.cpp:589-596   f = m*g; v += dt*w*f; p* = p + dt*v
    Flooring, x/y/z cell loop, bucket order NeighborSearch.cpp:154-234 (a brute-force all-pairs
                                           `distance2 <= h*h` pass, the hits of particle i listed by
                                           (cell dx, dy, dz of the 27-cell loop; particle index) --
                                           what the hash buckets deliver when no two of the 27 cells
                                           alias, which the script asserts)
    rho_i  = m_i W(0) + sum m_j W          Simulation.cpp:285-294, 625-633
    rho0   = (sum_i rho_i) / n             Simulation.cpp:294-297 (serial fp32)
    C_i    = rho_i / rho0 - 1              Simulation.cpp:635
    lambda = -C_i / (sum + 1e-6f)          Simulation.cpp:656-675
    scorr  = -0.1 * pow(W/W(0.3h), 4) * dt Simulation.cpp:698-702 (double)
    new_i += (1/rho0) * (l_i + l_j + (float)scorr) * gradW     Simulation.cpp:704-708

Runs only in the container that holds /root/reference; the fixtures are data and travel.

    make -C oracle && python -m tests.golden.make_ref_loops
"""
from __future__ import annotations

import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from oracle import pyoracle  # noqa: E402  (only its loader of oracle/_ref is used)

HERE = Path(__file__).resolve().parent
F = np.float32
DT = F(0.0005)          # GLFWApp.cpp:393
H = F(1.0)              # effective_radius = 1.f, Simulation.cpp:88
SPACING = F(1.0)        # m_grid_spacing, NeighborSearch.cpp:17
GRAVITY = F(-9.8)       # Simulation.cpp:16
MASS = F(0.1)           # GLFWApp.cpp:451
EPSILON = F(1.0e-6)     # Simulation.cpp:644
# the five plane colliders of the shipped scene (GLFWApp.cpp:177-191, Plane.cpp:49)
PLANES = [((0, 1, 0), 0.0), ((1, 0, 0), -10.0), ((-1, 0, 0), -10.0), ((0, 0, 1), -15.0), ((0, 0, -1), 1.0)]


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def lattice(nx, ny, nz):
    """GLFWApp::GenerateFluidParticles (GLFWApp.cpp:443-458), index k + nz*j + nz*ny*i"""
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    x = F(0.0) + F(0.5) * i.astype(F)
    y = F(0.5) + F(0.5) * j.astype(F)
    z = F(-10.0) + F(0.5) * k.astype(F)
    return np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(F)


def predict(ref, pos, vel):
    """PredictPositions (glue) + the particle loop of CollisionDetection (compiled reference)"""
    n = pos.shape[0]
    w = F(1.0) / MASS                                   # m_massInv, Particle.cpp:184
    f = np.array([MASS * F(0.0), MASS * GRAVITY, MASS * F(0.0)], F)   # m_mass * vec3(0, g, 0)
    s = DT * w
    v = (vel + s * f[None, :]).astype(F)                # v + dt * w * f  (fp32, one rounding per op)
    pred = (pos + DT * v).astype(F)
    new = pred.copy()
    ctype = np.full(len(PLANES), 4, np.int32)           # Collider::ColliderTypes::PLANE
    a = np.array([p[0] for p in PLANES], F)
    b = np.zeros_like(a)
    d = np.array([p[1] for p in PLANES], F)
    rc = ref.ref_particleset_collide(n, fp(pos), fp(v), fp(pred), fp(new), float(DT), len(PLANES),
                                     ctype.ctypes.data_as(C.POINTER(C.c_int32)), fp(a), fp(b), fp(d))
    assert rc == 0 and np.array_equal(pred, new)
    return v, pred


def neighbour_lists(ref, pred):
    """all pairs with glm::distance2 <= h*h, per particle in the order of the 27-cell loop"""
    n = pred.shape[0]
    cell = np.floor(pred / SPACING).astype(np.int32)    # Flooring, NeighborSearch.cpp:243-252
    # the hash buckets deliver exactly the 27 cells' occupants unless two of them alias
    key = (np.int64(73856093) * cell[:, 0] + np.int64(19349663) * cell[:, 1] + np.int64(83492791) * cell[:, 2]) \
        & 0xFFFFFFFF
    cells_u, first = np.unique(cell, axis=0, return_index=True)
    assert len(np.unique(key[first])) == len(cells_u), "two occupied cells share a hash value"
    square_h = F(H * H)
    d2 = np.empty(n, F)
    lists = []
    for i in range(n):
        ref.ref_glm_distance2_many(fp(pred[i]), fp(pred), n, fp(d2))
        hit = np.flatnonzero(d2 <= square_h)
        hit = hit[hit != i]
        off = cell[hit] - cell[i]
        assert np.abs(off).max(initial=0) <= 1, "a neighbour outside the 27 cells (h <= spacing?)"
        # x outermost, z innermost, bucket (= index) order inside a cell
        order = np.lexsort((hit, off[:, 2], off[:, 1], off[:, 0]))
        lists.append(hit[order].astype(np.uint32))
    return cell, lists


def density(ref, x, mass, lists, w0):
    """m_i W(0) + sum_j m_j W(|x_i - x_j|), accumulated in list order in fp32"""
    n = x.shape[0]
    out = np.empty(n, F)
    for i in range(n):
        nb = lists[i]
        k = len(nb)
        pts = np.ascontiguousarray(x[nb])
        dist, w = np.empty(k, F), np.empty(k, F)
        ref.ref_glm_distance_many(fp(x[i]), fp(pts), k, fp(dist))
        ref.ref_poly6_w_many(fp(dist), float(H), k, fp(w))
        acc = F(mass[i] * w0)
        for j in range(k):
            acc = F(acc + F(mass[nb[j]] * w[j]))
        out[i] = acc
    return out


def lambdas(ref, pred, Cc, lists, rho0):
    n = pred.shape[0]
    out = np.empty(n, F)
    inv = F(F(1.0) / rho0)
    zero = np.zeros((1, 3), F)
    g0, s0 = np.empty((1, 3), F), np.empty(1, F)
    ref.ref_poly6_w_gradient_many(fp(zero[0]), fp(zero), fp(np.zeros(1, F)), float(H), 1, fp(g0))
    ref.ref_glm_scaled_dot_self_many(float(inv), fp(g0), 1, fp(s0))     # gradientC_i term, :657-658
    for i in range(n):
        nb = lists[i]
        k = len(nb)
        pts = np.ascontiguousarray(pred[nb])
        dist, grad, dots = np.empty(k, F), np.empty((k, 3), F), np.empty(k, F)
        ref.ref_glm_distance_many(fp(pred[i]), fp(pts), k, fp(dist))
        ref.ref_poly6_w_gradient_many(fp(pred[i]), fp(pts), fp(dist), float(H), k, fp(grad))
        ref.ref_glm_scaled_dot_self_many(float(inv), fp(grad), k, fp(dots))
        acc = F(s0[0])
        for j in range(k):
            acc = F(acc + dots[j])
        lam = F(-Cc[i])
        out[i] = F(lam / F(acc + EPSILON))
    return out


def correction(ref, pred, lam, lists, rho0):
    n = pred.shape[0]
    new = pred.copy()
    inv = F(F(1.0) / rho0)
    w_dq = F(ref.ref_poly6_w(float(F(F(0.3) * H)), float(H)))
    for i in range(n):
        nb = lists[i]
        k = len(nb)
        pts = np.ascontiguousarray(pred[nb])
        dist, w, grad, res = np.empty(k, F), np.empty(k, F), np.empty((k, 3), F), np.empty((k, 3), F)
        ref.ref_glm_distance_many(fp(pred[i]), fp(pts), k, fp(dist))
        ref.ref_poly6_w_many(fp(dist), float(H), k, fp(w))
        ref.ref_poly6_w_gradient_many(fp(pred[i]), fp(pts), fp(dist), float(H), k, fp(grad))
        s = np.empty(k, F)
        for j in range(k):
            x = float(F(w[j] / w_dq))                       # float / float, promoted to double
            x = ref.ref_glm_pow(x, 4)
            scorr = (-0.1 * x) * float(DT)                  # double
            s[j] = F(inv * F(F(lam[i] + lam[nb[j]]) + F(scorr)))
        ref.ref_glm_scale_many(fp(s), fp(grad), k, fp(res))
        acc = new[i].copy()
        for j in range(k):
            acc = (acc + res[j]).astype(F)                  # m_new_positions[i] += result
        new[i] = acc
    return new


def case(ref, pos, vel):
    n = pos.shape[0]
    mass = np.full(n, MASS, F)
    v, pred = predict(ref, pos, vel)
    cell, lists = neighbour_lists(ref, pred)
    w0 = F(ref.ref_poly6_w(0.0, float(H)))
    # first frame: rest density on the CURRENT positions with the lists of the predicted ones
    rho_first = density(ref, pos, mass, lists, w0)
    acc = F(0.0)
    for i in range(n):
        acc = F(acc + rho_first[i])
    rho0 = F(acc / F(n))
    rho = density(ref, pred, mass, lists, w0)
    Cc = (rho / rho0 - F(1.0)).astype(F)
    lam = lambdas(ref, pred, Cc, lists, rho0)
    new = correction(ref, pred, lam, lists, rho0)
    counts = np.array([len(l) for l in lists], np.uint32)
    table = np.zeros((n, int(counts.max())), np.uint32)
    for i, l in enumerate(lists):
        table[i, :len(l)] = l
    assert np.isfinite(new).all() and (new != pred).any()
    return dict(pos=pos, vel=vel, dt=DT, h=H, mass=MASS, vel_predicted=v, pred=pred, cell=cell,
                neighbor_counts=counts, neighbors=table, rest_density=rho0, density=rho, C=Cc,
                lambdas=lam, new_positions=new)


def main():
    ref = pyoracle.load_ref()
    if ref is None or not ref.ref_has_particles() or not hasattr(ref, "ref_glm_distance2_many"):
        raise SystemExit("oracle/_ref is needed: make -C oracle (in the container that holds /root/reference)")
    # (a) the first sub-step of the full C1 scene (10 x 100 x 10 lattice at rest)
    pos = lattice(10, 100, 10)
    np.savez_compressed(HERE / "ref_loops_c1_first_substep.npz", **case(ref, pos, np.zeros_like(pos)))
    # (b) an 8 x 16 x 8 lattice, jittered by up to a fifth of the spacing, with random velocities
    rng = np.random.default_rng(2024)
    pos = (lattice(8, 16, 8) + rng.uniform(-0.1, 0.1, (8 * 16 * 8, 3))).astype(F)
    vel = rng.normal(0.0, 2.0, pos.shape).astype(F)
    np.savez_compressed(HERE / "ref_loops_jitter_8x16x8.npz", **case(ref, pos, vel))
    print("wrote ref_loops_c1_first_substep.npz, ref_loops_jitter_8x16x8.npz")


if __name__ == "__main__":
    main()
