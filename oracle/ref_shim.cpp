// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY, in-container only.
//
// extern "C" doors into the reference's OWN sources, compiled where they lie under
// /root/reference (never copied): SPHKernel.h (header-only), CollisionDetection.cpp and
// Collider.cpp; and, when the image holds NVIDIA's own cuda_runtime.h (it ships inside the
// installed triton wheel; Particle.h:7 includes it for the float3 type) also Particle.cpp,
// Constraints.cpp and ConstraintSolver.cpp (-DREF_WITH_PARTICLES, see oracle/Makefile).
// Every symbol these objects reference is defined by them, libstdc++ or libm: nothing is
// stubbed.  NeighborSearch.cpp / ParticleSystem.cpp / Simulation.cpp stay out: they call into
// libcudart and GLEW, which the image lacks (treated as unbuildable), and use
// MSVC-only library names (std::floorf, steady_clock = high_resolution_clock).  The only
// accommodation is -D__forceinline=inline for the MSVC keyword at Collider.h:26.  Output goes
// to oracle/_ref/ (git-ignored).
#include <cstdint>
#include <memory>
#include <vector>

#include "SPHKernel.h"
#include <glm/gtx/norm.hpp>
#include "CollisionDetection.h"
#ifdef REF_WITH_PARTICLES
#include "Particle.h"
#include "Constraints.h"
#include "ConstraintSolver.h"
#endif

extern "C" {

float ref_poly6_w(float distance, float h) { return SPHKernel::Poly6_W(distance, h); }

void ref_poly6_w_gradient(const float* diff, float distance, float h, float* out) {
    glm::vec3 g = SPHKernel::Poly6_W_Gradient(glm::vec3(diff[0], diff[1], diff[2]), distance, h);
    out[0] = g.x; out[1] = g.y; out[2] = g.z;
}

float ref_spiky_w(float distance, float h) { return SPHKernel::Spiky_W(distance, h); }

void ref_spiky_w_gradient(const float* diff, float distance, float h, float* out) {
    glm::vec3 g = SPHKernel::Spiky_W_Gradient(glm::vec3(diff[0], diff[1], diff[2]), distance, h);
    out[0] = g.x; out[1] = g.y; out[2] = g.z;
}

int ref_point_plane(const float* p, const float* n, float d) {
    PlaneCollider plane(glm::vec3(n[0], n[1], n[2]), d);
    return CollisionDetection::PointPlaneIntersection(glm::vec3(p[0], p[1], p[2]), &plane) ? 1 : 0;
}

int ref_point_aabb(const float* p, const float* mn, const float* mx) {
    AABB box(glm::vec3(mn[0], mn[1], mn[2]), glm::vec3(mx[0], mx[1], mx[2]));
    return CollisionDetection::PointAABBIntersection(glm::vec3(p[0], p[1], p[2]), &box) ? 1 : 0;
}

// glm helpers the step loop leans on, evaluated by the vendored glm itself
float ref_glm_distance(const float* a, const float* b) {
    return glm::distance(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
}

float ref_glm_dot(const float* a, const float* b) {
    return glm::dot(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
}

double ref_glm_pow(double x, int e) { return glm::pow(x, e); }

// glm::distance2 as NeighborSearch.cpp:219 calls it (gtx/norm.hpp, reached through common.h)
float ref_glm_distance2(const float* a, const float* b) {
    return glm::distance2(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
}

// The same leaf functions applied to one particle against an array of others -- each element is
// ONE call of the reference's function; the loops of Simulation.cpp that combine them are not here
// (tests/golden/make_ref_loops.py walks them pair by pair).
void ref_glm_distance2_many(const float* pi, const float* pts, uint32_t n, float* out) {
    const glm::vec3 a(pi[0], pi[1], pi[2]);
    for (uint32_t j = 0; j < n; ++j)
        out[j] = glm::distance2(a, glm::vec3(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]));
}

void ref_glm_distance_many(const float* pi, const float* pts, uint32_t n, float* out) {
    const glm::vec3 a(pi[0], pi[1], pi[2]);
    for (uint32_t j = 0; j < n; ++j)
        out[j] = glm::distance(a, glm::vec3(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]));
}

void ref_poly6_w_many(const float* distance, float h, uint32_t n, float* out) {
    for (uint32_t j = 0; j < n; ++j) out[j] = SPHKernel::Poly6_W(distance[j], h);
}

// diff = p_i - p_j (glm vec3 subtraction, Simulation.cpp:662,694), then Poly6_W_Gradient(diff, d, h)
void ref_poly6_w_gradient_many(const float* pi, const float* pts, const float* distance, float h, uint32_t n,
                               float* out) {
    const glm::vec3 a(pi[0], pi[1], pi[2]);
    for (uint32_t j = 0; j < n; ++j) {
        const glm::vec3 diff = a - glm::vec3(pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]);
        const glm::vec3 g = SPHKernel::Poly6_W_Gradient(diff, distance[j], h);
        out[3 * j] = g.x; out[3 * j + 1] = g.y; out[3 * j + 2] = g.z;
    }
}

// float * vec3 and dot(v, v) of glm, per element: s * v[j], then glm::dot of the product with itself
// (gradientC_j = (1.f / rho0) * grad; dot_value = dot(gradientC_j, gradientC_j), Simulation.cpp:665-667)
void ref_glm_scaled_dot_self_many(float s, const float* v, uint32_t n, float* out) {
    for (uint32_t j = 0; j < n; ++j) {
        const glm::vec3 g = s * glm::vec3(v[3 * j], v[3 * j + 1], v[3 * j + 2]);
        out[j] = glm::dot(g, g);
    }
}

// float * vec3 of glm, per element (result = s[j] * grad[j], Simulation.cpp:704-706)
void ref_glm_scale_many(const float* s, const float* v, uint32_t n, float* out) {
    for (uint32_t j = 0; j < n; ++j) {
        const glm::vec3 g = s[j] * glm::vec3(v[3 * j], v[3 * j + 1], v[3 * j + 2]);
        out[3 * j] = g.x; out[3 * j + 1] = g.y; out[3 * j + 2] = g.z;
    }
}

#ifdef REF_WITH_PARTICLES
int ref_has_particles(void) { return 1; }

static Collider* make_collider(int32_t type, const float* a, const float* b, float d) {
    if (type == (int32_t)Collider::ColliderTypes::PLANE) return new PlaneCollider(glm::vec3(a[0], a[1], a[2]), d);
    if (type == (int32_t)Collider::ColliderTypes::AABB)
        return new AABB(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2]));
    return nullptr;
}

// The particle loop of Simulation::CollisionDetection (Simulation.cpp:731-739) over the
// reference's ParticleSet::TestCollision / OnCollision (Particle.cpp:217-338).
// type/a/b/d: ncol colliders (a, b: 3 floats each).  vel, pred, newp are updated in place.
int ref_particleset_collide(uint32_t n, const float* pos, float* vel, float* pred, float* newp, float dt,
                            uint32_t ncol, const int32_t* type, const float* a, const float* b, const float* d) {
    ParticleSet set(n, 1.0f);
    for (uint32_t i = 0; i < n; ++i) {
        set.m_positions[i] = glm::vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        set.m_velocity[i] = glm::vec3(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]);
        set.m_predict_positions[i] = glm::vec3(pred[3 * i], pred[3 * i + 1], pred[3 * i + 2]);
        set.m_new_positions[i] = glm::vec3(newp[3 * i], newp[3 * i + 1], newp[3 * i + 2]);
    }
    std::vector<Collider*> cols;
    for (uint32_t j = 0; j < ncol; ++j) {
        Collider* c = make_collider(type[j], a + 3 * j, b + 3 * j, d[j]);
        if (!c) return 1;
        cols.push_back(c);
    }
    for (size_t i = 0; i < set.m_size; ++i)
        for (size_t j = 0; j < cols.size(); ++j)
            if (set.TestCollision(i, cols[j])) set.OnCollision(i, cols[j], dt);
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            vel[3 * i + k] = set.m_velocity[i][k];
            pred[3 * i + k] = set.m_predict_positions[i][k];
            newp[3 * i + k] = set.m_new_positions[i][k];
        }
    for (Collider* c : cols) delete c;
    return 0;
}

// ParticleSet::Update (Particle.cpp:207-215): v = (new - p) / dt, p = new
void ref_particleset_update(uint32_t n, float* pos, float* vel, const float* newp, float dt) {
    ParticleSet set(n, 1.0f);
    for (uint32_t i = 0; i < n; ++i) {
        set.m_positions[i] = glm::vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        set.m_new_positions[i] = glm::vec3(newp[3 * i], newp[3 * i + 1], newp[3 * i + 2]);
    }
    set.Update(dt);
    for (uint32_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            pos[3 * i + k] = set.m_positions[i][k];
            vel[3 * i + k] = set.m_velocity[i][k];
        }
}

// ConstraintSolver::SolveConstraints (ConstraintSolver.cpp:12-72) over DistanceConstraint
// objects (Constraints.cpp:50-120) holding the reference's Particle objects.
// mode: 0 = PBD_MODE::PBD, 1 = PBD_MODE::XPBD.  newp (n x 3) is updated in place; lambdas (k) out.
int ref_solve_distance_constraints(int32_t mode, float dt, uint32_t iterations, uint32_t n, float* newp,
                                   const float* mass, uint32_t k, const uint32_t* p0, const uint32_t* p1,
                                   const float* rest, const float* stiffness, const float* compliance,
                                   float* lambdas) {
    std::vector<Particle_Ptr> parts;
    for (uint32_t i = 0; i < n; ++i) {
        const glm::vec3 x(newp[3 * i], newp[3 * i + 1], newp[3 * i + 2]);
        Particle_Ptr p = std::make_shared<Particle>(x, mass[i]);
        p->m_new_position = x;
        parts.push_back(p);
    }
    std::vector<Constraint*> list, none;
    for (uint32_t c = 0; c < k; ++c) {
        if (p0[c] >= n || p1[c] >= n) return 1;
        DistanceConstraint* dc = new DistanceConstraint(parts[p0[c]], parts[p1[c]], rest[c]);
        dc->setStiffness(stiffness[c]);
        dc->setCompliance(compliance[c]);
        list.push_back(dc);
    }
    ConstraintSolver solver(mode == 1 ? PBD_MODE::XPBD : PBD_MODE::PBD);
    solver.setSolverIteration(iterations);
    const bool ok = solver.SolveConstraints(dt, list, none);
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) newp[3 * i + c] = parts[i]->m_new_position[c];
    for (uint32_t c = 0; c < k; ++c) {
        lambdas[c] = list[c]->m_lambda;
        delete static_cast<DistanceConstraint*>(list[c]);
    }
    return ok ? 0 : 2;
}
#else
int ref_has_particles(void) { return 0; }
#endif

}  // extern "C"
