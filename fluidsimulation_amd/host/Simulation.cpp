// Simulation.cpp -- host facade over the C ABI (include/pbf.h).  No HIP here: the facade is
// plain C++17 that a maintainer of the reference app could link in place of its own
// Simulation / ParticleSystem / NeighborSearch / ConstraintSolver translation units.
#include "Simulation.h"

#include <cstring>

// ---------------------------------------------------------------- ParticleSystem

ParticleSystem::~ParticleSystem() { Release(); }

ParticleSet* ParticleSystem::AllocateParticles(size_t n, float particle_mass) {
    m_particles = std::make_unique<ParticleSet>(n, particle_mass);
    return m_particles.get();
}

ParticleSet* ParticleSystem::AllocateBoundaryParticles(size_t n, float particle_mass) {
    m_boundary_particles = std::make_unique<ParticleSet>(n, particle_mass);
    return m_boundary_particles.get();
}

bool ParticleSystem::InitializeDevice(const pbf_params& params, int device) {
    if (!m_particles) {
        m_error = "AllocateParticles has not been called";
        return false;
    }
    Release();
    pbf_params p = params;
    if (p.capacity < m_particles->m_size) p.capacity = (uint32_t)m_particles->m_size;
    if (p.capacity == 0) p.capacity = 1;
    int rc = pbf_create(&p, device, &m_ctx);
    if (rc != PBF_OK) {
        m_error = pbf_last_error(nullptr);
        m_ctx = nullptr;
        return false;
    }
    ParticleSet& s = *m_particles;
    // uniform mass is the only case the reference constructs (Particle.cpp:183); keep the
    // per-particle array path for callers that edited m_mass
    bool uniform = true;
    for (size_t i = 1; i < s.m_size; ++i)
        if (s.m_mass[i] != s.m_mass[0]) { uniform = false; break; }
    rc = pbf_upload(m_ctx, s.m_size ? &s.m_positions[0].x : nullptr,
                    s.m_size ? &s.m_velocity[0].x : nullptr,
                    uniform ? nullptr : s.m_mass.data(), s.m_size ? s.m_mass[0] : 1.f,
                    (uint32_t)s.m_size);
    if (rc != PBF_OK) {
        m_error = pbf_last_error(m_ctx);
        return false;
    }
    if (m_boundary_particles && m_boundary_particles->m_size) {
        ParticleSet& b = *m_boundary_particles;
        bool b_uniform = true;
        for (size_t i = 1; i < b.m_size; ++i)
            if (b.m_mass[i] != b.m_mass[0]) { b_uniform = false; break; }
        rc = pbf_set_boundary(m_ctx, &b.m_positions[0].x, b_uniform ? nullptr : b.m_mass.data(),
                              b.m_mass[0], (uint32_t)b.m_size, 0);
        if (rc != PBF_OK) {
            m_error = pbf_last_error(m_ctx);
            return false;
        }
        pbf_get_boundary(m_ctx, b.m_volume.data(), nullptr, nullptr);
    }
    return true;
}

bool ParticleSystem::Update() {
    if (!m_ctx || !m_particles) return false;
    ParticleSet& s = *m_particles;
    if (!s.m_size) return true;
    int rc = pbf_download(m_ctx, &s.m_positions[0].x, &s.m_velocity[0].x, &s.m_new_positions[0].x);
    if (rc == PBF_OK) {
        s.m_predict_positions = s.m_new_positions;  // UpdatePredictPosition, Simulation.cpp:714-721
        rc = pbf_get_scalars(m_ctx, s.m_density.data(), s.m_C.data(), s.m_lambda.data());
    }
    if (rc == PBF_OK && m_boundary_particles && m_boundary_particles->m_size) {
        ParticleSet& b = *m_boundary_particles;
        rc = pbf_get_boundary(m_ctx, b.m_volume.data(), b.m_density.data(), b.m_lambda.data());
    }
    if (rc != PBF_OK) m_error = pbf_last_error(m_ctx);
    return rc == PBF_OK;
}

void ParticleSystem::Release() {
    if (m_ctx) {
        pbf_destroy(m_ctx);
        m_ctx = nullptr;
    }
}

// ---------------------------------------------------------------- NeighborSearch

NeighborSearch::NeighborSearch(std::shared_ptr<ParticleSystem> particle_system, GridSize grid_size)
    : m_grid_size(grid_size),
      m_num_grid_cells(grid_size.x * grid_size.y * grid_size.z),
      m_particle_system(std::move(particle_system)) {}

void NeighborSearch::Initialize() {
    ParticleSet* p = m_particle_system ? m_particle_system->getParticles() : nullptr;
    m_search_cache.assign(p ? p->m_size : 0, std::vector<size_t>());
    m_cache_valid = false;
}

bool NeighborSearch::SpatialSearch(float /*effective_radius*/) {
    pbf_ctx* c = m_particle_system ? m_particle_system->context() : nullptr;
    if (!c) return false;
    m_cache_valid = false;
    return pbf_build_grid(c) == PBF_OK;
}

bool NeighborSearch::refresh_cache() {
    pbf_ctx* c = m_particle_system ? m_particle_system->context() : nullptr;
    if (!c) return false;
    const uint32_t n = pbf_num_particles(c);
    m_search_cache.assign(n, std::vector<size_t>());
    if (!n) return true;
    // stride: the table bound the ctx was created with is not visible here; ask for the widest
    // the stats report, then download
    pbf_stats st;
    if (pbf_get_stats(c, &st) != PBF_OK) return false;
    const uint32_t stride = st.max_neighbor_count ? st.max_neighbor_count : 1;
    std::vector<uint32_t> counts(n), nb((size_t)n * stride);
    if (pbf_get_neighbors(c, counts.data(), nb.data(), stride) != PBF_OK) return false;
    for (uint32_t i = 0; i < n; ++i)
        m_search_cache[i].assign(nb.begin() + (size_t)i * stride,
                                 nb.begin() + (size_t)i * stride + counts[i]);
    m_cache_valid = true;
    return true;
}

const std::vector<size_t>& NeighborSearch::FetchNeighbors(size_t i) {
    static const std::vector<size_t> empty;
    if (!m_cache_valid && !refresh_cache()) return empty;
    return i < m_search_cache.size() ? m_search_cache[i] : empty;
}

// ---------------------------------------------------------------- Simulation

Simulation::Simulation() : m_world_desc(-9.8f, 0.f) {
    pbf_default_params(&m_params);
}

bool Simulation::Initialize(PBD_MODE mode, std::shared_ptr<ParticleSystem> particle_system, int device) {
    m_particle_system = std::move(particle_system);
    if (!m_particle_system) {
        m_error = "null particle system";
        return false;
    }
    GridSize gs{(uint32_t)m_params.grid_dim[0], (uint32_t)m_params.grid_dim[1], (uint32_t)m_params.grid_dim[2]};
    m_neighbor_searcher = std::make_shared<NeighborSearch>(m_particle_system, gs);
    if (!m_solver) m_solver = std::make_shared<ConstraintSolver>(mode);
    else m_solver->setPBDMode(mode);
    m_params.gravity[0] = 0.f;
    m_params.gravity[1] = m_world_desc.gravity;  // vec3(0, gravity, 0), Simulation.cpp:592
    m_params.gravity[2] = 0.f;
    m_params.solver_iterations = (int32_t)m_solver->getSolverIteration();
    if (!m_particle_system->InitializeDevice(m_params, device)) {
        m_error = m_particle_system->error();
        return false;
    }
    m_neighbor_searcher->Initialize();
    if (!push_colliders()) return false;
    if (!push_constraints()) return false;
    m_initialized = true;
    return true;
}

bool Simulation::push_colliders() {
    pbf_ctx* c = m_particle_system ? m_particle_system->context() : nullptr;
    if (!c) return true;  // pushed at Initialize
    std::vector<pbf_collider> cols;
    for (Collider* col : m_colliders) {
        pbf_collider pc;
        std::memset(&pc, 0, sizeof(pc));
        if (col->getColliderTypes() == Collider::ColliderTypes::PLANE) {
            auto* pl = static_cast<PlaneCollider*>(col);
            pc.type = PBF_COLLIDER_PLANE;
            pc.a[0] = pl->m_normal.x; pc.a[1] = pl->m_normal.y; pc.a[2] = pl->m_normal.z;
            pc.d = pl->m_d;
        } else {
            auto* bx = static_cast<AABB*>(col);
            pc.type = PBF_COLLIDER_AABB;
            pc.a[0] = bx->m_min.x; pc.a[1] = bx->m_min.y; pc.a[2] = bx->m_min.z;
            pc.b[0] = bx->m_max.x; pc.b[1] = bx->m_max.y; pc.b[2] = bx->m_max.z;
        }
        cols.push_back(pc);
    }
    if (pbf_set_colliders(c, cols.data(), (uint32_t)cols.size()) != PBF_OK) {
        m_error = pbf_last_error(c);
        return false;
    }
    return true;
}

bool Simulation::push_constraints() {
    pbf_ctx* c = m_particle_system ? m_particle_system->context() : nullptr;
    if (!c) return true;  // pushed at Initialize
    std::vector<pbf_distance_constraint> list;
    for (Constraint* k : m_static_constraints)
        list.push_back(static_cast<DistanceConstraint*>(k)->pod());
    const int mode = (m_solver && m_solver->getPBDMode() == PBD_MODE::PBD) ? PBF_CONSTRAINTS_PBD : PBF_CONSTRAINTS_XPBD;
    if (pbf_set_distance_constraints(c, list.data(), (uint32_t)list.size(), mode) != PBF_OK) {
        m_error = pbf_last_error(c);
        return false;
    }
    return true;
}

bool Simulation::AddStaticConstraint(Constraint* constraint) {
    if (!constraint || constraint->getConstraintType() != CONSTRAINT_TYPE::CONSTRAINT_DISTANCE) {
        m_error = "only DistanceConstraint is implemented (BendConstraint is a stub in the reference)";
        return false;
    }
    m_static_constraints.push_back(constraint);
    return push_constraints();
}

bool Simulation::AddStaticConstraints(const std::vector<Constraint*>& constraints) {
    for (Constraint* k : constraints)
        if (!k || k->getConstraintType() != CONSTRAINT_TYPE::CONSTRAINT_DISTANCE) {
            m_error = "only DistanceConstraint is implemented (BendConstraint is a stub in the reference)";
            return false;
        }
    m_static_constraints.insert(m_static_constraints.end(), constraints.begin(), constraints.end());
    return push_constraints();
}

bool Simulation::AddCollider(Collider* collider) {
    if (!collider) return false;
    auto t = collider->getColliderTypes();
    if (t != Collider::ColliderTypes::PLANE && t != Collider::ColliderTypes::AABB) {
        m_error = "only PLANE and AABB colliders have a defined particle response";
        return false;
    }
    if (m_colliders.size() >= PBF_MAX_COLLIDERS) {
        m_error = "too many colliders";
        return false;
    }
    m_colliders.push_back(collider);
    return push_colliders();
}

void Simulation::SetSolverIteration(uint32_t iter_count) {
    if (!m_solver) m_solver = std::make_shared<ConstraintSolver>(PBD_MODE::XPBD);
    m_solver->setSolverIteration(iter_count);
    if (m_particle_system && m_particle_system->context())
        pbf_set_solver_iterations(m_particle_system->context(), (int32_t)iter_count);
}

void Simulation::setGravity(float gravity) {
    m_world_desc.gravity = gravity;
    if (m_particle_system && m_particle_system->context())
        pbf_set_gravity(m_particle_system->context(), 0.f, gravity, 0.f);
}

bool Simulation::Step(float dt) {
    if (!m_initialized) return false;  // Simulation.cpp:82-83
    if (m_pause) return true;          // Simulation.cpp:85-86
    pbf_ctx* c = m_particle_system->context();
    m_neighbor_searcher->invalidate();
    // the reference re-reads the solver's iteration count every step (Simulation.cpp:112)
    pbf_set_solver_iterations(c, (int32_t)m_solver->getSolverIteration());
    int rc = pbf_step(c, dt);
    if (rc == PBF_OK) m_solver->SolveConstraints(dt);  // ProjectConstraints, Simulation.cpp:121
    if (rc != PBF_OK) {
        m_error = pbf_last_error(c);
        return false;
    }
    return true;
}

bool Simulation::StepDEM(float dt) {
    if (!m_initialized) return false;
    if (m_pause) return true;
    pbf_ctx* c = m_particle_system->context();
    m_neighbor_searcher->invalidate();
    m_dem.iterations = (int32_t)m_solver->getSolverIteration();
    if (pbf_step_dem(c, dt, &m_dem) != PBF_OK) {
        m_error = pbf_last_error(c);
        return false;
    }
    return true;
}

float Simulation::getRestDensity() {
    float v = 0.f;
    if (m_particle_system && m_particle_system->context())
        pbf_get_rest_density(m_particle_system->context(), &v);
    return v;
}

bool Simulation::getStats(pbf_stats* out) {
    if (!m_particle_system || !m_particle_system->context()) return false;
    return pbf_get_stats(m_particle_system->context(), out) == PBF_OK;
}
