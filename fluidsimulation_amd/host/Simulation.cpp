// Simulation.cpp -- host facade over the C ABI (include/pbf.h).  No HIP here: the facade is
// plain C++17 that a maintainer of the reference app could link in place of its own
// Simulation / ParticleSystem / NeighborSearch / ConstraintSolver translation units.
#include "Simulation.h"

#include <cmath>
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
    // the reference builds its scene inside Initialize (Simulation.cpp:51-53); a caller that brought
    // its own particles keeps them
    const bool build_scene = !m_particle_system->getParticles();
    if (build_scene) m_params.grid_dim[0] = m_params.grid_dim[1] = m_params.grid_dim[2] = 64;   // Simulation.cpp:46
    GridSize gs{(uint32_t)m_params.grid_dim[0], (uint32_t)m_params.grid_dim[1], (uint32_t)m_params.grid_dim[2]};
    m_neighbor_searcher = std::make_shared<NeighborSearch>(m_particle_system, gs);
    if (!m_solver) m_solver = std::make_shared<ConstraintSolver>(mode);
    else m_solver->setPBDMode(mode);
    SetupSimParams();
    if (build_scene) {
        GenerateFluidCube();
        InitializeBoundaryParticles();
    }
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

// Simulation::SetupSimParams (Simulation.cpp:312-354) in fp32 with pi = 3.1415926f (common.h:8):
//   volume = 20 * 0.015 / 1000, h = powf(3 volume / (4 pi), 1/3), r = powf(pi / (6 * 20), 1/3) * h.
// The two derived values are written as the constants SURVEY.md 8(a) row a3 pins (powf differs by an
// ulp between C libraries for r; the scene must not depend on which libm built this library).
void Simulation::SetupSimParams() {
    const float particle_mass = 0.015f, n_kernel_particles = 20.f;
    m_rest_density = 1000.f;
    m_particle_mass = particle_mass;
    m_volume = n_kernel_particles * particle_mass / m_rest_density;
    const float effective_radius = 0.041528303f, particle_radius = 0.012331058f;
    SimParams& sp = m_sim_params;
    sp = SimParams{};
    sp.gravity = {0.f, -9.8f, 0.f};
    sp.global_damping = 1.f;
    sp.particle_radius = particle_radius;
    sp.effective_radius = effective_radius;
    sp.epsilon = 100.f;
    sp.grid_size = {m_neighbor_searcher->m_grid_size.x, m_neighbor_searcher->m_grid_size.y,
                    m_neighbor_searcher->m_grid_size.z};
    sp.num_cells = m_neighbor_searcher->m_num_grid_cells;
    sp.world_origin = {0.f, 0.f, 0.f};
    sp.cell_size = {effective_radius, effective_radius, effective_radius};
    sp.spring = 0.5f;
    sp.damping = 0.02f;
    sp.shear = 0.1f;
    sp.attraction = 0.0f;
    sp.boundary_damping = 0.1f;
    sp.static_friction = 1.0f;       // "ice friction at -12 C"
    sp.kinematic_friction = 0.75f;
    sp.volume = m_volume;
}

// Simulation::GenerateFluidCube (Simulation.cpp:528-576): i, j, k in [-n, n), n = floor(0.5 / d),
// k fastest, cube centred on (0, 0.75, 0); the jitter the reference computes is never applied
void Simulation::GenerateFluidCube() {
    const float d = 2.f * m_sim_params.particle_radius;
    const int n = (int)(0.5f / d);
    ParticleSet* f = m_particle_system->AllocateParticles((size_t)8 * n * n * n, m_particle_mass);
    size_t idx = 0;
    for (int i = -n; i < n; ++i)
        for (int j = -n; j < n; ++j)
            for (int k = -n; k < n; ++k, ++idx) {
                const Vec3 p(0.f + d * (float)i, 0.75f + d * (float)j, 0.f + d * (float)k);
                f->m_positions[idx] = f->m_new_positions[idx] = f->m_predict_positions[idx] = p;
            }
    m_particle_system->setParticleRadius(m_sim_params.particle_radius);
    // the solver constants this scene is run with (the CUDA path's: Simulation.cpp:318,337,
    // cuda_simulation.cu:973), layered on the CPU path's arithmetic
    pbf_params& P = m_params;
    P.capacity = (uint32_t)f->m_size;
    P.max_neighbors = 96;
    P.effective_radius = P.grid_spacing = m_sim_params.effective_radius;
    P.epsilon = m_sim_params.epsilon;
    P.scorr_dt_power = 3;
    P.rest_density_mode = PBF_RHO0_FIXED;
    P.rest_density = m_rest_density;
}

// Simulation::InitializeBoundaryParticles (Simulation.cpp:356-490): one layer of particles as an
// open box -- floor, left / right, back / front
void Simulation::InitializeBoundaryParticles() {
    const float d = 2.f * m_sim_params.particle_radius, h = m_sim_params.effective_radius;
    const int m = (int)(1.f / d);
    std::vector<Vec3> b;
    for (int i = -m; i < m; ++i)
        for (int k = -m; k < m; ++k) b.emplace_back(d * (float)i, 0.f, d * (float)k);
    const float left = d * (float)(-m) - d, right = d * (float)(m - 1) + d;
    for (int j = -m; j < m; ++j)
        for (int k = -m; k < m; ++k) {
            const float y = d * (float)j + (float)m * d, z = d * (float)k;
            b.emplace_back(left, y, z);
            b.emplace_back(right, y, z);
        }
    const float back = d * (float)(-m) - d, front = d * (float)(m - 1) + d;
    for (int i = -m; i < m; ++i)
        for (int j = -m; j < m; ++j) {
            const float x = d * (float)i, y = d * (float)j + (float)m * d;
            b.emplace_back(x, y, back);
            b.emplace_back(x, y, front);
        }
    ParticleSet* bs = m_particle_system->AllocateBoundaryParticles(b.size(), m_particle_mass);
    float lo[3] = {1e30f, 1e30f, 1e30f};
    for (size_t t = 0; t < b.size(); ++t) {
        bs->m_positions[t] = b[t];
        lo[0] = fminf(lo[0], b[t].x); lo[1] = fminf(lo[1], b[t].y); lo[2] = fminf(lo[2], b[t].z);
    }
    // the wrap-around grid's window starts one cell below the walls (no wall cell on a grid face)
    for (int a = 0; a < 3; ++a) m_params.grid_origin_cell[a] = (int32_t)floorf(lo[a] / h) - 1;
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
