// Simulation.h -- the step loop facade: same method names and error behaviour as the
// reference's Simulation (reference Simulation.h:50-77), headless, over the C ABI.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "Collider.h"
#include "ConstraintSolver.h"
#include "Constraints.h"
#include "NeighborSearch.h"
#include "ParticleSystem.h"
#include "pbf.h"

struct SimWorldDesc {  // Simulation.h:14-20
    SimWorldDesc(float g, float damping) : gravity(g), global_velocity_damping(damping) {}
    float gravity;
    float global_velocity_damping;
};

// Simulation.h:22-48, member for member (float3 / uint3 are plain triples here: no CUDA/HIP header)
struct SimParams {
    struct F3 { float x, y, z; };
    struct U3 { uint32_t x, y, z; };
    F3 collider_pos;
    float collider_radius;
    F3 gravity;
    float global_damping;
    float effective_radius;
    float particle_radius;
    float epsilon;
    U3 grid_size;
    uint32_t num_cells;
    F3 world_origin;
    F3 cell_size;
    float spring;
    float damping;
    float shear;
    float attraction;
    float boundary_damping;
    float static_friction;
    float kinematic_friction;
    float volume;
};

class Simulation {
public:
    Simulation();  // gravity -9.8, paused, first frame (Simulation.cpp:14-22)

    // Solver parameters for the next Initialize (defaults = the reference CPU path).
    pbf_params& params() { return m_params; }

    // Simulation.cpp:42-72.  An EMPTY particle system is filled as the reference's Initialize fills
    // it: SetupSimParams -> GenerateFluidCube (64 000 particles) -> InitializeBoundaryParticles
    // (32 000), and params() is set to that scene's solver constants (h = cell = 0.0415, rho0 1000,
    // epsilon 100, s_corr ~ dt^3 as the CUDA kernels that scene is run by).  A particle system that
    // already holds particles (AllocateParticles + positions, as GLFWApp::GenerateFluidParticles
    // leaves it) is taken as it is, with params() as the caller set them.
    bool Initialize(PBD_MODE mode, std::shared_ptr<ParticleSystem> particle_system, int device = -1);
    // Simulation.cpp:80-138: false if not initialised, true (and nothing done) while paused
    bool Step(float dt);
    bool StepCUDA(float dt) { return Step(dt); }  // Simulation.cpp:140-250: one path here
    // the granular variant of StepCUDA: solve_pbd_dem in place of solve_sph_fluid (the call the
    // reference keeps commented out at Simulation.cpp:202-213); iterations = the solver's count
    bool StepDEM(float dt);
    pbf_dem_params& demParams() { return m_dem; }  // SimParams' radius / friction, Simulation.cpp:335,349-350

    bool AddCollider(Collider* collider);          // Simulation.cpp:252-256
    // Simulation.cpp:258-266; only DistanceConstraint has an implementation in the reference
    // (BendConstraint is an empty stub, Constraints.cpp:36-44).  Projected every Step in list
    // order with the solver's PBD / XPBD mode and iteration count.
    bool AddStaticConstraint(Constraint* constraint);
    bool AddStaticConstraints(const std::vector<Constraint*>& constraints);
    void SetSolverIteration(uint32_t iter_count);  // Simulation.cpp:268-271
    void Pause() { m_pause = !m_pause; }           // Simulation.cpp:302-305
    void setGravity(float gravity);                // Simulation.cpp:307-310
    float getGravity() const { return m_world_desc.gravity; }
    SimParams* getSimParams() { return &m_sim_params; }   // Simulation.h:76; filled by Initialize
    bool isPause() const { return m_pause; }
    float getRestDensity();

    std::shared_ptr<NeighborSearch> getNeighborSearch() { return m_neighbor_searcher; }
    std::shared_ptr<ConstraintSolver> getSolver() { return m_solver; }
    bool getStats(pbf_stats* out);
    const std::string& error() const { return m_error; }

private:
    void SetupSimParams();               // Simulation.cpp:312-354
    void GenerateFluidCube();            // Simulation.cpp:528-576
    void InitializeBoundaryParticles();  // Simulation.cpp:356-490
    bool push_colliders();
    bool push_constraints();
    std::vector<Constraint*> m_static_constraints;
    SimWorldDesc m_world_desc;
    bool m_initialized = false;
    bool m_pause = true;
    pbf_params m_params;
    SimParams m_sim_params{};
    float m_rest_density = 0.8f, m_particle_mass = 0.015f, m_volume = 0.f;   // Simulation.cpp:18,316-322
    pbf_dem_params m_dem{0.012331058f, 1.0f, 0.75f, 3};
    std::shared_ptr<ConstraintSolver> m_solver;
    std::shared_ptr<ParticleSystem> m_particle_system;
    std::shared_ptr<NeighborSearch> m_neighbor_searcher;
    std::vector<Collider*> m_colliders;
    std::string m_error;
};
