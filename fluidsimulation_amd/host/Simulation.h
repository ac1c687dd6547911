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

class Simulation {
public:
    Simulation();  // gravity -9.8, paused, first frame (Simulation.cpp:14-22)

    // Solver parameters for the next Initialize (defaults = the reference CPU path).
    pbf_params& params() { return m_params; }

    // Simulation.cpp:42-72.  The particle system must already hold its particles
    // (AllocateParticles + positions), as GLFWApp::GenerateFluidParticles leaves it.
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
    bool isPause() const { return m_pause; }
    float getRestDensity();

    std::shared_ptr<NeighborSearch> getNeighborSearch() { return m_neighbor_searcher; }
    std::shared_ptr<ConstraintSolver> getSolver() { return m_solver; }
    bool getStats(pbf_stats* out);
    const std::string& error() const { return m_error; }

private:
    bool push_colliders();
    bool push_constraints();
    std::vector<Constraint*> m_static_constraints;
    SimWorldDesc m_world_desc;
    bool m_initialized = false;
    bool m_pause = true;
    pbf_params m_params;
    pbf_dem_params m_dem{0.012331058f, 1.0f, 0.75f, 3};
    std::shared_ptr<ConstraintSolver> m_solver;
    std::shared_ptr<ParticleSystem> m_particle_system;
    std::shared_ptr<NeighborSearch> m_neighbor_searcher;
    std::vector<Collider*> m_colliders;
    std::string m_error;
};
