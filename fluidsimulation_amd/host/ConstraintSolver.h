// ConstraintSolver.h -- mirrors the reference's ConstraintSolver (reference
// ConstraintSolver.h:8-53).  For fluids the reference uses it only as the holder of the
// solver iteration count (Simulation.cpp:112) and PBD / XPBD mode.  The Gauss-Seidel sweep over
// the static constraint list (ConstraintSolver.cpp:35-72) runs on the device inside pbf_step
// (pbf_project_constraints) with this object's mode and iteration count, so SolveConstraints on
// the host has nothing left to do.
#pragma once
#include <cstdint>

enum class PBD_MODE { PBD, XPBD };

class ConstraintSolver {
public:
    explicit ConstraintSolver(PBD_MODE mode) : m_solver_iteration(1), m_mode(mode) {}  // .cpp:3-6
    bool SolveConstraints(const float& /*dt*/) { return true; }  // .cpp:12-23 over empty lists
    void setSolverIteration(uint32_t iteration_num) { m_solver_iteration = iteration_num; }
    void setPBDMode(PBD_MODE mode) { m_mode = mode; }
    uint32_t getSolverIteration() const { return m_solver_iteration; }
    PBD_MODE getPBDMode() const { return m_mode; }

private:
    uint32_t m_solver_iteration;
    PBD_MODE m_mode;
};
