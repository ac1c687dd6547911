// Constraints.h -- host mirror of the reference's constraint classes (reference
// Constraints.h:15-84).  A DistanceConstraint here names its two particles by index into the
// fluid ParticleSet (the reference holds pointers to legacy Particle objects); the projection
// itself runs on the device (include/pbf.h: pbf_set_distance_constraints).
#pragma once
#include <cstddef>
#include "pbf.h"

enum class CONSTRAINT_TYPE { CONSTRAINT_DISTANCE, CONSTRAINT_BEND };

class Constraint {
public:
    virtual ~Constraint() = default;
    void setStiffness(float stiffness) {  // Constraints.cpp:25-33
        m_stiffness = stiffness > 1.f ? 1.f : (stiffness < 0.f ? 0.f : stiffness);
    }
    void setCompliance(float compliance) { m_compliance = compliance; }
    virtual CONSTRAINT_TYPE getConstraintType() = 0;
    float m_lambda = 0.f;
    float m_stiffness = 1.f;
    float m_compliance = 0.f;
};

class DistanceConstraint final : public Constraint {
public:
    DistanceConstraint(size_t p0, size_t p1, float rest_length)
        : m_p0(p0), m_p1(p1), m_rest_length(rest_length) {}
    CONSTRAINT_TYPE getConstraintType() override { return CONSTRAINT_TYPE::CONSTRAINT_DISTANCE; }
    pbf_distance_constraint pod() const {
        return {(uint32_t)m_p0, (uint32_t)m_p1, m_rest_length, m_stiffness, m_compliance};
    }
    size_t m_p0, m_p1;
    float m_rest_length;
};
