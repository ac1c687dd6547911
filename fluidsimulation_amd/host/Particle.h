// Particle.h -- host-side mirror of the reference's ParticleSet (reference Particle.h:36-88).
// The host vectors keep the reference's member names; the device twins live behind the
// C ABI (pbf_ctx) as float4 arrays in sorted order and are not exposed.
#pragma once
#include <cstddef>
#include <vector>
#include "Vec3.h"

class ParticleSet {
public:
    ParticleSet() = default;
    // Particle.cpp:172-188: zero state, uniform mass, massInv = 1/m
    ParticleSet(size_t n, float particle_mass)
        : m_size(n), m_positions(n), m_predict_positions(n), m_new_positions(n), m_velocity(n),
          m_force(n), m_mass(n, particle_mass), m_massInv(n, 1.f / particle_mass), m_density(n, 0.f),
          m_C(n, 0.f), m_lambda(n, 0.f), m_volume(n, 0.f) {}

    size_t m_size = 0;
    std::vector<Vec3> m_positions;
    std::vector<Vec3> m_predict_positions;
    std::vector<Vec3> m_new_positions;
    std::vector<Vec3> m_velocity;
    std::vector<Vec3> m_force;
    std::vector<float> m_mass;
    std::vector<float> m_massInv;
    std::vector<float> m_density;
    std::vector<float> m_C;
    std::vector<float> m_lambda;
    std::vector<float> m_volume;  // boundary particles: Akinci volume (Particle.h:68)
};
