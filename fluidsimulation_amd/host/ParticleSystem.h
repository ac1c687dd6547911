// ParticleSystem.h -- owner of the fluid ParticleSet and of its device twin.
// Mirrors the call surface of the reference's ParticleSystem (reference
// ParticleSystem.h:11-46) without its GL VAO/VBO and CUDA-GL interop members.
#pragma once
#include <memory>
#include <string>
#include "Particle.h"
#include "pbf.h"

class ParticleSystem {
public:
    ParticleSystem() = default;
    ~ParticleSystem();
    ParticleSystem(const ParticleSystem&) = delete;
    ParticleSystem& operator=(const ParticleSystem&) = delete;

    // ParticleSystem.cpp:99-103
    ParticleSet* AllocateParticles(size_t n, float particle_mass);
    ParticleSet* getParticles() { return m_particles.get(); }
    // ParticleSystem.cpp:105-109; the boundary set is uploaded by InitializeDevice
    // (pbf_set_boundary: the reference's InitializeBoundaryCudaData, Simulation.cpp:492-526)
    ParticleSet* AllocateBoundaryParticles(size_t n, float particle_mass);
    ParticleSet* getBoundaryParticles() { return m_boundary_particles.get(); }
    void setParticleRadius(float r) { m_particle_radius = r; }
    float getParticleRadius() const { return m_particle_radius; }

    // replaces Initialize()/InitializeCUDA() (ParticleSystem.cpp:30-36,116-348): creates the
    // device arrays for `params` and uploads the host vectors.  false on failure (see error()).
    bool InitializeDevice(const pbf_params& params, int device = -1);
    // replaces Update()/UpdateGLBUfferData (ParticleSystem.cpp:38-49,372-406): instead of
    // re-uploading positions to a GL buffer, refreshes the HOST vectors from the device.
    bool Update();
    // the reference's VBO contents (ParticleSystem.cpp:372-389, drawn by Renderer.cpp:121-127):
    // m_size tightly packed float3 positions in particle-index order, written on the device
    // into caller-owned device memory (e.g. the mapped GL buffer of RegisterCUDAVBO's
    // counterpart), asynchronous on the ctx stream.
    bool UpdateDeviceVBO(void* d_xyz) {
        if (!m_ctx) return false;
        if (pbf_export_positions(m_ctx, d_xyz) == PBF_OK) return true;
        m_error = pbf_last_error(m_ctx);
        return false;
    }
    // raw SoA snapshot / restart (include/pbf.h: pbf_snapshot_save/load)
    bool SaveSnapshot(const char* path) {
        if (!m_ctx) return false;
        if (pbf_snapshot_save(m_ctx, path) == PBF_OK) return true;
        m_error = pbf_last_error(m_ctx);
        return false;
    }
    bool LoadSnapshot(const char* path) {
        if (!m_ctx) return false;
        if (pbf_snapshot_load(m_ctx, path) == PBF_OK) return true;
        m_error = pbf_last_error(m_ctx);
        return false;
    }
    void Release();  // ParticleSystem.cpp:59-97

    pbf_ctx* context() { return m_ctx; }
    const std::string& error() const { return m_error; }

private:
    std::unique_ptr<ParticleSet> m_particles;
    std::unique_ptr<ParticleSet> m_boundary_particles;
    float m_particle_radius = 0.f;
    pbf_ctx* m_ctx = nullptr;
    std::string m_error;
};
