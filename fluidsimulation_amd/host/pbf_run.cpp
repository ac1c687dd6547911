// pbf_run.cpp -- headless driver: what main.cpp + GLFWApp::Initialize/Run/Update do in the
// reference (main.cpp:9-35, GLFWApp.cpp:177-216,383-398), minus the window, renderer and GUI.
//
//   pbf_run [--scene c1|c2] [--dims nx ny nz] [--steps N] [--iters K] [--dump file.bin]
//           [--resume snap.pbf] [--save snap.pbf]   (PBFSNAP1 restart files, include/pbf.h)
//
// Builds the reference's lattice scene (GLFWApp::GenerateFluidParticles, GLFWApp.cpp:437-465),
// adds its five plane colliders, steps with dt = 0.0005, prints stats, and optionally dumps
// positions and velocities (n, then n*3 floats each) for the parity tests.
// --scene c2 builds the README CUDA scene instead: SetupSimParams' constants, GenerateFluidCube's
// 64 000 particles and InitializeBoundaryParticles' 32 000-particle open box
// (Simulation.cpp:312-576), held by the boundary-particle coupling (no plane colliders).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "Simulation.h"

int main(int argc, char** argv) {
    int nx = 10, ny = 100, nz = 10, steps = 10, iters = 2;
    std::string dump, save, resume, scene = "c1";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--dims" && i + 3 < argc) { nx = atoi(argv[++i]); ny = atoi(argv[++i]); nz = atoi(argv[++i]); }
        else if (a == "--steps" && i + 1 < argc) steps = atoi(argv[++i]);
        else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else if (a == "--save" && i + 1 < argc) save = argv[++i];
        else if (a == "--resume" && i + 1 < argc) resume = argv[++i];
        else if (a == "--scene" && i + 1 < argc) scene = argv[++i];
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    auto particle_system = std::make_shared<ParticleSystem>();
    auto simulator = std::make_shared<Simulation>();

    size_t n = 0;
    ParticleSet* particles = nullptr;
    PlaneCollider ground(Vec3(0, 1, 0), 0), c1(Vec3(1, 0, 0), -10), c2(Vec3(-1, 0, 0), -10),
        c3(Vec3(0, 0, 1), -15), c4(Vec3(0, 0, -1), 1);
    if (scene == "c2") {
        // nothing to do here: Initialize fills an empty particle system as the reference's does
        // (SetupSimParams, GenerateFluidCube, InitializeBoundaryParticles; Simulation.cpp:42-72)
    } else if (scene == "c1") {
        // GLFWApp::GenerateFluidParticles, GLFWApp.cpp:437-465
        n = (size_t)nx * ny * nz;
        particles = particle_system->AllocateParticles(n, 0.1f);
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < ny; ++j)
                for (int k = 0; k < nz; ++k) {
                    size_t idx = (size_t)k + (size_t)j * nz + (size_t)i * nz * ny;
                    Vec3 pos(0.f + 0.5f * (float)i, 0.5f + 0.5f * (float)j, -10.f + 0.5f * (float)k);
                    particles->m_positions[idx] = pos;
                    particles->m_new_positions[idx] = pos;
                    particles->m_predict_positions[idx] = pos;
                }
        // colliders, GLFWApp.cpp:177-191 and Plane.cpp:49
        for (Collider* c : {(Collider*)&ground, (Collider*)&c1, (Collider*)&c2, (Collider*)&c3, (Collider*)&c4})
            if (!simulator->AddCollider(c)) { fprintf(stderr, "AddCollider: %s\n", simulator->error().c_str()); return 1; }
    } else {
        fprintf(stderr, "unknown scene %s\n", scene.c_str());
        return 2;
    }

    if (simulator->params().capacity < n) simulator->params().capacity = (uint32_t)n;
    if (!simulator->Initialize(PBD_MODE::XPBD, particle_system)) {  // GLFWApp.cpp:215
        fprintf(stderr, "Initialize failed: %s\n", simulator->error().c_str());
        return 1;
    }
    if (scene == "c2") {
        particles = particle_system->getParticles();
        n = particles->m_size;
        const SimParams* sp = simulator->getSimParams();   // Simulation.h:76
        printf("SimParams: effective_radius %.9g particle_radius %.9g epsilon %g grid %ux%ux%u boundary %zu\n",
               sp->effective_radius, sp->particle_radius, sp->epsilon, sp->grid_size.x, sp->grid_size.y,
               sp->grid_size.z, particle_system->getBoundaryParticles()->m_size);
    }
    simulator->SetSolverIteration((uint32_t)iters);  // GLFWApp.cpp:216
    if (!resume.empty() && !particle_system->LoadSnapshot(resume.c_str())) {
        fprintf(stderr, "LoadSnapshot failed: %s\n", particle_system->error().c_str());
        return 1;
    }
    if (!simulator->Step(0.0005f) || !simulator->isPause()) { fprintf(stderr, "pause contract broken\n"); return 1; }
    simulator->Pause();  // the SPACE key, GLFWApp.cpp:285-290

    const float time_step = 0.0005f;  // GLFWApp.cpp:393
    auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s)
        if (!simulator->Step(time_step)) { fprintf(stderr, "Step failed: %s\n", simulator->error().c_str()); return 1; }
    if (!save.empty() && !particle_system->SaveSnapshot(save.c_str())) {
        fprintf(stderr, "SaveSnapshot failed: %s\n", particle_system->error().c_str());
        return 1;
    }
    if (!particle_system->Update()) { fprintf(stderr, "Update failed: %s\n", particle_system->error().c_str()); return 1; }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    pbf_stats st;
    simulator->getStats(&st);
    const auto& nb0 = simulator->getNeighborSearch()->FetchNeighbors(0);
    printf("particles %zu steps %d iterations %d rest_density %.9g\n", n, steps, iters, simulator->getRestDensity());
    printf("mean|C| %.9g max|C| %.9g com %.9g %.9g %.9g neighbours(0) %zu max_neighbours %u\n", st.mean_abs_C,
           st.max_abs_C, st.com[0], st.com[1], st.com[2], nb0.size(), st.max_neighbor_count);
    printf("%.3f ms/step (incl. final download)\n", 1e3 * sec / (steps > 0 ? steps : 1));
    if (!dump.empty()) {
        FILE* f = fopen(dump.c_str(), "wb");
        if (!f) { perror("dump"); return 1; }
        uint64_t nn = n;
        fwrite(&nn, sizeof(nn), 1, f);
        fwrite(particles->m_positions.data(), sizeof(Vec3), n, f);
        fwrite(particles->m_velocity.data(), sizeof(Vec3), n, f);
        uint64_t k = nb0.size();
        fwrite(&k, sizeof(k), 1, f);
        for (size_t v : nb0) { uint64_t u = v; fwrite(&u, sizeof(u), 1, f); }
        fclose(f);
    }
    return 0;
}
