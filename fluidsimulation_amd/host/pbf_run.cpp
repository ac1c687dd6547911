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

// Simulation::SetupSimParams (Simulation.cpp:312-354) in fp32 with pi = 3.1415926f:
//   volume = 20 * 0.015 / 1000, h = powf(3 volume / (4 pi), 1/3), r = powf(pi / (6 * 20), 1/3) * h.
// The two derived values are written as the constants SURVEY.md 8(a) row a3 pins (powf differs by
// an ulp between C libraries for r; the scene must not depend on which libm built this driver).
static void sim_params_constants(float& h, float& r, float& mass, float& rho0) {
    mass = 0.015f;
    rho0 = 1000.f;
    h = 0.041528303f;
    r = 0.012331058f;
}

// Simulation::GenerateFluidCube + InitializeBoundaryParticles (Simulation.cpp:356-576)
static void build_c2(ParticleSystem& ps, pbf_params& P) {
    float h, r, mass, rho0;
    sim_params_constants(h, r, mass, rho0);
    const float d = 2.f * r;
    const int n = (int)(0.5f / d);
    ParticleSet* f = ps.AllocateParticles((size_t)8 * n * n * n, mass);
    size_t idx = 0;
    for (int i = -n; i < n; ++i)
        for (int j = -n; j < n; ++j)
            for (int k = -n; k < n; ++k, ++idx) {
                Vec3 p(0.f + d * (float)i, 0.75f + d * (float)j, 0.f + d * (float)k);
                f->m_positions[idx] = f->m_new_positions[idx] = f->m_predict_positions[idx] = p;
            }
    const int m = (int)(1.f / d);
    std::vector<Vec3> b;
    for (int i = -m; i < m; ++i)
        for (int k = -m; k < m; ++k) b.emplace_back(d * (float)i, 0.f, d * (float)k);   // floor
    const float left = d * (float)(-m) - d, right = d * (float)(m - 1) + d;
    for (int j = -m; j < m; ++j)
        for (int k = -m; k < m; ++k) {                                                  // left, right
            const float y = d * (float)j + (float)m * d, z = d * (float)k;
            b.emplace_back(left, y, z);
            b.emplace_back(right, y, z);
        }
    const float back = d * (float)(-m) - d, front = d * (float)(m - 1) + d;
    for (int i = -m; i < m; ++i)
        for (int j = -m; j < m; ++j) {                                                  // back, front
            const float x = d * (float)i, y = d * (float)j + (float)m * d;
            b.emplace_back(x, y, back);
            b.emplace_back(x, y, front);
        }
    ParticleSet* bs = ps.AllocateBoundaryParticles(b.size(), mass);
    float lo[3] = {1e30f, 1e30f, 1e30f};
    for (size_t t = 0; t < b.size(); ++t) {
        bs->m_positions[t] = b[t];
        lo[0] = fminf(lo[0], b[t].x); lo[1] = fminf(lo[1], b[t].y); lo[2] = fminf(lo[2], b[t].z);
    }
    ps.setParticleRadius(r);
    P.capacity = (uint32_t)f->m_size;
    P.max_neighbors = 96;
    P.effective_radius = P.grid_spacing = h;
    P.grid_dim[0] = P.grid_dim[1] = P.grid_dim[2] = 64;              // Simulation.cpp:46
    for (int a = 0; a < 3; ++a) P.grid_origin_cell[a] = (int32_t)floorf(lo[a] / h) - 1;
    P.epsilon = 100.f;                                               // Simulation.cpp:337
    P.scorr_dt_power = 3;                                            // cuda_simulation.cu:973
    P.rest_density_mode = PBF_RHO0_FIXED;
    P.rest_density = rho0;
}

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
        build_c2(*particle_system, simulator->params());
        particles = particle_system->getParticles();
        n = particles->m_size;
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
