// pbf_slab_run.cpp -- headless multi-slab driver in plain C++ over the C ABI's native slab
// runtime (pbf_slab_attach / pbf_slab_step); no Python, no torch.  The reference is single-GPU
// (SURVEY.md 8(e)); the scene is the synthetic dam-break block of BASELINE.json's C3-C5
// (fluidsimulation_amd/scenes.py::dam_break restated).
//
//   pbf_slab_run --dims nx ny nz [--steps N] [--iters K] [--ghost-layers L]
//       [--overlap] [--profile K]
//       --slabs W [--check] [--rebalance K]          W rank THREADS of this process over the library's loopback
//                                     transport, all on device 0 (what a one-GPU box can run);
//                                     --check steps ONE ctx over the whole scene as well and
//                                     compares bit for bit
//       --rank r --world W --id-file F [--device d]
//                                     one PROCESS per GPU over RCCL: rank 0 writes the RCCL
//                                     unique id to F, the others wait for it
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "SlabDriver.h"

namespace {

struct Scene {
    pbf_params P;
    std::vector<Vec3> pos;
    float mass;
    std::vector<pbf_collider> cols;
};

int grid_dim_for(int extent_cells) { return std::max(16, (extent_cells + 3 + 7) / 8 * 8); }

pbf_collider plane(float nx, float ny, float nz, float d) {
    pbf_collider c;
    std::memset(&c, 0, sizeof(c));
    c.type = PBF_COLLIDER_PLANE;
    c.a[0] = nx; c.a[1] = ny; c.a[2] = nz;
    c.d = d;
    return c;
}

// scenes.py::dam_break: nx*ny*nz block at spacing d = 2r in the low-x end of an open-top tank
// 2.5 times longer in x; SetupSimParams' constants (Simulation.cpp:312-354) as pinned by
// SURVEY.md 8(a) a3; fixed rho0, eps = 100, s_corr ~ dt^3 as on the reference's GPU path
Scene dam_break(int nx, int ny, int nz, int iterations) {
    Scene s;
    const float h = 0.041528303f, r = 0.012331058f, tank = 2.5f;
    s.mass = 0.015f;
    const float d = 2.f * r, half = 0.5f * d;
    s.pos.reserve((size_t)nx * ny * nz);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k)
                s.pos.emplace_back(0.f + half + d * (float)i, 0.f + half + d * (float)j, 0.f + half + d * (float)k);
    const float hi[3] = {tank * d * (float)nx, 4.0f * d * (float)ny, d * (float)nz};
    pbf_default_params(&s.P);
    pbf_params& P = s.P;
    P.capacity = (uint32_t)s.pos.size();
    P.max_neighbors = 96;
    P.effective_radius = P.grid_spacing = h;
    const float ext[3] = {hi[0] / h, hi[1] / h, hi[2] / h};
    P.grid_dim[0] = grid_dim_for((int)ext[0]);
    P.grid_dim[1] = grid_dim_for((int)std::min(ext[1], 2.0f * (float)ny * d / h));
    P.grid_dim[2] = grid_dim_for((int)ext[2]);
    P.grid_origin_cell[0] = P.grid_origin_cell[1] = P.grid_origin_cell[2] = -1;
    P.epsilon = 100.f;
    P.scorr_dt_power = 3;
    P.rest_density_mode = PBF_RHO0_FIXED;
    P.rest_density = 1000.f;
    P.solver_iterations = iterations;
    P.container_enabled = 1;
    P.container_min[0] = P.container_min[1] = P.container_min[2] = 0.f;
    P.container_max[0] = hi[0];
    P.container_max[1] = 1.0e9f;
    P.container_max[2] = hi[2];
    s.cols = {plane(0, 1, 0, 0.f), plane(1, 0, 0, 0.f), plane(-1, 0, 0, -hi[0]), plane(0, 0, 1, 0.f),
              plane(0, 0, -1, -hi[2])};
    return s;
}

int g_profile_steps = 0;   // --profile K

struct RankResult {
    std::vector<float> records;   // owned particles at the end
    std::string error;
    double seconds = 0;
    int planes_moved = 0;
    std::string phases;           // device time per phase and sub-step of a profiled pass (--profile)
};

// one rank: its slab of the scene, stepped natively
void run_rank(const Scene& sc, const std::vector<int32_t>& planes0, int rank, int world, int device,
              int layers, int steps, float dt, const void* comm_id, RankResult* out, int rebalance_every = 0,
              bool overlap = false) {
    std::vector<int32_t> planes = planes0;   // this rank's copy: pbf_slab_rebalance keeps all copies equal
    pbf_ctx* ctx = nullptr;
    pbf_comm* comm = nullptr;
    auto fail = [&](const char* what) {
        out->error = std::string(what) + ": " + (ctx ? pbf_last_error(ctx) : pbf_last_error(nullptr));
        if (ctx) pbf_destroy(ctx);
        if (comm) pbf_comm_destroy(comm);
    };
    pbf_params P = slab::rank_local_params(sc.P, sc.pos, planes, rank, world, 1.5, layers);
    if (pbf_create(&P, device, &ctx) != PBF_OK) return fail("pbf_create");
    if (pbf_upload(ctx, nullptr, nullptr, nullptr, sc.mass, 0) != PBF_OK) return fail("pbf_upload");
    if (pbf_set_colliders(ctx, sc.cols.data(), (uint32_t)sc.cols.size()) != PBF_OK) return fail("pbf_set_colliders");
    if (pbf_slab_set_bounds(ctx, planes[rank], planes[rank + 1], layers) != PBF_OK) return fail("pbf_slab_set_bounds");
    {
        std::vector<float> rec = slab::records_in(sc.pos, sc.mass, sc.P.grid_spacing, planes[rank], planes[rank + 1]);
        if (pbf_slab_append_owned_host(ctx, rec.data(), (uint32_t)(rec.size() / PBF_RECORD_FLOATS)) != PBF_OK)
            return fail("pbf_slab_append_owned_host");
    }
    if (pbf_comm_create(comm_id, rank, world, device, &comm) != PBF_OK) {
        out->error = std::string("pbf_comm_create: ") + pbf_comm_last_error(nullptr);
        pbf_destroy(ctx);
        return;
    }
    if (pbf_slab_attach(ctx, comm, rank == 0 ? -1 : rank - 1, rank == world - 1 ? -1 : rank + 1, 0) != PBF_OK)
        return fail("pbf_slab_attach");
    if (overlap && pbf_slab_set_overlap(ctx, 1) != PBF_OK) return fail("pbf_slab_set_overlap");
    auto t0 = std::chrono::steady_clock::now();
    // re-balancing histograms cover the tank's cell planes (SlabDriver.h: plane_range)
    int32_t x_first = 0;
    uint32_t n_planes = 1;
    slab::plane_range(sc.P, sc.pos, &x_first, &n_planes);
    for (int s = 0; s < steps; ++s) {
        if (pbf_slab_step(ctx, dt) != PBF_OK) return fail("pbf_slab_step");
        if (rebalance_every > 0 && layers >= 2 && world > 1 && (s + 1) % rebalance_every == 0) {
            int32_t moved = 0;
            if (pbf_slab_rebalance(ctx, planes.data(), x_first, n_planes, layers - 1, &moved) != PBF_OK)
                return fail("pbf_slab_rebalance");
            out->planes_moved += moved;
        }
    }
    if (pbf_sync(ctx) != PBF_OK) return fail("pbf_sync");
    out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (g_profile_steps > 0) {   // a pass of its own: HIP events between the launches
        pbf_profile_enable(ctx, 1);
        pbf_profile_reset(ctx);
        for (int s = 0; s < g_profile_steps; ++s)
            if (pbf_slab_step(ctx, dt) != PBF_OK) return fail("pbf_slab_step");
        double ms[16];
        uint64_t calls[16];
        const uint32_t k = pbf_num_phases() < 16 ? pbf_num_phases() : 16;
        if (pbf_profile_read(ctx, ms, calls, k) != PBF_OK) return fail("pbf_profile_read");
        pbf_profile_enable(ctx, 0);
        char buf[96];
        for (uint32_t ph = 0; ph < k; ++ph)
            if (calls[ph]) {
                snprintf(buf, sizeof buf, " %s %.1f us x %.1f", pbf_phase_name(ph), 1e3 * ms[ph] / (double)calls[ph],
                         (double)calls[ph] / g_profile_steps);
                out->phases += buf;
            }
    }
    uint32_t k = 0;
    out->records.resize((size_t)P.capacity * PBF_RECORD_FLOATS);
    if (pbf_slab_export_host(ctx, out->records.data(), P.capacity, &k) != PBF_OK) return fail("pbf_slab_export_host");
    out->records.resize((size_t)k * PBF_RECORD_FLOATS);
    pbf_destroy(ctx);
    pbf_comm_destroy(comm);
}

}  // namespace

int main(int argc, char** argv) {
    int nx = 48, ny = 12, nz = 12, steps = 5, iters = 4, slabs = 0, layers = 4, rank = -1, world = 0, device = -1;
    bool check = false;
    int rebalance = 0;   // --rebalance K: pbf_slab_rebalance every K sub-steps
    bool overlap = false;   // --overlap: pbf_slab_set_overlap (ghost refresh behind the interior lambda sweep)
    std::string id_file;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--dims" && i + 3 < argc) { nx = atoi(argv[++i]); ny = atoi(argv[++i]); nz = atoi(argv[++i]); }
        else if (a == "--steps" && i + 1 < argc) steps = atoi(argv[++i]);
        else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
        else if (a == "--ghost-layers" && i + 1 < argc) layers = atoi(argv[++i]);
        else if (a == "--slabs" && i + 1 < argc) slabs = atoi(argv[++i]);
        else if (a == "--rank" && i + 1 < argc) rank = atoi(argv[++i]);
        else if (a == "--world" && i + 1 < argc) world = atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--id-file" && i + 1 < argc) id_file = argv[++i];
        else if (a == "--check") check = true;
        else if (a == "--rebalance" && i + 1 < argc) rebalance = atoi(argv[++i]);
        else if (a == "--overlap") overlap = true;
        else if (a == "--profile" && i + 1 < argc) g_profile_steps = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    const float dt = 0.0005f;  // GLFWApp.cpp:393
    const Scene sc = dam_break(nx, ny, nz, iters);
    const size_t n = sc.pos.size();

    if (rank >= 0) {   // one process per GPU over RCCL
        if (world < 1 || rank >= world || id_file.empty()) { fprintf(stderr, "--rank needs --world and --id-file\n"); return 2; }
        char id[PBF_COMM_ID_BYTES];
        if (rank == 0) {
            if (pbf_comm_unique_id(id) != PBF_OK) { fprintf(stderr, "%s\n", pbf_comm_last_error(nullptr)); return 1; }
            std::ofstream f(id_file + ".tmp", std::ios::binary);
            f.write(id, sizeof(id));
            f.close();
            std::rename((id_file + ".tmp").c_str(), id_file.c_str());
        } else {
            for (int tries = 0;; ++tries) {
                std::ifstream f(id_file, std::ios::binary);
                if (f.read(id, sizeof(id))) break;
                if (tries > 600) { fprintf(stderr, "no communicator id in %s\n", id_file.c_str()); return 1; }
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        const std::vector<int32_t> planes = slab::split_planes(sc.pos, sc.P.grid_spacing, world, 2 * layers + 1);
        RankResult res;
        run_rank(sc, planes, rank, world, device < 0 ? rank : device, layers, steps, dt, id, &res, rebalance);
        if (!res.error.empty()) { fprintf(stderr, "rank %d: %s\n", rank, res.error.c_str()); return 1; }
        printf("rank %d of %d: %zu owned particles after %d sub-steps, %.3f ms/sub-step\n", rank, world,
               res.records.size() / PBF_RECORD_FLOATS, steps, 1e3 * res.seconds / (steps > 0 ? steps : 1));
        return 0;
    }

    if (slabs < 1) { fprintf(stderr, "give --slabs W (threads, loopback) or --rank/--world/--id-file (RCCL)\n"); return 2; }
    char id[PBF_COMM_ID_BYTES];
    pbf_comm_local_id(id);
    const std::vector<int32_t> planes = slab::split_planes(sc.pos, sc.P.grid_spacing, slabs, 2 * layers + 1);
    std::vector<RankResult> res((size_t)slabs);
    std::vector<std::thread> th;
    for (int r = 0; r < slabs; ++r)
        th.emplace_back(run_rank, std::cref(sc), std::cref(planes), r, slabs, device < 0 ? 0 : device, layers, steps,
                        dt, (const void*)id, &res[(size_t)r], rebalance, overlap);
    for (auto& t : th) t.join();
    size_t total = 0;
    double slowest = 0;
    for (int r = 0; r < slabs; ++r) {
        if (!res[(size_t)r].error.empty()) { fprintf(stderr, "rank %d: %s\n", r, res[(size_t)r].error.c_str()); return 1; }
        total += res[(size_t)r].records.size() / PBF_RECORD_FLOATS;
        slowest = std::max(slowest, res[(size_t)r].seconds);
    }
    printf("particles %zu in %d slabs (%d ghost layers%s), %d sub-steps, %d iterations: %.3f ms/sub-step (slowest rank)\n",
           n, slabs, layers, overlap ? ", overlapped refresh" : "", steps, iters, 1e3 * slowest / (steps > 0 ? steps : 1));
    if (g_profile_steps > 0)
        for (int r = 0; r < slabs; ++r) printf("rank %d phases (us per launch x launches per sub-step):%s\n", r, res[(size_t)r].phases.c_str());
    if (rebalance > 0) {
        printf("re-balanced every %d sub-steps: %d plane moves; owned per rank at the end:", rebalance, res[0].planes_moved);
        for (int r = 0; r < slabs; ++r) printf(" %zu", res[(size_t)r].records.size() / PBF_RECORD_FLOATS);
        printf("\n");
    }
    if (total != n) { fprintf(stderr, "particle count changed: %zu -> %zu\n", n, total); return 1; }
    if (check) {
        // ONE ctx over the whole scene, same steps; compare position and velocity by particle id
        pbf_ctx* ctx = nullptr;
        if (pbf_create(&sc.P, device < 0 ? 0 : device, &ctx) != PBF_OK) { fprintf(stderr, "%s\n", pbf_last_error(nullptr)); return 1; }
        std::vector<float> p3(n * 3), v3(n * 3);
        bool ok = pbf_upload(ctx, &sc.pos[0].x, nullptr, nullptr, sc.mass, (uint32_t)n) == PBF_OK &&
                  pbf_set_colliders(ctx, sc.cols.data(), (uint32_t)sc.cols.size()) == PBF_OK;
        for (int s = 0; ok && s < steps; ++s) ok = pbf_step(ctx, dt) == PBF_OK;
        ok = ok && pbf_download(ctx, p3.data(), v3.data(), nullptr) == PBF_OK;
        if (!ok) { fprintf(stderr, "single ctx: %s\n", pbf_last_error(ctx)); return 1; }
        pbf_destroy(ctx);
        size_t bad = 0;
        for (int r = 0; r < slabs; ++r) {
            const std::vector<float>& rec = res[(size_t)r].records;
            for (size_t k = 0; k < rec.size(); k += PBF_RECORD_FLOATS) {
                const size_t id_ = slab::record_id(&rec[k]);
                if (id_ >= n || std::memcmp(&rec[k], &p3[3 * id_], 12) != 0 || std::memcmp(&rec[k + 4], &v3[3 * id_], 12) != 0)
                    ++bad;
            }
        }
        printf("check against one ctx: %zu of %zu particles differ\n", bad, n);
        if (bad) return 1;
    }
    return 0;
}
