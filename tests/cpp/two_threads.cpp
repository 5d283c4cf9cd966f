// tests/cpp/two_threads.cpp — `LiquidWorld: Send + Sync` on the C side (the reference pins it at compile time,
// /root/reference/src/liquid_world.rs:283-287; here it is a run-time property of libsalva_hip.so: every entry point that takes a
// world holds the world's lock, capi.hip).  One world, three threads at once: an AABB query (scratch buffers + a stream), a
// read-back of the particles (lazily refreshed staging arrays), contact counts (field read-back), while the main thread steps.
// Every answer a reader gets must be one a single-threaded run could have got: the query hits and the positions of SOME
// completed step — never a torn state, a crash, or a HIP error.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/salva_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, salva_hip_last_error()); exit(2); } } while (0)

int main() {
    const float r = 0.025f;
    SalvaHipParams p;
    salva_hip_default_params(&p);
    p.particle_radius = r;
    SalvaHipWorld* w = nullptr;
    CHECK(salva_hip_create(&p, &w));
    const int side = 24;
    const uint64_t n = (uint64_t)side * side * side;
    std::vector<float> pos(3 * n), vel(3 * n, 0.0f);
    uint64_t k = 0;
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j)
            for (int l = 0; l < side; ++l, ++k) { pos[3 * k] = 2 * r * i; pos[3 * k + 1] = 2 * r * j; pos[3 * k + 2] = 2 * r * l; }
    CHECK(salva_hip_set_fluid(w, 0, n, pos.data(), vel.data(), nullptr, nullptr, nullptr, 1000.0f, 1u, 0xffffffffu, SALVA_HIP_DIRTY_ALL));
    SalvaHipForceDesc xsph;
    memset(&xsph, 0, sizeof xsph);
    xsph.kind = SALVA_HIP_FORCE_XSPH; xsph.p[0] = 0.5f;
    CHECK(salva_hip_set_fluid_forces(w, 0, &xsph, 1));
    const float g[3] = {0.0f, -9.81f, 0.0f};
    SalvaHipStepStats st;
    CHECK(salva_hip_step(w, 1.0f / 200.0f, g, &st));

    std::atomic<bool> stop{false};
    std::atomic<long> queries{0}, reads{0}, fields{0}, bad{0};
    const float lo[3] = {-1.0f, -100.0f, -1.0f}, hi[3] = {0.3f, 100.0f, 0.3f};
    std::thread q([&] {
        std::vector<uint32_t> kinds(n), slots(n), idx(n);
        while (!stop.load()) {
            const int64_t m = salva_hip_particles_intersecting_aabb(w, lo, hi, n, kinds.data(), slots.data(), idx.data());
            if (m < 0) { ++bad; fprintf(stderr, "query: %s\n", salva_hip_last_error()); break; }
            for (int64_t a = 0; a < m; ++a) if (kinds[a] != 0u || slots[a] != 0u || idx[a] >= n) { ++bad; break; }
            if (m < (int64_t)(7 * 24 * 7) || m > (int64_t)n) ++bad;  // the block's corner column is in the box at every step
            ++queries;
        }
    });
    std::thread rd([&] {
        std::vector<float> a(3 * n), b(3 * n);
        while (!stop.load()) {
            if (salva_hip_get_fluid(w, 0, a.data(), b.data()) < 0) { ++bad; fprintf(stderr, "get_fluid: %s\n", salva_hip_last_error()); break; }
            // a free-falling lattice: every particle has the same velocity (+- the XSPH / solver noise of a block at rest density 0.8)
            double vmin = 1e30, vmax = -1e30;
            for (uint64_t i = 0; i < n; ++i) { const double vy = b[3 * i + 1]; if (!(vy == vy)) { ++bad; break; } vmin = vy < vmin ? vy : vmin; vmax = vy > vmax ? vy : vmax; }
            if (vmax - vmin > 0.5) ++bad;  // (a torn read — half the particles of another step — shows up as a velocity jump of g dt k)
            ++reads;
        }
    });
    std::thread fd([&] {
        std::vector<float> c(n);
        while (!stop.load()) {
            if (salva_hip_get_fluid_field(w, 0, SALVA_HIP_FIELD_NUM_FLUID_CONTACTS, c.data()) < 0) { ++bad; fprintf(stderr, "field: %s\n", salva_hip_last_error()); break; }
            for (uint64_t i = 0; i < n; ++i) if (!(c[i] >= 1.0f && c[i] <= 80.0f)) { ++bad; break; }  // self + lattice neighbours
            ++fields;
        }
    });
    for (int s = 0; s < 150; ++s) {
        CHECK(salva_hip_step(w, 1.0f / 200.0f, g, &st));
        // (a frame loop does other things between two steps; a std::mutex is not fair, and a thread that re-locks at once starves the others)
        std::this_thread::sleep_for(std::chrono::microseconds(500));
    }
    stop = true;
    q.join(); rd.join(); fd.join();
    printf("two_threads: 150 steps beside %ld queries, %ld read-backs, %ld field reads; %ld bad\n", queries.load(), reads.load(), fields.load(), bad.load());
    salva_hip_destroy(w);
    return (bad.load() == 0 && queries.load() > 3 && reads.load() > 3 && fields.load() > 3) ? 0 : 1;
}
