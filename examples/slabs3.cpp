// slabs3.cpp — the x-slab decomposition through the C++ mirror (salva::Comm, LiquidWorld::set_domain / owned / delete_owned):
// a block of fluid in a long lattice tank, cut into NRANKS slabs of grid-cell planes, one LiquidWorld per slab.  Here the ranks
// are host threads of one process over the in-process loopback transport, so the example runs on a single GPU; a multi-GPU run
// is the same code with one process per GPU and Comm::rccl(...) (or Comm::peer(...)) in place of Comm::loopback — see
// bench.py for the id / handle distribution.  Checks that every particle is owned by exactly one rank after the run and that
// all ranks took the same solver iterations (the convergence test is global), then deletes a band of particles collectively.
// No counterpart in the reference (salva is single-process): the scene is examples3d/basic3.rs's block in a longer tank.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../include/salva_hip.hpp"

using namespace salva;

static const Real R = 0.025f, SF = 2.0f, H = R * SF * 2.0f;

static int cell_x(const Vec3& p) { return (int)std::floor(p[0] / H); }  // hgrid.rs:63-71, f32 like the device

int main(int argc, char** argv) {
    const int nranks = argc > 1 ? atoi(argv[1]) : 2, nsteps = argc > 2 ? atoi(argv[2]) : 10;
    try {
        // ---- the undivided scene: a 40 x 12 x 12 block per rank, side by side, in one open tank
        std::vector<Vec3> pos, bpos;
        const int nx = 40 * nranks, ny = 12, nz = 12;
        const Real d = 2.0f * R;
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < ny; ++j)
                for (int k = 0; k < nz; ++k) pos.push_back(Vec3{(i + 0.5f) * d, (j + 0.5f) * d + 2 * d, (k + 0.5f) * d});
        for (int i = -2; i < nx + 2; ++i)  // floor and the two long walls
            for (int k = -2; k < nz + 2; ++k) {
                bpos.push_back(Vec3{(i + 0.5f) * d, 0.5f * d, (k + 0.5f) * d});
                if (k == -2 || k == nz + 1)
                    for (int j = 1; j < ny + 6; ++j) bpos.push_back(Vec3{(i + 0.5f) * d, (j + 0.5f) * d, (k + 0.5f) * d});
            }
        for (int j = 1; j < ny + 6; ++j)  // the two end walls
            for (int k = -1; k < nz + 1; ++k) {
                bpos.push_back(Vec3{(-2 + 0.5f) * d, (j + 0.5f) * d, (k + 0.5f) * d});
                bpos.push_back(Vec3{(nx + 1 + 0.5f) * d, (j + 0.5f) * d, (k + 0.5f) * d});
            }
        // ---- cut the occupied cell planes into equal slabs (salva_amd/dist.py split_slabs balances by count; equal planes do here)
        int lo = 1 << 30, hi = -(1 << 30);
        for (const Vec3& p : pos) { lo = std::min(lo, cell_x(p)); hi = std::max(hi, cell_x(p)); }
        std::vector<std::pair<int, int>> slabs((size_t)nranks);
        for (int r = 0; r < nranks; ++r) slabs[r] = {lo + (hi - lo + 1) * r / nranks, lo + (hi - lo + 1) * (r + 1) / nranks - 1};
        auto owner = [&](const Vec3& p) {
            const int c = cell_x(p);
            for (int r = 0; r + 1 < nranks; ++r)
                if (c <= slabs[r].second) return r;
            return nranks - 1;
        };
        std::vector<uint32_t> offsets((size_t)nranks + 1, 0);
        for (const Vec3& p : pos) ++offsets[(size_t)owner(p) + 1];
        for (int r = 0; r < nranks; ++r) offsets[r + 1] += offsets[r];

        std::vector<Comm> comms = Comm::loopback(nranks);
        std::vector<std::vector<uint32_t>> owned_ids((size_t)nranks);
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> iters((size_t)nranks);
        std::vector<int64_t> left((size_t)nranks, -1);
        std::vector<std::string> errors((size_t)nranks);
        auto rank_main = [&](int r) {
            try {
                LiquidWorld world(DFSPHSolver(), R, SF);
                std::vector<Vec3> mine;
                for (const Vec3& p : pos)
                    if (owner(p) == r) mine.push_back(p);
                Fluid fluid(mine, R, 1000.0f);
                for (Vec3& v : fluid.velocities) v = Vec3{1.5f, 0.0f, 0.0f};  // drifting along x: particles change owner
                fluid.nonpressure_forces.push_back(std::make_shared<XSPHViscosity>(0.5f, 0.0f));
                world.add_fluid(std::move(fluid));
                // the boundary particles within three cell planes of the slab (open-ended at the two outer ranks)
                std::vector<Vec3> near;
                for (const Vec3& p : bpos) {
                    const int c = cell_x(p);
                    if ((r == 0 || c >= slabs[r].first - 3) && (r == nranks - 1 || c <= slabs[r].second + 3)) near.push_back(p);
                }
                world.add_boundary(Boundary(near));
                world.set_domain(comms[r], slabs[r].first, slabs[r].second, offsets[r]);
                const Vec3 gravity{0.0f, -9.81f, 0.0f};
                for (int s = 0; s < nsteps; ++s) {
                    world.step(1.0f / 200.0f, gravity);
                    iters[r].push_back({world.counters().n_divergence_iters, world.counters().n_pressure_iters});
                }
                owned_ids[r] = world.owned().gids;
                // collective removal: every rank passes the same list, each deletes what it owns
                std::vector<uint32_t> band;
                for (uint32_t g = 0; g < (uint32_t)pos.size(); g += 7) band.push_back(g);
                left[r] = world.delete_owned(band);
                world.step(1.0f / 200.0f, gravity);
            } catch (const std::exception& e) {
                errors[r] = e.what();
            }
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < nranks; ++r) threads.emplace_back(rank_main, r);
        for (auto& t : threads) t.join();
        for (int r = 0; r < nranks; ++r)
            if (!errors[r].empty()) { fprintf(stderr, "rank %d: %s\n", r, errors[r].c_str()); return 1; }

        std::vector<int> seen(pos.size(), 0);
        size_t total = 0;
        for (int r = 0; r < nranks; ++r) {
            for (uint32_t g : owned_ids[r]) ++seen[g];
            total += owned_ids[r].size();
            printf("rank %d: slab [%d, %d], owns %zu particles after %d steps (%u uploaded), %lld after the removal\n", r, slabs[r].first,
                   slabs[r].second, owned_ids[r].size(), nsteps, offsets[r + 1] - offsets[r], (long long)left[r]);
        }
        for (int c : seen)
            if (c != 1) { fprintf(stderr, "a particle is owned %d times\n", c); return 1; }
        for (int r = 1; r < nranks; ++r)
            if (iters[r] != iters[0]) { fprintf(stderr, "ranks took different solver iterations\n"); return 1; }
        int64_t kept = 0;
        for (int64_t k : left) kept += k;
        const size_t deleted = (pos.size() + 6) / 7;
        if ((size_t)kept != pos.size() - deleted) { fprintf(stderr, "removal: %lld left, expected %zu\n", (long long)kept, pos.size() - deleted); return 1; }
        printf("slabs3 OK: %zu particles on %d ranks, every one owned once, lock-step iterations, %zu removed collectively\n", total, nranks, deleted);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
