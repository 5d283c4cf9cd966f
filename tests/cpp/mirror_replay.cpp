// mirror_replay.cpp — test helper: replays an operation script (written by tests/test_mirrors_gpu.py) through the C++ mirror
// of the salva3d API (include/salva_hip.hpp) and dumps the final state.  The Python mirror replays the same script; both
// drive the same library, so the two dumps must be identical bit for bit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "../../include/salva_hip.hpp"

using namespace salva;

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: mirror_replay script out\n"); return 2; }
    std::ifstream in(argv[1]);
    FILE* out = fopen(argv[2], "wb");
    if (!in || !out) { fprintf(stderr, "cannot open files\n"); return 2; }
    try {
        std::string solver;
        Real radius;
        in >> solver >> radius;
        PressureSolver ps = solver == "iisph" ? (PressureSolver)IISPHSolver() : (PressureSolver)DFSPHSolver();
        LiquidWorld world(ps, radius, 2.0f);
        std::string op;
        while (in >> op) {
            if (op == "ADD_FLUID") {
                Real density; size_t n;
                in >> density >> n;
                std::vector<Vec3> pos(n), vel(n);
                for (size_t i = 0; i < n; ++i) in >> pos[i][0] >> pos[i][1] >> pos[i][2] >> vel[i][0] >> vel[i][1] >> vel[i][2];
                Fluid f(pos, radius, density, InteractionGroups{});
                f.velocities = vel;
                f.nonpressure_forces.push_back(std::make_shared<XSPHViscosity>(0.5f, 0.2f));
                world.add_fluid(std::move(f));
            } else if (op == "ADD_PARTICLES") {
                size_t slot, n; int has_vel;
                in >> slot >> n >> has_vel;
                std::vector<Vec3> pos(n), vel(n, Vec3{0, 0, 0});
                for (size_t i = 0; i < n; ++i) {
                    in >> pos[i][0] >> pos[i][1] >> pos[i][2];
                    if (has_vel) in >> vel[i][0] >> vel[i][1] >> vel[i][2];
                }
                world.fluids()[slot].add_particles(pos, has_vel ? &vel : nullptr);
            } else if (op == "DELETE") {
                size_t slot, k;
                in >> slot >> k;
                for (size_t j = 0; j < k; ++j) { size_t i; in >> i; world.fluids()[slot].delete_particle_at_next_timestep(i); }
            } else if (op == "REMOVE_FLUID") {
                size_t slot; in >> slot;
                world.remove_fluid(slot);
            } else if (op == "SHIFT_VELOCITIES") {
                size_t slot; Real dx;
                in >> slot >> dx;
                Fluid& f = world.fluids()[slot];
                for (Vec3& v : f.velocities) v[0] += dx;
                f.mark_dirty(SALVA_HIP_DIRTY_VELOCITIES);
            } else if (op == "ADD_BOUNDARY") {
                size_t n; in >> n;
                std::vector<Vec3> pos(n);
                for (size_t i = 0; i < n; ++i) in >> pos[i][0] >> pos[i][1] >> pos[i][2];
                world.add_boundary(Boundary(pos));
            } else if (op == "REMOVE_BOUNDARY") {
                size_t slot; in >> slot;
                world.remove_boundary(slot);
            } else if (op == "STEP") {
                Real dt; Vec3 g;
                in >> dt >> g[0] >> g[1] >> g[2];
                world.step(dt, g);
            } else {
                fprintf(stderr, "unknown op %s\n", op.c_str());
                return 2;
            }
        }
        const uint64_t nf = world.fluids().size();
        fwrite(&nf, sizeof nf, 1, out);
        for (Fluid& f : world.fluids()) {
            const uint64_t n = f.num_particles();
            fwrite(&n, sizeof n, 1, out);
            if (n) { fwrite(f.positions[0].data(), sizeof(Real) * 3, n, out); fwrite(f.velocities[0].data(), sizeof(Real) * 3, n, out); }
        }
    } catch (const Error& e) {
        fprintf(stderr, "salva_hip error %d: %s\n", e.code, e.what());
        return 1;
    }
    fclose(out);
    return 0;
}
