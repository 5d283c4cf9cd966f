// faucet3.cpp — the control flow of /root/reference/examples3d/faucet3.rs through the C++ mirror: every few steps a
// nozzle adds a sheet of particles with a downward velocity (`Fluid::add_particles`, fluid.rs:126-150) and everything that
// fell below a kill plane is deleted (`delete_particle_at_next_timestep`, fluid.rs:71-86).  Both edits are replayed on the
// device (salva_hip_add_particles / salva_hip_delete_particles): the particles the fluid already holds never cross PCIe.
// XSPHViscosity(0.5, 0.0) + Akinci2013SurfaceTension(1.0, 10.0) as in the original (faucet3.rs:37-38).
#include <cstdio>
#include <cstdlib>

#include "../include/salva_hip.hpp"

using namespace salva;

int main(int argc, char** argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 300;
    const Real r = 0.025f, d = 2.0f * r;
    try {
        LiquidWorld world(DFSPHSolver(), r, 2.0f);
        Fluid fluid(std::vector<Vec3>{}, r, 1000.0f, InteractionGroups{});
        fluid.nonpressure_forces.push_back(std::make_shared<XSPHViscosity>(0.5f, 0.0f));
        fluid.nonpressure_forces.push_back(std::make_shared<Akinci2013SurfaceTension>(1.0f, 10.0f));
        const FluidHandle fh = world.add_fluid(std::move(fluid));
        std::vector<Vec3> floor;  // a 1 m x 1 m plate at y = 0
        for (int i = -10; i <= 10; ++i) for (int k = -10; k <= 10; ++k) floor.push_back(Vec3{i * d, 0.0f, k * d});
        world.add_boundary(Boundary(floor));
        std::vector<Vec3> nozzle, nozzle_vel;
        for (int i = -2; i <= 2; ++i) for (int k = -2; k <= 2; ++k) { nozzle.push_back(Vec3{i * d, 0.6f, k * d}); nozzle_vel.push_back(Vec3{0.0f, -1.0f, 0.0f}); }
        const Vec3 gravity{0.0f, -9.81f, 0.0f};
        size_t added = 0, deleted = 0;
        for (int s = 0; s < nsteps; ++s) {
            Fluid& f = world.fluids()[fh];  // fluids_mut().get_mut(handle) in the reference
            if (s % 10 == 0) { f.add_particles(nozzle, &nozzle_vel); added += nozzle.size(); }
            for (size_t i = 0; i < f.num_particles(); ++i)
                if (f.positions[i][1] < -0.3f || f.positions[i][0] * f.positions[i][0] + f.positions[i][2] * f.positions[i][2] > 1.0f) {
                    f.delete_particle_at_next_timestep(i);
                    ++deleted;
                }
            world.step(1.0f / 200.0f, gravity);
            if (s % 100 == 99 || s == nsteps - 1) {
                const Fluid& g = world.fluids()[fh];
                Real ymin = 1e9f, ymax = -1e9f;
                for (const Vec3& p : g.positions) { ymin = p[1] < ymin ? p[1] : ymin; ymax = p[1] > ymax ? p[1] : ymax; }
                printf("step %d: %zu particles (added %zu, deleted %zu), y in [%.3f, %.3f], %d div / %d pressure iterations\n", s + 1,
                       g.num_particles(), added, deleted, ymin, ymax, world.counters().n_divergence_iters, world.counters().n_pressure_iters);
            }
        }
    } catch (const Error& e) {
        fprintf(stderr, "salva_hip error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
