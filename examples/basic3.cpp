// basic3.cpp — the scene of /root/reference/examples3d/basic3.rs (dam break: 15^3 particles, r = 0.05, DFSPH +
// ArtificialViscosity(1.0, 0.0), gravity -9.81 y, dt = 1/200) driven through the C++ mirror of the salva3d API.
// The rapier colliders of the original (ground + 4 walls, sampled by ray casting) are replaced by lattice shells at the
// same 2r spacing; rendering is replaced by a one-line summary per 50 steps.
#include <cstdio>
#include <cstdlib>

#include "../include/salva_hip.hpp"

using namespace salva;

static const Real PARTICLE_RADIUS = 0.05f, SMOOTHING_FACTOR = 2.0f;

// examples3d/helper.rs:4-20
static Fluid cube_fluid(size_t ni, size_t nj, size_t nk, Real particle_rad, Real density) {
    std::vector<Vec3> points;
    const Vec3 half_extents{(Real)ni * particle_rad, (Real)nj * particle_rad, (Real)nk * particle_rad};
    for (size_t i = 0; i < ni; ++i)
        for (size_t j = 0; j < nj; ++j)
            for (size_t k = 0; k < nk; ++k) {
                const Real x = (Real)i * particle_rad * 2.0f, y = (Real)j * particle_rad * 2.0f, z = (Real)k * particle_rad * 2.0f;
                points.push_back(Vec3{x + particle_rad - half_extents[0], y + particle_rad - half_extents[1],
                                      z + particle_rad - half_extents[2]});
            }
    return Fluid(points, particle_rad, density, InteractionGroups{});
}

static std::vector<Vec3> plate(Real x0, Real x1, Real y0, Real y1, Real z0, Real z1) {  // lattice points of a box region
    std::vector<Vec3> pts;
    const Real d = 2.0f * PARTICLE_RADIUS;
    for (Real x = x0; x <= x1 + 1e-4f; x += d)
        for (Real y = y0; y <= y1 + 1e-4f; y += d)
            for (Real z = z0; z <= z1 + 1e-4f; z += d) pts.push_back(Vec3{x, y, z});
    return pts;
}

int main(int argc, char** argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 200;
    try {
        LiquidWorld world(DFSPHSolver(), PARTICLE_RADIUS, SMOOTHING_FACTOR);
        const Real ground_thickness = 0.2f, ground_half_width = 2.5f, ground_half_height = 0.7f;
        const size_t nparticles = 15;
        Fluid fluid = cube_fluid(nparticles, nparticles, nparticles, PARTICLE_RADIUS, 1000.0f);
        fluid.transform_by(Vec3{0.0f, ground_thickness + (Real)nparticles * PARTICLE_RADIUS, 0.0f});
        fluid.nonpressure_forces.push_back(std::make_shared<ArtificialViscosity>(1.0f, 0.0f));
        const FluidHandle fh = world.add_fluid(std::move(fluid));
        // ground (top face of the 2.5 x 0.2 x 2.5 cuboid) and the inner faces of the four walls
        const Real w = ground_half_width;
        world.add_boundary(Boundary(plate(-w, w, ground_thickness, ground_thickness, -w, w)));
        world.add_boundary(Boundary(plate(-w, w, ground_thickness, 2 * ground_half_height, w, w)));
        world.add_boundary(Boundary(plate(-w, w, ground_thickness, 2 * ground_half_height, -w, -w)));
        world.add_boundary(Boundary(plate(w, w, ground_thickness, 2 * ground_half_height, -w, w)));
        world.add_boundary(Boundary(plate(-w, -w, ground_thickness, 2 * ground_half_height, -w, w)));
        const Vec3 gravity{0.0f, -9.81f, 0.0f};
        for (int s = 0; s < nsteps; ++s) {
            world.step(1.0f / 200.0f, gravity);
            if (s % 50 == 49 || s == nsteps - 1) {
                const Fluid& f = world.fluids()[fh];
                Real ymin = 1e9f, ymax = -1e9f, xspan = 0;
                for (const Vec3& p : f.positions) { ymin = p[1] < ymin ? p[1] : ymin; ymax = p[1] > ymax ? p[1] : ymax; xspan = (p[0] > xspan) ? p[0] : xspan; }
                const SalvaHipStepStats& c = world.counters();
                printf("step %d: %zu particles, y in [%.3f, %.3f], max x %.3f, contacts %llu, iters (div %d, press %d), %.3f ms\n", s + 1,
                       f.num_particles(), ymin, ymax, xspan, (unsigned long long)c.ncontacts, c.n_divergence_iters,
                       c.n_pressure_iters, c.step_ms);
                if (!(ymin > 0.0f)) { fprintf(stderr, "fluid fell through the ground\n"); return 2; }
            }
        }
    } catch (const Error& e) {
        fprintf(stderr, "salva error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
