// host_shape3.cpp — DynamicContactSampling for a collider whose geometry the device library has no code for: a torus.
// The loop of /root/reference/src/integrations/rapier/fluids_pipeline.rs:193-259 runs on the device; its two calls into the shape
// — compute_aabb and project_point_and_get_feature, which a salva3d binding forwards to parry — come back to the host once per
// step through SalvaHipHostShape (include/salva_hip.h).  The solver is DFSPHSolver<Poly6Kernel, SpikyKernel> to show the
// KernelDensity / KernelGradient type parameters (dfsph_solver.rs:17-20) at the same time.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../include/salva_hip.hpp"

using namespace salva;

struct Torus {  // about the world y axis, centred at `c`
    float c[3], R, r;
};
static void torus_aabb(void* user, float* mins, float* maxs) {
    const Torus& t = *static_cast<const Torus*>(user);
    const float e[3] = {t.R + t.r, t.r, t.R + t.r};
    for (int a = 0; a < 3; ++a) { mins[a] = t.c[a] - e[a]; maxs[a] = t.c[a] + e[a]; }
}
static void torus_project(void* user, uint32_t n, const float* pts, float* proj, uint8_t* inside) {
    const Torus& t = *static_cast<const Torus*>(user);
    for (uint32_t k = 0; k < n; ++k) {
        const double x = pts[3 * k] - t.c[0], y = pts[3 * k + 1] - t.c[1], z = pts[3 * k + 2] - t.c[2];
        const double planar = std::sqrt(x * x + z * z);
        const double ux = planar > 1e-12 ? x / planar : 1.0, uz = planar > 1e-12 ? z / planar : 0.0;
        const double dx = x - ux * t.R, dy = y, dz = z - uz * t.R;  // from the nearest point of the ring
        const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
        const double nx = d > 1e-12 ? dx / d : 0.0, ny = d > 1e-12 ? dy / d : 1.0, nz = d > 1e-12 ? dz / d : 0.0;
        proj[3 * k] = (float)(t.c[0] + ux * t.R + nx * t.r);
        proj[3 * k + 1] = (float)(t.c[1] + ny * t.r);
        proj[3 * k + 2] = (float)(t.c[2] + uz * t.R + nz * t.r);
        inside[k] = d <= t.r ? 1 : 0;
    }
}

int main(int argc, char** argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 120;
    const Real r = 0.025f, d = 2.0f * r, dt = 1.0f / 200.0f;
    try {
        LiquidWorld world(DFSPHSolverT<Poly6Kernel, SpikyKernel>(), r, 2.0f);
        Torus torus{{0.0f, 0.0f, 0.0f}, 0.22f, 0.07f};
        std::vector<Vec3> block;
        const int n = 12;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k)
            block.push_back(Vec3{(i - n / 2) * d + r, torus.r + 0.01f + j * d + r, (k - n / 2) * d + r});
        Fluid fluid(block, r, 1000.0f, InteractionGroups{});
        fluid.nonpressure_forces.push_back(std::make_shared<XSPHViscosity>(0.5f, 0.5f));
        const FluidHandle fh = world.add_fluid(std::move(fluid));
        const BoundaryHandle bh = world.add_boundary(Boundary::dynamic_host_shape(SalvaHipHostShape{torus_aabb, torus_project, &torus}));
        const Vec3 gravity{0.0f, -9.81f, 0.0f};
        size_t most = 0;
        for (int s = 0; s < nsteps; ++s) {
            world.step(dt, gravity);
            world.sync_boundary(bh);
            const Boundary& b = world.boundaries()[bh];
            most = b.num_particles() > most ? b.num_particles() : most;
            double off = 0.0;
            for (const Vec3& p : b.positions) {  // every emitted boundary particle lies on the torus
                const double planar = std::sqrt((double)p[0] * p[0] + (double)p[2] * p[2]);
                off = std::fmax(off, std::fabs(std::sqrt((planar - torus.R) * (planar - torus.R) + (double)p[1] * p[1]) - torus.r));
            }
            if (s % 30 == 29 || s == nsteps - 1) {
                double deepest = 1e9;
                size_t below = 0;
                for (const Vec3& p : world.fluids()[fh].positions) {
                    const double planar = std::sqrt((double)p[0] * p[0] + (double)p[2] * p[2]);
                    deepest = std::fmin(deepest, std::sqrt((planar - torus.R) * (planar - torus.R) + (double)p[1] * p[1]) - torus.r);
                    below += p[1] < -0.3f;
                }
                printf("step %d: %zu boundary samples (max %zu), worst sample off the surface %.2e, deepest fluid particle %.3f r, %zu particles below the ring\n",
                       s + 1, b.num_particles(), most, off, deepest / r, below);
            }
        }
    } catch (const Error& e) {
        fprintf(stderr, "salva_hip error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
