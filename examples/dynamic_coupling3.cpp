// dynamic_coupling3.cpp — coupling3.cpp with ColliderSampling::DynamicContactSampling instead of sample points
// (/root/reference/src/integrations/rapier/fluids_pipeline.rs:42-43, 193-259): the dynamic body is a ball whose boundary
// particles are the projections of the nearby fluid particles onto it, re-emitted inside every step on the device
// (salva_hip_set_boundary_dynamic_sampling).  Per step the coupling still sends one pose and receives one wrench.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../include/salva_hip.hpp"

using namespace salva;

struct Body {  // the slice of rapier's RigidBody the coupling touches
    Vec3 translation{0, 0, 0}, linvel{0, 0, 0}, angvel{0, 0, 0};
    Real mass = 1.0f, inertia = 1.0f;  // isotropic inertia: no frame change needed for the torque impulse
    SalvaHipRigidPose pose() const {
        SalvaHipRigidPose p{};
        for (int k = 0; k < 3; ++k) { p.translation[k] = translation[k]; p.linvel[k] = linvel[k]; p.angvel[k] = angvel[k]; p.world_com[k] = translation[k]; }
        p.rotation[3] = 1.0f;  // the example keeps the box axis aligned (small angular velocities are only reported)
        p.has_body = 1; p.is_dynamic = 1;
        return p;
    }
};

int main(int argc, char** argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 200;
    const Real r = 0.025f, d = 2.0f * r, dt = 1.0f / 200.0f;
    try {
        LiquidWorld world(DFSPHSolver(), r, 2.0f);
        std::vector<Vec3> pool, shell;
        const int nx = 16, ny = 8, nz = 16;
        for (int i = 0; i < nx; ++i) for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k)
            pool.push_back(Vec3{(i - nx / 2) * d + r, j * d + r + d, (k - nz / 2) * d + r});
        for (int i = -1; i <= nx; ++i) for (int j = 0; j <= ny + 6; ++j) for (int k = -1; k <= nz; ++k)
            if (i == -1 || i == nx || j == 0 || k == -1 || k == nz) shell.push_back(Vec3{(i - nx / 2) * d + r, j * d + r, (k - nz / 2) * d + r});
        Fluid fluid(pool, r, 1000.0f, InteractionGroups{});
        fluid.nonpressure_forces.push_back(std::make_shared<ArtificialViscosity>(1.0f, 0.5f));
        const FluidHandle fh = world.add_fluid(std::move(fluid));
        world.add_boundary(Boundary(shell));
        const Real ball_radius = 2.0f * d;
        const BoundaryHandle bh = world.add_boundary(Boundary::dynamic_ball(ball_radius));
        Body body;
        body.translation = Vec3{0.0f, (ny + 5) * d, 0.0f};
        body.mass = 0.5f * 1000.0f * 4.18879f * ball_radius * ball_radius * ball_radius;  // half the density of the fluid
        body.inertia = 0.4f * body.mass * ball_radius * ball_radius;
        ColliderCouplingSet coupling;
        coupling.register_coupling(bh, [&] { return body.pose(); }, [&](const Vec3& j, const Vec3& tj) {
            for (int k = 0; k < 3; ++k) { body.linvel[k] += j[k] / body.mass; body.angvel[k] += tj[k] / body.inertia; }
        });
        const Vec3 gravity{0.0f, -9.81f, 0.0f};
        for (int s = 0; s < nsteps; ++s) {
            world.step_with_coupling(dt, gravity, coupling);
            for (int k = 0; k < 3; ++k) { body.linvel[k] += gravity[k] * dt; body.translation[k] += body.linvel[k] * dt; }
            if (s % 50 == 49 || s == nsteps - 1) {
                world.sync_boundary(bh);
                const Boundary& b = world.boundaries()[bh];
                Real ymin = b.positions.empty() ? 0.0f : 1e9f;
                for (const Vec3& p : b.positions) ymin = p[1] < ymin ? p[1] : ymin;
                printf("step %d: ball y %.4f vy %.4f |angvel| %.4f, lowest sample y %.4f, %zu samples, fluid %zu particles, %d pressure iterations\n",
                       s + 1, body.translation[1], body.linvel[1],
                       std::sqrt(body.angvel[0] * body.angvel[0] + body.angvel[1] * body.angvel[1] + body.angvel[2] * body.angvel[2]), ymin,
                       b.num_particles(), world.fluids()[fh].num_particles(), world.counters().n_pressure_iters);
            }
        }
    } catch (const Error& e) {
        fprintf(stderr, "salva_hip error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
