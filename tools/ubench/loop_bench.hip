// Speed of light of the neighbour loop itself: the pred-density pair loop of dfsph.hip run from registers and LDS only
// (no global traffic, no barriers inside the timed region), W waves per SIMD, one or two workgroups per CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -I../../salva_amd/csrc loop_bench.hip -o loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "tile.h"

using namespace salva;
struct RecPW { float4 p, w; };

// MODE 0: for_each_ff2 as in k_pred_density; 1: four-contact interleaved step; 2: as 0 but every lane reads slot 0 (no bank
// conflicts, broadcast); 3: as 0 with scalar (unpacked) arithmetic; 4: LDS reads only (no arithmetic)
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_loop(StepCtx c, const uint32_t* __restrict__ lists, float* out, unsigned long long* cyc,
                                              uint32_t S, int reps, uint32_t nq) {
    float4* Lp = reinterpret_cast<float4*>(tile_smem);
    float4* Lw = Lp + S;
    for (uint32_t s = threadIdx.x; s < S; s += blockDim.x) {
        const float fx = (float)(s % 12) * 0.05f, fy = (float)((s / 12) % 12) * 0.05f, fz = (float)(s / 144) * 0.05f;
        Lp[s] = make_float4(fx, fy, fz, 0.1f);
        Lw[s] = make_float4(0.01f * fx, -0.02f * fy, 0.03f * fz, 0.0f);
    }
    ListRegs lr;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < LIST_REGS; ++k) lr.d[k] = lists[((size_t)(blockIdx.x * (blockDim.x >> 6) + wv) * LIST_REGS + k) * 64 + lane];
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < LIST_REGS; ++k) lr.d[k] = 0u;
    }
    const float4 pi = make_float4(0.3f + 0.001f * lane, 0.3f, 0.3f, 0.1f), wi = make_float4(0.01f, 0.02f, 0.03f, 0.0f);
    __syncthreads();
    f2 acc2 = {0.0f, 0.0f};
    float accs = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (MODE != 2) {  // the list is opaque per pass: no address arithmetic may be hoisted out of the timed loop
#pragma unroll
            for (int k = 0; k < LIST_REGS; ++k) asm volatile("" : "+v"(lr.d[k]));
        }
        if (MODE == 0 || MODE == 2) {
            for_each_ff2(c, 0, nq, lr, [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; }, [&](const RecPW& A, const RecPW& B) {
                asm volatile("" ::"v"(A.w.w), "v"(B.w.w));
                const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
                const f2 mj = {A.p.w, B.p.w};
                acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
            });
        } else if (MODE == 1) {
            auto c2 = [&](const RecPW& A, const RecPW& B) {
                asm volatile("" ::"v"(A.w.w), "v"(B.w.w));
                const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
                const f2 mj = {A.p.w, B.p.w};
                acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
            };
            for_each_ff4(c, 0, nq, lr, [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; },
                         [&](const RecPW& A, const RecPW& B, const RecPW& C, const RecPW& D) {
                asm volatile("" ::"v"(A.w.w), "v"(B.w.w), "v"(C.w.w), "v"(D.w.w));
                const f2 dxa = {pi.x - A.p.x, pi.x - B.p.x}, dxb = {pi.x - C.p.x, pi.x - D.p.x};
                const f2 dya = {pi.y - A.p.y, pi.y - B.p.y}, dyb = {pi.y - C.p.y, pi.y - D.p.y};
                const f2 dza = {pi.z - A.p.z, pi.z - B.p.z}, dzb = {pi.z - C.p.z, pi.z - D.p.z};
                f2 ga, gb;
                kernel_grad2x2(dxa * dxa + dya * dya + dza * dza, dxb * dxb + dyb * dyb + dzb * dzb, c.sc, ga, gb);
                const f2 uxa = {wi.x - A.w.x, wi.x - B.w.x}, uxb = {wi.x - C.w.x, wi.x - D.w.x};
                const f2 uya = {wi.y - A.w.y, wi.y - B.w.y}, uyb = {wi.y - C.w.y, wi.y - D.w.y};
                const f2 uza = {wi.z - A.w.z, wi.z - B.w.z}, uzb = {wi.z - C.w.z, wi.z - D.w.z};
                const f2 ma = {A.p.w, B.p.w}, mb = {C.p.w, D.p.w};
                acc2 += (uxa * dxa + uya * dya + uza * dza) * ga * ma;
                acc2 += (uxb * dxb + uyb * dyb + uzb * dzb) * gb * mb;
            }, c2);
        } else if (MODE == 3) {
            for_each_ff2(c, 0, nq, lr, [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; }, [&](const RecPW& A, const RecPW& B) {
                asm volatile("" ::"v"(A.w.w), "v"(B.w.w));
                {
                    const float dx = pi.x - A.p.x, dy = pi.y - A.p.y, dz = pi.z - A.p.z;
                    const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                    accs += ((wi.x - A.w.x) * dx + (wi.y - A.w.y) * dy + (wi.z - A.w.z) * dz) * g * A.p.w;
                }
                {
                    const float dx = pi.x - B.p.x, dy = pi.y - B.p.y, dz = pi.z - B.p.z;
                    const float g = kernel_grad(dx * dx + dy * dy + dz * dz, c.sc);
                    accs += ((wi.x - B.w.x) * dx + (wi.y - B.w.y) * dy + (wi.z - B.w.z) * dz) * g * B.p.w;
                }
            });
        } else if (MODE == 7 || MODE == 8) {
            // independent streams: the arithmetic of MODE 2 (loop-invariant records, no LDS dependence) next to the LDS reads
            // of MODE 4 (results discarded) — do they overlap when nothing ties them together?
            const RecPW I0{Lp[1], Lw[1]}, I1{Lp[2], Lw[2]};
            for_each_ff2(c, 0, nq, lr, [&](uint32_t s) {
                if (MODE == 7) return RecPW{Lp[s], Lw[s]};
                return RecPW{make_float4(reinterpret_cast<const float*>(Lp)[s], 0.f, 0.f, 0.f), make_float4(reinterpret_cast<const float*>(Lw)[s], 0.f, 0.f, 0.f)};
            }, [&](const RecPW& A, const RecPW& B) {
                if (MODE == 7) {
                    asm volatile("" ::"v"(A.p.x), "v"(A.p.y), "v"(A.p.z), "v"(A.p.w), "v"(A.w.x), "v"(A.w.y), "v"(A.w.z), "v"(A.w.w));
                    asm volatile("" ::"v"(B.p.x), "v"(B.p.y), "v"(B.p.z), "v"(B.p.w), "v"(B.w.x), "v"(B.w.y), "v"(B.w.z), "v"(B.w.w));
                } else {
                    asm volatile("" ::"v"(A.p.x), "v"(A.w.x), "v"(B.p.x), "v"(B.w.x));
                }
                float px = pi.x;
                asm volatile("" : "+v"(px));  // keep the arithmetic inside the loop
                const f2 dx = {px - I0.p.x, px - I1.p.x}, dy = {pi.y - I0.p.y, pi.y - I1.p.y}, dz = {pi.z - I0.p.z, pi.z - I1.p.z};
                const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                const f2 ux = {wi.x - I0.w.x, wi.x - I1.w.x}, uy = {wi.y - I0.w.y, wi.y - I1.w.y}, uz = {wi.z - I0.w.z, wi.z - I1.w.z};
                const f2 mj = {I0.p.w, I1.p.w};
                acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
            });
        } else if (MODE == 9 || MODE == 10) {
            // depth-1 software pipeline with the LDS reads of step k+1 spread between the arithmetic of step k
            auto ld = [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; };
            auto c2 = [&](const RecPW& A, const RecPW& B) {
                asm volatile("" ::"v"(A.w.w), "v"(B.w.w));
                const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
                const f2 mj = {A.p.w, B.p.w};
                acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
            };
            constexpr int NS = 9;
            RecPW q[NS + 1][4];
            q[0][0] = ld(lr.d[0] & 0xffffu); q[0][1] = ld(lr.d[0] >> 16);
            q[0][2] = ld(lr.d[1] & 0xffffu); q[0][3] = ld(lr.d[1] >> 16);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (k + 1 < NS) {
                    q[k + 1][0] = ld(lr.d[2 * k + 2] & 0xffffu); q[k + 1][1] = ld(lr.d[2 * k + 2] >> 16);
                    q[k + 1][2] = ld(lr.d[2 * k + 3] & 0xffffu); q[k + 1][3] = ld(lr.d[2 * k + 3] >> 16);
                }
                c2(q[k][0], q[k][1]);
                c2(q[k][2], q[k][3]);
                if (k + 1 < NS) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // one DS read
                        __builtin_amdgcn_sched_group_barrier(0x2, MODE == 9 ? 12 : 6, 0);     // then some VALU
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE == 5 || MODE == 6) {
            // software pipeline with a compile-time trip count (9 steps of 4 contacts): the loads of step k+1 (MODE 5) or
            // k+2 (MODE 6) are issued before the arithmetic of step k
            auto ld = [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; };
            auto c2 = [&](const RecPW& A, const RecPW& B) {
                asm volatile("" ::"v"(A.w.w), "v"(B.w.w));
                const f2 dx = {pi.x - A.p.x, pi.x - B.p.x}, dy = {pi.y - A.p.y, pi.y - B.p.y}, dz = {pi.z - A.p.z, pi.z - B.p.z};
                const f2 g = kernel_grad2(dx * dx + dy * dy + dz * dz, c.sc);
                const f2 ux = {wi.x - A.w.x, wi.x - B.w.x}, uy = {wi.y - A.w.y, wi.y - B.w.y}, uz = {wi.z - A.w.z, wi.z - B.w.z};
                const f2 mj = {A.p.w, B.p.w};
                acc2 += (ux * dx + uy * dy + uz * dz) * g * mj;
            };
            constexpr int NS = 9, D = (MODE == 5) ? 1 : 2;
            RecPW q[NS + 2][4];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                q[k][0] = ld(lr.d[2 * k] & 0xffffu); q[k][1] = ld(lr.d[2 * k] >> 16);
                q[k][2] = ld(lr.d[2 * k + 1] & 0xffffu); q[k][3] = ld(lr.d[2 * k + 1] >> 16);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (k + D < NS) {
                    q[k + D][0] = ld(lr.d[2 * (k + D)] & 0xffffu); q[k + D][1] = ld(lr.d[2 * (k + D)] >> 16);
                    q[k + D][2] = ld(lr.d[2 * (k + D) + 1] & 0xffffu); q[k + D][3] = ld(lr.d[2 * (k + D) + 1] >> 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
                c2(q[k][0], q[k][1]);
                c2(q[k][2], q[k][3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for_each_ff2(c, 0, nq, lr, [&](uint32_t s) { return RecPW{Lp[s], Lw[s]}; }, [&](const RecPW& A, const RecPW& B) {
                asm volatile("" ::"v"(A.p.x), "v"(A.p.y), "v"(A.p.z), "v"(A.p.w), "v"(A.w.x), "v"(A.w.y), "v"(A.w.z), "v"(A.w.w));
                asm volatile("" ::"v"(B.p.x), "v"(B.p.y), "v"(B.p.z), "v"(B.p.w), "v"(B.w.x), "v"(B.w.y), "v"(B.w.z), "v"(B.w.w));
            });
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc2.x + acc2.y + accs;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wv] = t1 - t0;
}

template <int MODE>
void run(const char* name, const StepCtx& c, const uint32_t* lists, float* out, unsigned long long* cyc, uint32_t S, uint32_t nq) {
    const int reps = 40;
    for (int cfg = 1; cfg < 4; cfg += 2) {
        // waves per workgroup, workgroups per CU
        static const int W[5] = {4, 8, 8, 8, 8}, G[5] = {1, 1, 2, 2, 2};
        const int threads = 64 * W[cfg], blocks = 256 * G[cfg];
        const uint32_t lds = S * 32;
        hipFuncSetAttribute((const void*)k_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(threads), lds, 0, c, lists, out, cyc, S, reps, nq);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_loop<MODE>, dim3(blocks), dim3(threads), lds, 0, c, lists, out, cyc, S, reps, nq);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * W[cfg]);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
        const double waves_per_simd = W[cfg] * G[cfg] / 4.0;
        // one "slice pass" = one wave walking nq dwords; a CU runs W*G of them concurrently
        printf("%-34s %2d waves x %d WG/CU (%.0f/SIMD): slice pass %.0f cycles (memtime) per wave -> %.0f cycles per slice pass per SIMD; wall %.1f us -> %.2f us per slice pass per SIMD\n",
               name, W[cfg], G[cfg], waves_per_simd, avg / reps, avg / reps / waves_per_simd, ms * 1e3, ms * 1e3 / reps / waves_per_simd);
    }
}

// index of a lane inside its ds_read_b128 conflict group (MI355X_MICROARCH.md: 4 groups of 16 lanes:
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same +32)
static int group_index(int lane) {
    static const int g0[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
    static const int g1[16] = {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
    const int l = lane & 31;
    for (int k = 0; k < 16; ++k) { if (g0[k] == l) return k; if (g1[k] == l) return k; }
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t S = 1728, nq = 18;
    const int order = argc > 1 ? atoi(argv[1]) : 0;  // 0: ascending slots (as the list builder writes them); 1: rotated by residue;
                                                     // 2: conflict-free by construction (16 consecutive slots per group)
    StepCtx c{};
    c.sc = make_sph_consts(0.1f);
    c.cap_ff = 24;
    std::vector<uint32_t> h((size_t)512 * 16 * LIST_REGS * 64);
    srand(1);
    for (size_t w = 0; w < (size_t)512 * 16; ++w)
        for (int lane = 0; lane < 64; ++lane) {
            // neighbour slots of one particle: ~34 of the ~216 candidates of its 27 cells, ascending; lanes of a wave are
            // consecutive particles (8 per cell), whose candidate windows nearly coincide
            std::vector<uint32_t> sl;
            const uint32_t cell = (uint32_t)(lane / 8), base = (uint32_t)((w * 7919u) % (S - 700)) + cell * 8u;
            for (int row = 0; row < 9 && sl.size() < 2 * nq; ++row) {
                const uint32_t rb = base + (uint32_t)row * 72u;  // rows of 3 cells = 24 candidates
                for (int q = 0; q < 24 && sl.size() < 2 * nq; ++q)
                    if (rand() % 100 < 16) sl.push_back((rb + (uint32_t)q) % S);
            }
            while (sl.size() < 2 * LIST_REGS) sl.push_back(sl.empty() ? 0u : sl.back());
            if (order == 1) {
                const uint32_t rot = (uint32_t)group_index(lane);
                std::vector<uint32_t> head(sl.begin(), sl.begin() + 2 * nq);
                std::stable_sort(head.begin(), head.end(), [&](uint32_t a, uint32_t b) { return ((a + 16u - rot) & 15u) < ((b + 16u - rot) & 15u); });
                std::copy(head.begin(), head.end(), sl.begin());
            } else if (order == 2) {
                for (size_t k = 0; k < sl.size(); ++k) sl[k] = (uint32_t)((w * 131u + k * 37u) % (S - 16)) / 16u * 16u + (uint32_t)group_index(lane);
            }
            for (int k = 0; k < LIST_REGS; ++k) h[(w * LIST_REGS + k) * 64 + lane] = sl[2 * k] | (sl[2 * k + 1] << 16);
        }
    uint32_t* lists; float* out; unsigned long long* cyc;
    hipMalloc(&lists, h.size() * 4); hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&cyc, 512 * 16 * 8);
    hipMemcpy(lists, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    c.nbr_ff = lists;
    printf("list order %d\n", order);
    run<0>("packed pair loop (as shipped)", c, lists, out, cyc, S, nq);
    run<4>("LDS reads only", c, lists, out, cyc, S, nq);
    run<9>("pipelined, 1 read per 12 VALU", c, lists, out, cyc, S, nq);
    run<10>("pipelined, 1 read per 6 VALU", c, lists, out, cyc, S, nq);
    if (order == 0) {
        run<2>("as shipped, all lanes slot 0", c, lists, out, cyc, S, nq);
        run<3>("scalar arithmetic", c, lists, out, cyc, S, nq);
    }
    return 0;
}
