// VALU issue-cost microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the instruction kinds the
// neighbour loops are made of.  One workgroup per CU, W waves per SIMD; every wave runs REP x 64 independent
// instructions of one kind (8 independent register chains) between two s_memtime reads.
//   hipcc -O3 --offload-arch=gfx950 valu_bench.hip -o valu_bench && ./valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int reps, float seed) {
    float a[8], b[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], q[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f + i; p[i] = f2{a[i], b[i]}; q[i] = f2{b[i], a[i]}; }
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 2654435761u + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define OP(i)                                                                                                   \
    if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(a[(i + 1) & 7]));          \
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i]), "v"(p[(i + 1) & 7]));       \
    if (KIND == 2) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));                                   \
    if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(q[i]));                                \
    if (KIND == 4) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));                                   \
    if (KIND == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(q[i]));                                \
    if (KIND == 6) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));                                                   \
    if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]));                          \
    if (KIND == 8) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));                                   \
    if (KIND == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");                       \
    if (KIND == 10) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(u[i]));                                           \
    if (KIND == 11) asm volatile("v_and_b32 %0, 0xffff0, %0" : "+v"(u[i]));                                         \
    if (KIND == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));                                      \
    if (KIND == 13) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                        \
    if (KIND == 14) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));                \
    if (KIND == 15) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                                                 \
    if (KIND == 16) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                                  \
    if (KIND == 17) asm volatile("v_bfe_u32 %0, %0, 16, 16" : "+v"(u[i]));                                          \
    if (KIND == 18) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[i]) : "v"(q[i]));               \
    if (KIND == 19) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]), "v"(a[(i + 1) & 7]) : "vcc"); \
    if (KIND == 20) asm volatile("v_cmp_gt_f32 s[20:21], %1, %2\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b[i]), "v"(a[(i + 1) & 7]) : "s20", "s21"); \
    if (KIND == 21) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));    \
    if (KIND == 22) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(a[i]) : "v"(b[i]));                          \
    if (KIND == 23) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));                                   \
    if (KIND == 24) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));         \
    if (KIND == 25) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));             \
    if (KIND == 26) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));                                   \
    if (KIND == 27) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %4, %4, %1" : "+v"(p[i]), "+v"(a[i]) : "v"(q[i]), "v"(p[(i + 1) & 7]), "v"(b[i])); \
    if (KIND == 28) asm volatile("v_mul_f32 %0, %2, %0\n\tv_max_f32 %1, %2, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(b[(i + 1) & 7]));  \
    if (KIND == 29) asm volatile("v_mul_f32 %0, %2, %0\n\tv_lshlrev_b32 %1, 4, %1" : "+v"(a[i]), "+v"(u[i]) : "v"(b[i]));          \
    if (KIND == 30) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc", "s20", "s21");   \
    if (KIND == 31) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a[i]), "v"(u[i]) : "vcc");                    \
    if (KIND == 32) asm volatile("v_mul_f32 %0, %1, %0 \n s_nop 0" : "+v"(a[i]) : "v"(b[i]));                          \
    if (KIND == 33) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));              \
    if (KIND == 34) asm volatile("v_sub_f32 %0, s20, %1" : "=v"(a[i]) : "v"(b[i]));                                  \
    if (KIND == 35) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p[i]) : "v"(q[i]), "v"(q[(i + 1) & 7])); \
    if (KIND == 36) asm volatile("v_and_b32 %0, 0xffff, %1\n\tv_lshrrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
            R8(OP)
#undef OP
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int reps = 200;
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps, blocks = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
        const double ninst = (double)reps * 64;
        // per SIMD: wps waves each issue ninst instructions in `avg` cycles
        printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (memtime), kernel %.1f us -> %.2f ns per instr-slot\n", name, wps,
               avg / (ninst * wps), ms * 1e3, ms * 1e6 / (ninst * wps));
    }
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<2>("v_mul_f32", out, cyc);
    run<3>("v_pk_mul_f32", out, cyc);
    run<4>("v_sub_f32", out, cyc);
    run<5>("v_pk_add_f32", out, cyc);
    run<6>("v_rsq_f32", out, cyc);
    run<15>("v_sqrt_f32", out, cyc);
    run<16>("v_rcp_f32", out, cyc);
    run<7>("v_cndmask_b32", out, cyc);
    run<8>("v_max_f32", out, cyc);
    run<9>("v_cmp_gt_f32", out, cyc);
    run<10>("v_lshlrev_b32", out, cyc);
    run<11>("v_and_b32 (literal)", out, cyc);
    run<12>("v_mov_b32", out, cyc);
    run<13>("v_add_u32", out, cyc);
    run<14>("v_lshl_add_u32", out, cyc);
    run<17>("v_bfe_u32", out, cyc);
    run<18>("v_pk_mul_f32 op_sel_hi", out, cyc);
    run<19>("[x2] v_cmp vcc + v_cndmask vcc", out, cyc);
    run<20>("[x2] v_cmp sgpr + v_cndmask sgpr", out, cyc);
    run<21>("v_cndmask_b32 d!=s (vcc)", out, cyc);
    run<22>("v_cndmask_b32 0,v (vcc)", out, cyc);
    run<23>("v_min_f32", out, cyc);
    run<24>("v_med3_f32", out, cyc);
    run<25>("v_fmac_f32", out, cyc);
    run<26>("v_add_f32", out, cyc);
    run<27>("[x2] v_pk_fma + v_fma", out, cyc);
    run<28>("[x2] v_mul + v_max", out, cyc);
    run<29>("[x2] v_mul + v_lshlrev", out, cyc);
    run<30>("[x2] v_cmp vcc + v_cmp sgpr", out, cyc);
    run<31>("v_cmp_class_f32", out, cyc);
    run<32>("v_mul_f32 + s_nop 0", out, cyc);
    run<33>("v_sub_f32 d!=s", out, cyc);
    run<34>("v_sub_f32 sgpr src", out, cyc);
    run<35>("v_pk_add_f32 neg (sub)", out, cyc);
    run<36>("[x2] v_and + v_lshrrev", out, cyc);
    return 0;
}
