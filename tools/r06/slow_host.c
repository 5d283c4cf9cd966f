/* tools/r06/slow_host.c — LD_PRELOAD shim that makes the HOST slower, by a known amount, at every point where it talks to the GPU:
 * a busy-wait of SLOW_HOST_US microseconds before every hipLaunchKernel / hipMemsetAsync / hipMemcpyAsync.
 *
 * Why: the driver's box of round 5 ran identical kernels 8.8 % slower per step — the step was host-paced
 * there.  Pinning the process to a far socket and loading the other cores does not reproduce that on the boxes this round got
 * (tools/r06/hostile_host.py: within 1 %), a slower launch path does.  The slope d(ms per step) / d(us per launch) is the number
 * of launches that sit on the GPU's critical path; a step that is enqueued ahead of the GPU has slope 0 until the host itself
 * becomes the bottleneck (launches x delay > kernel time).
 *
 *   gcc -O2 -shared -fPIC -o slow_host.so slow_host.c -ldl
 *   SLOW_HOST_US=5 LD_PRELOAD=$PWD/slow_host.so python tools/r06/hostile_host.py
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef struct { uint32_t x, y, z; } dim3_t;
typedef int hipError_t;
typedef void* hipStream_t;

static double delay_us = -1.0;
/* the HIP runtime comes in as a dependency of a dlopen()ed library (RTLD_LOCAL): RTLD_NEXT does not see it — ask the library itself */
static void* real_sym(const char* name) {
    static void* h = NULL;
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    void* p = h ? dlvsym(h, name, "hip_4.2") : NULL;
    if (!p && h) p = dlsym(h, name);
    if (!p) abort();
    return p;
}
static unsigned long long calls = 0;

static void slow(void) {
    if (delay_us < 0.0) {
        const char* e = getenv("SLOW_HOST_US");
        delay_us = e ? atof(e) : 0.0;
    }
    ++calls;
    if (delay_us <= 0.0) return;
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    do {
        clock_gettime(CLOCK_MONOTONIC, &t);
    } while ((t.tv_sec - t0.tv_sec) * 1e6 + (t.tv_nsec - t0.tv_nsec) * 1e-3 < delay_us);
}

hipError_t hipLaunchKernel(const void* f, dim3_t grid, dim3_t block, void** args, size_t shmem, hipStream_t s) {
    static hipError_t (*real)(const void*, dim3_t, dim3_t, void**, size_t, hipStream_t) = NULL;
    if (!real) real = (hipError_t(*)(const void*, dim3_t, dim3_t, void**, size_t, hipStream_t))real_sym("hipLaunchKernel");
    slow();
    return real(f, grid, block, args, shmem, s);
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s) {
    static hipError_t (*real)(void*, int, size_t, hipStream_t) = NULL;
    if (!real) real = (hipError_t(*)(void*, int, size_t, hipStream_t))real_sym("hipMemsetAsync");
    slow();
    return real(dst, value, bytes, s);
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, int kind, hipStream_t s) {
    static hipError_t (*real)(void*, const void*, size_t, int, hipStream_t) = NULL;
    if (!real) real = (hipError_t(*)(void*, const void*, size_t, int, hipStream_t))real_sym("hipMemcpyAsync");
    slow();
    return real(dst, src, bytes, kind, s);
}
unsigned long long slow_host_calls(void) { return calls; }
