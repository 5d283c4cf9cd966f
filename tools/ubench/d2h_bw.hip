// d2h_bw.hip — how fast do 24 MB leave the device?  hipMemcpyAsync (pageable / pinned destination, the world's kind of stream) against
// a copy kernel storing into host-mapped pinned memory.  hipcc --offload-arch=gfx950 -O2 -o d2h_bw d2h_bw.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 24u << 20;
    void *dev, *pin;
    CK(hipMalloc(&dev, bytes));
    CK(hipMemset(dev, 1, bytes));
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    void* page = malloc(bytes);
    memset(page, 0, bytes);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(page, dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        double t1 = now();
        CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev));
        double t3 = now();
        for (int blocks : {16, 64, 256}) {
            double a = now();
            k_copy<<<blocks, 256, 0, s>>>((const uint4*)dev, (uint4*)pin, bytes / 16); CK(hipStreamSynchronize(s));
            double b = now();
            printf("  copy kernel %3d blocks -> pinned: %.2f ms = %.1f GB/s\n", blocks, (b - a) * 1e3, bytes / (b - a) / 1e9);
        }
        printf("rep %d: memcpyAsync -> pageable %.2f ms = %.1f GB/s | -> pinned %.2f ms = %.1f GB/s | -> pinned + event sync %.2f ms\n", rep,
               (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
    }
    return 0;
}
