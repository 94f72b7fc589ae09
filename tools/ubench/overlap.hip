// Micro-benchmark: does a host->device copy on one stream overlap a kernel on another?  (Behind NOTEBOOK.md's statement
// on the pipelined host path.)  build: hipcc --offload-arch=gfx950 -O2 -o overlap overlap.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin(long long cycles, int *sink) {
    const long long t0 = clock64();
    int x = 0;
    while (clock64() - t0 < cycles) x += 1;
    if (x == -1) *sink = x;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = 32u << 20;
    char *h = nullptr, *d = nullptr;
    int *sink = nullptr;
    hipHostMalloc((void **)&h, bytes, hipHostMallocDefault);
    hipMalloc((void **)&d, bytes);
    hipMalloc((void **)&sink, 4);
    hipStream_t sk, sc;
    hipStreamCreateWithFlags(&sk, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sc, hipStreamNonBlocking);
    const long long cycles = 2000000;  // ~1 ms at 100 MHz clock64 ... calibrated below
    for (int blocks : {256, 1024, 4096}) {
        for (int threads : {64, 256}) {
            // warm
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, sk, 1000, sink);
            hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sc);
            hipDeviceSynchronize();
            double t = now_us();
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, sk, cycles, sink);
            hipStreamSynchronize(sk);
            const double tk = now_us() - t;
            t = now_us();
            hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sc);
            hipStreamSynchronize(sc);
            const double tc = now_us() - t;
            t = now_us();
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(threads), 0, sk, cycles, sink);
            hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sc);
            hipStreamSynchronize(sc);
            const double tc_done = now_us() - t;
            hipStreamSynchronize(sk);
            const double both = now_us() - t;
            printf("grid %5d x %3d: kernel alone %8.1f us, 32 MB H2D alone %7.1f us, together %8.1f us (copy done at %8.1f us)\n",
                   blocks, threads, tk, tc, both, tc_done);
        }
    }
    return 0;
}
