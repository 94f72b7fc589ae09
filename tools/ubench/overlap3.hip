// Micro-benchmark: three streams, each [H2D 20 MB -> kernel ~2.5 ms -> D2H 1.7 MB], enqueued back to back: the shape of
// the pipelined host path.  Does chunk i+1's H2D overlap chunk i's kernel?  Variants move the D2H elsewhere.
// build: hipcc --offload-arch=gfx950 -O2 -o overlap3 overlap3.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
    const size_t in_bytes = 20u << 20, out_bytes = 1700u << 10;
    const int N = 6, S = 3;
    char *h[S], *d[S], *ho[S], *dout[S];
    int *sink = nullptr;
    hipMalloc((void **)&sink, 4);
    hipStream_t st[S], copy_out;
    hipEvent_t done_k[N];
    for (int i = 0; i < S; ++i) {
        hipHostMalloc((void **)&h[i], in_bytes + out_bytes, hipHostMallocDefault);
        hipHostMalloc((void **)&ho[i], out_bytes, hipHostMallocDefault);
        hipMalloc((void **)&d[i], in_bytes + out_bytes);
        hipMalloc((void **)&dout[i], out_bytes);
        hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    hipStreamCreateWithFlags(&copy_out, hipStreamNonBlocking);
    for (int i = 0; i < N; ++i) hipEventCreateWithFlags(&done_k[i], hipEventDisableTiming);
    const long long cycles = 5800000;  // ~2.5 ms (clock64 counts shader clocks, ~2.3 GHz)
    for (int variant = 0; variant < 8; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            const double t = now_us();
            for (int i = 0; i < N; ++i) {
                const int s = i % S;
                if (variant >= 4 && i) {  // the host needs ~450 us to plan and stage the next chunk
                    const double w = now_us();
                    while (now_us() - w < 450) {}
                }
                hipMemcpyAsync(d[s], h[s], in_bytes, hipMemcpyHostToDevice, st[s]);
                hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st[s], cycles, sink);
                if (variant == 0 || variant == 4) hipMemcpyAsync(ho[s], dout[s], out_bytes, hipMemcpyDeviceToHost, st[s]);
                if (variant == 7)  // results land in the allocation the inputs came from (one arena + mirror per slot)
                    hipMemcpyAsync(h[s] + in_bytes, d[s] + in_bytes, out_bytes, hipMemcpyDeviceToHost, st[s]);
                if (variant == 2 || variant == 6) {  // D2H on a stream of its own, ordered by an event
                    hipEventRecord(done_k[i], st[s]);
                    hipStreamWaitEvent(copy_out, done_k[i], 0);
                    hipMemcpyAsync(ho[s], dout[s], out_bytes, hipMemcpyDeviceToHost, copy_out);
                }
                if (variant == 3) {  // D2H only after the host has seen the kernel finish (what a late enqueue would do)
                    if (i >= 1) {
                        const int p = (i - 1) % S;
                        hipStreamSynchronize(st[p]);
                        hipMemcpyAsync(ho[p], dout[p], out_bytes, hipMemcpyDeviceToHost, st[p]);
                    }
                }
            }
            hipDeviceSynchronize();
            if (rep)
                printf("variant %d (%s): %d chunks in %8.1f us (serial would be ~%d us, overlapped ~%d us)\n", variant,
                       variant == 0   ? "H2D, kernel, D2H on the chunk's stream"
                       : variant == 1 ? "no D2H at all"
                       : variant == 2 ? "D2H on its own stream after an event"
                       : variant == 3 ? "D2H enqueued after the host saw the kernel finish"
                       : variant == 4 ? "as 0, 450 us of host work between chunks"
                       : variant == 5 ? "as 1 (no D2H), 450 us of host work between chunks"
                       : variant == 7 ? "as 4, D2H into the pinned allocation the inputs came from"
                                      : "as 2 (D2H on its own stream), 450 us of host work between chunks",
                       N, now_us() - t, N * (380 + 2500 + 35), 380 + N * 2500);
        }
    }
    return 0;
}
