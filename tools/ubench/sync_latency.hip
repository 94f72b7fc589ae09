// How late does the host learn that a kernel is through?  A kernel spins ~50 us, then stores a word into pinned host memory
// (system-scope release) as its last instruction.  The host (a) polls that word, (b) sits in hipStreamSynchronize, (c) polls
// hipStreamQuery; the difference between (a) and the others is what completion through the runtime costs per call.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/sync_latency.hip -o tools/ubench/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>

__global__ void spin_then_flag(volatile uint32_t *flag, uint32_t value, long long clocks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks) __builtin_amdgcn_s_sleep(8);
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store((uint32_t *)flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    uint32_t *flag;
    hipHostMalloc((void **)&flag, 64, hipHostMallocDefault);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const long long clocks = 5000;  // wall_clock64 ticks at 100 MHz: 50 us
    const int n = 2000;
    std::vector<double> poll(n), sync(n), query(n), both(n);
    for (int mode = 0; mode < 4; ++mode)
        for (int i = 0; i < n + 100; ++i) {
            *flag = 0;
            const double t0 = now_us();
            hipLaunchKernelGGL(spin_then_flag, dim3(1), dim3(64), 0, s, flag, 1u, clocks);
            double t_flag = 0, t_done = 0;
            if (mode == 0) {
                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == 0) {}
                t_flag = now_us();
                hipStreamSynchronize(s);
                t_done = now_us();
            } else if (mode == 1) {
                hipStreamSynchronize(s);
                t_done = now_us();
            } else if (mode == 2) {
                while (hipStreamQuery(s) == hipErrorNotReady) {}
                t_done = now_us();
            } else {  // the next launch goes out as soon as the flag is seen: launch-to-launch period
                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == 0) {}
                t_done = now_us();
            }
            if (i >= 100) {
                if (mode == 0) poll[i - 100] = t_flag - t0, both[i - 100] = t_done - t_flag;
                if (mode == 1) sync[i - 100] = t_done - t0;
                if (mode == 2) query[i - 100] = t_done - t0;
                if (mode == 3) both[i - 100] = t_done - t0;
            }
        }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("launch -> flag seen by polling host memory: %.1f us (median); launch -> hipStreamSynchronize returns: %.1f us; launch -> hipStreamQuery says ready: %.1f us; flag-polled calls back to back (no runtime wait): %.1f us per call\n",
           med(poll), med(sync), med(query), med(both));
    return 0;
}
