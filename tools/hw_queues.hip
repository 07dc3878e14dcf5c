// tools/hw_queues.hip - which HIP streams share a hardware queue on this runtime?  (the encoder host's stream creation order)
// N streams are created in order; for every ordered pair (i, j) a kernel that spins ~10 ms is launched on stream j and a trivial kernel on stream i right after it: if the
// trivial kernel takes as long as the spin, the two streams are served by one in-order hardware queue.  Prints the matrix ('X' = serialised).
// build: hipcc --offload-arch=gfx950 -O2 -o hw_queues tools/hw_queues.hip      run: ./hw_queues [nstreams] [index of a high-priority stream or -1]
// inside another process (e.g. after `import torch`, whose runtime state changes the picture): hipcc -DHW_QUEUES_LIB -shared -fPIC -o libhw_queues.so ...; ctypes: hw_queues_probe(n, prio)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void spin_kernel(long long ticks, int *out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } if (out) *out = 1; }
__global__ void tiny_kernel(int *out) { if (out) *out = 2; }
extern "C" int hw_queues_probe_pattern(const char *pattern);
extern "C" int hw_queues_probe(int n, int prio)
{
    char pat[65]; int k = 0;
    for (; k < n && k < 64; ++k) pat[k] = k == prio ? 'P' : 'n';
    pat[k] = 0;
    return hw_queues_probe_pattern(pat);
}
// pattern: one letter per stream in creation order, 'n' = normal priority, 'P' = high priority
extern "C" int hw_queues_probe_pattern(const char *pattern)
{
    const int n = (int)strlen(pattern);
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    std::vector<hipStream_t> s(n);
    for (int i = 0; i < n; ++i) {
        if (pattern[i] == 'P') (void)hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, hi);
        else (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    }
    int *d = nullptr;
    (void)hipMalloc(&d, 64);
    int rate_khz = 100000;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const long long ticks = (long long)rate_khz * 10;                   // 10 ms
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s[i], d); }   // every stream has its queue before the measurement
    (void)hipDeviceSynchronize();
    printf("streams %s (creation order; P = high priority), GPU_MAX_HW_QUEUES=%s; row = the waiting stream, column = the stream that spins\n    ", pattern, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    for (int j = 0; j < n; ++j) printf("%2d%s", j, pattern[j] == 'P' ? "p" : " ");
    printf("\n");
    for (int i = 0; i < n; ++i) {
        printf("%2d%s ", i, pattern[i] == 'P' ? "p" : " ");
        for (int j = 0; j < n; ++j) {
            if (i == j) { printf(" . "); continue; }
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s[j], ticks, d + 1);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s[i], d);
            (void)hipStreamSynchronize(s[i]);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            (void)hipStreamSynchronize(s[j]);
            printf(" %c ", ms > 5.0 ? 'X' : '-');
        }
        printf("\n");
    }
    fflush(stdout);
    for (int i = 0; i < n; ++i) (void)hipStreamDestroy(s[i]);
    (void)hipFree(d);
    return 0;
}
#ifndef HW_QUEUES_LIB
int main(int argc, char **argv) { return argc > 1 && (argv[1][0] == 'n' || argv[1][0] == 'P') ? hw_queues_probe_pattern(argv[1]) : hw_queues_probe(argc > 1 ? atoi(argv[1]) : 8, argc > 2 ? atoi(argv[2]) : -1); }
#endif
