// overlap_bench.hip -- what overlaps with what on this box: page-locked H2D / D2H copies and kernels on separate streams.
// hipcc --offload-arch=gfx950 -O2 -o tools/overlap_bench tools/overlap_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(unsigned long long cycles, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned x = 0;
    while (wall_clock64() - t0 < cycles) x++;
    if (x == 0xFFFFFFFFu) *sink = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t N = 512ull << 20;
    void *h1, *h2, *d1, *d2; unsigned *sink;
    CK(hipHostMalloc(&h1, N, hipHostMallocDefault)); CK(hipHostMalloc(&h2, N, hipHostMallocDefault));
    memset(h1, 1, N); memset(h2, 2, N);
    CK(hipMalloc(&d1, N)); CK(hipMalloc(&d2, N)); CK(hipMalloc(&sink, 4));
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    const unsigned long long kc = 100000000ull * 20 / 1000;   // wall_clock64 ticks at 100 MHz: 20 ms
    auto sync = [&]() { CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s3)); };
    auto run = [&](const char *name, auto fn) {
        fn(); sync();
        double best = 1e9;
        for (int r = 0; r < 3; r++) { const double t0 = now(); fn(); const double t1 = now(); sync(); const double t2 = now(); if (t2 - t0 < best) best = t2 - t0; if (r == 2) printf("%-58s %7.2f ms (enqueue %.2f ms)\n", name, best * 1e3, (t1 - t0) * 1e3); }
    };
    run("H2D 512 MB", [&]() { CK(hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1)); });
    run("D2H 512 MB", [&]() { CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2)); });
    run("spin kernel 20 ms, 256 blocks", [&]() { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s3, kc, sink); });
    run("spin kernel 20 ms, 8192 blocks x 256 thr", [&]() { hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, s3, kc / 4, sink); });
    run("H2D (s1) || D2H (s2)", [&]() { CK(hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2)); });
    run("H2D (s1) || kernel (s3)", [&]() { CK(hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1)); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s3, kc, sink); });
    run("kernel (s3) || D2H (s2)", [&]() { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s3, kc, sink); CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2)); });
    run("kernel (s3) || H2D (s1) || D2H (s2)", [&]() { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s3, kc, sink); CK(hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2)); });
    run("kernel (s1) || kernel (s3), 128 blocks each", [&]() { hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s1, kc, sink); hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s3, kc, sink); });
    run("full-chip kernel (s3) || H2D (s1) || D2H (s2)", [&]() { hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, s3, kc / 4, sink); CK(hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2)); });
    run("one stream: H2D, kernel, D2H", [&]() { CK(hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1)); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, kc, sink); CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s1)); });
    run("3 streams x (H2D, kernel, D2H) of 1/3 each", [&]() {
        hipStream_t ss[3] = {s1, s2, s3};
        for (int k = 0; k < 3; k++) {
            CK(hipMemcpyAsync((char *)d1 + k * (N / 4), (char *)h1 + k * (N / 4), N / 4, hipMemcpyHostToDevice, ss[k]));
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, ss[k], kc / 3, sink);
            CK(hipMemcpyAsync((char *)h2 + k * (N / 4), (char *)d2 + k * (N / 4), N / 4, hipMemcpyDeviceToHost, ss[k]));
        }
    });
    return 0;
}
