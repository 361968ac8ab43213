// DIAGNOSTIC micro-benchmark (not part of the product): back-to-back launch cost of an (almost) empty kernel as a
// function of grid shape, dynamic LDS, register budget and kernel-argument size -- the floor under a per-step launch.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[76]; };   // ~304 B, like DevBatch
template <int VG>
__global__ __launch_bounds__(1024) void k_small(int *o, int n) { if (n == -1) o[threadIdx.x] = VG; }
__global__ __launch_bounds__(256, 4) void k_big(Big b, int *o, int n) { if (n == -1) o[threadIdx.x] = b.v[threadIdx.x % 76]; }
__global__ __launch_bounds__(256, 4) void k_regs(int *o, int n) {
    // force a large register allocation without doing work
    asm volatile("" ::: "v127");
    if (n == -1) o[threadIdx.x] = 1;
}
template <class F> static float timeit(F &&launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 2000; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / 2000;
}
int main() {
    int *o; hipMalloc(&o, 4096);
    Big b{};
    printf("1024 x 256, no LDS, small args      %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_small<0>, dim3(1024), dim3(256), 0, 0, o, 0); }));
    printf("1024 x 256, 9 KB LDS, small args    %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_small<0>, dim3(1024), dim3(256), 9216, 0, o, 0); }));
    printf("1024 x 256, no LDS, 304 B args      %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), 0, 0, b, o, 0); }));
    printf("1024 x 256, 9 KB LDS, 304 B args    %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_big, dim3(1024), dim3(256), 9216, 0, b, o, 0); }));
    printf("1024 x 256, 128 VGPRs               %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_regs, dim3(1024), dim3(256), 0, 0, o, 0); }));
    printf("1024 x 256, 128 VGPRs, 9 KB LDS     %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_regs, dim3(1024), dim3(256), 9216, 0, o, 0); }));
    printf("512 x 512, no LDS                   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_small<1>, dim3(512), dim3(512), 0, 0, o, 0); }));
    printf("256 x 1024, no LDS                  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_small<2>, dim3(256), dim3(1024), 0, 0, o, 0); }));
    printf("256 x 1024, 36 KB LDS               %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_small<2>, dim3(256), dim3(1024), 36864, 0, o, 0); }));
    printf("256 x 256                           %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_small<0>, dim3(256), dim3(256), 0, 0, o, 0); }));
    printf("4096 x 64                           %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_small<3>, dim3(4096), dim3(64), 0, 0, o, 0); }));
    return 0;
}
