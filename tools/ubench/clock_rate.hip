// DIAGNOSTIC: shader clock under load.  s_memtime ticks at the shader clock, s_memrealtime at a constant 100 MHz;
// their ratio over a busy loop (idle-ish VALU loop, and a loop of back-to-back f32 MFMAs on every CU) gives the
// clock the kernels really run at.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void spin(int mode, int iters, long long *out, float *sink) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 64; ++j) a = a * 1.0001f + b;
        }
    }
    float s = a;
    for (int r = 0; r < 16; ++r) s += acc[r];
    long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
int main() {
    long long *out; float *sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(mode == 1 ? 256 : 256), dim3(256), 0, 0, mode, mode == 1 ? 20000 : 20000, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("%s: %.3f ms, s_memtime %lld ticks, s_memrealtime %lld ticks -> memtime %.1f MHz, realtime %.1f MHz%s\n",
                   mode ? "f32 MFMA on 256 CUs x 4 waves" : "VALU chain", ms, h[0], h[1], h[0] / ms / 1e3, h[1] / ms / 1e3,
                   mode ? "" : "");
            if (mode == 1) printf("   MFMA: %.1f cycles(memtime) per MFMA, %.1f ns per MFMA -> %.1f TFLOP/s chip\n", (double)h[0] / (20000.0 * 16), ms * 1e6 / (20000.0 * 16),
                                  256 * 4 * 20000.0 * 16 * 4096 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
