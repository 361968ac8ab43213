// DIAGNOSTIC micro-benchmark (not part of the product): issue rate of scalar / vector / mixed instruction streams as
// a function of the waves per SIMD -- is the scalar unit shared by the four SIMDs of a CU?  Cycles (s_memtime) per
// instruction per wave for W = 1, 2, 4 waves per SIMD (every CU filled the same way).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 1024
template <int MODE>
__global__ void k(int *out, unsigned long long *cyc, int seed) {
    int s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    float v0 = (float)threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        if (MODE == 0 || MODE == 2) {      // 4 independent scalar adds
            asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5\n\ts_add_u32 %2, %2, 7\n\ts_add_u32 %3, %3, 9"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
        if (MODE == 1 || MODE == 2) {      // 4 independent vector adds
            asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
        if (MODE == 3) {                   // dependent scalar chain
            asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7\n\ts_add_u32 %0, %0, 9" : "+s"(s0) : : "scc");
        }
        if (MODE == 4) {                   // s_nop
            asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
        }
        if (MODE == 5) {                   // scalar compare + (not taken) branch
            asm volatile("s_cmp_eq_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 .Lnever_%=\n\ts_cmp_eq_u32 %1, 0x7ffffffe\n\ts_cbranch_scc1 .Lnever_%=\n.Lnever_%=:" : : "s"(s0), "s"(s1) : "scc");
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3 + (int)(v0 + v1 + v2 + v3);
}
template <int MODE> static void run(const char *name, int per_iter, int *out, unsigned long long *cyc) {
    for (int W : {1, 2, 4}) {
        const int threads = 64 * 4 * W;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 1);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 4 * W);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        printf("%-34s waves/SIMD %d: %.2f cycles per instruction per wave; CU-wide %.2f instructions per cycle\n", name, W,
               s / h.size() / (REP / 4 * per_iter), (double)(REP / 4 * per_iter) * 4 * W / (s / h.size()));
    }
}
int main() {
    int *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("independent s_add_u32", 4, out, cyc);
    run<3>("dependent s_add_u32", 4, out, cyc);
    run<1>("independent v_add_f32", 4, out, cyc);
    run<2>("4 s_add + 4 v_add interleaved", 8, out, cyc);
    run<4>("s_nop 0", 4, out, cyc);
    run<5>("s_cmp + s_cbranch (not taken)", 4, out, cyc);
    return 0;
}
