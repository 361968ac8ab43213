// DIAGNOSTIC micro-benchmark (not part of the product): dependent-chain latency and co-issue behaviour of the
// f64 VALU instructions the observation tail is made of, on gfx950.  For each op: cycles per dependent
// instruction with W waves per SIMD issuing the same chain (W = 1, 2, 4), measured with s_memtime around 256
// dependent instructions; and the LDS-fed chain of lds_chain_sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHAIN 256
template <int OP>
__global__ void chain_kernel(double *out, unsigned long long *cyc, double seed) {
    double a = seed + threadIdx.x, b = seed * 0.5;
    float af = (float)a, bf = (float)b;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < CHAIN; ++i) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
        if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));
        if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
        if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(af) : "v"(bf));
        if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));
        if (OP == 5) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a));
        if (OP == 6) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));
        if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(af) : "v"(bf));
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + af;
}

__global__ void lds_chain_kernel(double *out, unsigned long long *cyc, int n8) {
    extern __shared__ double sm[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = 1.0 + i;
    __syncthreads();
    const double *src = sm + (threadIdx.x & 7) * 64 + (threadIdx.x >> 6) * 512;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double acc = 0.0;
    for (int i = 0; i < n8; i += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = acc + src[i + q];
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
static void run(const char *name, double *out, unsigned long long *cyc) {
    // one CU's worth is enough: grid of 256 blocks x (64 * W * 4) threads -> W waves per SIMD on every CU
    for (int W : {1, 2, 4}) {
        const int threads = 64 * 4 * W;
        hipLaunchKernelGGL(chain_kernel<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.25);
        hipLaunchKernelGGL(chain_kernel<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.25);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 4 * W);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        printf("%-12s waves/SIMD %d: %.2f memtime ticks per dependent instruction (per wave)\n", name, W, s / h.size() / CHAIN);
    }
}

int main() {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 16 * 8);
    run<0>("v_add_f64", out, cyc); run<1>("v_fma_f64", out, cyc); run<2>("v_mul_f64", out, cyc); run<6>("v_max_f64", out, cyc);
    run<3>("v_add_f32", out, cyc); run<7>("v_add_u32", out, cyc); run<4>("v_rcp_f64", out, cyc); run<5>("v_sqrt_f64", out, cyc);
    for (int W : {1, 4}) for (int n8 : {8, 40, 56}) {
        const int threads = 64 * 4 * W;
        hipLaunchKernelGGL(lds_chain_kernel, dim3(256), dim3(threads), 32768, 0, out, cyc, n8);
        hipLaunchKernelGGL(lds_chain_kernel, dim3(256), dim3(threads), 32768, 0, out, cyc, n8);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 4 * W);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        printf("lds chain n8=%d waves/SIMD %d: %.1f ticks total, %.2f per element\n", n8, W, s / h.size(), s / h.size() / n8);
    }
    // s_memtime tick rate vs wall: 1e6 dependent adds timed with events
    return 0;
}
