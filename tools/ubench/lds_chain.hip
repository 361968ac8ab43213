// DIAGNOSTIC micro-benchmark (not part of the product): cost of the LDS-fed sequential f64 chain of the
// observation tail as a function of the row layout (bank conflicts between the rows that the chain lanes
// walk), the number of distinct rows per wave and the waves per CU that walk at the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// lane L walks row (L % rows_per_wave) of its wave's slice; all other lanes alias row 0 (as in the kernel)
template <int WIDE, int MASK>
__global__ void k(double *out, unsigned long long *cyc, int n8, int row_stride_b, int rows, int slice_b, int active_waves) {
    extern __shared__ unsigned char sm[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<double *>(sm)[i] = 1.0 + i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= active_waves) return;
    const int r = lane < rows ? lane : 0;
    const double *src = reinterpret_cast<const double *>(sm + (size_t)wave * slice_b + (size_t)r * row_stride_b);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double acc = 0.0, a[8], b[8];
    if (MASK && lane >= MASK) { out[blockIdx.x * blockDim.x + threadIdx.x] = 0; return; }
    if (WIDE) {
        // same chain, operands fetched as 16-byte pairs (ds_read_b128)
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
        double2 A[4], B[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) A[q] = s2[q];
        int i = 8;
        for (;;) {
            if (i < n8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) B[q] = s2[i / 2 + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc = acc + A[q].x; acc = acc + A[q].y; }
            if (i >= n8) break;
            i += 8;
            if (i < n8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) A[q] = s2[i / 2 + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc = acc + B[q].x; acc = acc + B[q].y; }
            if (i >= n8) break;
            i += 8;
        }
        asm volatile("s_nop 0" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = src[q];
    int i = 8;
    for (;;) {
        if (i < n8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = src[i + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = acc + a[q];
        if (i >= n8) break;
        i += 8;
        if (i < n8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = src[i + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = acc + b[q];
        if (i >= n8) break;
        i += 8;
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 1024 * 16 * 8);
    const int n8 = 48;
    struct Cfg { const char *name; int row_stride, rows, slice, waves_per_block, blocks_per_cu; };
    // 4-wave blocks, 4 blocks per CU (the step kernel's shape): every wave walks 3 rows
    const Cfg cfgs[] = {
        {"3 rows/wave, stride 512 B (today), 16 waves/CU", 512, 3, 2240, 4, 4},
        {"3 rows/wave, stride 528 B (skewed), 16 waves/CU", 528, 3, 2240, 4, 4},
        {"3 rows/wave, stride 512 B, 4 waves/CU", 512, 3, 2240, 4, 1},
        {"12 rows in 1 walker wave, stride 512 B, slice 2240 (1 walker per block, 4 blocks/CU)", 512, 12, 0, 1, 4},
        {"12 rows in 1 walker wave, stride 528 B (1 walker per block, 4 blocks/CU)", 528, 12, 0, 1, 4},
        {"12 rows in 1 walker wave, stride 592 B (1 walker per block, 4 blocks/CU)", 592, 12, 0, 1, 4},
        {"1 row/wave, 16 waves/CU", 512, 1, 2240, 4, 4},
    };
    for (const Cfg &c : cfgs) {
        const int blocks = 256 * c.blocks_per_cu;
      for (int variant = 0; variant < 4; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            auto kk = variant == 0 ? k<0, 0> : (variant == 1 ? k<1, 0> : (variant == 2 ? k<1, 32> : k<1, 16>));
            hipLaunchKernelGGL(kk, dim3(blocks), dim3(256), 65536 / c.blocks_per_cu > 40960 ? 40960 : 16384, 0, out, cyc, n8, c.row_stride, c.rows,
                               c.slice, c.waves_per_block);
        }
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int bI = 0; bI < blocks; ++bI) for (int w = 0; w < c.waves_per_block; ++w) { s += (double)h[bI * 16 + w]; n++; }
        const char *vn[] = {"read2_b64", "read_b128", "read_b128, exec = 32 lanes", "read_b128, exec = 16 lanes"};
        printf("%-90s %-28s n8=%d: %.0f cycles per chain, %.1f per element\n", c.name, vn[variant], n8, s / n, s / n / n8);
      }
    }
    return 0;
}
