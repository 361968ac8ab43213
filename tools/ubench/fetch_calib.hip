// DIAGNOSTIC micro-benchmark (not part of the product): calibrates rocprofv3's FETCH_SIZE on gfx950 for the access
// widths step_kernel uses.  /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE reads exactly 1/2 of the bytes of a
// wide (16 B/lane) coalesced stream and is uncalibrated for other widths.  Each kernel below reads a known number
// of bytes ONCE (buffers far larger than L2, touched by no earlier launch of the same run except the writer that
// initialised them long before), in rows laid out like the environment records: a wave reads one 64-lane row of
// W bytes per lane, rows 4 KB apart.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T>
__global__ void rows(const unsigned char *base, size_t row_stride, int rows_per_wave, T *sink) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lane = threadIdx.x & 63;
    T acc{};
    for (int r = 0; r < rows_per_wave; ++r) {
        const T v = reinterpret_cast<const T *>(base + (wave * rows_per_wave + r) * row_stride)[lane];
        if constexpr (sizeof(T) == 4) acc += v;
        else if constexpr (sizeof(T) == 8) acc += v;
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    // (a data-dependent store keeps the loads alive; the buffer holds 0x01 bytes, so it never fires)
    bool hit;
    if constexpr (sizeof(T) == 16) hit = acc.x == 0x12345678u; else hit = acc == (T)0x12345678u;
    if (hit) sink[0] = acc;
}
int main() {
    const size_t waves = 65536, rpw = 16, stride = 4096;
    const size_t bytes = waves * rpw * stride;     // 4 GiB footprint: nothing stays in L2 / Infinity Cache between kernels
    unsigned char *buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    void *sink; hipMalloc(&sink, 64);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(rows<uint32_t>, dim3(waves / 4), dim3(256), 0, 0, buf, stride, (int)rpw, (uint32_t *)sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(rows<unsigned long long>, dim3(waves / 4), dim3(256), 0, 0, buf + 1024, stride, (int)rpw, (unsigned long long *)sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(rows<uint4>, dim3(waves / 4), dim3(256), 0, 0, buf + 2048, stride, (int)rpw, (uint4 *)sink);
    hipDeviceSynchronize();
    printf("true bytes per kernel: 4 B/lane rows %zu, 8 B/lane rows %zu, 16 B/lane rows %zu\n", waves * rpw * 256, waves * rpw * 512,
           waves * rpw * 1024);
    return 0;
}
