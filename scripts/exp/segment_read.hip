// Experiment (not part of the library): how fast can the access pattern of dict_build be READ?
//  A: genome-major layout, one wave per (genome, bucket) segment of ~610 keys, 4 keys per lane in flight,
//     workgroup b walks genomes w, w+8, ... (what dict_build does, minus all LDS work)
//  B: bucket-major layout: workgroup b streams one contiguous range of G*610 keys
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/segment_read scripts/exp/segment_read.hip ; run: /tmp/segment_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIF>
__global__ __launch_bounds__(512) void read_segments(const uint64_t *__restrict__ keys, uint32_t G, uint32_t B, uint32_t seg, uint64_t *out)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint64_t acc = 0;
    for (uint32_t g = wave; g < G; g += nw) {
        const uint64_t s0 = ((uint64_t)g * B + b) * seg;
        for (uint32_t i0 = lane; i0 < seg; i0 += 64 * KIF) {
            uint64_t kv[KIF];
#pragma unroll
            for (int j = 0; j < KIF; j++) { const uint32_t i = i0 + 64 * j; kv[j] = i < seg ? keys[s0 + i] : 0; }
#pragma unroll
            for (int j = 0; j < KIF; j++) acc ^= kv[j] * 0x9E3779B97F4A7C15ull;
        }
    }
    if (acc == 0x1234567) out[b] = acc;
}
template <int KIF>
__global__ __launch_bounds__(512) void read_stream(const uint64_t *__restrict__ keys, uint64_t per_wg, uint64_t *out)
{
    const uint64_t base = (uint64_t)blockIdx.x * per_wg;
    uint64_t acc = 0;
    for (uint64_t i0 = threadIdx.x; i0 < per_wg; i0 += 512ull * KIF) {
        uint64_t kv[KIF];
#pragma unroll
        for (int j = 0; j < KIF; j++) { const uint64_t i = i0 + 512ull * j; kv[j] = i < per_wg ? keys[base + i] : 0; }
#pragma unroll
        for (int j = 0; j < KIF; j++) acc ^= kv[j] * 0x9E3779B97F4A7C15ull;
    }
    if (acc == 0x1234567) out[blockIdx.x] = acc;
}
int main()
{
    const uint32_t G = 1000, B = 8192, seg = 610;
    const uint64_t n = (uint64_t)G * B * seg;
    uint64_t *keys, *out;
    CHK(hipMalloc(&keys, n * 8 + 4096));
    CHK(hipMalloc(&out, B * 8));
    CHK(hipMemset(keys, 1, n * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; r++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 3;
        printf("%-34s %7.2f ms  %6.2f TB/s\n", name, ms, n * 8 / (ms * 1e-3) / 1e12);
    };
    run("A genome-major, wave/segment, KIF4", [&] { hipLaunchKernelGGL(read_segments<4>, dim3(B), dim3(512), 0, 0, keys, G, B, seg, out); });
    run("A genome-major, wave/segment, KIF8", [&] { hipLaunchKernelGGL(read_segments<8>, dim3(B), dim3(512), 0, 0, keys, G, B, seg, out); });
    run("A genome-major, wave/segment, KIF10", [&] { hipLaunchKernelGGL(read_segments<10>, dim3(B), dim3(512), 0, 0, keys, G, B, seg, out); });
    run("B bucket-major stream, KIF4", [&] { hipLaunchKernelGGL(read_stream<4>, dim3(B), dim3(512), 0, 0, keys, (uint64_t)G * seg, out); });
    run("B bucket-major stream, KIF8", [&] { hipLaunchKernelGGL(read_stream<8>, dim3(B), dim3(512), 0, 0, keys, (uint64_t)G * seg, out); });
    return 0;
}
