// dictionary sort candidates on 10.3M (u64 key < 2^62, u32 index) pairs:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp/sort_bench.hip -o /tmp/sortb && /tmp/sortb
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void msd_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n, int shift, uint32_t nb, uint32_t *__restrict__ hist)
{
    extern __shared__ uint32_t lh[];
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&lh[(uint32_t)(keys[i] >> shift)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
__global__ void msd_scatter_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, uint64_t n, int shift, uint32_t nb,
                                   uint32_t *__restrict__ cursor, uint64_t *__restrict__ okeys, uint32_t *__restrict__ ovals)
{
    extern __shared__ uint32_t lds[];
    uint32_t *cnt = lds, *base = lds + nb;
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t c0 = min((uint64_t)blockIdx.x * per, n), c1 = min(c0 + per, n);
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    for (uint64_t i = c0 + threadIdx.x; i < c1; i += blockDim.x) atomicAdd(&cnt[(uint32_t)(keys[i] >> shift)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) {
        const uint32_t c = cnt[i];
        base[i] = c ? atomicAdd(&cursor[i], c) : 0u;
        cnt[i] = 0;
    }
    __syncthreads();
    for (uint64_t i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
        const uint64_t k = keys[i];
        const uint32_t d = (uint32_t)(k >> shift);
        const uint32_t at = base[d] + atomicAdd(&cnt[d], 1u);
        okeys[at] = k;
        ovals[at] = vals[i];
    }
}

int main()
{
    const uint64_t n = 10300000;
    const int kbits = 62, pbits = 12, shift = kbits - pbits;
    const uint32_t nb = 1u << pbits;
    std::vector<uint64_t> h(n);
    std::mt19937_64 rng(1);
    for (auto &v : h) { uint64_t a = rng() >> 2, b = rng() >> 2; v = std::min(a, b); }       // canonical-like skew
    std::vector<uint32_t> hv(n);
    for (uint64_t i = 0; i < n; i++) hv[i] = (uint32_t)i;
    uint64_t *k0, *k1, *k2; uint32_t *v0, *v1, *v2, *hist, *start, *cursor;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&k2, n * 8));
    CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4)); CK(hipMalloc(&v2, n * 4));
    CK(hipMalloc(&hist, (nb + 1) * 4)); CK(hipMalloc(&start, (nb + 1) * 4)); CK(hipMalloc(&cursor, (nb + 1) * 4));
    CK(hipMemcpy(k0, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    void *tmp = nullptr; size_t tb = 0, tb2 = 0, tb3 = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, n, 0u, 64u, s));
    CK(rocprim::segmented_radix_sort_pairs(nullptr, tb2, k1, k2, v1, v2, n, nb, start, start + 1, 0u, (unsigned)shift, s));
    CK(rocprim::exclusive_scan(nullptr, tb3, hist, start, 0u, (size_t)nb + 1, rocprim::plus<uint32_t>(), s));
    tb = std::max(tb, std::max(tb2, tb3));
    CK(hipMalloc(&tmp, tb));
    for (int end_bit : {64, 62}) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0, s));
            CK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, n, 0u, (unsigned)end_bit, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("rocprim radix_sort_pairs 0..%d: %.3f ms\n", end_bit, ms);
        }
    }
    std::vector<uint64_t> ref(n);
    CK(hipMemcpy(ref.data(), k1, n * 8, hipMemcpyDeviceToHost));
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t ea, eb, ec; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb)); CK(hipEventCreate(&ec));
        CK(hipEventRecord(e0, s));
        CK(hipMemsetAsync(hist, 0, (nb + 1) * 4, s));
        hipLaunchKernelGGL(msd_hist_kernel, dim3(1024), dim3(256), nb * 4, s, k0, n, shift, nb, hist);
        CK(rocprim::exclusive_scan(tmp, tb, hist, start, 0u, (size_t)nb + 1, rocprim::plus<uint32_t>(), s));
        CK(hipMemcpyAsync(cursor, start, (nb + 1) * 4, hipMemcpyDeviceToDevice, s));
        CK(hipEventRecord(ea, s));
        hipLaunchKernelGGL(msd_scatter_kernel, dim3(1024), dim3(256), nb * 8, s, k0, v0, n, shift, nb, cursor, k1, v1);
        CK(hipEventRecord(eb, s));
        CK(rocprim::segmented_radix_sort_pairs(tmp, tb, k1, k2, v1, v2, n, nb, start, start + 1, 0u, (unsigned)shift, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms, m1, m2, m3; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&m1, e0, ea)); CK(hipEventElapsedTime(&m2, ea, eb)); CK(hipEventElapsedTime(&m3, eb, e1));
        if (rep == 2) printf("msd %d bits: total %.3f ms (hist+scan %.3f, scatter %.3f, segmented sort %.3f)\n", pbits, ms, m1, m2, m3);
    }
    std::vector<uint64_t> got(n);
    CK(hipMemcpy(got.data(), k2, n * 8, hipMemcpyDeviceToHost));
    printf("sorted equal: %d\n", (int)(got == ref));
    return 0;
}
