// grm_dictsort.hip -- the dictionary's column order: U distinct canonical k-mers (one word, k <= 32) sorted by value, each with the index
// it had before (the local entry it stems from).
//
// A general radix sort takes 8 passes over the 12-byte pairs for 62-bit keys (rocPRIM: 0.9-1.05 ms for the 10.3 M entries of 1000 x 5 Mbp,
// 5 % of the pass, and a rank of an 8-GPU run repeats it in full).  The keys here are distinct k-mers: nearly uniform below their top
// bits.  So: ONE split by the top `pbits` bits into key ranges of ~2500 entries (per-workgroup counts, offsets by a scan over the count
// matrix -- no returning global atomics, which serialise on 4096 addresses), then every range is sorted inside LDS by a counting sort on
// its next 12 bits (0.6 entries per slot on average) and a thread per slot that orders the few entries it holds.
// A range that does not fit LDS (keys crowding under one 12-bit prefix: a degenerate input) raises a flag and the caller takes the general sort.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "grm_internal.h"
#include "grm_coop.h"

namespace grm {

constexpr int DS_THREADS = 256;            // split kernels
constexpr int DS_MAX_PBITS = 12;
constexpr int DS_SORT_THREADS = 1024;      // one range per workgroup
constexpr uint32_t DS_RANGE_CAP = 8192;    // entries of a range in LDS: 8192 x (8 + 4) bytes twice over would not fit -- sorted through ONE image, see below
constexpr int DS_SUB_BITS = 12;            // counting sort inside a range

struct DsArgs {
    const uint64_t *keys;                  // [n]
    uint64_t n;
    int shift;                             // key >> shift = range (pbits bits)
    uint32_t n_ranges;                     // 2^pbits
    uint32_t n_chunks;                     // workgroups of the split = rows of the count matrix
};

// counts[chunk][range]
__global__ __launch_bounds__(DS_THREADS) void ds_count_kernel(DsArgs a, uint32_t *__restrict__ counts)
{
    extern __shared__ uint32_t lh[];
    for (uint32_t i = threadIdx.x; i < a.n_ranges; i += DS_THREADS) lh[i] = 0;
    __syncthreads();
    const uint64_t per = (a.n + a.n_chunks - 1) / a.n_chunks;
    const uint64_t c0 = min((uint64_t)blockIdx.x * per, a.n), c1 = min(c0 + per, a.n);
    for (uint64_t i = c0 + threadIdx.x; i < c1; i += DS_THREADS) atomicAdd(&lh[(uint32_t)(a.keys[i] >> a.shift)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.n_ranges; i += DS_THREADS) counts[(uint64_t)blockIdx.x * a.n_ranges + i] = lh[i];
}
// per range: exclusive prefix of its counts over the chunks (in place), its total.  (16 loads in flight per thread: read one after the
// other, with a store between them, the 256 steps of this loop are 256 memory round trips.)
__global__ void ds_offsets_kernel(uint32_t *__restrict__ counts, uint32_t n_chunks, uint32_t n_ranges, uint32_t *__restrict__ total)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranges) return;
    uint32_t run = 0;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = c0 + j < n_chunks ? counts[(uint64_t)(c0 + j) * n_ranges + r] : 0u;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (c0 + j < n_chunks) counts[(uint64_t)(c0 + j) * n_ranges + r] = run;
            run += v[j];
        }
    }
    total[r] = run;
}
// start[0 .. n_ranges]: exclusive scan of the range totals; *too_big = 1 when a range exceeds what the range sort holds
__global__ __launch_bounds__(1024) void ds_starts_kernel(const uint32_t *__restrict__ total, uint32_t n_ranges, uint32_t *__restrict__ start,
                                                          int *__restrict__ too_big)
{
    __shared__ uint32_t scratch[32];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_ranges; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_ranges ? total[i] : 0u;
        if (v > DS_RANGE_CAP) atomicExch(too_big, 1);
        uint32_t sum;
        const uint32_t pre = block_scan_sum(v, scratch, &sum);
        if (i < n_ranges) start[i] = carry + pre;
        carry += sum;
    }
    if (threadIdx.x == 0) start[n_ranges] = carry;
}
// every entry to its range: position = start of the range + what the chunks before this one put there + an LDS cursor
__global__ __launch_bounds__(DS_THREADS) void ds_scatter_kernel(DsArgs a, const uint32_t *__restrict__ base, const uint32_t *__restrict__ start,
                                                                 uint64_t *__restrict__ okeys, uint32_t *__restrict__ oidx)
{
    extern __shared__ uint32_t cur[];
    for (uint32_t i = threadIdx.x; i < a.n_ranges; i += DS_THREADS) cur[i] = start[i] + base[(uint64_t)blockIdx.x * a.n_ranges + i];
    __syncthreads();
    const uint64_t per = (a.n + a.n_chunks - 1) / a.n_chunks;
    const uint64_t c0 = min((uint64_t)blockIdx.x * per, a.n), c1 = min(c0 + per, a.n);
    for (uint64_t i = c0 + threadIdx.x; i < c1; i += DS_THREADS) {
        const uint64_t k = a.keys[i];
        const uint32_t at = atomicAdd(&cur[(uint32_t)(k >> a.shift)], 1u);
        okeys[at] = k;
        oidx[at] = (uint32_t)i;
    }
}
// One range per workgroup, in place.  The entries go into an LDS image ordered by the next DS_SUB_BITS bits of the key (counting sort:
// histogram, scan, placement), then thread s orders the entries of slot s (insertion sort: they are few) and the image leaves in order.
__global__ __launch_bounds__(DS_SORT_THREADS) void ds_sort_ranges_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ idx,
                                                                         const uint32_t *__restrict__ start, uint32_t n_ranges, int shift)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    constexpr uint32_t NS = 1u << DS_SUB_BITS;
    uint64_t *sk = reinterpret_cast<uint64_t *>(lds_raw);                          // [DS_RANGE_CAP]
    uint32_t *si = reinterpret_cast<uint32_t *>(lds_raw + (size_t)DS_RANGE_CAP * 8);      // [DS_RANGE_CAP]
    uint32_t *hist = si + DS_RANGE_CAP;                                              // [NS]: counts, then running positions
    uint32_t *first = hist + NS;                                                     // [NS + 1]
    __shared__ uint32_t scratch[32];
    const int sub_shift = shift > DS_SUB_BITS ? shift - DS_SUB_BITS : 0;
    const uint32_t sub_mask = shift > DS_SUB_BITS ? NS - 1 : (1u << shift) - 1u;
    for (uint32_t r = blockIdx.x; r < n_ranges; r += gridDim.x) {
        const uint32_t s0 = start[r], m = start[r + 1] - s0;
        if (m <= 1 || m > DS_RANGE_CAP) continue;            // (too large: the caller redoes everything with the general sort)
        for (uint32_t i = threadIdx.x; i < NS; i += DS_SORT_THREADS) hist[i] = 0;
        __syncthreads();
        // (a thread's entries stay in registers between the histogram and the placement: at most DS_RANGE_CAP / threads = 8)
        uint64_t mk[DS_RANGE_CAP / DS_SORT_THREADS];
        uint32_t mi[DS_RANGE_CAP / DS_SORT_THREADS];
#pragma unroll
        for (int j = 0; j < (int)(DS_RANGE_CAP / DS_SORT_THREADS); j++) {
            const uint32_t i = (uint32_t)j * DS_SORT_THREADS + threadIdx.x;
            mk[j] = i < m ? keys[s0 + i] : 0ull;
            mi[j] = i < m ? idx[s0 + i] : 0u;
            if (i < m) atomicAdd(&hist[(uint32_t)(mk[j] >> sub_shift) & sub_mask], 1u);
        }
        __syncthreads();
        // exclusive scan of the NS counts: NS / threads per thread
        {
            constexpr int PER = NS / DS_SORT_THREADS;
            uint32_t v[PER], sum = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) { v[q] = hist[threadIdx.x * PER + q]; sum += v[q]; }
            uint32_t all;
            uint32_t pre = block_scan_sum(sum, scratch, &all);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                first[threadIdx.x * PER + q] = pre;
                hist[threadIdx.x * PER + q] = pre;
                pre += v[q];
            }
            if (threadIdx.x == DS_SORT_THREADS - 1) first[NS] = pre;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (int)(DS_RANGE_CAP / DS_SORT_THREADS); j++) {
            const uint32_t i = (uint32_t)j * DS_SORT_THREADS + threadIdx.x;
            if (i < m) {
                const uint32_t at = atomicAdd(&hist[(uint32_t)(mk[j] >> sub_shift) & sub_mask], 1u);
                sk[at] = mk[j];
                si[at] = mi[j];
            }
        }
        __syncthreads();
        // the entries of one slot, in order
        for (uint32_t s = threadIdx.x; s < NS; s += DS_SORT_THREADS) {
            const uint32_t a = first[s], b = first[s + 1];
            for (uint32_t i = a + 1; i < b; i++) {
                const uint64_t kk = sk[i];
                const uint32_t ii = si[i];
                uint32_t j = i;
                while (j > a && sk[j - 1] > kk) { sk[j] = sk[j - 1]; si[j] = si[j - 1]; j--; }
                sk[j] = kk;
                si[j] = ii;
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += DS_SORT_THREADS) { keys[s0 + i] = sk[i]; idx[s0 + i] = si[i]; }
        __syncthreads();
    }
}

size_t dict_sort_scratch_bytes(uint64_t n)
{
    const uint32_t n_ranges = 1u << DS_MAX_PBITS;
    const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>(512, (n + 4095) / 4096));
    return (size_t)n_chunks * n_ranges * 4 + (size_t)(2 * n_ranges + 2) * 4 + 64;
}

// keys[n] (values below 2^key_bits, distinct or not) -> okeys ascending, oidx = the position every key had in `keys`.
// scratch: dict_sort_scratch_bytes(n); *too_big (device int, zeroed by the caller) = 1: a key range did not fit, the output is NOT sorted
hipError_t launch_dict_sort(hipStream_t s, const uint64_t *keys, uint64_t n, int key_bits, uint64_t *okeys, uint32_t *oidx, void *scratch, int *too_big)
{
    if (!n) return hipSuccess;
    // ranges of ~2048 entries, at most 2^12 of them
    int pbits = 0;
    while (pbits < DS_MAX_PBITS && pbits < key_bits && (n >> pbits) > 2048) pbits++;
    if (pbits == 0) pbits = 1;               // (a shift by the full key width is no shift)
    DsArgs a;
    a.keys = keys; a.n = n; a.shift = key_bits - pbits; a.n_ranges = 1u << pbits;
    a.n_chunks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(512, (n + 4095) / 4096));
    uint32_t *counts = reinterpret_cast<uint32_t *>(scratch);
    uint32_t *total = counts + (size_t)a.n_chunks * a.n_ranges;
    uint32_t *start = total + a.n_ranges;
    hipLaunchKernelGGL(ds_count_kernel, dim3(a.n_chunks), dim3(DS_THREADS), (size_t)a.n_ranges * 4, s, a, counts);
    hipLaunchKernelGGL(ds_offsets_kernel, dim3((a.n_ranges + 63) / 64), dim3(64), 0, s, counts, a.n_chunks, a.n_ranges, total);
    hipLaunchKernelGGL(ds_starts_kernel, dim3(1), dim3(1024), 0, s, total, a.n_ranges, start, too_big);
    hipLaunchKernelGGL(ds_scatter_kernel, dim3(a.n_chunks), dim3(DS_THREADS), (size_t)a.n_ranges * 4, s, a, counts, start, okeys, oidx);
    static std::atomic<uint64_t> lds_set{0};
    const size_t lds = (size_t)DS_RANGE_CAP * 12 + ((size_t)2 << DS_SUB_BITS) * 4 + 16;
    const hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void *>(ds_sort_ranges_kernel), (int)lds, lds_set);
    if (ea != hipSuccess) return ea;
    const uint32_t grid = a.n_ranges < 2048u ? a.n_ranges : 2048u;
    hipLaunchKernelGGL(ds_sort_ranges_kernel, dim3(grid), dim3(DS_SORT_THREADS), lds, s, okeys, oidx, start, a.n_ranges, a.shift);
    return hipGetLastError();
}

}  // namespace grm
