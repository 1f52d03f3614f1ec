// grm_wide.hip -- two-word k-mers (33 <= k <= 64; BASELINE config C5: k = 63).
//
// Sort-based path, correctness first: 128-bit keys have no 128-bit LDS compare-and-swap, so
// the hash-table pipeline of grm_kernels.hip does not carry over unchanged.  Instead:
//   extract   every start position -> canonical (hi, lo), sentinel where invalid  (hand-written)
//   sort      stable LSD: rocPRIM radix sort by lo, then by hi (index payload)      (library)
//   reduce    runs of equal (key, genome) -> counts -> abundance filter; runs of equal key ->
//             carrier count -> singleton filter -> column ids; dictionary + presence bits
// Entries stay in genome-major input order under the stable sort, so inside a run of equal keys
// the genomes appear in ascending order.  All index arithmetic is 32-bit: a batch is limited
// to 2^32-1 symbols on this path (loud error beyond).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grm_device_fns.h"
#include "grm_internal.h"

namespace grm {

constexpr int WIDE_PPT = 16;

__device__ __forceinline__ uint32_t wide_genome_of(const uint64_t *__restrict__ gso, uint32_t n_genomes, uint64_t p)
{
    uint32_t lo = 0, hi = n_genomes;
    while (hi - lo > 1) {
        const uint32_t m = (lo + hi) >> 1;
        if (gso[m] <= p) lo = m; else hi = m;
    }
    return lo;
}

// one (hi, lo) slot per symbol position; invalid positions get the all-ones sentinel, which
// sorts last and is never a canonical k-mer.  n_valid counts the real k-mers.
__global__ __launch_bounds__(256) void wide_extract_kernel(const uint64_t *__restrict__ sym2, const uint64_t *__restrict__ inv,
                                                           uint64_t total_syms, int k, uint64_t *__restrict__ khi,
                                                           uint64_t *__restrict__ klo, unsigned long long *__restrict__ n_valid)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = q * WIDE_PPT;
    if (p0 >= total_syms) return;
    const uint64_t grp = p0 >> 6;
    const int64_t nv = (int64_t)total_syms - k + 1 - (int64_t)p0;
    uint32_t valid = 0;
    if (nv > 0) {
        valid = (uint32_t)(valid_starts(inv[grp], inv[grp + 1], k) >> (p0 & 63)) & 0xffffu;
        if (nv < WIDE_PPT) valid &= (1u << nv) - 1;
    }
    const uint64_t n_here = min((uint64_t)WIDE_PPT, total_syms - p0);
    for (uint64_t i = 0; i < n_here; i++) {
        if (!((valid >> i) & 1u)) { khi[p0 + i] = ~0ull; klo[p0 + i] = ~0ull; }
    }
    if (valid) {
        const uint64_t wi = p0 >> 5;
        for_each_kmer_wide<WIDE_PPT>(sym2[wi], sym2[wi + 1], sym2[wi + 2], (int)(p0 & 31), valid, k, [&](int i, K128 c) {
            khi[p0 + i] = c.hi;
            klo[p0 + i] = c.lo;
        });
        atomicAdd(n_valid, (unsigned long long)__popc(valid));
    }
}

// head flags over the sorted entries [0, n): new key / new (key, genome)
__global__ void wide_mark_kernel(const uint64_t *__restrict__ khi, const uint64_t *__restrict__ klo,
                                 const uint32_t *__restrict__ pos, const uint64_t *__restrict__ gso, uint32_t n_genomes,
                                 uint32_t n, uint32_t *__restrict__ key_head, uint32_t *__restrict__ kg_head)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t kh = 1, gh = 1;
        if (i > 0) {
            kh = (khi[i] != khi[i - 1]) || (klo[i] != klo[i - 1]);
            gh = kh || (wide_genome_of(gso, n_genomes, pos[i]) != wide_genome_of(gso, n_genomes, pos[i - 1]));
        }
        key_head[i] = kh;
        kg_head[i] = gh;
    }
}
// sub_start[r] = first entry of (key, genome) run r ; sub_start[n_sub] = n
__global__ void wide_sub_start_kernel(const uint32_t *__restrict__ kg_head, const uint32_t *__restrict__ sub_id, uint32_t n,
                                      uint32_t n_sub, uint32_t *__restrict__ sub_start)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (kg_head[i]) sub_start[sub_id[i]] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) sub_start[n_sub] = n;
}
// per (key, genome) run: starts a new key?  passes the abundance filter?
__global__ void wide_sub_kernel(const uint32_t *__restrict__ sub_start, const uint32_t *__restrict__ key_head, uint32_t n_sub,
                                uint32_t abundance_min, uint32_t *__restrict__ sub_key_head, uint32_t *__restrict__ sub_ok)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_sub; r += gridDim.x * blockDim.x) {
        const uint32_t i = sub_start[r];
        sub_key_head[r] = key_head[i];
        sub_ok[r] = (sub_start[r + 1] - i) >= abundance_min ? 1u : 0u;
    }
}
// carriers per key: key id of run r = (inclusive scan of sub_key_head)[r] - 1
__global__ void wide_key_count_kernel(const uint32_t *__restrict__ key_incl, const uint32_t *__restrict__ sub_ok, uint32_t n_sub,
                                      uint32_t *__restrict__ carriers)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_sub; r += gridDim.x * blockDim.x)
        if (sub_ok[r]) atomicAdd(&carriers[key_incl[r] - 1], 1u);
}
__global__ void wide_keep_kernel(const uint32_t *__restrict__ carriers, uint32_t n_keys, uint32_t min_carriers,
                                 uint32_t *__restrict__ keep)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += gridDim.x * blockDim.x)
        keep[i] = carriers[i] >= min_carriers ? 1u : 0u;
}
// dictionary (hi, lo interleaved) + presence bits
__global__ void wide_emit_kernel(const uint64_t *__restrict__ khi, const uint64_t *__restrict__ klo,
                                 const uint32_t *__restrict__ pos, const uint64_t *__restrict__ gso, uint32_t n_genomes,
                                 const uint32_t *__restrict__ sub_start, const uint32_t *__restrict__ sub_key_head,
                                 const uint32_t *__restrict__ sub_ok, const uint32_t *__restrict__ key_incl,
                                 const uint32_t *__restrict__ keep, const uint32_t *__restrict__ col, uint32_t n_sub,
                                 uint64_t *__restrict__ dict, uint64_t *__restrict__ matrix, uint64_t n_cols)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_sub; r += gridDim.x * blockDim.x) {
        const uint32_t kid = key_incl[r] - 1;
        if (!keep[kid]) continue;
        const uint32_t c = col[kid];
        const uint32_t i = sub_start[r];
        if (sub_key_head[r]) { dict[2ull * c] = khi[i]; dict[2ull * c + 1] = klo[i]; }
        if (sub_ok[r]) {
            const uint32_t g = wide_genome_of(gso, n_genomes, pos[i]);
            atomicOr((unsigned long long *)&matrix[(uint64_t)(g >> 6) * n_cols + c], 1ull << (63 - (g & 63)));
        }
    }
}
// counted set of a single-genome batch: (key, count) of the runs that pass the filter
__global__ void wide_set_kernel(const uint64_t *__restrict__ khi, const uint64_t *__restrict__ klo,
                                const uint32_t *__restrict__ sub_start, const uint32_t *__restrict__ sub_ok,
                                const uint32_t *__restrict__ out_pos, uint32_t n_sub, uint64_t *__restrict__ kmers,
                                uint32_t *__restrict__ counts)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_sub; r += gridDim.x * blockDim.x) {
        if (!sub_ok[r]) continue;
        const uint32_t i = sub_start[r], o = out_pos[r];
        kmers[2ull * o] = khi[i];
        kmers[2ull * o + 1] = klo[i];
        counts[o] = sub_start[r + 1] - i;
    }
}

static inline uint32_t wgrid(uint64_t n)
{
    uint64_t g = (n + 255) / 256;
    return (uint32_t)(g < 1 ? 1 : (g > 256u * 32u ? 256u * 32u : g));
}

void launch_wide_extract(hipStream_t s, const uint64_t *sym2, const uint64_t *inv, uint64_t total_syms, int k, uint64_t *khi,
                         uint64_t *klo, unsigned long long *n_valid)
{
    if (!total_syms) return;
    const uint64_t n_threads = (total_syms + WIDE_PPT - 1) / WIDE_PPT;
    hipLaunchKernelGGL(wide_extract_kernel, dim3((uint32_t)((n_threads + 255) / 256)), dim3(256), 0, s, sym2, inv, total_syms, k, khi,
                       klo, n_valid);
}
void launch_wide_mark(hipStream_t s, const uint64_t *khi, const uint64_t *klo, const uint32_t *pos, const uint64_t *gso,
                      uint32_t n_genomes, uint32_t n, uint32_t *key_head, uint32_t *kg_head)
{
    if (n) hipLaunchKernelGGL(wide_mark_kernel, dim3(wgrid(n)), dim3(256), 0, s, khi, klo, pos, gso, n_genomes, n, key_head, kg_head);
}
void launch_wide_sub_start(hipStream_t s, const uint32_t *kg_head, const uint32_t *sub_id, uint32_t n, uint32_t n_sub,
                           uint32_t *sub_start)
{
    hipLaunchKernelGGL(wide_sub_start_kernel, dim3(wgrid(n)), dim3(256), 0, s, kg_head, sub_id, n, n_sub, sub_start);
}
void launch_wide_sub(hipStream_t s, const uint32_t *sub_start, const uint32_t *key_head, uint32_t n_sub, uint32_t abundance_min,
                     uint32_t *sub_key_head, uint32_t *sub_ok)
{
    if (n_sub) hipLaunchKernelGGL(wide_sub_kernel, dim3(wgrid(n_sub)), dim3(256), 0, s, sub_start, key_head, n_sub, abundance_min,
                                  sub_key_head, sub_ok);
}
void launch_wide_key_count(hipStream_t s, const uint32_t *key_incl, const uint32_t *sub_ok, uint32_t n_sub, uint32_t *carriers)
{
    if (n_sub) hipLaunchKernelGGL(wide_key_count_kernel, dim3(wgrid(n_sub)), dim3(256), 0, s, key_incl, sub_ok, n_sub, carriers);
}
void launch_wide_keep(hipStream_t s, const uint32_t *carriers, uint32_t n_keys, uint32_t min_carriers, uint32_t *keep)
{
    if (n_keys) hipLaunchKernelGGL(wide_keep_kernel, dim3(wgrid(n_keys)), dim3(256), 0, s, carriers, n_keys, min_carriers, keep);
}
void launch_wide_emit(hipStream_t s, const uint64_t *khi, const uint64_t *klo, const uint32_t *pos, const uint64_t *gso,
                      uint32_t n_genomes, const uint32_t *sub_start, const uint32_t *sub_key_head, const uint32_t *sub_ok,
                      const uint32_t *key_incl, const uint32_t *keep, const uint32_t *col, uint32_t n_sub, uint64_t *dict,
                      uint64_t *matrix, uint64_t n_cols)
{
    if (n_sub) hipLaunchKernelGGL(wide_emit_kernel, dim3(wgrid(n_sub)), dim3(256), 0, s, khi, klo, pos, gso, n_genomes, sub_start,
                                  sub_key_head, sub_ok, key_incl, keep, col, n_sub, dict, matrix, n_cols);
}
void launch_wide_set(hipStream_t s, const uint64_t *khi, const uint64_t *klo, const uint32_t *sub_start, const uint32_t *sub_ok,
                     const uint32_t *out_pos, uint32_t n_sub, uint64_t *kmers, uint32_t *counts)
{
    if (n_sub) hipLaunchKernelGGL(wide_set_kernel, dim3(wgrid(n_sub)), dim3(256), 0, s, khi, klo, sub_start, sub_ok, out_pos, n_sub,
                                  kmers, counts);
}

}  // namespace grm
