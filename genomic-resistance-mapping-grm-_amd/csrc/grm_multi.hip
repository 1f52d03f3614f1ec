// grm_multi.hip -- k-mers of 65 .. 128 bases (three and four 64-bit words; the reference's --kmer-size range ends
// at 128: bin/kover/kover:114, GUI spinbox 3-128 at src/app.py:1648-1652).
//
// Sort-based path, the same shape as grm_wide.hip with the word count as a template parameter:
//   extract   every start position -> canonical key as W words (structure of arrays, word 0 most significant);
//             positions without a valid window get the all-ones sentinel, which sorts last and is never canonical
//             (the reverse complement of a k-mer of G's is a k-mer of C's, which is smaller)
//   sort      stable LSD: one rocPRIM radix sort per word, least significant first, index payload   (library)
//   reduce    runs of equal (key, genome) -> counts -> abundance filter; runs of equal key -> carrier count ->
//             singleton filter -> columns; dictionary + presence bits.  The run bookkeeping kernels are the
//             width-independent ones of grm_wide.hip.
// 32-bit index arithmetic: a batch is limited to 2^32-1 symbols on this path (loud error beyond).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grm_device_fns.h"
#include "grm_internal.h"

namespace grm {

constexpr int MULTI_PPT = 32;       // start positions per thread: k - 1 + 32 symbol steps for 32 k-mers

template <int W>
struct KeyW {
    uint64_t w[W];       // w[0] most significant
};
template <int W>
__device__ __forceinline__ bool key_less(const KeyW<W> &a, const KeyW<W> &b)
{
#pragma unroll
    for (int j = 0; j < W; j++) {
        if (a.w[j] != b.w[j]) return a.w[j] < b.w[j];
    }
    return false;
}

__device__ __forceinline__ uint32_t multi_genome_of(const uint64_t *__restrict__ gso, uint32_t n_genomes, uint64_t p)
{
    uint32_t lo = 0, hi = n_genomes;
    while (hi - lo > 1) {
        const uint32_t m = (lo + hi) >> 1;
        if (gso[m] <= p) lo = m; else hi = m;
    }
    return lo;
}

// One thread walks the symbols [p0, p0 + 32 + k - 1): rolling forward / reverse-complement keys and the length of
// the current run of valid symbols; start position p0 + i is valid iff the k symbols from it are.
template <int W>
__global__ __launch_bounds__(256) void multi_extract_kernel(const uint64_t *__restrict__ sym2, const uint64_t *__restrict__ inv,
                                                            uint64_t total_syms, int k, MultiWordsOut out,
                                                            unsigned long long *__restrict__ n_valid)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = q * MULTI_PPT;
    if (p0 >= total_syms) return;
    const int top_bits = 2 * k - 64 * (W - 1);                 // significant bits of word 0
    const uint64_t top_mask = top_bits >= 64 ? ~0ull : ((1ull << top_bits) - 1);
    const int rc_word = W - 1 - (2 * (k - 1)) / 64, rc_shift = (2 * (k - 1)) & 63;     // where the complement enters
    KeyW<W> fwd, rc;
#pragma unroll
    for (int j = 0; j < W; j++) { fwd.w[j] = 0; rc.w[j] = 0; }
    uint32_t run = 0, n_ok = 0;
    const uint64_t p_end = min(p0 + MULTI_PPT + (uint64_t)k - 1, total_syms);
    uint64_t sw = 0, iw = 0;
    for (uint64_t p = p0; p < p0 + MULTI_PPT + (uint64_t)k - 1; p++) {
        bool bad = true;
        uint64_t sym = 0;
        if (p < p_end) {
            if (p == p0 || (p & 31) == 0) sw = sym2[p >> 5];
            if (p == p0 || (p & 63) == 0) iw = inv[p >> 6];
            sym = (sw >> (62 - 2 * (p & 31))) & 3ull;
            bad = (iw >> (p & 63)) & 1ull;
        }
        // fwd = (fwd << 2 | sym) & mask ; rc = rc >> 2 | (sym ^ 2) << 2(k-1)
#pragma unroll
        for (int j = 0; j < W - 1; j++) fwd.w[j] = (fwd.w[j] << 2) | (fwd.w[j + 1] >> 62);
        fwd.w[W - 1] = (fwd.w[W - 1] << 2) | sym;
        fwd.w[0] &= top_mask;
#pragma unroll
        for (int j = W - 1; j > 0; j--) rc.w[j] = (rc.w[j] >> 2) | (rc.w[j - 1] << 62);
        rc.w[0] >>= 2;
#pragma unroll
        for (int j = 0; j < W; j++) { if (j == rc_word) rc.w[j] |= (sym ^ 2ull) << rc_shift; }
        run = bad ? 0u : run + 1u;
        if (p + 1 >= p0 + (uint64_t)k) {
            const uint64_t s = p + 1 - (uint64_t)k;              // start position of the window that ends at p
            if (s < total_syms) {
                const bool ok = run >= (uint32_t)k;
                const KeyW<W> &c = key_less<W>(fwd, rc) ? fwd : rc;
#pragma unroll
                for (int j = 0; j < W; j++) out.w[j][s] = ok ? c.w[j] : ~0ull;
                n_ok += ok ? 1u : 0u;
            }
        }
    }
    if (n_ok) atomicAdd(n_valid, (unsigned long long)n_ok);
}

// head flags over the sorted entries [0, n): new key / new (key, genome)
template <int W>
__global__ void multi_mark_kernel(MultiWords S, const uint32_t *__restrict__ pos, const uint64_t *__restrict__ gso, uint32_t n_genomes,
                                  uint32_t n, uint32_t *__restrict__ key_head, uint32_t *__restrict__ kg_head)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t kh = 1, gh = 1;
        if (i > 0) {
            kh = 0;
#pragma unroll
            for (int j = 0; j < W; j++) kh |= (S.w[j][i] != S.w[j][i - 1]) ? 1u : 0u;
            gh = kh || (multi_genome_of(gso, n_genomes, pos[i]) != multi_genome_of(gso, n_genomes, pos[i - 1]));
        }
        key_head[i] = kh;
        kg_head[i] = gh;
    }
}
// dictionary (W words per column, most significant first) + presence bits
template <int W>
__global__ void multi_emit_kernel(MultiWords S, const uint32_t *__restrict__ pos, const uint64_t *__restrict__ gso, uint32_t n_genomes,
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
        if (sub_key_head[r]) {
#pragma unroll
            for (int j = 0; j < W; j++) dict[(uint64_t)W * c + j] = S.w[j][i];
        }
        if (sub_ok[r]) {
            const uint32_t g = multi_genome_of(gso, n_genomes, pos[i]);
            atomicOr((unsigned long long *)&matrix[(uint64_t)(g >> 6) * n_cols + c], 1ull << (63 - (g & 63)));
        }
    }
}
// counted set of a single-genome batch: (key, count) of the runs that pass the filter
template <int W>
__global__ void multi_set_kernel(MultiWords S, const uint32_t *__restrict__ sub_start, const uint32_t *__restrict__ sub_ok,
                                 const uint32_t *__restrict__ out_pos, uint32_t n_sub, uint64_t *__restrict__ kmers,
                                 uint32_t *__restrict__ counts)
{
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_sub; r += gridDim.x * blockDim.x) {
        if (!sub_ok[r]) continue;
        const uint32_t i = sub_start[r], o = out_pos[r];
#pragma unroll
        for (int j = 0; j < W; j++) kmers[(uint64_t)W * o + j] = S.w[j][i];
        counts[o] = sub_start[r + 1] - i;
    }
}
// n keys of W interleaved words -> W separate arrays (sets handed to grm_build_matrix)
template <int W>
__global__ void multi_split_kernel(const uint64_t *__restrict__ keys, uint64_t n, MultiWordsOut out, uint64_t at)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int j = 0; j < W; j++) out.w[j][at + i] = keys[(uint64_t)W * i + j];
    }
}

static inline uint32_t mgrid(uint64_t n)
{
    uint64_t g = (n + 255) / 256;
    return (uint32_t)(g < 1 ? 1 : (g > 256u * 32u ? 256u * 32u : g));
}

// (W = 2: two-word k-mers take this path only where the hash-partition pipeline does not apply in the staged calls -- abundance-min > 1)
#define GRM_MULTI_KERNEL(W_, KERNEL, GRID, ...)                                                             \
    do {                                                                                                    \
        if ((W_) == 2) hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), 0, s, __VA_ARGS__);                   \
        else if ((W_) == 3) hipLaunchKernelGGL(KERNEL<3>, GRID, dim3(256), 0, s, __VA_ARGS__);              \
        else hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, s, __VA_ARGS__);                             \
    } while (0)

void launch_multi_extract(hipStream_t s, int words, const uint64_t *sym2, const uint64_t *inv, uint64_t total_syms, int k,
                          const MultiWordsOut &out, unsigned long long *n_valid)
{
    if (!total_syms) return;
    const uint64_t n_threads = (total_syms + MULTI_PPT - 1) / MULTI_PPT;
    const dim3 grid((uint32_t)((n_threads + 255) / 256));
    GRM_MULTI_KERNEL(words, multi_extract_kernel, grid, sym2, inv, total_syms, k, out, n_valid);
}
void launch_multi_mark(hipStream_t s, int words, const MultiWords &S, const uint32_t *pos, const uint64_t *gso, uint32_t n_genomes, uint32_t n,
                       uint32_t *key_head, uint32_t *kg_head)
{
    if (!n) return;
    GRM_MULTI_KERNEL(words, multi_mark_kernel, dim3(mgrid(n)), S, pos, gso, n_genomes, n, key_head, kg_head);
}
void launch_multi_emit(hipStream_t s, int words, const MultiWords &S, const uint32_t *pos, const uint64_t *gso, uint32_t n_genomes,
                       const uint32_t *sub_start, const uint32_t *sub_key_head, const uint32_t *sub_ok, const uint32_t *key_incl,
                       const uint32_t *keep, const uint32_t *col, uint32_t n_sub, uint64_t *dict, uint64_t *matrix, uint64_t n_cols)
{
    if (!n_sub) return;
    GRM_MULTI_KERNEL(words, multi_emit_kernel, dim3(mgrid(n_sub)), S, pos, gso, n_genomes, sub_start, sub_key_head, sub_ok, key_incl, keep, col, n_sub,
                     dict, matrix, n_cols);
}
void launch_multi_set(hipStream_t s, int words, const MultiWords &S, const uint32_t *sub_start, const uint32_t *sub_ok, const uint32_t *out_pos,
                      uint32_t n_sub, uint64_t *kmers, uint32_t *counts)
{
    if (!n_sub) return;
    GRM_MULTI_KERNEL(words, multi_set_kernel, dim3(mgrid(n_sub)), S, sub_start, sub_ok, out_pos, n_sub, kmers, counts);
}
void launch_multi_split(hipStream_t s, int words, const uint64_t *keys, uint64_t n, const MultiWordsOut &out, uint64_t at)
{
    if (!n) return;
    GRM_MULTI_KERNEL(words, multi_split_kernel, dim3(mgrid(n)), keys, n, out, at);
}

// ---- the staged calls (partition / local_dict / export / set_global_dict / fill) over this path -------------------------------
// flag of every column of a batch's own matrix: 1 = carried by one of its genomes, 2 = by several
__global__ void multi_flags_kernel(const uint64_t *__restrict__ matrix, uint64_t n_rows, uint64_t n_cols, uint8_t *__restrict__ flags)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t n = 0;
        for (uint64_t r = 0; r < n_rows && n < 2; r++) n += (uint32_t)__popcll(matrix[r * n_cols + c]);
        flags[c] = (uint8_t)(n < 2 ? n : 2);
    }
}
void launch_multi_flags(hipStream_t s, const uint64_t *matrix, uint64_t n_rows, uint64_t n_cols, uint8_t *flags)
{
    if (!n_cols) return;
    hipLaunchKernelGGL(multi_flags_kernel, dim3(mgrid(n_cols)), dim3(256), 0, s, matrix, n_rows, n_cols, flags);
}
// mark[i] = entry i of a gathered list says "several carriers" (flag 2): those entries are laid out a second time behind the n
// others (at n + pos[i], pos = exclusive scan of mark), so that "in two lists, or flagged in one" becomes "twice in the sorted whole"
__global__ void multi_flag_mark_kernel(const uint8_t *__restrict__ flags, uint64_t n, uint32_t *__restrict__ mark)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) mark[i] = flags[i] >= 2 ? 1u : 0u;
}
void launch_multi_flag_mark(hipStream_t s, const uint8_t *flags, uint64_t n, uint32_t *mark)
{
    if (!n) return;
    hipLaunchKernelGGL(multi_flag_mark_kernel, dim3(mgrid(n)), dim3(256), 0, s, flags, n, mark);
}
template <int W>
__global__ void multi_split_marked_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ mark, const uint32_t *__restrict__ pos,
                                          uint64_t n, MultiWordsOut out, uint64_t at)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!mark[i]) continue;
#pragma unroll
        for (int j = 0; j < W; j++) out.w[j][at + pos[i]] = keys[(uint64_t)W * i + j];
    }
}
void launch_multi_split_marked(hipStream_t s, int words, const uint64_t *keys, const uint32_t *mark, const uint32_t *pos, uint64_t n,
                               const MultiWordsOut &out, uint64_t at)
{
    if (!n) return;
    GRM_MULTI_KERNEL(words, multi_split_marked_kernel, dim3(mgrid(n)), keys, mark, pos, n, out, at);
}
// column of every key of a sorted list `mine` (n x W words) in the sorted dictionary `dict` (n_dict x W): binary search; ~0 = not there
template <int W>
__global__ void multi_lookup_kernel(const uint64_t *__restrict__ mine, uint64_t n, const uint64_t *__restrict__ dict, uint64_t n_dict,
                                    uint32_t *__restrict__ col)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        KeyW<W> key;
#pragma unroll
        for (int j = 0; j < W; j++) key.w[j] = mine[(uint64_t)W * i + j];
        uint64_t lo = 0, hi = n_dict;                  // first entry >= key
        while (lo < hi) {
            const uint64_t m = (lo + hi) >> 1;
            KeyW<W> d;
#pragma unroll
            for (int j = 0; j < W; j++) d.w[j] = dict[(uint64_t)W * m + j];
            if (key_less<W>(d, key)) lo = m + 1; else hi = m;
        }
        bool same = lo < n_dict;
        if (same) {
#pragma unroll
            for (int j = 0; j < W; j++) same = same && dict[(uint64_t)W * lo + j] == key.w[j];
        }
        col[i] = same ? (uint32_t)lo : ~0u;
    }
}
void launch_multi_lookup(hipStream_t s, int words, const uint64_t *mine, uint64_t n, const uint64_t *dict, uint64_t n_dict, uint32_t *col)
{
    if (!n) return;
    GRM_MULTI_KERNEL(words, multi_lookup_kernel, dim3(mgrid(n)), mine, n, dict, n_dict, col);
}
// the batch's own columns moved to their places in the global matrix (zeroed by the caller): out[r][col[c]] = own[r][c]
__global__ void multi_scatter_cols_kernel(const uint64_t *__restrict__ own, uint64_t n_own, const uint32_t *__restrict__ col, uint64_t n_rows,
                                          uint64_t *__restrict__ out, uint64_t n_cols)
{
    const uint64_t total = n_rows * n_own;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = t / n_own, c = t - r * n_own;
        const uint32_t g = col[c];
        if (g != ~0u) out[r * n_cols + g] = own[t];
    }
}
void launch_multi_scatter_cols(hipStream_t s, const uint64_t *own, uint64_t n_own, const uint32_t *col, uint64_t n_rows, uint64_t *out, uint64_t n_cols)
{
    if (!n_own || !n_rows) return;
    hipLaunchKernelGGL(multi_scatter_cols_kernel, dim3(mgrid(n_rows * n_own)), dim3(256), 0, s, own, n_own, col, n_rows, out, n_cols);
}

}  // namespace grm
