// grm_superkmer.hip -- record form of the partition (11 <= k <= 32, abundance-min 1, matrix path).
//
// The key form (grm_kernels.hip, levels 1 and 2) moves every canonical k-mer through HBM three times as an 8-byte key
// (write, read + write, read: 160 GB per 1000 x 5 Mbp).  Here the bucket of a k-mer is a function of its MINIMIZER --
// the canonical 11-mer with the smallest hash among the k - 10 it contains -- so consecutive k-mers of a sequence fall
// into the same bucket for (k - 9) / 2 positions on average (what DSK itself does with its minimizer partitions [EXT]).
// A run of consecutive valid k-mer starts with the same bucket travels as ONE 16-byte record that carries its own
// bases: 2 bits x (len + k - 1 <= 60 bases) MSB-first in x and the upper 56 bits of y, len in the low byte of y.
// One kernel, one level: 8 B per k-mer become about 2 B, and dict_build (record form) rebuilds the canonical k-mers
// from the records while it unions the bucket.  A k-mer and its reverse complement contain the same canonical
// 11-mers, so the bucket is a function of the canonical k-mer, on every rank alike.
#include "grm_internal.h"
#include "grm_device_fns.h"
#include "grm_coop.h"
#include <algorithm>
#include <cstdlib>

namespace grm {

constexpr int SK_THREADS = 512;
constexpr int SK_PPT = 32;                            // k-mer start positions per thread and step: one packed word
constexpr int SK_MAX_BITS = 14;                       // 2^14 packed 16-bit cursors = 32 KB of LDS

struct SkArgs {
    const uint64_t *sym2;
    const uint64_t *inv;
    uint64_t total_syms;
    const uint64_t *genome_sym_off;
    uint32_t n_genomes;
    int k, bb, lmax;
    int part_bits;                                    // a genome is cut into 2^part_bits parts, one workgroup each
};

// One workgroup per (genome, part): it owns the part's 2^bb record segments, so the slot of a record comes from an LDS
// cursor -- a returning GLOBAL atomic per record (6e8 of them, each to its own address) measured 20 ms of the first
// form's 35, and an add per wave to one global counter another 28 (same-address atomics serialise in L2).
// Per step a thread takes one packed word (32 start positions): minimizer bucket of every valid k-mer start, then run by
// run one LDS atomic and one 16-byte store.  W = k - SK_M + 1 m-mers per k-mer (template: the window minimum is a fixed
// pattern of register moves).
template <int W>
__global__ __launch_bounds__(SK_THREADS) void superkmer_kernel(SkArgs a, uint32_t *__restrict__ rcount, ulonglong2 *__restrict__ recs,
                                                                uint32_t rcap, uint32_t *__restrict__ part_kmers, int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint32_t *s_bk = reinterpret_cast<uint32_t *>(lds_raw);                               // [SK_THREADS * 16]: 16-bit bucket per position
    uint32_t *cur = s_bk + SK_THREADS * (SK_PPT / 2);                                     // [2^bb / 2]: two 16-bit cursors per word
    uint32_t *scratch = cur + ((1u << a.bb) + 1) / 2;                                     // [16]
    constexpr int NM = SK_PPT + W - 1;                 // m-mer positions a thread looks at
    const uint32_t B = 1u << a.bb;
    const uint32_t vg = blockIdx.x;                    // virtual genome = genome * parts + part
    const uint32_t gen = vg >> a.part_bits, part = vg & ((1u << a.part_bits) - 1u);
    const uint64_t lo = a.genome_sym_off[gen], hi = a.genome_sym_off[gen + 1];
    const uint64_t w_lo = lo >> 5, w_hi = (hi + 31) >> 5;
    const uint64_t per_part = (w_hi - w_lo + (1u << a.part_bits) - 1) >> a.part_bits;
    const uint64_t w_a = min(w_lo + (uint64_t)part * per_part, w_hi), w_b = min(w_a + per_part, w_hi);
    for (uint32_t i = threadIdx.x; i < (B + 1) / 2; i += SK_THREADS) cur[i] = 0;
    __syncthreads();
    const uint16_t *my_bk = reinterpret_cast<const uint16_t *>(s_bk) + threadIdx.x * SK_PPT;
    ulonglong2 *seg0 = recs + (uint64_t)vg * B * rcap;
    uint32_t n_valid = 0;
    bool over = false;
    for (uint64_t wbase = w_a; wbase < w_b; wbase += SK_THREADS) {
        const uint64_t wi = wbase + threadIdx.x;
        const uint64_t p0 = wi << 5;
        uint32_t valid = 0;
        uint64_t w0 = 0, w1 = 0;
        const int64_t nv = (int64_t)a.total_syms - a.k + 1 - (int64_t)p0;
        if (wi < w_b && nv > 0) {
            const uint64_t grp = p0 >> 6;
            w0 = a.sym2[wi];
            w1 = a.sym2[wi + 1];
            valid = (uint32_t)valid_starts_at(a.inv[grp], a.inv[grp + 1], (int)(p0 & 63), a.k);
            if (nv < SK_PPT) valid &= (1u << nv) - 1;
            // the first and the last word of a genome also hold positions of its neighbours
            if (p0 < lo) valid &= ~0u << (uint32_t)(lo - p0);
            if (p0 + SK_PPT > hi) valid &= hi > p0 ? ((1u << (uint32_t)(hi - p0)) - 1u) : 0u;
        }
        n_valid += (uint32_t)__popc(valid);
        uint32_t heads = 0;
        if (valid) {
            // hashes of the canonical m-mers at positions 0 .. NM-1 (rolling forward / reverse-complement words)
            uint32_t h[NM];
            constexpr uint32_t mmask = (1u << (2 * SK_M)) - 1;
            uint32_t f = (uint32_t)(w0 >> (64 - 2 * (SK_M - 1)));
            uint32_t r = (uint32_t)(revcomp_m(f, SK_M - 1) << 2);
#pragma unroll
            for (int q = 0; q < NM; q++) {
                const int si = q + SK_M - 1;               // index of the symbol that completes m-mer q (static)
                const uint32_t s = si < 32 ? (uint32_t)(w0 >> (62 - 2 * si)) & 3u : (uint32_t)(w1 >> (62 - 2 * (si - 32))) & 3u;
                f = ((f << 2) | s) & mmask;
                r = (r >> 2) | ((s ^ 2u) << (2 * (SK_M - 1)));
                h[q] = minimizer_hash(f < r ? f : r);
            }
            // minimum over windows of W: doubling (h[i] = min over [i, i + span)), then two overlapping spans
            constexpr int LV = W >= 16 ? 4 : W >= 8 ? 3 : W >= 4 ? 2 : W >= 2 ? 1 : 0;
            constexpr int SPAN = 1 << LV;
#pragma unroll
            for (int l = 0; l < LV; l++) {
#pragma unroll
                for (int i = 0; i < NM; i++)
                    if (i + (2 << l) <= NM) h[i] = min(h[i], h[i + (1 << l)]);
            }
            uint32_t bk[SK_PPT];
#pragma unroll
            for (int i = 0; i < SK_PPT; i++) bk[i] = minimizer_bucket(min(h[i], h[i + W - SPAN]), a.bb);
#pragma unroll
            for (int i = 0; i < SK_PPT; i++) {
                const bool vi = (valid >> i) & 1u;
                const bool hd = vi && (i == 0 || !((valid >> (i > 0 ? i - 1 : 0)) & 1u) || bk[i] != bk[i > 0 ? i - 1 : 0]);
                heads |= (uint32_t)hd << i;
            }
            // a record holds at most lmax k-mers (its bases must fit 120 bits): split a longer run
            if (a.lmax < SK_PPT) {
#pragma unroll
                for (int i = 29; i < SK_PPT; i++) {
                    if (i >= a.lmax && ((valid >> i) & 1u) && ((heads >> (i + 1 - a.lmax)) & ((1u << a.lmax) - 1u)) == 0) heads |= 1u << i;
                }
            }
            // the bucket of a run's first position is the only per-position value the emission needs, and it is
            // indexed by a run-time position: through LDS (each thread reads back its own 64 bytes)
#pragma unroll
            for (int i = 0; i < SK_PPT; i += 2) s_bk[threadIdx.x * (SK_PPT / 2) + i / 2] = bk[i] | (bk[i + 1] << 16);
        }
        // ends of runs: the next head, the next invalid start, or the end of the word
        const uint64_t bnd = (uint64_t)(heads | ~valid) | (1ull << SK_PPT);
        // the runs of a lane leave one per round (a lane has ~4, at most 32)
        uint32_t hd = heads;
        while (hd) {
            const int i = __ffs(hd) - 1;
            hd &= hd - 1;
            const uint32_t len = (uint32_t)__ffsll((unsigned long long)(bnd >> (i + 1)));
            ulonglong2 rec;
            rec.x = i ? ((w0 << (2 * i)) | (w1 >> (64 - 2 * i))) : w0;
            rec.y = ((w1 << (2 * i)) & ~0xffull) | len;
            const uint32_t bkt = my_bk[i];
            const uint32_t sh = (bkt & 1u) * 16u;
            const uint32_t slot = (atomicAdd(&cur[bkt >> 1], 1u << sh) >> sh) & 0xffffu;
            if (slot < rcap) seg0[(uint64_t)bkt * rcap + slot] = rec;
            else over = true;
        }
    }
    if (over) atomicExch(overflow, 1);
    __syncthreads();
    // records per segment; k-mer occurrences (= valid starts) of the part
    for (uint32_t b2 = threadIdx.x; b2 < B; b2 += SK_THREADS) {
        const uint32_t c = (cur[b2 >> 1] >> ((b2 & 1u) * 16u)) & 0xffffu;
        rcount[(uint64_t)vg * B + b2] = min(c, rcap);
    }
    uint32_t total;
    (void)block_scan_sum(n_valid, scratch, &total);
    if (threadIdx.x == 0) part_kmers[vg] = total;
}

template <int W>
static void launch_sk(hipStream_t s, const SkArgs &a, uint32_t *rcount, ulonglong2 *recs, uint32_t rcap, uint32_t *part_kmers, int *overflow)
{
    const size_t lds = (size_t)SK_THREADS * (SK_PPT / 2) * 4 + (((size_t)1 << a.bb) + 1) / 2 * 4 + 64;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(superkmer_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(superkmer_kernel<W>, dim3(a.n_genomes << a.part_bits), dim3(SK_THREADS), lds, s, a, rcount, recs, rcap, part_kmers, overflow);
}

int superkmer_lmax(int k) { return std::min(SK_PPT, 61 - k); }
int superkmer_max_bits() { return SK_MAX_BITS; }

void launch_superkmer_scatter(hipStream_t s, const KmerLaunch &L, int part_bits, uint32_t *rcount, void *recs, uint32_t rcap, uint32_t *part_kmers,
                              int *overflow)
{
    if (!L.total_syms || !L.n_genomes) return;
    SkArgs a;
    a.sym2 = L.sym2; a.inv = L.inv; a.total_syms = L.total_syms; a.genome_sym_off = L.genome_sym_off;
    a.n_genomes = L.n_genomes; a.k = L.k; a.bb = L.bb; a.lmax = superkmer_lmax(L.k);
    a.part_bits = part_bits;
    ulonglong2 *r = reinterpret_cast<ulonglong2 *>(recs);
    switch (L.k - SK_M + 1) {
#define GRM_SK_CASE(W) case W: launch_sk<W>(s, a, rcount, r, rcap, part_kmers, overflow); break;
        GRM_SK_CASE(1) GRM_SK_CASE(2) GRM_SK_CASE(3) GRM_SK_CASE(4) GRM_SK_CASE(5) GRM_SK_CASE(6) GRM_SK_CASE(7) GRM_SK_CASE(8)
        GRM_SK_CASE(9) GRM_SK_CASE(10) GRM_SK_CASE(11) GRM_SK_CASE(12) GRM_SK_CASE(13) GRM_SK_CASE(14) GRM_SK_CASE(15)
        GRM_SK_CASE(16) GRM_SK_CASE(17) GRM_SK_CASE(18) GRM_SK_CASE(19) GRM_SK_CASE(20) GRM_SK_CASE(21) GRM_SK_CASE(22)
#undef GRM_SK_CASE
        default: break;     // the host only asks for 11 <= k <= 32
    }
}

}  // namespace grm
