// grm_coop.h -- cooperative (block / wave level) device helpers shared by the kernel files.
// wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace grm {

// ------------------------------------------------------------------------------------
// small cooperative helpers (256- or 1024-thread blocks, wave64)
// ------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only: waits for this wave's outstanding LDS operations
// (lgkmcnt) but not for global memory (vmcnt), so a returning global atomic or a load issued before
// the barrier stays in flight across it.  s_waitcnt simm16 (gfx9): vmcnt = 63 (no wait), expcnt = 7,
// lgkmcnt = 0.
__device__ __forceinline__ void lds_barrier()
{
    __asm__ volatile("" ::: "memory");      // compiler: no memory access moves across (no instruction emitted)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    __asm__ volatile("" ::: "memory");
}

// A read of LDS that must be done again every time it is written (another wave may have stored there).  Not `volatile`:
// the compiler sends a volatile access through the FLAT path (flat_load + s_waitcnt vmcnt(0) lgkmcnt(0): an address-space
// check per access, and every global load in flight is waited for); a relaxed workgroup-scope atomic load stays a ds_read.
template <class T>
__device__ __forceinline__ T lds_peek(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// exclusive prefix sum of one value per lane inside a wave (DPP-free shuffle form)
__device__ __forceinline__ uint32_t wave_scan_excl(uint32_t v)
{
    const int lane = (int)(threadIdx.x & 63);
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    return inc - v;
}

// inclusive prefix sum inside a wave with DPP moves (row_shr 1/2/4/8 inside the rows of 16, then row_bcast:15 / :31
// across them): six dependent VALU instructions instead of six ds_bpermute round trips
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t v)
{
    int s = (int)v;
    s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);
    return (uint32_t)s;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// exclusive "rightmost non-zero" scan over the block, in thread order.
// returns the carry for this thread (0 if no non-zero value precedes it in the block);
// *block_last receives the rightmost non-zero value of the whole block (0 if none).
// scratch: >= 16 ints of LDS.  Contains two __syncthreads().
__device__ __forceinline__ int block_scan_last_nonzero(int v, int *scratch, int *block_last)
{
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(inc, d);
        if (lane >= d && inc == 0) inc = o;
    }
    int exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 0;
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    int prefix = 0, last = 0;
    for (int w = 0; w < nw; w++) {
        int t = scratch[w];
        if (w < wave && t) prefix = t;
        if (t) last = t;
    }
    __syncthreads();
    *block_last = last;
    return exc ? exc : prefix;
}

// exclusive sum scan over the block (uint32); *block_total = sum of all.
__device__ __forceinline__ uint32_t block_scan_sum(uint32_t v, uint32_t *scratch, uint32_t *block_total)
{
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint32_t prefix = 0, total = 0;
    for (int w = 0; w < nw; w++) {
        uint32_t t = scratch[w];
        if (w < wave) prefix += t;
        total += t;
    }
    __syncthreads();
    *block_total = total;
    return prefix + inc - v;
}

__device__ __forceinline__ uint64_t block_scan_sum64(uint64_t v, uint64_t *scratch, uint64_t *block_total)
{
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    uint64_t prefix = 0, total = 0;
    for (int w = 0; w < nw; w++) {
        uint64_t t = scratch[w];
        if (w < wave) prefix += t;
        total += t;
    }
    __syncthreads();
    *block_total = total;
    return prefix + inc - v;
}

// genome of symbol position p: last g with genome_sym_off[g] <= p
__device__ __forceinline__ uint32_t genome_of(const uint64_t *__restrict__ gso, uint32_t n_genomes, uint64_t p)
{
    uint32_t lo = 0, hi = n_genomes;   // invariant: gso[lo] <= p < gso[hi]
    while (hi - lo > 1) {
        uint32_t m = (lo + hi) >> 1;
        if (gso[m] <= p) lo = m; else hi = m;
    }
    return lo;
}

// XCD-aware span order: workgroups are dealt round-robin to the 8 XCDs, so block b and
// b+8 share an L2.  Give each XCD one contiguous eighth of the spans: the partition's open
// write lines (one per bucket of the genome being scattered) then live in ONE L2.
__device__ __forceinline__ uint64_t xcd_span(uint32_t block, uint32_t n_spans)
{
    const uint32_t per = (n_spans + 7) / 8;
    return (uint64_t)(block & 7u) * per + (block >> 3);
}

// block-wide ordered compaction of flagged table slots: returns exclusive position of this
// thread's element within the current sweep; *sweep_total = #flagged in the sweep.
__device__ __forceinline__ uint32_t sweep_compact(bool flag, uint32_t *scratch, uint32_t *sweep_total)
{
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
    const uint64_t m = __ballot(flag);
    const uint32_t before = __popcll(m & ((1ull << lane) - 1));
    if (lane == 0) scratch[wave] = __popcll(m);
    __syncthreads();
    uint32_t prefix = 0, total = 0;
    for (int w = 0; w < nw; w++) {
        const uint32_t t = scratch[w];
        if (w < wave) prefix += t;
        total += t;
    }
    __syncthreads();
    *sweep_total = total;
    return prefix + before;
}


}  // namespace grm
