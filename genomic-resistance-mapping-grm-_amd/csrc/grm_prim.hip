// grm_prim.hip -- thin rocPRIM wrappers for the dictionary-sized (U ~ 1e7) plumbing steps:
// sorting the k-mer dictionary by value and prefix sums.  These are not the hot path
// (they touch U keys, the hot kernels touch every k-mer occurrence); the hand-written
// kernels are in grm_kernels.hip.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "grm_internal.h"

namespace grm {

hipError_t sort_pairs_u64_u8(hipStream_t s, const uint64_t *kin, uint64_t *kout, const uint8_t *vin, uint8_t *vout,
                             uint64_t n, void *tmp, size_t &tmp_bytes)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, 64u, s);
}
hipError_t sort_pairs_u64_u32(hipStream_t s, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                              uint64_t n, void *tmp, size_t &tmp_bytes)
{
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, 64u, s);
}
hipError_t sort_pairs_u32_u32(hipStream_t s, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                              uint64_t n, int end_bit, void *tmp, size_t &tmp_bytes)
{
    if (end_bit < 1) end_bit = 1;
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, s);
}
hipError_t exclusive_scan_u32_u64(hipStream_t s, const uint32_t *in, uint64_t *out, uint64_t n, void *tmp,
                                  size_t &tmp_bytes)
{
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), s);
}

hipError_t exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n, void *tmp, size_t &tmp_bytes)
{
    return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (uint32_t)0, (size_t)n, rocprim::plus<uint32_t>(), s);
}
hipError_t inclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n, void *tmp, size_t &tmp_bytes)
{
    return rocprim::inclusive_scan(tmp, tmp_bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), s);
}

}  // namespace grm
