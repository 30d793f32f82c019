// msm_sort.hip -- the key sort of the MSM's bucket sort.
//
// Step of VariableBaseMSM::multi_scalar_mul (arkworks/algebra/ec/src/msm/variable_base.rs:47-76: "for each scalar, add the
// base to bucket[digit - 1]") restated as a sort of (bucket, table index) pairs, so that one lane can own one bucket.
//
// msm.hip's counting sort (histogram with one atomicAdd per digit, scatter with one returning atomicSub per digit) is kept
// for small MSMs.  For a 2^20-scalar MSM over a table with window multiples (13 digits per scalar into 2^19 buckets at n = 2^20: 13.6 M pairs) its
// 27 M device-scope atomics execute at the memory side, take 1.26 ms and slow whatever runs beside them (the witness map's
// first transform pass: 0.6 ms instead of 0.09).  rocPRIM's LSD radix sort (block-local LDS ranking, no global atomics)
// sorts the same pairs on c key bits in 0.40 ms (tools/ubench_radix.hip: 34 G pairs/s).  This file only wraps that call:
// rocPRIM's templates take ~20 s to compile and msm.hip should stay quick to rebuild.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "internal.hpp"

// Bytes of temporary storage for `n` pairs (host-side query, no device work).
size_t zk_sort_pairs_temp_bytes(size_t n, unsigned key_bits) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, n, 0u, key_bits, (hipStream_t)0);
    return bytes;
}

// keys_out / vals_out = the pairs sorted by the low `key_bits` bits of the key (stable), stream-ordered on `st`.
int zk_sort_pairs(hipStream_t st, void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                  const uint32_t* vals_in, uint32_t* vals_out, size_t n, unsigned key_bits) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, key_bits, st) == hipSuccess ? 0 : -1;
}
