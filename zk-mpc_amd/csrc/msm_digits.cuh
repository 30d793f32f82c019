// msm_digits.cuh -- the signed window digits of an MSM scalar, shared by the bucket sort (msm_sort.hip) and the small-MSM counting
// sort (msm.hip).
//
// arkworks cuts the canonical scalar into unsigned c-bit windows (ec/src/msm/variable_base.rs:52-62: `scalar.divn(w_start);
// scalar % (1 << c)`).  Here a bias sum_w 2^(off[w+1]-1) is added once, after which every window, read on its own, is the signed
// digit d + 2^(cw-1) with d in [-2^(cw-1), 2^(cw-1) - 1]: half the buckets, no carry between windows.  253-bit scalars + 2 bits of
// headroom for the bias = 255 bits in 9 words.
#pragma once
#include "devutil.cuh"
#include "ctx.hpp"

namespace zk {

struct WinOff { uint16_t off[66]; };     // window w covers bits [off[w], off[w+1])
struct Bias { uint32_t w[9]; };

// the 9 words of (canonical value of scalar i) + bias
__device__ __forceinline__ void scalar_biased_words(const void* scalars, size_t i, const Bias& bias, uint32_t (&out)[9]) {
    const Fr s = fp_ext_to_canon<FrParams>(fr_load(scalars, i));
    uint32_t w8[8];
    fp_pack<FrParams>(w8, s);
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint64_t t = (uint64_t)(k < 8 ? w8[k] : 0u) + bias.w[k] + carry;
        out[k] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
}
// two = the two 32-bit words that hold the window starting at bit `bit` (cw bits wide)
__device__ __forceinline__ int32_t signed_digit(uint64_t two, uint32_t bit, uint32_t cw) {
    const uint32_t half = 1u << (cw - 1), mask = (1u << cw) - 1;
    return (int32_t)((uint32_t)(two >> (bit & 31)) & mask) - (int32_t)half;
}

}  // namespace zk

// What msm_sort.hip::zk_msm_group needs to know about one MSM: the scalars, the window plan, how a digit becomes a bucket id and a
// table entry, and where the results go.
struct ZkGroupArgs {
    const void* scalars;      // n field elements in the reference's layout, on the device
    size_t n;
    zk::WinOff wo;
    zk::Bias bias;
    uint32_t W, NB;           // windows, buckets per window
    bool merged;              // one bucket set for all windows (a table with window multiples): entry = w * n_tab + tab_off + i
    uint32_t n_tab, tab_off;
    uint32_t NBt;             // buckets in all (NB, or W * NB)
    uint32_t lanes, seg_max;  // resident lanes of the accumulate kernel, and the segment length of a full-density input
    uint32_t* sorted;         // out: the entries grouped by bucket (room for W * n)
    uint32_t* offs;           // out: NBt + 1 bucket starts (offs[NBt] = the number of non-zero digits)
    uint32_t* ctr;            // out: ctr[3] = the segment length for this input, ctr[4] = the number of non-zero digits
};
bool zk_msm_group_supported(const ZkGroupArgs& a);
int zk_msm_group(zk_ctx* ctx, hipStream_t st, int slot, const ZkGroupArgs& a);
