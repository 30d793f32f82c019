// she.hip -- the SHE (BV11 / DPSZ11 section 6) ring arithmetic of the preprocessing phase on gfx950.
//
// Replaces (reference, relative to /root/reference):
//   src/she/texts.rs:43-127                 Texts<Fq> add / sub / neg
//   src/she/encodedtext.rs:93-134           Encodedtext * {BigUint, Fq, Encodedtext}
//   src/she/polynomial.rs:152-168           poly_remainder2 (the mod X^N + 1 step of Encodedtext::mul)
//   src/she/ciphertext.rs:46-79,113-122     Ciphertext::{encrypt_from, decrypt, mul}
//   src/she/plaintext.rs:45-59, src/she/encodedtext.rs:24-52, src/she/polynomial.rs:21-69,107-119
//                                           Plaintexts::encode / Encodedtext::decode (interpolation at / evaluation on
//                                           the roots of X^N + 1 in Fr)
//
// The ring is F_q[X]/(X^N + 1), q = the MNT4-753 base prime (ark_mnt4_753::Fq, 12 x u64 Montgomery words,
// R = 2^768).  The reference multiplies with three size-2N FFTs plus an O(N^2) long division per product
// (DensePolynomial::mul, then divide_with_q_and_r); the result is just the negacyclic convolution, which is
// unique, so this file computes it with a size-N negacyclic NTT whose twist by the 2N-th root psi is merged
// into the butterflies (Cooley-Tukey forward natural -> bit-reversed with psi^brv(k) twiddles, Gentleman-Sande
// inverse bit-reversed -> natural), so no bit-reversal pass and no separate twist exist.  Ciphertext::mul
// needs 4 forward transforms and 3 inverse ones (the reference: 4 full products = 12 FFTs of twice the size).
//
// Kernel shape: field elements are 26 limbs of 29 bits (fp29.cuh); a tile of 1024 elements sits in LDS
// limb-major (26 x 1024 x 4 B = 104 KiB) and carries up to 10 butterfly levels per pass; polynomials with
// N < 1024 are packed several per tile; N > 1024 (up to 2^14, the 2-adicity limit 2N <= 2^15 of the field)
// take their top levels in global memory first.  Non power-of-two N (the reference uses N = 3 in
// src/main.rs:99-114) goes through an O(N^2) schoolbook kernel.  Everything is integer-ALU bound
// (1 352 v_mad_u64_u32 per 753-bit Montgomery product).
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "internal.hpp"

using namespace zk;

namespace {

using F7 = Fp<Fq753Params>;
constexpr int L7 = 26;
constexpr int TILE = 1024;
constexpr int LOG_TILE = 10;
constexpr int TILE_THREADS = 512;   // 2 waves per SIMD next to the 104 KiB tile (one workgroup per CU)
constexpr uint32_t SHE_MAX_LOG = FQ753_TWO_ADICITY - 1;  // 2N <= 2^15

__device__ __forceinline__ F7 f7_load(const void* base, size_t i) {
    const uint4* p = reinterpret_cast<const uint4*>(base) + 6 * i;
    uint32_t w[24];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint4 v = p[k];
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
    return fp_unpack<Fq753Params>(w);
}

__device__ __forceinline__ void f7_store(void* base, size_t i, const F7& a) {
    uint32_t w[24];
    fp_pack<Fq753Params>(w, a);
    uint4* p = reinterpret_cast<uint4*>(base) + 6 * i;
#pragma unroll
    for (int k = 0; k < 6; k++) p[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

__device__ __forceinline__ F7 f7_mul(const F7& a, const F7& b) { return fp_mul<Fq753Params>(a, b); }
__device__ __forceinline__ F7 f7_add(const F7& a, const F7& b) { return fp_add<Fq753Params>(a, b); }
__device__ __forceinline__ F7 f7_sub(const F7& a, const F7& b) { return fp_sub<Fq753Params>(a, b); }

// ---- the lazy domain of the transform butterflies ---------------------------------------------------------------------------
// 26 limbs of 29 bits hold 754 bits and q = 0.4427 * 2^754: a value below 2.26 q keeps EVERY limb below 2^29, so it is a legal
// operand of the product-scanning Montgomery product (fp29.cuh: 52 * 2^58 < 2^64 per column) without being reduced, and that
// product (RI = 2^754) of x < 2.26 q by y < q lands below 0.4427 x + q < 2 q by itself: no conditional subtraction anywhere.
//   f7l_red(a)      any a < 7.9 q with u32 limbs -> a - k q in [0, 2.01 q), limbs < 2^29: k = floor(top * MQ / 2^56) <= a / q
//                   estimated from the top limb (k >= a / q - 2), the subtraction folded into the carry pass as a + k (2^754 - q)
//   f7l_sub<K>(a,b) a + K q - b limb by limb, no borrow (OFF<K>: K q written with limbs 0..24 in [2^29, 2^30)): b's limbs < 2^29,
//                   b < 1.89 q for K = 2 (a product), b < 2.26 q for K = 3
// Forward (Cooley-Tukey):   V = x[j+t] S < 1.89 q;  x[j] = red(U + V),  x[j+t] = red(U + 2q - V)          (inputs < 2.01 q)
// Inverse (Gentleman-Sande): x[j] = red(U + X),  x[j+t] = red(U + 3q - X) S < 1.89 q
// Bounds and the worst product column are recomputed by tests/test_abi.py::test_she_lazy_domain_bounds.
using LZ7 = Fq753Lazy;
__device__ __forceinline__ F7 f7l_add(const F7& a, const F7& b) {
    F7 r;
#pragma unroll
    for (int i = 0; i < L7; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <int K>
__device__ __forceinline__ F7 f7l_sub(const F7& a, const F7& b) {
    static_assert(K == 2 || K == 3, "offsets generated: 2 q, 3 q");
    F7 r;
#pragma unroll
    for (int i = 0; i < L7; i++) r.l[i] = a.l[i] + (K == 2 ? LZ7::OFF2[i] : LZ7::OFF3[i]) - b.l[i];
    return r;
}
__device__ __forceinline__ F7 f7l_red(const F7& a) {
    const uint32_t t = a.l[L7 - 1] + (a.l[L7 - 2] >> 29);
    const uint32_t k = (uint32_t)(((uint64_t)t * LZ7::MQ) >> 56);
    F7 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L7; i++) {
        const uint64_t acc = (uint64_t)k * LZ7::RC[i] + (uint64_t)(a.l[i] + c);     // a.l[i] + c < 2^32: limbs < 2^31, c < 2^7
        r.l[i] = (uint32_t)acc & MASK29;             // (the top limb too: k 2^754 drops off)
        c = (uint32_t)(acc >> 29);
    }
    return r;
}
__device__ __forceinline__ F7 f7l_mul(const F7& a, const F7& b) { return fp_mul_lazy<Fq753Params>(a, b); }
// a < 2.01 q, limbs < 2^29 -> a mod q
__device__ __forceinline__ F7 f7l_canon(const F7& a) {
    const F7 t = fp_reduce_once<Fq753Params>(a.l);
    return fp_reduce_once<Fq753Params>(t.l);
}

// limb-major table of n elements (internal form)
__device__ __forceinline__ F7 tab_load(const uint32_t* tab, uint32_t n, uint32_t i) {
    F7 r;
#pragma unroll
    for (int k = 0; k < L7; k++) r.l[k] = tab[(size_t)k * n + i];
    return r;
}
__device__ __forceinline__ void tab_store(uint32_t* tab, uint32_t n, uint32_t i, const F7& a) {
#pragma unroll
    for (int k = 0; k < L7; k++) tab[(size_t)k * n + i] = a.l[k];
}
__device__ __forceinline__ F7 lds_get(const uint32_t* s, uint32_t i) {
    F7 r;
#pragma unroll
    for (int k = 0; k < L7; k++) r.l[k] = s[k * TILE + i];
    return r;
}
__device__ __forceinline__ void lds_put(uint32_t* s, uint32_t i, const F7& a) {
#pragma unroll
    for (int k = 0; k < L7; k++) s[k * TILE + i] = a.l[k];
}

struct F7K { uint32_t l[L7]; };
__device__ __forceinline__ F7 f7k(const F7K& k) {
    F7 r;
#pragma unroll
    for (int i = 0; i < L7; i++) r.l[i] = k.l[i];
    return r;
}

// psi[k] = psi^brv(k), psi_inv[k] = psi^-brv(k) over log_n bits; scale[0] = RI^2 / (RE n) (RAW operand that both
// divides by n and repairs the ext*ext pointwise products).  psi = ROOT^(2^(14 - log_n)) is a primitive 2n-th root.
__global__ void __launch_bounds__(64) k_she_tables(uint32_t* psi, uint32_t* psi_inv, uint32_t* scale, uint32_t log_n) {
    uint32_t n = 1u << log_n;
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    F7 w = fp_const<Fq753Params>(Fq753Params::TWO_ADIC_ROOT), wi = fp_const<Fq753Params>(Fq753Params::TWO_ADIC_ROOT_INV);
    for (uint32_t i = 0; i < SHE_MAX_LOG - log_n; i++) { w = f7_mul(w, w); wi = f7_mul(wi, wi); }
    uint32_t e = log_n ? (__brev(k) >> (32 - log_n)) : 0;
    F7 a = fp_one<Fq753Params>(), b = a;
    for (uint32_t bit = 0; bit < log_n; bit++) {
        if ((e >> bit) & 1) { a = f7_mul(a, w); b = f7_mul(b, wi); }
        w = f7_mul(w, w);
        wi = f7_mul(wi, wi);
    }
    tab_store(psi, n, k, a);
    tab_store(psi_inv, n, k, b);
    if (k == 0) {
        F7 ninv = fp_one<Fq753Params>(), half = fp_const<Fq753Params>(Fq753Params::INV2);
        for (uint32_t i = 0; i < log_n; i++) ninv = f7_mul(ninv, half);
        F7 s = f7_mul(fp_const<Fq753Params>(Fq753Params::EXT_TO_INT), ninv);
        for (int i = 0; i < L7; i++) scale[i] = s.l[i];
    }
}

// A "row" is one polynomial of n coefficients.  Row r of an operand lives at
//   base + ((r / rpg) * gstride + (r % rpg) * n) elements,
// which addresses c0/c1/c2 of a batch of ciphertexts (rpg = rows taken per ciphertext, gstride = 3n) as well as a
// plain batch (rpg = 1, gstride = n) and a single shared polynomial (gstride = 0, rpg = 1).
struct RowMap {
    const void* base;
    uint32_t rpg;
    uint64_t gstride;
};
__device__ __forceinline__ size_t row_elem(const RowMap& m, uint32_t n, uint64_t row, uint32_t j) {
    return (size_t)((row / m.rpg) * m.gstride + (row % m.rpg) * (uint64_t)n + j);
}

// Forward: global levels with t >= TILE (n > TILE only).  One butterfly per thread, in place on dst (packed rows).
__global__ void __launch_bounds__(256) k_she_fwd_global(void* data, const uint32_t* psi, uint32_t log_n, uint32_t log_t,
                                                        uint64_t n_bfly) {
    uint32_t n = 1u << log_n, t = 1u << log_t, m = n >> (log_t + 1);
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < n_bfly; g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t row = g >> (log_n - 1);
        uint32_t b = (uint32_t)(g & ((n >> 1) - 1));
        uint32_t i = b >> log_t, off = b & (t - 1);
        size_t j = (size_t)row * n + 2 * (size_t)i * t + off;
        F7 S = tab_load(psi, n, m + i);
        F7 U = f7_load(data, j), V = f7l_mul(f7_load(data, j + t), S);
        f7_store(data, j, f7l_red(f7l_add(U, V)));
        f7_store(data, j + t, f7l_red(f7l_sub<2>(U, V)));
    }
}

// Forward LDS pass: levels t = min(n, TILE)/2 ... 1 on one tile.  FIRST: read through the row map (n <= TILE), else
// read dst in place (the global levels already copied).
// The transform-domain values it writes are lazy representatives (< 2.01 q, limbs < 2^29: they fit the 768-bit words) unless
// `canon` asks for reduced ones (a consumer outside the transform kernels: the s * s of decrypt).
template <bool FIRST>
__global__ void __launch_bounds__(TILE_THREADS) k_she_fwd_tile(RowMap src, void* dst, const uint32_t* psi, uint32_t log_n,
                                                      uint64_t n_elems, int canon) {
    extern __shared__ uint32_t lds[];
    uint32_t n = 1u << log_n;
    uint64_t e0 = (uint64_t)blockIdx.x * TILE;
    for (uint32_t l = threadIdx.x; l < TILE; l += TILE_THREADS) {
        uint64_t e = e0 + l;
        if (e < n_elems) {
            F7 v = FIRST ? f7_load(src.base, row_elem(src, n, e >> log_n, (uint32_t)(e & (n - 1)))) : f7_load(dst, e);
            lds_put(lds, l, v);
        }
    }
    __syncthreads();
    int top = (int)(log_n < LOG_TILE ? log_n : LOG_TILE) - 1;
    for (int log_t = top; log_t >= 0; log_t--) {
        uint32_t t = 1u << log_t, m = n >> (log_t + 1);
        for (uint32_t lb = threadIdx.x; lb < TILE / 2; lb += TILE_THREADS) {
            uint64_t g = (e0 >> 1) + lb;                     // butterfly index over the whole batch
            if (2 * g >= n_elems) continue;
            uint64_t row = log_n ? (g >> (log_n - 1)) : g;
            uint32_t b = (uint32_t)(g & ((n >> 1) - 1));
            uint32_t i = b >> log_t, off = b & (t - 1);
            uint32_t j = (uint32_t)(row * n + 2 * (uint64_t)i * t + off - e0);
            F7 S = tab_load(psi, n, m + i);
            F7 U = lds_get(lds, j), V = f7l_mul(lds_get(lds, j + t), S);
            lds_put(lds, j, f7l_red(f7l_add(U, V)));
            lds_put(lds, j + t, f7l_red(f7l_sub<2>(U, V)));
        }
        __syncthreads();
    }
    for (uint32_t l = threadIdx.x; l < TILE; l += TILE_THREADS) {
        uint64_t e = e0 + l;
        if (e < n_elems) f7_store(dst, e, canon ? f7l_canon(lds_get(lds, l)) : lds_get(lds, l));
    }
}

// Inverse LDS pass with the point-wise stage fused into the load:
//   v = x0*y0 (+ x1*y1) (negated if NEG), all operands in the transform domain, then levels t = 1 ... min(n,TILE)/2.
// LAST: multiply by the scale constant and write through the output row map; else write the packed work buffer.
struct InvArgs {
    RowMap x0, y0, x1, y1;   // x1.base == nullptr: single product
    RowMap out;
    void* work;              // packed rows (used when n > TILE)
    int negate;
};
template <bool LAST>
__global__ void __launch_bounds__(TILE_THREADS) k_she_inv_tile(InvArgs a, const uint32_t* psi_inv, const uint32_t* scale,
                                                      uint32_t log_n, uint64_t n_elems) {
    extern __shared__ uint32_t lds[];
    uint32_t n = 1u << log_n;
    uint64_t e0 = (uint64_t)blockIdx.x * TILE;
    for (uint32_t l = threadIdx.x; l < TILE; l += TILE_THREADS) {
        uint64_t e = e0 + l;
        if (e < n_elems) {
            uint64_t row = e >> log_n;
            uint32_t j = (uint32_t)(e & (n - 1));
            // operands < 2.01 q: each product < 2.79 q, their sum < 5.6 q
            F7 v = f7l_mul(f7_load(a.x0.base, row_elem(a.x0, n, row, j)), f7_load(a.y0.base, row_elem(a.y0, n, row, j)));
            if (a.x1.base)
                v = f7l_add(v, f7l_mul(f7_load(a.x1.base, row_elem(a.x1, n, row, j)), f7_load(a.y1.base, row_elem(a.y1, n, row, j))));
            v = f7l_red(v);
            if (a.negate) v = f7l_red(f7l_sub<3>(fp_zero<Fq753Params>(), v));
            lds_put(lds, l, v);
        }
    }
    __syncthreads();
    int top = (int)(log_n < LOG_TILE ? log_n : LOG_TILE);
    for (int log_t = 0; log_t < top; log_t++) {
        uint32_t t = 1u << log_t, h = n >> (log_t + 1);
        for (uint32_t lb = threadIdx.x; lb < TILE / 2; lb += TILE_THREADS) {
            uint64_t g = (e0 >> 1) + lb;
            if (2 * g >= n_elems) continue;
            uint64_t row = log_n ? (g >> (log_n - 1)) : g;
            uint32_t b = (uint32_t)(g & ((n >> 1) - 1));
            uint32_t i = b >> log_t, off = b & (t - 1);
            uint32_t j = (uint32_t)(row * n + 2 * (uint64_t)i * t + off - e0);
            F7 S = tab_load(psi_inv, n, h + i);
            F7 U = lds_get(lds, j), V = lds_get(lds, j + t);
            lds_put(lds, j, f7l_red(f7l_add(U, V)));
            lds_put(lds, j + t, f7l_mul(f7l_red(f7l_sub<3>(U, V)), S));
        }
        __syncthreads();
    }
    F7 sc;
    if (LAST) {
#pragma unroll
        for (int k = 0; k < L7; k++) sc.l[k] = scale[k];
    }
    for (uint32_t l = threadIdx.x; l < TILE; l += TILE_THREADS) {
        uint64_t e = e0 + l;
        if (e >= n_elems) continue;
        F7 v = lds_get(lds, l);
        if (LAST) {
            f7_store(const_cast<void*>(a.out.base), row_elem(a.out, n, e >> log_n, (uint32_t)(e & (n - 1))), f7_mul(v, sc));
        } else {
            f7_store(a.work, e, v);
        }
    }
}

// Inverse global level (n > TILE): t >= TILE.  LAST applies the scale and writes through the row map.
template <bool LAST>
__global__ void __launch_bounds__(256) k_she_inv_global(void* work, RowMap out, const uint32_t* psi_inv, const uint32_t* scale,
                                                        uint32_t log_n, uint32_t log_t, uint64_t n_bfly) {
    uint32_t n = 1u << log_n, t = 1u << log_t, h = n >> (log_t + 1);
    F7 sc;
    if (LAST) {
#pragma unroll
        for (int k = 0; k < L7; k++) sc.l[k] = scale[k];
    }
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < n_bfly; g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t row = g >> (log_n - 1);
        uint32_t b = (uint32_t)(g & ((n >> 1) - 1));
        uint32_t i = b >> log_t, off = b & (t - 1);
        uint32_t jj = 2 * i * t + off;
        size_t j = (size_t)row * n + jj;
        F7 S = tab_load(psi_inv, n, h + i);
        F7 U = f7_load(work, j), V = f7_load(work, j + t);
        F7 lo = f7l_red(f7l_add(U, V)), hi = f7l_mul(f7l_red(f7l_sub<3>(U, V)), S);      // lazy: < 2.01 q; f7_mul by sc reduces
        if (LAST) {
            f7_store(const_cast<void*>(out.base), row_elem(out, n, row, jj), f7_mul(lo, sc));
            f7_store(const_cast<void*>(out.base), row_elem(out, n, row, jj + t), f7_mul(hi, sc));
        } else {
            f7_store(work, j, lo);
            f7_store(work, j + t, hi);
        }
    }
}

__global__ void __launch_bounds__(256) k_she_gather(RowMap src, void* dst, uint32_t n, uint64_t n_elems) {
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_elems; e += (uint64_t)gridDim.x * blockDim.x)
        f7_store(dst, e, f7_load(src.base, row_elem(src, n, e / n, (uint32_t)(e % n))));
}

// Schoolbook negacyclic product for any n: out_k = sum_{i<=k} a_i b_{k-i} - sum_{i>k} a_i b_{n+k-i}.
__global__ void __launch_bounds__(256) k_she_schoolbook(RowMap x0, RowMap y0, RowMap x1, RowMap y1, RowMap out, int negate,
                                                        uint32_t n, uint64_t n_elems) {
    const F7 fix = fp_const<Fq753Params>(Fq753Params::EXT_TO_INT);
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_elems; e += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t row = e / n;
        uint32_t k = (uint32_t)(e % n);
        F7 acc = fp_zero<Fq753Params>();
        for (int term = 0; term < 2; term++) {
            const RowMap& X = term ? x1 : x0;
            const RowMap& Y = term ? y1 : y0;
            if (!X.base) continue;
            for (uint32_t i = 0; i < n; i++) {
                F7 p = i <= k ? f7_mul(f7_load(X.base, row_elem(X, n, row, i)), f7_load(Y.base, row_elem(Y, n, row, k - i)))
                              : f7_mul(f7_load(X.base, row_elem(X, n, row, i)), f7_load(Y.base, row_elem(Y, n, row, n + k - i)));
                acc = i <= k ? f7_add(acc, p) : f7_sub(acc, p);
            }
        }
        acc = f7_mul(acc, fix);
        if (negate) acc = fp_neg<Fq753Params>(acc);
        f7_store(const_cast<void*>(out.base), row_elem(out, n, row, k), acc);
    }
}

enum { SHE_ADD = 1, SHE_SUB = 2, SHE_MUL = 0, SHE_NEG = 3 };

__global__ void __launch_bounds__(256) k_she_vec_op(int op, const void* a, const void* b, void* out, size_t n) {
    const F7 fix = fp_const<Fq753Params>(Fq753Params::EXT_TO_INT);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F7 x = f7_load(a, i), z;
        if (op == SHE_NEG) z = fp_neg<Fq753Params>(x);
        else {
            F7 y = f7_load(b, i);
            z = op == SHE_ADD ? f7_add(x, y) : op == SHE_SUB ? f7_sub(x, y) : f7_mul(f7_mul(x, y), fix);
        }
        f7_store(out, i, z);
    }
}

__global__ void __launch_bounds__(256) k_she_vec_scale(const void* a, F7K k, void* out, size_t n) {
    const F7 kk = f7k(k);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        f7_store(out, i, f7_mul(f7_load(a, i), kk));
}

// Ciphertext::encrypt_from tail: c0 = bv + w*p + e, c1 = av + u*p, c2 = 0.  ct holds bv, av in c0, c1 on entry.
__global__ void __launch_bounds__(256) k_she_encrypt_tail(void* ct, const void* e, const void* r, F7K p, uint32_t n, uint64_t batch) {
    const F7 pp = f7k(p);
    uint64_t total = batch * n;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c = g / n;
        uint32_t j = (uint32_t)(g % n);
        size_t o = (size_t)c * 3 * n + j;
        F7 u = f7_load(r, o), w = f7_load(r, o + 2 * (size_t)n);
        f7_store(ct, o, f7_add(f7_add(f7_load(ct, o), f7_mul(w, pp)), f7_load(e, (size_t)c * n + j)));
        f7_store(ct, o + n, f7_add(f7_load(ct, o + n), f7_mul(u, pp)));
        f7_store(ct, o + 2 * (size_t)n, fp_zero<Fq753Params>());
    }
}

// out = c0 - t   (t = s*c1 + s*s*c2, packed rows)
__global__ void __launch_bounds__(256) k_she_decrypt_tail(const void* ct, const void* t, void* out, uint32_t n, uint64_t batch) {
    uint64_t total = batch * n;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c = g / n;
        uint32_t j = (uint32_t)(g % n);
        f7_store(out, g, f7_sub(f7_load(ct, (size_t)c * 3 * n + j), f7_load(t, g)));
    }
}

// ---- Fr <-> Fq753 for encode / decode ----
// Plaintexts::encode tail: Fr coefficient (ext) -> canonical integer -> Fq element (ext), after the twist by zeta^-j.
__global__ void __launch_bounds__(256) k_she_fr_to_fq(const void* fr, void* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr c = fp_ext_to_canon<FrParams>(fr_load(fr, i));
        F7 v = fp_zero<Fq753Params>();
#pragma unroll
        for (int k = 0; k < 9; k++) v.l[k] = c.l[k];
        f7_store(out, i, fp_canon_to_ext<Fq753Params>(v));
    }
}

// Encodedtext::decode head (src/she/encodedtext.rs:29-45): canonical integer, minus (q mod p) when above q/2, mod p.
__global__ void __launch_bounds__(256) k_she_fq_to_fr(const void* fq, void* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F7 c = fp_ext_to_canon<Fq753Params>(f7_load(fq, i));
        // c > HALF ?
        int32_t borrow = 0;
#pragma unroll
        for (int k = 0; k < L7; k++) borrow = ((int32_t)Fq753Params::HALF[k] - (int32_t)c.l[k] + borrow) >> 29;
        if (borrow < 0) {
            int32_t cy = 0;
#pragma unroll
            for (int k = 0; k < L7; k++) {
                int32_t s = (int32_t)c.l[k] - (int32_t)Fq753Params::MOD_FR[k] + cy;
                c.l[k] = (uint32_t)s & MASK29;
                cy = s >> 29;
            }
        }
        Fr acc = fp_zero<FrParams>();
#pragma unroll
        for (int k = 0; k < L7; k++) {
            Fr limb = fp_zero<FrParams>();
            limb.l[0] = c.l[k];
            Fr kk;
#pragma unroll
            for (int q = 0; q < 9; q++) kk.l[q] = FR_FOLD29[k][q];
            acc = fr_add(acc, fr_mul(limb, kk));
        }
        fr_store(out, i, acc);
    }
}

// v[row][j] *= tw[j]  (tw: n Fr elements, ext form)
__global__ void __launch_bounds__(256) k_fr_rowscale(void* v, const void* tw, uint32_t n, uint64_t total) {
    const Fr fix = fp_const<FrParams>(FrParams::EXT_TO_INT);
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x)
        fr_store(v, g, fr_mul(fr_mul(fr_load(v, g), fr_load(tw, g % n)), fix));
}

F7K to_f7k(const F7& a) {
    F7K k;
    for (int i = 0; i < L7; i++) k.l[i] = a.l[i];
    return k;
}

struct SheTables { uint32_t *psi, *psi_inv, *scale; };

int she_tables(zk_ctx* ctx, uint32_t log_n, SheTables* t) {
    uint32_t n = 1u << log_n;
    char name[32];
    snprintf(name, sizeof name, "she_tab_%u", log_n);
    bool fresh = ctx->slots.find(name) == ctx->slots.end();
    size_t words = (size_t)2 * L7 * n + 32;
    void* p;
    ZK_TRY(zk_scratch(ctx, name, words * 4, &p));
    t->psi = (uint32_t*)p;
    t->psi_inv = t->psi + (size_t)L7 * n;
    t->scale = t->psi_inv + (size_t)L7 * n;
    if (fresh) {
        hipLaunchKernelGGL(k_she_tables, (n + 63) / 64, 64, 0, ctx->stream, t->psi, t->psi_inv, t->scale, log_n);
        ZK_HIP(ctx, hipGetLastError());
    }
    return ZK_OK;
}

bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
uint32_t ilog2(size_t n) { uint32_t l = 0; while (((size_t)1 << l) < n) l++; return l; }

constexpr size_t LDS_BYTES = (size_t)L7 * TILE * 4;

int she_lds_attr(zk_ctx* ctx) {
    if (ctx->flags["she_lds"]) return ZK_OK;
    ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_she_fwd_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_she_fwd_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_she_inv_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_she_inv_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    ctx->flags["she_lds"] = 1;
    return ZK_OK;
}

// dst (packed rows) <- forward negacyclic NTT of `rows` polynomials read through src.
int she_forward(zk_ctx* ctx, const SheTables& tb, RowMap src, void* dst, uint32_t log_n, uint64_t rows, int canon = 0) {
    uint32_t n = 1u << log_n;
    uint64_t n_elems = rows * n;
    unsigned tiles = (unsigned)((n_elems + TILE - 1) / TILE);
    ZK_TRY(she_lds_attr(ctx));
    if (log_n <= LOG_TILE) {
        hipLaunchKernelGGL(k_she_fwd_tile<true>, tiles, TILE_THREADS, LDS_BYTES, ctx->stream, src, dst, tb.psi, log_n, n_elems, canon);
    } else {
        hipLaunchKernelGGL(k_she_gather, zk_grid(n_elems, 256), 256, 0, ctx->stream, src, dst, n, n_elems);
        for (int log_t = (int)log_n - 1; log_t >= LOG_TILE; log_t--)
            hipLaunchKernelGGL(k_she_fwd_global, zk_grid(n_elems / 2, 256), 256, 0, ctx->stream, dst, tb.psi, log_n, (uint32_t)log_t,
                               n_elems / 2);
        hipLaunchKernelGGL(k_she_fwd_tile<false>, tiles, TILE_THREADS, LDS_BYTES, ctx->stream, src, dst, tb.psi, log_n, n_elems, canon);
    }
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

// out rows <- inverse transform of x0*y0 (+ x1*y1), optionally negated.  Operands are transform-domain rows.
int she_inverse(zk_ctx* ctx, const SheTables& tb, InvArgs a, uint32_t log_n, uint64_t rows) {
    uint32_t n = 1u << log_n;
    uint64_t n_elems = rows * n;
    unsigned tiles = (unsigned)((n_elems + TILE - 1) / TILE);
    ZK_TRY(she_lds_attr(ctx));
    if (log_n <= LOG_TILE) {
        hipLaunchKernelGGL(k_she_inv_tile<true>, tiles, TILE_THREADS, LDS_BYTES, ctx->stream, a, tb.psi_inv, tb.scale, log_n, n_elems);
    } else {
        ZK_TRY(zk_scratch(ctx, "she_inv_work", n_elems * 96, &a.work));
        hipLaunchKernelGGL(k_she_inv_tile<false>, tiles, TILE_THREADS, LDS_BYTES, ctx->stream, a, tb.psi_inv, tb.scale, log_n, n_elems);
        for (uint32_t log_t = LOG_TILE; log_t < log_n; log_t++) {
            if (log_t + 1 == log_n)
                hipLaunchKernelGGL(k_she_inv_global<true>, zk_grid(n_elems / 2, 256), 256, 0, ctx->stream, a.work, a.out, tb.psi_inv,
                                   tb.scale, log_n, log_t, n_elems / 2);
            else
                hipLaunchKernelGGL(k_she_inv_global<false>, zk_grid(n_elems / 2, 256), 256, 0, ctx->stream, a.work, a.out, tb.psi_inv,
                                   tb.scale, log_n, log_t, n_elems / 2);
        }
    }
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

RowMap rows_of(const void* base, uint32_t rpg, uint64_t gstride) { return RowMap{base, rpg, gstride}; }
const RowMap NO_ROWS{nullptr, 1, 0};

bool ntt_path(size_t n) { return is_pow2(n) && n >= 4 && ilog2(n) <= SHE_MAX_LOG; }

// out <- x0 (*) y0 [+ x1 (*) y1] in F_q[X]/(X^n+1), time-domain operands given through row maps.
// Forward transforms of the distinct operands are the caller's business on the NTT path; this helper is the
// schoolbook path used for the n the transform does not cover.
int she_schoolbook(zk_ctx* ctx, RowMap x0, RowMap y0, RowMap x1, RowMap y1, RowMap out, int negate, uint32_t n, uint64_t rows) {
    uint64_t n_elems = rows * n;
    hipLaunchKernelGGL(k_she_schoolbook, zk_grid(n_elems, 256), 256, 0, ctx->stream, x0, y0, x1, y1, out, negate, n, n_elems);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

int check_n(zk_ctx* ctx, size_t n, const char* who) {
    if (n == 0 || n > ((size_t)1 << SHE_MAX_LOG)) {
        ctx->last_error = std::string(who) + ": degree must be in 1..2^14 (2N <= 2^15, the 2-adicity of the MNT4-753 base field)";
        return ZK_ERR_ARG;
    }
    return ZK_OK;
}

}  // namespace

extern "C" int zk_she_vec_op_dev(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (n && (!a || !out))) return ZK_ERR_ARG;
    if (op != ZK_OP_MUL && op != ZK_OP_ADD && op != ZK_OP_SUB && op != ZK_OP_NEG) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_she_vec_op_dev: unknown op");
    if (op != ZK_OP_NEG && n && !b) return ZK_ERR_ARG;
    if (n == 0) return ZK_OK;
    hipLaunchKernelGGL(k_she_vec_op, zk_grid(n, 256), 256, 0, ctx->stream, op, a, b, out, n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_she_vec_scale_dev(zk_ctx* ctx, const void* a, const zk_fq753* k, void* out, size_t n) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !k || (n && (!a || !out))) return ZK_ERR_ARG;
    if (n == 0) return ZK_OK;
    F7 kk = fp_ext_to_int<Fq753Params>(host_load_ext<Fq753Params>(k->l));
    hipLaunchKernelGGL(k_she_vec_scale, zk_grid(n, 256), 256, 0, ctx->stream, a, to_f7k(kk), out, n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_she_negacyclic_mul_dev(zk_ctx* ctx, const void* a, const void* b, void* out, size_t n, size_t batch) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (batch && (!a || !b || !out))) return ZK_ERR_ARG;
    ZK_TRY(check_n(ctx, n, "zk_she_negacyclic_mul_dev"));
    if (batch == 0) return ZK_OK;
    if (!ntt_path(n))
        return she_schoolbook(ctx, rows_of(a, 1, n), rows_of(b, 1, n), NO_ROWS, NO_ROWS, rows_of(out, 1, n), 0, (uint32_t)n, batch);
    uint32_t log_n = ilog2(n);
    SheTables tb;
    ZK_TRY(she_tables(ctx, log_n, &tb));
    void *fa, *fb;
    ZK_TRY(zk_scratch(ctx, "she_fa", batch * n * 96, &fa));
    ZK_TRY(zk_scratch(ctx, "she_fb", batch * n * 96, &fb));
    ZK_TRY(she_forward(ctx, tb, rows_of(a, 1, n), fa, log_n, batch));
    ZK_TRY(she_forward(ctx, tb, rows_of(b, 1, n), fb, log_n, batch));
    InvArgs ia{rows_of(fa, 1, n), rows_of(fb, 1, n), NO_ROWS, NO_ROWS, rows_of(out, 1, n), nullptr, 0};
    return she_inverse(ctx, tb, ia, log_n, batch);
    ZK_API_END
}

extern "C" int zk_she_ciphertext_mul_dev(zk_ctx* ctx, const void* x, const void* y, void* out, size_t n, size_t batch) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (batch && (!x || !y || !out))) return ZK_ERR_ARG;
    ZK_TRY(check_n(ctx, n, "zk_she_ciphertext_mul_dev"));
    if (batch == 0) return ZK_OK;
    const char* xb = (const char*)x;
    const char* yb = (const char*)y;
    char* ob = (char*)out;
    size_t poly = n * 96;
    if (!ntt_path(n)) {
        uint32_t nn = (uint32_t)n;
        ZK_TRY(she_schoolbook(ctx, rows_of(xb, 1, 3 * n), rows_of(yb, 1, 3 * n), NO_ROWS, NO_ROWS, rows_of(ob, 1, 3 * n), 0, nn, batch));
        ZK_TRY(she_schoolbook(ctx, rows_of(xb, 1, 3 * n), rows_of(yb + poly, 1, 3 * n), rows_of(xb + poly, 1, 3 * n), rows_of(yb, 1, 3 * n),
                              rows_of(ob + poly, 1, 3 * n), 0, nn, batch));
        return she_schoolbook(ctx, rows_of(xb + poly, 1, 3 * n), rows_of(yb + poly, 1, 3 * n), NO_ROWS, NO_ROWS, rows_of(ob + 2 * poly, 1, 3 * n),
                              1, nn, batch);
    }
    uint32_t log_n = ilog2(n);
    SheTables tb;
    ZK_TRY(she_tables(ctx, log_n, &tb));
    char *fx, *fy;   // [batch][2][n] transforms of (c0, c1)
    ZK_TRY(zk_scratch(ctx, "she_fa", batch * 2 * poly, (void**)&fx));
    ZK_TRY(zk_scratch(ctx, "she_fb", batch * 2 * poly, (void**)&fy));
    ZK_TRY(she_forward(ctx, tb, rows_of(xb, 2, 3 * n), fx, log_n, 2 * batch));
    ZK_TRY(she_forward(ctx, tb, rows_of(yb, 2, 3 * n), fy, log_n, 2 * batch));
    RowMap x0 = rows_of(fx, 1, 2 * n), x1 = rows_of(fx + poly, 1, 2 * n);
    RowMap y0 = rows_of(fy, 1, 2 * n), y1 = rows_of(fy + poly, 1, 2 * n);
    // c0 = x0 y0 ; c1 = x0 y1 + x1 y0 ; c2 = -(x1 y1)    (src/she/ciphertext.rs:116-120)
    ZK_TRY(she_inverse(ctx, tb, InvArgs{x0, y0, NO_ROWS, NO_ROWS, rows_of(ob, 1, 3 * n), nullptr, 0}, log_n, batch));
    ZK_TRY(she_inverse(ctx, tb, InvArgs{x0, y1, x1, y0, rows_of(ob + poly, 1, 3 * n), nullptr, 0}, log_n, batch));
    return she_inverse(ctx, tb, InvArgs{x1, y1, NO_ROWS, NO_ROWS, rows_of(ob + 2 * poly, 1, 3 * n), nullptr, 1}, log_n, batch);
    ZK_API_END
}

extern "C" int zk_she_encrypt_dev(zk_ctx* ctx, const void* e, const void* pk_a, const void* pk_b, const void* r, const zk_fq753* p,
                                  void* out, size_t n, size_t batch) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !p || (batch && (!e || !pk_a || !pk_b || !r || !out))) return ZK_ERR_ARG;
    ZK_TRY(check_n(ctx, n, "zk_she_encrypt_dev"));
    if (batch == 0) return ZK_OK;
    const char* rb = (const char*)r;
    char* ob = (char*)out;
    size_t poly = n * 96;
    RowMap v = rows_of(rb + poly, 1, 3 * n);   // r = u | v | w  (src/she/ciphertext.rs:53-66)
    if (!ntt_path(n)) {
        uint32_t nn = (uint32_t)n;
        ZK_TRY(she_schoolbook(ctx, rows_of(pk_b, 1, 0), v, NO_ROWS, NO_ROWS, rows_of(ob, 1, 3 * n), 0, nn, batch));
        ZK_TRY(she_schoolbook(ctx, rows_of(pk_a, 1, 0), v, NO_ROWS, NO_ROWS, rows_of(ob + poly, 1, 3 * n), 0, nn, batch));
    } else {
        uint32_t log_n = ilog2(n);
        SheTables tb;
        ZK_TRY(she_tables(ctx, log_n, &tb));
        char *fv, *fk;
        ZK_TRY(zk_scratch(ctx, "she_fa", batch * poly, (void**)&fv));
        ZK_TRY(zk_scratch(ctx, "she_fb", 2 * poly, (void**)&fk));
        ZK_TRY(she_forward(ctx, tb, v, fv, log_n, batch));
        ZK_TRY(she_forward(ctx, tb, rows_of(pk_a, 1, n), fk, log_n, 1));
        ZK_TRY(she_forward(ctx, tb, rows_of(pk_b, 1, n), fk + poly, log_n, 1));
        ZK_TRY(she_inverse(ctx, tb, InvArgs{rows_of(fk + poly, 1, 0), rows_of(fv, 1, n), NO_ROWS, NO_ROWS, rows_of(ob, 1, 3 * n), nullptr, 0},
                           log_n, batch));
        ZK_TRY(she_inverse(ctx, tb, InvArgs{rows_of(fk, 1, 0), rows_of(fv, 1, n), NO_ROWS, NO_ROWS, rows_of(ob + poly, 1, 3 * n), nullptr, 0},
                           log_n, batch));
    }
    F7 pp = fp_ext_to_int<Fq753Params>(host_load_ext<Fq753Params>(p->l));
    hipLaunchKernelGGL(k_she_encrypt_tail, zk_grid(batch * n, 256), 256, 0, ctx->stream, out, e, r, to_f7k(pp), (uint32_t)n, (uint64_t)batch);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_she_decrypt_dev(zk_ctx* ctx, const void* ct, const void* sk, void* out, size_t n, size_t batch) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (batch && (!ct || !sk || !out))) return ZK_ERR_ARG;
    ZK_TRY(check_n(ctx, n, "zk_she_decrypt_dev"));
    if (batch == 0) return ZK_OK;
    const char* cb = (const char*)ct;
    size_t poly = n * 96;
    char* t;   // s*c1 + s*s*c2, packed rows
    ZK_TRY(zk_scratch(ctx, "she_dec_t", batch * poly, (void**)&t));
    if (!ntt_path(n)) {
        uint32_t nn = (uint32_t)n;
        char* ss;
        ZK_TRY(zk_scratch(ctx, "she_dec_ss", poly, (void**)&ss));
        ZK_TRY(she_schoolbook(ctx, rows_of(sk, 1, 0), rows_of(sk, 1, 0), NO_ROWS, NO_ROWS, rows_of(ss, 1, n), 0, nn, 1));
        ZK_TRY(she_schoolbook(ctx, rows_of(sk, 1, 0), rows_of(cb + poly, 1, 3 * n), rows_of(ss, 1, 0), rows_of(cb + 2 * poly, 1, 3 * n),
                              rows_of(t, 1, n), 0, nn, batch));
    } else {
        uint32_t log_n = ilog2(n);
        SheTables tb;
        ZK_TRY(she_tables(ctx, log_n, &tb));
        char *fc, *fs;   // fc: [batch][2][n] transforms of (c1, c2); fs: s, then s*s (transform domain)
        ZK_TRY(zk_scratch(ctx, "she_fa", batch * 2 * poly, (void**)&fc));
        ZK_TRY(zk_scratch(ctx, "she_fb", 2 * poly, (void**)&fs));
        ZK_TRY(she_forward(ctx, tb, rows_of(cb + poly, 2, 3 * n), fc, log_n, 2 * batch));
        ZK_TRY(she_forward(ctx, tb, rows_of(sk, 1, n), fs, log_n, 1, 1));       // reduced: k_she_vec_op multiplies it by itself
        hipLaunchKernelGGL(k_she_vec_op, zk_grid(n, 256), 256, 0, ctx->stream, (int)SHE_MUL, (const void*)fs, (const void*)fs, (void*)(fs + poly), n);
        ZK_TRY(she_inverse(ctx, tb,
                           InvArgs{rows_of(fs, 1, 0), rows_of(fc, 1, 2 * n), rows_of(fs + poly, 1, 0), rows_of(fc + poly, 1, 2 * n),
                                   rows_of(t, 1, n), nullptr, 0},
                           log_n, batch));
    }
    hipLaunchKernelGGL(k_she_decrypt_tail, zk_grid(batch * n, 256), 256, 0, ctx->stream, ct, (const void*)t, out, (uint32_t)n, (uint64_t)batch);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

// zeta^(+-j) for the primitive 2n-th root zeta = cyclotomic_moduli's root (src/she/polynomial.rs:107-119), ext form.
static int she_fr_twist(zk_ctx* ctx, uint32_t log_n, bool inverse, void** out) {
    uint32_t n = 1u << log_n;
    char name[40];
    snprintf(name, sizeof name, "she_twist_%u_%d", log_n, (int)inverse);
    bool fresh = ctx->slots.find(name) == ctx->slots.end();
    ZK_TRY(zk_scratch(ctx, name, (size_t)n * 32, out));
    if (!fresh) return ZK_OK;
    // root = TWO_ADIC_ROOT^(2^(47 - (log_n + 1))), computed on the host in the device's internal form
    Fr z = fp_const<FrParams>(FrParams::TWO_ADIC_ROOT);
    for (uint32_t i = 0; i < FR_TWO_ADICITY - (log_n + 1); i++) z = fp_sqr<FrParams>(z);
    if (inverse) z = fp_inv<FrParams>(z);
    zk_fr base, start;
    host_store_ext<FrParams>(base.l, fp_int_to_ext<FrParams>(z));
    host_store_ext<FrParams>(start.l, fp_int_to_ext<FrParams>(fp_one<FrParams>()));
    return zk_fr_powers_dev(ctx, &base, &start, n, *out);
}

extern "C" int zk_she_encode_dev(zk_ctx* ctx, const void* plain_fr, void* out, size_t n, size_t batch) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (batch && (!plain_fr || !out))) return ZK_ERR_ARG;
    ZK_TRY(check_n(ctx, n, "zk_she_encode_dev"));
    if (!is_pow2(n)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_she_encode_dev: the slot count must be a power of two (X^N + 1 cyclotomic)");
    if (batch == 0) return ZK_OK;
    uint32_t log_n = ilog2(n);
    char* w;
    ZK_TRY(zk_scratch(ctx, "she_enc_fr", batch * n * 32, (void**)&w));
    ZK_HIP(ctx, hipMemcpyAsync(w, plain_fr, batch * n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    // interpolation on {zeta^(2i+1)} = inverse FFT over <zeta^2>, then coefficient j times zeta^-j
    for (size_t b = 0; b < batch && log_n; b++) ZK_TRY(zk_ntt_launch(ctx, w + b * n * 32, log_n, 1, 0));
    void* tw;
    ZK_TRY(she_fr_twist(ctx, log_n, true, &tw));
    hipLaunchKernelGGL(k_fr_rowscale, zk_grid(batch * n, 256), 256, 0, ctx->stream, (void*)w, (const void*)tw, (uint32_t)n, (uint64_t)(batch * n));
    hipLaunchKernelGGL(k_she_fr_to_fq, zk_grid(batch * n, 256), 256, 0, ctx->stream, (const void*)w, out, batch * n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_she_decode_dev(zk_ctx* ctx, const void* enc, void* out_fr, size_t n, size_t batch) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (batch && (!enc || !out_fr))) return ZK_ERR_ARG;
    ZK_TRY(check_n(ctx, n, "zk_she_decode_dev"));
    if (!is_pow2(n)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_she_decode_dev: the slot count must be a power of two (X^N + 1 cyclotomic)");
    if (batch == 0) return ZK_OK;
    uint32_t log_n = ilog2(n);
    hipLaunchKernelGGL(k_she_fq_to_fr, zk_grid(batch * n, 256), 256, 0, ctx->stream, enc, out_fr, batch * n);
    void* tw;
    ZK_TRY(she_fr_twist(ctx, log_n, false, &tw));
    hipLaunchKernelGGL(k_fr_rowscale, zk_grid(batch * n, 256), 256, 0, ctx->stream, out_fr, (const void*)tw, (uint32_t)n, (uint64_t)(batch * n));
    ZK_HIP(ctx, hipGetLastError());
    for (size_t b = 0; b < batch && log_n; b++) ZK_TRY(zk_ntt_launch(ctx, (char*)out_fr + b * n * 32, log_n, 0, 0));
    return ZK_OK;
    ZK_API_END
}
