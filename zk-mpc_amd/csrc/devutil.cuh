// devutil.cuh -- device-side load/store helpers (16-byte vector accesses) and host conversions.
#pragma once
#include "ec.cuh"

namespace zk {

// ---- Fr in the reference's layout: 8 x u32 (= 4 x u64 LE), Montgomery R = 2^256 ("ext") ----
__device__ __forceinline__ Fr fr_load(const void* base, size_t i) {
    const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * i;
    uint4 a = p[0], b = p[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fp_unpack<FrParams>(w);
}

__device__ __forceinline__ void fr_store(void* base, size_t i, const Fr& v) {
    uint32_t w[8];
    fp_pack<FrParams>(w, v);
    uint4* p = reinterpret_cast<uint4*>(base) + 2 * i;
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

__device__ __forceinline__ Fr fr_mul(const Fr& a, const Fr& b) { return fp_mul<FrParams>(a, b); }
__device__ __forceinline__ Fr fr_add(const Fr& a, const Fr& b) { return fp_add<FrParams>(a, b); }
__device__ __forceinline__ Fr fr_sub(const Fr& a, const Fr& b) { return fp_sub<FrParams>(a, b); }

// ---- packed points (internal form) via 16-byte loads; F::WORDS is a multiple of 4 ----
template <class F>
__device__ __forceinline__ typename F::T felt_load16(const uint32_t* w) {
    uint32_t t[F::WORDS];
    const uint4* p = reinterpret_cast<const uint4*>(w);
#pragma unroll
    for (int i = 0; i < F::WORDS / 4; i++) {
        uint4 v = p[i];
        t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w;
    }
    return F::load(t);
}

template <class F>
__device__ __forceinline__ void felt_store16(uint32_t* w, const typename F::T& a) {
    uint32_t t[F::WORDS];
    F::store(t, a);
    uint4* p = reinterpret_cast<uint4*>(w);
#pragma unroll
    for (int i = 0; i < F::WORDS / 4; i++) p[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
}

template <class F>
__device__ __forceinline__ Affine<F> aff_load16(const uint32_t* base, size_t i) {
    const uint32_t* w = base + i * (2 * F::WORDS);
    return Affine<F>{felt_load16<F>(w), felt_load16<F>(w + F::WORDS)};
}
template <class F>
__device__ __forceinline__ void aff_store16(uint32_t* base, size_t i, const Affine<F>& p) {
    uint32_t* w = base + i * (2 * F::WORDS);
    felt_store16<F>(w, p.x);
    felt_store16<F>(w + F::WORDS, p.y);
}
template <class F>
__device__ __forceinline__ XYZZ<F> xyzz_load16(const uint32_t* base, size_t i) {
    const uint32_t* w = base + i * (4 * F::WORDS);
    return XYZZ<F>{felt_load16<F>(w), felt_load16<F>(w + F::WORDS), felt_load16<F>(w + 2 * F::WORDS),
                   felt_load16<F>(w + 3 * F::WORDS)};
}
template <class F>
__device__ __forceinline__ void xyzz_store16(uint32_t* base, size_t i, const XYZZ<F>& p) {
    uint32_t* w = base + i * (4 * F::WORDS);
    felt_store16<F>(w, p.x);
    felt_store16<F>(w + F::WORDS, p.y);
    felt_store16<F>(w + 2 * F::WORDS, p.zz);
    felt_store16<F>(w + 3 * F::WORDS, p.zzz);
}

// ---- host conversions between the C-ABI structs (u64 limbs, ext Montgomery) and Fp ----
template <class P>
inline Fp<P> host_load_ext(const uint64_t* l) {  // raw bits of the external form
    uint32_t w[P::W];
    for (int i = 0; i < P::W / 2; i++) { w[2 * i] = (uint32_t)l[i]; w[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    return fp_unpack<P>(w);
}
template <class P>
inline void host_store_ext(uint64_t* l, const Fp<P>& a) {
    uint32_t w[P::W];
    fp_pack<P>(w, a);
    for (int i = 0; i < P::W / 2; i++) l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

}  // namespace zk
