// hostgroup.hpp -- host-side conversions between the C-ABI structs (u64 limbs, arkworks Montgomery
// form) and the internal field / curve types, plus arkworks' compressed point serialisation.
// The O(1)-per-proof group work of create_proof (src/groth16.rs:115-176) runs on the host with these.
#pragma once
#include "devutil.cuh"
#include <string.h>

namespace zk {

template <class F>
inline typename F::T host_felt_from_abi(const uint64_t* l) {  // ext words -> internal
    uint32_t w[F::WORDS];
    for (int i = 0; i < F::WORDS / 2; i++) { w[2 * i] = (uint32_t)l[i]; w[2 * i + 1] = (uint32_t)(l[i] >> 32); }
    return F::ext_to_int(F::load(w));
}

template <class F>
inline void host_felt_to_abi(uint64_t* l, const typename F::T& a) {  // internal -> ext words
    uint32_t w[F::WORDS];
    F::store(w, F::int_to_ext(a));
    for (int i = 0; i < F::WORDS / 2; i++) l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

template <class F>
inline Affine<F> host_aff_from_abi(const uint64_t* p) {
    constexpr int FE = F::WORDS / 2;
    bool zero = true;
    for (int i = 0; i < 2 * FE; i++) zero = zero && p[i] == 0;
    if (zero) return aff_inf<F>();
    return Affine<F>{host_felt_from_abi<F>(p), host_felt_from_abi<F>(p + FE)};
}

template <class F>
inline void host_aff_to_abi(uint64_t* p, const Affine<F>& a) {
    constexpr int FE = F::WORDS / 2;
    if (aff_is_inf<F>(a)) { memset(p, 0, 2 * FE * 8); return; }
    host_felt_to_abi<F>(p, a.x);
    host_felt_to_abi<F>(p + FE, a.y);
}

// Jacobian (X,Y,Z) -> XYZZ (X, Y, Z^2, Z^3)
template <class F>
inline XYZZ<F> host_proj_from_abi(const uint64_t* p) {
    constexpr int FE = F::WORDS / 2;
    typename F::T z = host_felt_from_abi<F>(p + 2 * FE);
    if (F::is_zero(z)) return xyzz_inf<F>();
    typename F::T zz = F::sqr(z);
    return XYZZ<F>{host_felt_from_abi<F>(p), host_felt_from_abi<F>(p + FE), zz, F::mul(zz, z)};
}

// affine -> Jacobian with Z = 1; zero = (1,1,0) like GroupProjective::zero()
template <class F>
inline void host_write_projective(const Affine<F>& a, uint64_t* out) {
    constexpr int FE = F::WORDS / 2;
    if (aff_is_inf<F>(a)) {
        host_felt_to_abi<F>(out, F::one());
        host_felt_to_abi<F>(out + FE, F::one());
        memset(out + 2 * FE, 0, FE * 8);
        return;
    }
    host_felt_to_abi<F>(out, a.x);
    host_felt_to_abi<F>(out + FE, a.y);
    host_felt_to_abi<F>(out + 2 * FE, F::one());
}

template <class F>
inline void host_write_projective(const XYZZ<F>& p, uint64_t* out) { host_write_projective<F>(xyzz_to_affine<F>(p), out); }

// canonical little-endian bytes of an internal-form Fq
inline void fq_canonical_bytes(const Fq& a, uint8_t out[48]) {
    uint32_t w[12];
    fp_pack<FqParams>(w, fp_int_to_canon<FqParams>(a));
    for (int i = 0; i < 12; i++) for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(w[i] >> (8 * b));
}

inline int cmp_le_bytes(const uint8_t* a, const uint8_t* b, int n) {
    for (int i = n - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}

// y > -y as canonical integers (short_weierstrass_jacobian.rs:856-857)
inline bool fq_gt_neg(const Fq& y) {
    uint8_t a[48], b[48];
    fq_canonical_bytes(y, a);
    fq_canonical_bytes(fp_neg<FqParams>(y), b);
    return cmp_le_bytes(a, b, 48) > 0;
}
// Fq2 ordering: c1 first, then c0 (quadratic_extension.rs:411-420)
inline bool fq2_gt_neg(const Fq2& y) {
    uint8_t a[48], b[48];
    fq_canonical_bytes(y.c1, a);
    fq_canonical_bytes(fp_neg<FqParams>(y.c1), b);
    int c = cmp_le_bytes(a, b, 48);
    if (c) return c > 0;
    fq_canonical_bytes(y.c0, a);
    fq_canonical_bytes(fp_neg<FqParams>(y.c0), b);
    return cmp_le_bytes(a, b, 48) > 0;
}

inline void g1_serialize(const Affine<G1Field>& p, uint8_t out[48]) {
    if (aff_is_inf<G1Field>(p)) { memset(out, 0, 48); out[47] |= 1 << 6; return; }
    fq_canonical_bytes(p.x, out);
    if (fq_gt_neg(p.y)) out[47] |= 1 << 7;
}
inline void g2_serialize(const Affine<G2Field>& p, uint8_t out[96]) {
    if (aff_is_inf<G2Field>(p)) { memset(out, 0, 96); out[95] |= 1 << 6; return; }
    fq_canonical_bytes(p.x.c0, out);
    fq_canonical_bytes(p.x.c1, out + 48);
    if (fq2_gt_neg(p.y)) out[95] |= 1 << 7;
}

// Fr scalar (ext form) -> canonical 8 x u32
inline void fr_abi_to_canon_words(const uint64_t l[4], uint32_t w[8]) {
    fp_pack<FrParams>(w, fp_ext_to_canon<FrParams>(host_load_ext<FrParams>(l)));
}

template <class F>
inline XYZZ<F> host_scalar_mul(const XYZZ<F>& p, const uint32_t k[8]) {
    XYZZ<F> r = xyzz_inf<F>();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            r = xyzz_dbl<F>(r);
            if ((k[i] >> b) & 1) r = xyzz_add<F>(r, p);
        }
    return r;
}

}  // namespace zk
