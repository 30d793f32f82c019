// hostfield64.hpp -- host-only Fq / Fq2 with 64-bit limbs (Montgomery R = 2^384, i.e. exactly the
// reference's in-memory form), used for the O(1)-per-proof tail of the prover: Horner over the
// window sums, the r/s scalar multiplications of create_proof (src/groth16.rs:115-174) and the
// affine conversion before serialisation.  Same static interface as FqField / Fq2Field in
// fp29.cuh, so the curve templates in ec.cuh are reused unchanged; about 3x faster on x86 than
// running the 29-bit device representation on the host.
#pragma once
#include <stdint.h>
#include "ec.cuh"

namespace zk {

struct Fq64 { uint64_t l[6]; };

namespace host64 {
static const uint64_t P[6] = {0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull,
                              0x1a22d9f300f5138full, 0xc63b05c06ca1493bull, 0x01ae3a4617c510eaull};
static const uint64_t ONE[6] = {202099033278250856ull, 5854854902718660529ull, 11492539364873682930ull,
                                8885205928937022213ull, 5545221690922665192ull, 39800542322357402ull};  // R mod p
static const uint64_t INV = 9586122913090633727ull;  // -p^-1 mod 2^64
typedef unsigned __int128 u128;

inline int cmp(const uint64_t* a, const uint64_t* b) {
    for (int i = 5; i >= 0; i--) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
inline void sub_raw(uint64_t* a, const uint64_t* b) {
    u128 br = 0;
    for (int i = 0; i < 6; i++) {
        u128 t = (u128)a[i] - b[i] - br;
        a[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
}  // namespace host64

struct Fq64Field {
    using T = Fq64;
    static T zero() { return T{{0, 0, 0, 0, 0, 0}}; }
    static T one() { T r; for (int i = 0; i < 6; i++) r.l[i] = host64::ONE[i]; return r; }
    static bool is_zero(const T& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
    static bool eq(const T& a, const T& b) {
        uint64_t o = 0;
        for (int i = 0; i < 6; i++) o |= a.l[i] ^ b.l[i];
        return o == 0;
    }
    static T add(const T& a, const T& b) {
        T r;
        host64::u128 c = 0;
        for (int i = 0; i < 6; i++) { c += (host64::u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
        if (host64::cmp(r.l, host64::P) >= 0) host64::sub_raw(r.l, host64::P);
        return r;
    }
    static T sub(const T& a, const T& b) {
        T r = a;
        if (host64::cmp(b.l, r.l) > 0) {
            host64::u128 c = 0;
            for (int i = 0; i < 6; i++) { c += (host64::u128)r.l[i] + host64::P[i]; r.l[i] = (uint64_t)c; c >>= 64; }
        }
        host64::sub_raw(r.l, b.l);
        return r;
    }
    static T dbl(const T& a) { return add(a, a); }
    static T neg(const T& a) { return is_zero(a) ? a : sub(zero(), a); }
    static T mul(const T& a, const T& b) {
        using host64::u128;
        uint64_t r[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
            u128 t = (u128)a.l[0] * b.l[i] + r[0];
            uint64_t r0 = (uint64_t)t, c1 = (uint64_t)(t >> 64);
            uint64_t k = r0 * host64::INV;
            t = (u128)k * host64::P[0] + r0;
            uint64_t c2 = (uint64_t)(t >> 64);
            for (int j = 1; j < 6; j++) {
                t = (u128)a.l[j] * b.l[i] + r[j] + c1;
                uint64_t rj = (uint64_t)t;
                c1 = (uint64_t)(t >> 64);
                t = (u128)k * host64::P[j] + rj + c2;
                r[j - 1] = (uint64_t)t;
                c2 = (uint64_t)(t >> 64);
            }
            r[5] = c1 + c2;
        }
        T o;
        for (int i = 0; i < 6; i++) o.l[i] = r[i];
        if (host64::cmp(o.l, host64::P) >= 0) host64::sub_raw(o.l, host64::P);
        return o;
    }
    static T sqr(const T& a) { return mul(a, a); }
    static T mulsub(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static T inv(const T& a) {  // a^(p-2)
        uint64_t e[6];
        for (int i = 0; i < 6; i++) e[i] = host64::P[i];
        e[0] -= 2;
        T r = one();
        for (int i = 5; i >= 0; i--)
            for (int b = 63; b >= 0; b--) {
                r = sqr(r);
                if ((e[i] >> b) & 1) r = mul(r, a);
            }
        return r;
    }
    static T select(bool c, const T& a, const T& b) { return c ? a : b; }
    static constexpr int WORDS = 12;
    static T load(const uint32_t* w) {  // ext-form words
        T r;
        for (int i = 0; i < 6; i++) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        return r;
    }
    static void store(uint32_t* w, const T& a) {
        for (int i = 0; i < 6; i++) { w[2 * i] = (uint32_t)a.l[i]; w[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
    }
    // conversions with the device representation (internal form, 29-bit limbs)
    static T from_dev(const Fq& a) {
        uint32_t w[12];
        fp_pack<FqParams>(w, fp_int_to_ext<FqParams>(a));
        return load(w);
    }
    static Fq to_dev(const T& a) {
        uint32_t w[12];
        store(w, a);
        return fp_ext_to_int<FqParams>(fp_unpack<FqParams>(w));
    }
};

struct Fq264 { Fq64 c0, c1; };

struct Fq264Field {
    using T = Fq264;
    using B = Fq64Field;
    static T zero() { return T{B::zero(), B::zero()}; }
    static T one() { return T{B::one(), B::zero()}; }
    static T add(const T& a, const T& b) { return T{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    static T sub(const T& a, const T& b) { return T{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    static T dbl(const T& a) { return T{B::dbl(a.c0), B::dbl(a.c1)}; }
    static T neg(const T& a) { return T{B::neg(a.c0), B::neg(a.c1)}; }
    static Fq64 mul5(const Fq64& a) { Fq64 t = B::dbl(B::dbl(a)); return B::add(t, a); }
    static T mul(const T& a, const T& b) {
        Fq64 v0 = B::mul(a.c0, b.c0), v1 = B::mul(a.c1, b.c1);
        Fq64 s = B::mul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
        return T{B::sub(v0, mul5(v1)), B::sub(B::sub(s, v0), v1)};
    }
    static T sqr(const T& a) { return mul(a, a); }
    static T mulsub(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static T inv(const T& a) {
        Fq64 n = B::add(B::sqr(a.c0), mul5(B::sqr(a.c1)));
        Fq64 ni = B::inv(n);
        return T{B::mul(a.c0, ni), B::neg(B::mul(a.c1, ni))};
    }
    static bool is_zero(const T& a) { return B::is_zero(a.c0) && B::is_zero(a.c1); }
    static bool eq(const T& a, const T& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
    static constexpr int WORDS = 24;
    static T load(const uint32_t* w) { return T{B::load(w), B::load(w + 12)}; }
    static void store(uint32_t* w, const T& a) { B::store(w, a.c0); B::store(w + 12, a.c1); }
    static T from_dev(const Fq2& a) { return T{B::from_dev(a.c0), B::from_dev(a.c1)}; }
    static Fq2 to_dev(const T& a) { return Fq2{B::to_dev(a.c0), B::to_dev(a.c1)}; }
};

// device-field <-> host-field mapping used by the generic tail code
template <class F> struct Host64Of;
template <> struct Host64Of<FqField> { using type = Fq64Field; };
template <> struct Host64Of<Fq2Field> { using type = Fq264Field; };

template <class F>
inline XYZZ<typename Host64Of<F>::type> xyzz_to_host64(const XYZZ<F>& p) {
    using H = typename Host64Of<F>::type;
    return XYZZ<H>{H::from_dev(p.x), H::from_dev(p.y), H::from_dev(p.zz), H::from_dev(p.zzz)};
}
template <class F>
inline Affine<typename Host64Of<F>::type> aff_to_host64(const Affine<F>& p) {
    using H = typename Host64Of<F>::type;
    return Affine<H>{H::from_dev(p.x), H::from_dev(p.y)};
}
template <class F>
inline Affine<F> aff_from_host64(const Affine<typename Host64Of<F>::type>& p) {
    using H = typename Host64Of<F>::type;
    return Affine<F>{H::to_dev(p.x), H::to_dev(p.y)};
}

// affine (host field, ext form) -> C-ABI Jacobian with Z = 1; zero = (1,1,0)
template <class H>
inline void host64_write_projective(const Affine<H>& a, uint64_t* out) {
    constexpr int FE = H::WORDS / 2;
    uint32_t w[H::WORDS];
    auto put = [&](uint64_t* dst, const typename H::T& v) {
        H::store(w, v);
        for (int i = 0; i < FE; i++) dst[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    };
    if (aff_is_inf<H>(a)) {
        put(out, H::one());
        put(out + FE, H::one());
        for (int i = 0; i < FE; i++) out[2 * FE + i] = 0;
        return;
    }
    put(out, a.x);
    put(out + FE, a.y);
    put(out + 2 * FE, H::one());
}

// C-ABI Jacobian (X,Y,Z) -> XYZZ over the host field
template <class H>
inline XYZZ<H> host64_proj_from_abi(const uint64_t* p) {
    constexpr int FE = H::WORDS / 2;
    uint32_t w[H::WORDS];
    auto get = [&](const uint64_t* src) {
        for (int i = 0; i < FE; i++) { w[2 * i] = (uint32_t)src[i]; w[2 * i + 1] = (uint32_t)(src[i] >> 32); }
        return H::load(w);
    };
    typename H::T z = get(p + 2 * FE);
    if (H::is_zero(z)) return xyzz_inf<H>();
    typename H::T zz = H::sqr(z);
    return XYZZ<H>{get(p), get(p + FE), zz, H::mul(zz, z)};
}

// ---- a point that arrives from a PEER (the collaborative provers' small opens) ---------------------------------------------
// The reference's MpcSerNet::broadcast deserialises every payload and unwraps (mpc-algebra/src/channel.rs:12-28); for a
// point that is GroupAffine::deserialize: coordinates canonical, on the curve, in the prime-order subgroup
// (short_weierstrass_jacobian.rs:888-905, :171-183).  The wire form here is the C-ABI Jacobian a party's own
// host64_write_projective produces: Z = 0 with X = Y = 1 (infinity) or Z = 1.  host64_peer_point accepts exactly that:
//   every coordinate word below q;  Z = 0 or Z = 1 (Montgomery one);  Z = 1: y^2 = x^3 + b;
//   `subgroup` (the malicious-security entry points): r P = O, one 253-bit scalar multiplication on the host.
// On failure `ok` is cleared and infinity returned; the caller turns that into ZK_ERR_STATE like a non-canonical scalar.
template <class H> struct Host64Curve;
template <> struct Host64Curve<Fq64Field> {
    static Fq64 b() { return Fq64Field::one(); }                                  // y^2 = x^3 + 1 (curves/g1.rs:22-25)
    static bool words_valid(const uint64_t* w) { return host64::cmp(w, host64::P) < 0; }
};
template <> struct Host64Curve<Fq264Field> {
    static Fq264 b() { return Fq264{Fq64Field::zero(), Fq64Field::from_dev(fp_const<FqParams>(FqParams::G2_B_C1))}; }   // b' = 1/u (curves/g2.rs:28-35)
    static bool words_valid(const uint64_t* w) { return host64::cmp(w, host64::P) < 0 && host64::cmp(w + 6, host64::P) < 0; }
};

template <class H>
inline XYZZ<H> host64_peer_point(const uint64_t* p, bool subgroup, bool& ok) {
    constexpr int FE = H::WORDS / 2;
    uint32_t w[H::WORDS];
    auto get = [&](const uint64_t* src) {
        for (int i = 0; i < FE; i++) { w[2 * i] = (uint32_t)src[i]; w[2 * i + 1] = (uint32_t)(src[i] >> 32); }
        return H::load(w);
    };
    if (!Host64Curve<H>::words_valid(p) || !Host64Curve<H>::words_valid(p + FE) || !Host64Curve<H>::words_valid(p + 2 * FE)) { ok = false; return xyzz_inf<H>(); }
    const typename H::T z = get(p + 2 * FE);
    if (H::is_zero(z)) return xyzz_inf<H>();
    if (!H::eq(z, H::one())) { ok = false; return xyzz_inf<H>(); }
    const typename H::T x = get(p), y = get(p + FE);
    if (!H::eq(H::sqr(y), H::add(H::mul(H::sqr(x), x), Host64Curve<H>::b()))) { ok = false; return xyzz_inf<H>(); }
    const XYZZ<H> pt{x, y, H::one(), H::one()};
    if (subgroup) {
        uint32_t HOST64_R_WORDS[8];                                               // r, from the generated constants (fr.rs:45-52)
        fp_pack<FrParams>(HOST64_R_WORDS, fp_const<FrParams>(FrParams::P));
        XYZZ<H> r = xyzz_inf<H>();
        bool started = false;
        for (int i = 7; i >= 0; i--)
            for (int b = 31; b >= 0; b--) {
                if (started) r = xyzz_dbl<H>(r);
                if ((HOST64_R_WORDS[i] >> b) & 1) { r = xyzz_add<H>(r, pt); started = true; }
            }
        if (!xyzz_is_inf<H>(r)) { ok = false; return xyzz_inf<H>(); }
    }
    return pt;
}

template <class H>
inline XYZZ<H> host64_scalar_mul(const XYZZ<H>& p, const uint32_t k[8]) {
    // fixed 4-bit windows from the top: 14 additions for the table, then per window four doublings and one addition (none for a
    // zero digit): ~250 doublings + ~75 additions against 250 + 127 for double-and-add (the O(1) tail of a small proof is a chain
    // of three of these: 0.8 ms of a 2 ms proof at 2^10 before round 5)
    XYZZ<H> tab[16];
    tab[0] = xyzz_inf<H>();
    tab[1] = p;
    for (int i = 2; i < 16; i++) tab[i] = (i & 1) ? xyzz_add<H>(tab[i - 1], p) : xyzz_dbl<H>(tab[i >> 1]);
    XYZZ<H> r = xyzz_inf<H>();
    bool started = false;
    for (int i = 7; i >= 0; i--)
        for (int b = 28; b >= 0; b -= 4) {
            const uint32_t d = (k[i] >> b) & 15u;
            if (started) { r = xyzz_dbl<H>(r); r = xyzz_dbl<H>(r); r = xyzz_dbl<H>(r); r = xyzz_dbl<H>(r); }
            if (d) { r = started ? xyzz_add<H>(r, tab[d]) : tab[d]; started = true; }
        }
    return r;
}

// k P in G1 through the curve's endomorphism (BLS12-377: y^2 = x^3 + 1 has phi(x, y) = (beta x, y) with beta a primitive cube root of
// unity in Fq, and phi(P) = lambda P on the prime-order subgroup for lambda = z^2 - 1, z the curve's seed: lambda^2 + lambda + 1 = r).
// k = k1 + k2 lambda by plain division (k2 = k / lambda, k1 = k mod lambda: both below 2^127 for k < r), then ONE joint chain over
// 2-bit windows of (k1, k2) with the table i P + j phi(P), i, j < 4: ~127 doublings + ~70 additions instead of ~250 + ~90.  The
// two scalar multiplications of a Groth16 proof's tail that need an MSM result (s g_a, r g1_b) are this chain: ~0.1 ms of a
// 0.65 ms proof at 2^10 before.  ONLY for points of the prime-order subgroup (phi(P) = lambda P fails on the cofactor torsion):
// the callers use it where the proving key came from generate_parameters (zk_pk::points_in_subgroup); the C ABI's zk_g1_mul, the
// subgroup test of a peer's point and keys that were deserialised (no subgroup check there) keep the plain chain.  beta / lambda
// are held to the oracle's scalar multiplication by tests/test_abi.py (zk_diag_g1_mul_glv).
namespace host64 {
static const uint64_t GLV_BETA[6] = {0xdacd106da5847973ull, 0xd8fe2454bac2a79aull, 0x1ada4fd6fd832edcull,
                                     0xfb9868449d150908ull, 0xd63eb8aeea32285eull, 0x0167d6a36f873fd0ull};   // Montgomery form, R = 2^384
static const uint64_t GLV_LAMBDA[2] = {0x0a11800000000000ull, 0x452217cc90000001ull};                         // z^2 - 1, 127 bits
}  // namespace host64

inline XYZZ<Fq64Field> host64_scalar_mul_glv(const XYZZ<Fq64Field>& p, const uint32_t k[8]) {
    using H = Fq64Field;
    using X = XYZZ<H>;
    typedef unsigned __int128 u128;
    const u128 lambda = ((u128)host64::GLV_LAMBDA[1] << 64) | host64::GLV_LAMBDA[0];
    // k2 = k / lambda, k1 = k mod lambda (bit-serial: the remainder stays below lambda < 2^127, so the shift cannot overflow)
    u128 rem = 0;
    uint32_t k2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 255; i >= 0; i--) {
        rem = (rem << 1) | ((k[i >> 5] >> (i & 31)) & 1u);
        if (rem >= lambda) { rem -= lambda; k2[i >> 5] |= 1u << (i & 31); }
    }
    const uint32_t k1[8] = {(uint32_t)rem, (uint32_t)(rem >> 32), (uint32_t)(rem >> 64), (uint32_t)(rem >> 96), 0, 0, 0, 0};
    int top = 255;                                   // the highest set bit of either half (k < r: <= 126; any 256-bit k still works)
    while (top > 0 && !(((k1[top >> 5] | k2[top >> 5]) >> (top & 31)) & 1u)) top--;
    H::T beta;
    for (int i = 0; i < 6; i++) beta.l[i] = host64::GLV_BETA[i];
    auto phi = [&](const X& a) { return X{H::mul(a.x, beta), a.y, a.zz, a.zzz}; };
    X tab[4][4];
    tab[0][0] = xyzz_inf<H>();
    tab[1][0] = p;
    tab[2][0] = xyzz_dbl<H>(p);
    tab[3][0] = xyzz_add<H>(tab[2][0], p);
    for (int j = 1; j < 4; j++) tab[0][j] = phi(tab[j][0]);
    for (int i = 1; i < 4; i++)
        for (int j = 1; j < 4; j++) tab[i][j] = xyzz_add<H>(tab[i][0], tab[0][j]);
    X r = xyzz_inf<H>();
    bool started = false;
    for (int w = top >> 1; w >= 0; w--) {
        const int b = 2 * w;
        const uint32_t d1 = (k1[b >> 5] >> (b & 31)) & 3u, d2 = (k2[b >> 5] >> (b & 31)) & 3u;
        if (started) { r = xyzz_dbl<H>(r); r = xyzz_dbl<H>(r); }
        if (d1 | d2) { r = started ? xyzz_add<H>(r, tab[d1][d2]) : tab[d1][d2]; started = true; }
    }
    return r;
}

}  // namespace zk
