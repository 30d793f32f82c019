/* fp_tmpl.h -- TEST INFRASTRUCTURE (oracle).  Prime-field template, included once per field with
 *   FP      : type/function prefix      FP_N : number of 64-bit limbs
 *   FP_MOD, FP_R, FP_R2 : const uint64_t[FP_N]      FP_INV : -p^-1 mod 2^64
 * Restates arkworks/algebra/ff/src/fields/{arithmetic.rs:7-83, macros.rs:255-475,638-730}:
 * elements are little-endian u64 limbs in Montgomery form (R = 2^(64 N)), always < p. */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(FP, name)

typedef struct { uint64_t l[FP_N]; } FP;

static inline int FN(is_zero)(const FP* a) {
    uint64_t o = 0;
    for (int i = 0; i < FP_N; i++) o |= a->l[i];
    return o == 0;
}
static inline int FN(eq)(const FP* a, const FP* b) {
    uint64_t o = 0;
    for (int i = 0; i < FP_N; i++) o |= a->l[i] ^ b->l[i];
    return o == 0;
}
/* integer comparison of the raw limbs */
static inline int FN(cmp_raw)(const uint64_t* a, const uint64_t* b) {
    for (int i = FP_N - 1; i >= 0; i--) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
static inline void FN(sub_raw)(uint64_t* a, const uint64_t* b) { /* a -= b */
    unsigned __int128 br = 0;
    for (int i = 0; i < FP_N; i++) {
        unsigned __int128 t = (unsigned __int128)a[i] - b[i] - br;
        a[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
/* reduce(): macros.rs:262-266 */
static inline void FN(reduce)(FP* a) {
    if (FN(cmp_raw)(a->l, FP_MOD) >= 0) FN(sub_raw)(a->l, FP_MOD);
}
/* add_assign: macros.rs:698-705 */
static inline void FN(add)(FP* r, const FP* a, const FP* b) {
    unsigned __int128 c = 0;
    for (int i = 0; i < FP_N; i++) {
        c += (unsigned __int128)a->l[i] + b->l[i];
        r->l[i] = (uint64_t)c;
        c >>= 64;
    }
    FN(reduce)(r);
}
/* sub_assign: macros.rs:708-717 (add modulus first when b > a) */
static inline void FN(sub)(FP* r, const FP* a, const FP* b) {
    FP t = *a;
    if (FN(cmp_raw)(b->l, t.l) > 0) {
        unsigned __int128 c = 0;
        for (int i = 0; i < FP_N; i++) {
            c += (unsigned __int128)t.l[i] + FP_MOD[i];
            t.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    FN(sub_raw)(t.l, b->l);
    *r = t;
}
static inline void FN(neg)(FP* r, const FP* a) { /* macros.rs:638-651 */
    if (FN(is_zero)(a)) { *r = *a; return; }
    FP t;
    for (int i = 0; i < FP_N; i++) t.l[i] = FP_MOD[i];
    FN(sub_raw)(t.l, a->l);
    *r = t;
}
static inline void FN(dbl)(FP* r, const FP* a) { FN(add)(r, a, a); }

/* mul_assign, CIOS: arithmetic.rs:7-57 */
static inline void FN(mul)(FP* out, const FP* a, const FP* b) {
    uint64_t r[FP_N];
    for (int i = 0; i < FP_N; i++) r[i] = 0;
    for (int i = 0; i < FP_N; i++) {
        unsigned __int128 t = (unsigned __int128)a->l[0] * b->l[i] + r[0];
        uint64_t r0 = (uint64_t)t, carry1 = (uint64_t)(t >> 64);
        uint64_t k = r0 * FP_INV;
        t = (unsigned __int128)k * FP_MOD[0] + r0;
        uint64_t carry2 = (uint64_t)(t >> 64);
        for (int j = 1; j < FP_N; j++) {
            t = (unsigned __int128)a->l[j] * b->l[i] + r[j] + carry1;
            uint64_t rj = (uint64_t)t;
            carry1 = (uint64_t)(t >> 64);
            t = (unsigned __int128)k * FP_MOD[j] + rj + carry2;
            r[j - 1] = (uint64_t)t;
            carry2 = (uint64_t)(t >> 64);
        }
        r[FP_N - 1] = carry1 + carry2;
    }
    for (int i = 0; i < FP_N; i++) out->l[i] = r[i];
    FN(reduce)(out);
}
static inline void FN(sqr)(FP* r, const FP* a) { FN(mul)(r, a, a); } /* same value as arithmetic.rs:85-172 */

static inline void FN(one)(FP* r) { for (int i = 0; i < FP_N; i++) r->l[i] = FP_R[i]; }
static inline void FN(zero)(FP* r) { for (int i = 0; i < FP_N; i++) r->l[i] = 0; }
static inline int FN(is_one)(const FP* a) {
    uint64_t o = 0;
    for (int i = 0; i < FP_N; i++) o |= a->l[i] ^ FP_R[i];
    return o == 0;
}
/* into_repr: arithmetic.rs:59-83 (Montgomery reduction of the limbs) */
static inline void FN(into_repr)(uint64_t* out, const FP* a) {
    FP one_raw;
    FN(zero)(&one_raw);
    one_raw.l[0] = 1;
    FP t;
    FN(mul)(&t, a, &one_raw);
    for (int i = 0; i < FP_N; i++) out[i] = t.l[i];
}
/* from_repr: macros.rs:464-474 (multiply by R2) */
static inline void FN(from_repr)(FP* out, const uint64_t* in) {
    FP t, r2;
    for (int i = 0; i < FP_N; i++) { t.l[i] = in[i]; r2.l[i] = FP_R2[i]; }
    FN(mul)(out, &t, &r2);
}
/* a^e, e = little-endian u64 limbs (plain integer) */
static inline void FN(pow)(FP* out, const FP* a, const uint64_t* e, int nl) {
    FP r;
    FN(one)(&r);
    for (int i = nl - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            FN(sqr)(&r, &r);
            if ((e[i] >> b) & 1) FN(mul)(&r, &r, a);
        }
    *out = r;
}
/* inverse: value of macros.rs:389-443 (binary EEA) == a^(p-2) */
static inline void FN(inv)(FP* out, const FP* a) {
    uint64_t e[FP_N];
    for (int i = 0; i < FP_N; i++) e[i] = FP_MOD[i];
    e[0] -= 2; /* both moduli end in ...0001 */
    FN(pow)(out, a, e, FP_N);
}

#undef FN
#undef FP
#undef FP_N
#undef FP_MOD
#undef FP_R
#undef FP_R2
#undef FP_INV
