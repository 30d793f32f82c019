/* ec_tmpl.h -- TEST INFRASTRUCTURE (oracle).  Jacobian short-Weierstrass (a = 0) template, included
 * once per group with
 *   G  : prefix (g1 / g2)         FE : coordinate field type (fq / fq2)
 *   FE_(op) : field function macro, e.g. FE_(mul)(&r,&a,&b)
 * Restates arkworks/algebra/ec/src/models/short_weierstrass_jacobian.rs:
 *   double_in_place :557-623 (a = 0 branch), add_assign_mixed :628-693, add_assign :721-784,
 *   From<Projective> for Affine :823-845, and ec/src/msm/variable_base.rs:11-106. */

#define GCAT_(a, b) a##_##b
#define GCAT(a, b) GCAT_(a, b)
#define GN(name) GCAT(G, name)
#define AFF GN(affine)
#define PRJ GN(proj)

typedef struct { FE x, y; int infinity; } AFF;
typedef struct { FE x, y, z; } PRJ;

static inline void GN(proj_zero)(PRJ* p) { FE_(one)(&p->x); FE_(one)(&p->y); FE_(zero)(&p->z); }
static inline int GN(proj_is_zero)(const PRJ* p) { return FE_(is_zero)(&p->z); }

static void GN(double_in_place)(PRJ* s) {
    if (GN(proj_is_zero)(s)) return;
    FE a, b, c, d, e, f, t;
    FE_(sqr)(&a, &s->x);                 /* A = X1^2 */
    FE_(sqr)(&b, &s->y);                 /* B = Y1^2 */
    FE_(sqr)(&c, &b);                    /* C = B^2 */
    FE_(add)(&t, &s->x, &b);             /* D = 2*((X1+B)^2-A-C) */
    FE_(sqr)(&t, &t);
    FE_(sub)(&t, &t, &a);
    FE_(sub)(&t, &t, &c);
    FE_(dbl)(&d, &t);
    FE_(dbl)(&t, &a);                    /* E = 3*A */
    FE_(add)(&e, &a, &t);
    FE_(sqr)(&f, &e);                    /* F = E^2 */
    FE_(mul)(&s->z, &s->z, &s->y);       /* Z3 = 2*Y1*Z1 */
    FE_(dbl)(&s->z, &s->z);
    FE_(sub)(&s->x, &f, &d);             /* X3 = F-2*D */
    FE_(sub)(&s->x, &s->x, &d);
    FE_(sub)(&t, &d, &s->x);             /* Y3 = E*(D-X3)-8*C */
    FE_(mul)(&t, &t, &e);
    FE_(dbl)(&c, &c); FE_(dbl)(&c, &c); FE_(dbl)(&c, &c);
    FE_(sub)(&s->y, &t, &c);
}

static void GN(add_assign_mixed)(PRJ* s, const AFF* o) {
    if (o->infinity) return;
    if (GN(proj_is_zero)(s)) { s->x = o->x; s->y = o->y; FE_(one)(&s->z); return; }
    FE z1z1, u2, s2, h, hh, i, j, r, v, t;
    FE_(sqr)(&z1z1, &s->z);              /* Z1Z1 = Z1^2 */
    FE_(mul)(&u2, &o->x, &z1z1);         /* U2 = X2*Z1Z1 */
    FE_(mul)(&s2, &o->y, &s->z);         /* S2 = Y2*Z1*Z1Z1 */
    FE_(mul)(&s2, &s2, &z1z1);
    if (FE_(eq)(&s->x, &u2) && FE_(eq)(&s->y, &s2)) { GN(double_in_place)(s); return; }
    FE_(sub)(&h, &u2, &s->x);            /* H = U2-X1 */
    FE_(sqr)(&hh, &h);                   /* HH = H^2 */
    FE_(dbl)(&i, &hh); FE_(dbl)(&i, &i); /* I = 4*HH */
    FE_(mul)(&j, &h, &i);                /* J = H*I */
    FE_(sub)(&r, &s2, &s->y);            /* r = 2*(S2-Y1) */
    FE_(dbl)(&r, &r);
    FE_(mul)(&v, &s->x, &i);             /* V = X1*I */
    FE x3;
    FE_(sqr)(&x3, &r);                   /* X3 = r^2 - J - 2*V */
    FE_(sub)(&x3, &x3, &j);
    FE_(sub)(&x3, &x3, &v);
    FE_(sub)(&x3, &x3, &v);
    FE_(mul)(&j, &j, &s->y);             /* Y3 = r*(V-X3)-2*Y1*J */
    FE_(dbl)(&j, &j);
    FE_(sub)(&t, &v, &x3);
    FE_(mul)(&t, &t, &r);
    FE_(sub)(&s->y, &t, &j);
    s->x = x3;
    FE_(add)(&t, &s->z, &h);             /* Z3 = (Z1+H)^2-Z1Z1-HH */
    FE_(sqr)(&t, &t);
    FE_(sub)(&t, &t, &z1z1);
    FE_(sub)(&s->z, &t, &hh);
}

static void GN(add_assign)(PRJ* s, const PRJ* o) {
    if (GN(proj_is_zero)(s)) { *s = *o; return; }
    if (GN(proj_is_zero)(o)) return;
    FE z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;
    FE_(sqr)(&z1z1, &s->z);
    FE_(sqr)(&z2z2, &o->z);
    FE_(mul)(&u1, &s->x, &z2z2);
    FE_(mul)(&u2, &o->x, &z1z1);
    FE_(mul)(&s1, &s->y, &o->z); FE_(mul)(&s1, &s1, &z2z2);
    FE_(mul)(&s2, &o->y, &s->z); FE_(mul)(&s2, &s2, &z1z1);
    if (FE_(eq)(&u1, &u2) && FE_(eq)(&s1, &s2)) { GN(double_in_place)(s); return; }
    FE_(sub)(&h, &u2, &u1);              /* H = U2-U1 */
    FE_(dbl)(&i, &h); FE_(sqr)(&i, &i);  /* I = (2*H)^2 */
    FE_(mul)(&j, &h, &i);                /* J = H*I */
    FE_(sub)(&r, &s2, &s1); FE_(dbl)(&r, &r);
    FE_(mul)(&v, &u1, &i);               /* V = U1*I */
    FE x3;
    FE_(sqr)(&x3, &r);                   /* X3 = r^2 - J - 2*V */
    FE_(sub)(&x3, &x3, &j);
    FE_(dbl)(&t, &v);
    FE_(sub)(&x3, &x3, &t);
    FE_(sub)(&t, &v, &x3);               /* Y3 = r*(V - X3) - 2*S1*J */
    FE_(mul)(&t, &t, &r);
    FE_(mul)(&s1, &s1, &j); FE_(dbl)(&s1, &s1);
    FE y3;
    FE_(sub)(&y3, &t, &s1);
    FE_(add)(&t, &s->z, &o->z);          /* Z3 = ((Z1+Z2)^2 - Z1Z1 - Z2Z2)*H */
    FE_(sqr)(&t, &t);
    FE_(sub)(&t, &t, &z1z1);
    FE_(sub)(&t, &t, &z2z2);
    FE_(mul)(&s->z, &t, &h);
    s->x = x3; s->y = y3;
}

static void GN(neg)(PRJ* p) { if (!GN(proj_is_zero)(p)) FE_(neg)(&p->y, &p->y); }

static void GN(into_affine)(AFF* a, const PRJ* p) {
    if (GN(proj_is_zero)(p)) { FE_(zero)(&a->x); FE_(one)(&a->y); a->infinity = 1; return; }
    FE zi, zi2, zi3;
    FE_(inv)(&zi, &p->z);
    FE_(sqr)(&zi2, &zi);
    FE_(mul)(&zi3, &zi2, &zi);
    FE_(mul)(&a->x, &p->x, &zi2);
    FE_(mul)(&a->y, &p->y, &zi3);
    a->infinity = 0;
}

/* mul_bits, MSB first (ec/src/lib.rs:216-227); k = 4 canonical u64 limbs */
static void GN(mul_bigint)(PRJ* out, const PRJ* p, const uint64_t k[4]) {
    PRJ r;
    GN(proj_zero)(&r);
    int started = 0;
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            if (started) GN(double_in_place)(&r);
            if ((k[i] >> b) & 1) { started = 1; GN(add_assign)(&r, p); }
        }
    *out = r;
}

/* VariableBaseMSM::multi_scalar_mul, scalars = canonical BigInteger256 (variable_base.rs:11-106).
 * Windows are the unit of parallelism, as cfg_into_iter!(window_starts) is with rayon. */
static void GN(msm)(PRJ* out, const AFF* bases, const uint64_t* scalars, size_t size, int threads) {
    size_t c;
    if (size < 32) c = 3;
    else {
        size_t lg = 0;
        while (((size_t)1 << lg) < size) lg++;       /* ark_std::log2 = ceil */
        c = lg * 69 / 100 + 2;                        /* ln_without_floats + 2 */
    }
    const size_t num_bits = 253;
    size_t nwin = (num_bits + c - 1) / c;
    PRJ* window_sums = (PRJ*)malloc(nwin * sizeof(PRJ));
    static const uint64_t fr_one[4] = {1, 0, 0, 0};
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t wi = 0; wi < nwin; wi++) {
        size_t w_start = wi * c;
        PRJ res;
        GN(proj_zero)(&res);
        size_t nb = ((size_t)1 << c) - 1;
        PRJ* buckets = (PRJ*)malloc(nb * sizeof(PRJ));
        for (size_t b = 0; b < nb; b++) GN(proj_zero)(&buckets[b]);
        for (size_t i = 0; i < size; i++) {
            const uint64_t* s = scalars + 4 * i;
            if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;                      /* filter(!is_zero) */
            if (s[0] == fr_one[0] && s[1] == 0 && s[2] == 0 && s[3] == 0) {     /* scalar == fr_one */
                if (w_start == 0) GN(add_assign_mixed)(&res, &bases[i]);
                continue;
            }
            size_t limb = w_start / 64, off = w_start % 64;                      /* divn(w_start); % 2^c */
            uint64_t v = s[limb] >> off;
            if (off && limb + 1 < 4) v |= s[limb + 1] << (64 - off);
            v &= ((uint64_t)1 << c) - 1;
            if (v) GN(add_assign_mixed)(&buckets[v - 1], &bases[i]);
        }
        PRJ running;
        GN(proj_zero)(&running);
        for (size_t b = nb; b-- > 0;) {
            GN(add_assign)(&running, &buckets[b]);
            GN(add_assign)(&res, &running);
        }
        free(buckets);
        window_sums[wi] = res;
    }
    PRJ total;
    GN(proj_zero)(&total);
    for (size_t wi = nwin; wi-- > 1;) {
        GN(add_assign)(&total, &window_sums[wi]);
        for (size_t k = 0; k < c; k++) GN(double_in_place)(&total);
    }
    GN(add_assign)(&total, &window_sums[0]);
    *out = total;
    free(window_sums);
}

#undef GN
#undef AFF
#undef PRJ
#undef G
#undef FE
#undef FE_
