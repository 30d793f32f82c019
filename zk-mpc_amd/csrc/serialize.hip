// serialize.hip -- arkworks CanonicalSerialize of curve-point tables on the device.
//
// Replaces (reference, relative to /root/reference/arkworks/algebra):
//   ec/src/models/short_weierstrass_jacobian.rs:847-883   GroupAffine::{serialize, serialize_uncompressed}
//   ec/src/models/short_weierstrass_jacobian.rs:930-942   GroupAffine::deserialize_unchecked
//   ec/src/models/short_weierstrass_jacobian.rs:888-905,110-125   GroupAffine::deserialize / get_point_from_x (compressed form)
//   ff/src/fields/macros.rs:3-55, serialize/src/flags.rs:61-139   field bytes + SWFlags (bit 7: y > -y, bit 6: infinity)
//   ff/src/fields/models/quadratic_extension.rs:411-420,659-669   Fq2 ordering (c1 first) and layout (c0 | c1, flags on c1)
// A proving key at 2^20 holds ~5 M points: one thread per point, HBM-bound (96-192 B in, 48-192 B out).
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "internal.hpp"

using namespace zk;

namespace {

// canonical limbs > (p - 1) / 2, i.e. y > -y for y != 0
__device__ __forceinline__ bool fq_gt_half(const Fq& c) {
    int32_t borrow = 0;
#pragma unroll
    for (int k = 0; k < FqParams::L; k++) borrow = ((int32_t)FqParams::HALF[k] - (int32_t)c.l[k] + borrow) >> 29;
    return borrow < 0;
}

__device__ __forceinline__ void fq_words(const Fq& internal, uint32_t w[12], Fq* canon_out = nullptr) {
    Fq c = fp_int_to_canon<FqParams>(internal);
    fp_pack<FqParams>(w, c);
    if (canon_out) *canon_out = c;
}

template <class F> struct Ser;
template <> struct Ser<G1Field> {
    static constexpr int FW = 12;   // words per base-field element
    __device__ static void words(const Fq& a, uint32_t* w) { fq_words(a, w); }
    __device__ static bool gt_neg(const Fq& y) { return fq_gt_half(fp_int_to_canon<FqParams>(y)); }
    __device__ static void one(uint32_t* w) { for (int i = 0; i < 12; i++) w[i] = 0; w[0] = 1; }
    __device__ static Fq from_words(const uint32_t* w) { return fp_canon_to_int<FqParams>(fp_unpack<FqParams>(w)); }
};
template <> struct Ser<G2Field> {
    static constexpr int FW = 24;
    __device__ static void words(const Fq2& a, uint32_t* w) { fq_words(a.c0, w); fq_words(a.c1, w + 12); }
    __device__ static bool gt_neg(const Fq2& y) {
        if (!fp_is_zero<FqParams>(y.c1)) return fq_gt_half(fp_int_to_canon<FqParams>(y.c1));
        return fq_gt_half(fp_int_to_canon<FqParams>(y.c0));
    }
    __device__ static void one(uint32_t* w) { for (int i = 0; i < 24; i++) w[i] = 0; w[0] = 1; }
    __device__ static Fq2 from_words(const uint32_t* w) {
        return Fq2{fp_canon_to_int<FqParams>(fp_unpack<FqParams>(w)), fp_canon_to_int<FqParams>(fp_unpack<FqParams>(w + 12))};
    }
};

// curve coefficient b of y^2 = x^3 + b: 1 on G1 (curves/g1.rs:22-26), 1/u = (0, -1/5) on the twist (curves/g2.rs:28-35)
template <class F> __device__ typename F::T curve_b();
template <> __device__ Fq curve_b<G1Field>() { return fp_one<FqParams>(); }
template <> __device__ Fq2 curve_b<G2Field>() { return Fq2{fp_zero<FqParams>(), fp_const<FqParams>(FqParams::G2_B_C1)}; }

constexpr uint32_t FLAG_POSITIVE = 1u << 31, FLAG_INFINITY = 1u << 30;   // bits 7 / 6 of the last byte

template <class F, bool COMPRESSED>
__global__ void __launch_bounds__(256) k_serialize(const uint32_t* bases, size_t n, uint32_t* out) {
    constexpr int FW = Ser<F>::FW, OW = COMPRESSED ? FW : 2 * FW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Affine<F> p = aff_load16<F>(bases, i);
        uint32_t w[OW];
        bool inf = F::is_zero(p.x) && F::is_zero(p.y);
        if (inf) {
#pragma unroll
            for (int k = 0; k < OW; k++) w[k] = 0;
            if (!COMPRESSED) Ser<F>::one(w + FW);               // GroupAffine::zero() = (0, 1, infinity)
            w[OW - 1] |= FLAG_INFINITY;
        } else {
            Ser<F>::words(p.x, w);
            if (COMPRESSED) {
                if (Ser<F>::gt_neg(p.y)) w[OW - 1] |= FLAG_POSITIVE;
            } else {
                Ser<F>::words(p.y, w + FW);
            }
        }
#pragma unroll
        for (int k = 0; k < OW; k++) out[i * OW + k] = w[k];
    }
}

template <class F>
__global__ void __launch_bounds__(256) k_deserialize_uncompressed(const uint32_t* in, size_t n, uint32_t* bases, uint32_t* bad) {
    constexpr int FW = Ser<F>::FW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t w[2 * FW];
#pragma unroll
        for (int k = 0; k < 2 * FW; k++) w[k] = in[i * 2 * FW + k];
        uint32_t flags = w[2 * FW - 1] & (FLAG_POSITIVE | FLAG_INFINITY);
        w[2 * FW - 1] &= ~(FLAG_POSITIVE | FLAG_INFINITY);
        Affine<F> p;
        if (flags & FLAG_INFINITY) {
            p.x = F::zero();
            p.y = F::zero();
        } else {
            p.x = Ser<F>::from_words(w);
            p.y = Ser<F>::from_words(w + FW);
            // y^2 = x^3 + b, so that garbage is not silently taken for a point (deserialize_unchecked itself does not check)
            typename F::T lhs = F::sqr(p.y), rhs = F::add(F::mul(F::sqr(p.x), p.x), curve_b<F>());
            if (!F::eq(lhs, rhs)) atomicOr(bad, 1u);
        }
        if (flags & FLAG_POSITIVE) atomicOr(bad, 2u);          // the uncompressed form never carries the sign flag
        aff_store16<F>(bases, i, p);
    }
}

// ---- square roots, for the compressed form (GroupAffine::get_point_from_x, short_weierstrass_jacobian.rs:110-125) ----
// Which root comes out does not matter: the sign flag of the encoding selects between y and -y afterwards.
// Fq: Tonelli-Shanks with q - 1 = 2^46 t (ff/src/fields/macros.rs sqrt_impl, eprint 2012/685 alg. 5); false = non-residue.
__device__ bool fq_sqrt(const Fq& a, Fq* out) {
    if (fp_is_zero<FqParams>(a)) { *out = a; return true; }
    const Fq one = fp_one<FqParams>();
    uint32_t e[FqParams::L];
#pragma unroll
    for (int i = 0; i < FqParams::L; i++) e[i] = FqParams::TS_T_MINUS1_DIV2[i];
    Fq z = fp_const<FqParams>(FqParams::TS_ROOT);
    Fq w = fp_pow_limbs<FqParams>(a, e, FqParams::L);       // a^((t-1)/2)
    Fq x = fp_mul<FqParams>(w, a);                            // a^((t+1)/2)
    Fq b = fp_mul<FqParams>(x, w);                            // a^t
    uint32_t v = FQ_TWO_ADICITY;
    while (!fp_eq<FqParams>(b, one)) {
        uint32_t k = 0;
        Fq b2k = b;
        while (!fp_eq<FqParams>(b2k, one)) {
            b2k = fp_sqr<FqParams>(b2k);
            if (++k == v) return false;
        }
        Fq ww = z;
        for (uint32_t i = 0; i + k + 1 < v; i++) ww = fp_sqr<FqParams>(ww);
        z = fp_sqr<FqParams>(ww);
        b = fp_mul<FqParams>(b, z);
        x = fp_mul<FqParams>(x, ww);
        v = k;
    }
    *out = x;
    return fp_eq<FqParams>(fp_sqr<FqParams>(x), a);
}
// Fq2 = Fq[u]/(u^2 + 5): through the norm (ff/src/fields/models/quadratic_extension.rs sqrt).
__device__ bool fq2_sqrt(const Fq2& a, Fq2* out) {
    using B = FqField;
    Fq r;
    if (B::is_zero(a.c1)) {
        if (fq_sqrt(a.c0, &r)) { *out = Fq2{r, B::zero()}; return true; }
        if (!fq_sqrt(B::mul(a.c0, fp_const<FqParams>(FqParams::G2_B_C1)), &r)) return false;   // c0 / (-5) = s^2  =>  (s u)^2 = c0
        *out = Fq2{B::zero(), r};
        return true;
    }
    Fq norm = B::add(B::sqr(a.c0), Fq2Field::mul5(B::sqr(a.c1)));
    Fq alpha;
    if (!fq_sqrt(norm, &alpha)) return false;
    const Fq inv2 = fp_const<FqParams>(FqParams::INV2);
    Fq delta = B::mul(B::add(alpha, a.c0), inv2);
    Fq c0;
    if (!fq_sqrt(delta, &c0)) {
        delta = B::sub(delta, alpha);
        if (!fq_sqrt(delta, &c0)) return false;
    }
    Fq2 y{c0, B::mul(B::mul(a.c1, inv2), B::inv(c0))};
    *out = y;
    return Fq2Field::eq(Fq2Field::sqr(y), a);
}
__device__ __forceinline__ bool field_sqrt(const Fq& a, Fq* out) { return fq_sqrt(a, out); }
__device__ __forceinline__ bool field_sqrt(const Fq2& a, Fq2* out) { return fq2_sqrt(a, out); }

// Compressed points (GroupAffine::deserialize, :888-905 without the subgroup check): x with the flags in its last byte.
template <class F>
__global__ void __launch_bounds__(64) k_deserialize_compressed(const uint32_t* in, size_t n, uint32_t* bases, uint32_t* bad) {
    constexpr int FW = Ser<F>::FW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t w[FW];
#pragma unroll
        for (int k = 0; k < FW; k++) w[k] = in[i * FW + k];
        const uint32_t flags = w[FW - 1] & (FLAG_POSITIVE | FLAG_INFINITY);
        w[FW - 1] &= ~(FLAG_POSITIVE | FLAG_INFINITY);
        Affine<F> p;
        p.x = F::zero();
        p.y = F::zero();
        if (!(flags & FLAG_INFINITY)) {
            p.x = Ser<F>::from_words(w);
            typename F::T y, rhs = F::add(F::mul(F::sqr(p.x), p.x), curve_b<F>());
            if (!field_sqrt(rhs, &y)) {
                atomicOr(bad, 1u);                       // x is not the abscissa of a curve point
                p.x = F::zero();
            } else {
                const bool positive = (flags & FLAG_POSITIVE) != 0;
                p.y = (Ser<F>::gt_neg(y) != positive) ? F::neg(y) : y;
            }
        }
        aff_store16<F>(bases, i, p);
    }
}

template <class F>
int serialize_t(zk_ctx* ctx, const zk_bases* b, size_t offset, size_t n, int compressed, uint8_t* out) {
    if (offset + n > b->n) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_bases_serialize: range out of bounds");
    if (!n) return ZK_OK;
    const size_t per = (size_t)Ser<F>::FW * 4 * (compressed ? 1 : 2);
    void* stage;
    ZK_TRY(zk_scratch(ctx, "ser_stage", n * per, &stage));
    const uint32_t* src = b->dev + offset * 2 * F::WORDS;
    if (compressed) hipLaunchKernelGGL((k_serialize<F, true>), zk_grid(n, 256), 256, 0, ctx->stream, src, n, (uint32_t*)stage);
    else hipLaunchKernelGGL((k_serialize<F, false>), zk_grid(n, 256), 256, 0, ctx->stream, src, n, (uint32_t*)stage);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out, stage, n * per, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

template <class F>
int deserialize_t(zk_ctx* ctx, int group, const uint8_t* bytes, size_t n, int compressed, zk_bases** out) {
    zk_bases* b = new zk_bases();
    b->group = group;
    b->n = n;
    const size_t in_bytes = n * (size_t)Ser<F>::FW * (compressed ? 4 : 8), dev_bytes = n * 2 * F::WORDS * 4;
    if (n) {
        char* stage;
        ZK_TRY(zk_scratch(ctx, "ser_stage", in_bytes + 16, (void**)&stage));
        if (hipMalloc((void**)&b->dev, dev_bytes) != hipSuccess) { delete b; ZK_FAIL(ctx, ZK_ERR_NOMEM, "zk_bases_deserialize_uncompressed: hipMalloc failed"); }
        b->owned = true;
        uint32_t* bad = (uint32_t*)(stage + in_bytes);
        ZK_HIP(ctx, hipMemcpyAsync(stage, bytes, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(ctx, hipMemsetAsync(bad, 0, 4, ctx->stream));
        if (compressed)
            hipLaunchKernelGGL(k_deserialize_compressed<F>, zk_grid(n, 64, 1 << 16), 64, 0, ctx->stream, (const uint32_t*)stage, n, b->dev, bad);
        else
            hipLaunchKernelGGL(k_deserialize_uncompressed<F>, zk_grid(n, 256), 256, 0, ctx->stream, (const uint32_t*)stage, n, b->dev, bad);
        ZK_HIP(ctx, hipGetLastError());
        uint32_t h = 0;
        ZK_HIP(ctx, hipMemcpyAsync(&h, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h) {
            (void)hipFree(b->dev);
            delete b;
            ZK_FAIL(ctx, ZK_ERR_ARG, h & 1 ? "zk_bases_deserialize_uncompressed: a point is not on the curve (SerializationError::InvalidData)"
                                           : "zk_bases_deserialize_uncompressed: unexpected flag bits");
        }
    }
    *out = b;
    return ZK_OK;
}

}  // namespace

extern "C" size_t zk_point_serialized_size(int group, int compressed) {
    return (size_t)(group == 2 ? 96 : 48) * (compressed ? 1 : 2);
}

extern "C" int zk_bases_serialize(zk_ctx* ctx, const zk_bases* b, size_t offset, size_t n, int compressed, uint8_t* out_host) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !b || (n && !out_host)) return ZK_ERR_ARG;
    return b->group == 2 ? serialize_t<G2Field>(ctx, b, offset, n, compressed, out_host)
                         : serialize_t<G1Field>(ctx, b, offset, n, compressed, out_host);
    ZK_API_END
}

extern "C" int zk_bases_deserialize_uncompressed(zk_ctx* ctx, int group, const uint8_t* bytes_host, size_t n, zk_bases** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !out || (n && !bytes_host) || (group != 1 && group != 2)) return ZK_ERR_ARG;
    return group == 2 ? deserialize_t<G2Field>(ctx, group, bytes_host, n, 0, out) : deserialize_t<G1Field>(ctx, group, bytes_host, n, 0, out);
    ZK_API_END
}

extern "C" int zk_bases_deserialize_compressed(zk_ctx* ctx, int group, const uint8_t* bytes_host, size_t n, zk_bases** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !out || (n && !bytes_host) || (group != 1 && group != 2)) return ZK_ERR_ARG;
    return group == 2 ? deserialize_t<G2Field>(ctx, group, bytes_host, n, 1, out) : deserialize_t<G1Field>(ctx, group, bytes_host, n, 1, out);
    ZK_API_END
}
