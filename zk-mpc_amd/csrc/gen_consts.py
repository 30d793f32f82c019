#!/usr/bin/env python3
"""Generate consts.cuh: radix-2^29 limb constants for the BLS12-377 fields used by the HIP kernels.

Primary inputs are the two moduli and the generator coordinates exactly as published in the
reference (arkworks/curves/bls12_377/src/fields/{fr,fq}.rs, curves/{g1,g2}.rs); everything
else is derived here with Python integers.

Representation (see fp29.cuh): an element is L limbs of 29 bits; the Montgomery radix of the
device arithmetic is RI = 2^(29*L) ("internal" form x*RI).  The reference keeps elements as
W 32-bit words in Montgomery form with RE = 2^(32*W) ("external" form x*RE).  RAW constants
below are plain integers k (no Montgomery factor), to be used as the second operand of
mmul(a, k) = a*k/RI.

Run: python3 gen_consts.py > consts.cuh
"""

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
G1X = 81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695
G1Y = 241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030
G2X0 = 233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294
G2X1 = 140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118
G2Y0 = 63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423
G2Y1 = 149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491
# MNT4-753 base field (the SHE ciphertext modulus, src/she.rs:17-18): arkworks/curves/mnt4_753/src/fields/fq.rs:51 (MODULUS),
# :105 (GENERATOR = 17), :14 (TWO_ADICITY = 15); TWO_ADIC_ROOT_OF_UNITY (:16) = 17^T is derived below.
Q753_MOD = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888253786114353726529584385201591605722013126468931404347949840543007986327743462853720628051692141265303114721689601
Q753_GENERATOR = 17
Q753_TWO_ADICITY = 15
FQ_TWO_ADICITY = 46          # fields/fq.rs:28
FQ_QNR = 15                  # a quadratic non-residue of Fq (checked below); any one gives a valid 2^46-th root of unity
assert pow(FQ_QNR, (Q_MOD - 1) // 2, Q_MOD) == Q_MOD - 1 and (Q_MOD - 1) % (1 << FQ_TWO_ADICITY) == 0 and ((Q_MOD - 1) >> FQ_TWO_ADICITY) % 2 == 1
FR_GENERATOR = 22
FR_TWO_ADICITY = 47
B = 29


def limbs(v, n):
    assert v < (1 << (B * n))
    return [(v >> (B * i)) & ((1 << B) - 1) for i in range(n)]


def arr(name, v, n, comment=""):
    return "    static constexpr uint32_t %s[%d] = {%s};%s" % (
        name, n, ", ".join("0x%08xu" % l for l in limbs(v, n)), ("  // " + comment) if comment else "")


def limbs_top(v, n):
    """n limbs, the last one holding everything above bit 29 (n - 1) (it may exceed 29 bits)."""
    assert v < (1 << (B * (n - 1) + 32))
    return [(v >> (B * i)) & ((1 << B) - 1) for i in range(n - 1)] + [v >> (B * (n - 1))]


def field(struct, p, L, W, internal=(), raw=(), LR=None):
    LR = LR or L
    RI = 1 << (B * LR)
    RE = 1 << (32 * W)
    assert RI > p
    kp = ["        {%s}," % ", ".join("0x%08xu" % l for l in limbs_top(k * p, L)) for k in range(9)]
    out = ["struct %s {" % struct,
           "    static constexpr int L = %d;      // 29-bit limbs" % L,
           "    static constexpr int LR = %d;     // Montgomery digits: the device's radix is RI = 2^(29 LR); LR = L + 1 leaves the" % LR,
           "                                      // product below p + 2^(29 (2L - LR) + 6) without a final subtraction (lazy domain)",
           "    static constexpr uint32_t KP[9][%d] = {   // k * p, k = 0..8; the top limb holds the excess over 29 bits" % L,
           ] + kp + ["    };",
           "    static constexpr int W = %d;      // 32-bit words of the external (arkworks) layout" % W,
           "    static constexpr int BITS = %d;" % p.bit_length(),
           arr("P", p, L),
           "    static constexpr uint32_t INV = 0x%08xu;  // -p^-1 mod 2^29" % ((-pow(p, -1, 1 << B)) % (1 << B)),
           arr("ONE", RI % p, L, "1 in internal form"),
           arr("RI2", (RI * RI) % p, L, "RAW: canonical -> internal"),
           arr("RAW_ONE", 1, L, "RAW: internal -> canonical"),
           arr("EXT_TO_CANON", (RI * pow(RE, -1, p)) % p, L, "RAW"),
           arr("CANON_TO_EXT", (RE * RI) % p, L, "RAW"),
           arr("EXT_TO_INT", (RI * RI * pow(RE, -1, p)) % p, L, "RAW (also fixes ext*ext products)"),
           arr("INT_TO_EXT", RE % p, L, "RAW"),
           arr("P_MINUS_2", p - 2, L, "Fermat inversion exponent (plain bits)"),
           ]
    for name, v in internal:
        out.append(arr(name, (v * RI) % p, L, "internal form"))
    for name, v in raw:
        out.append(arr(name, v % p, L, "RAW"))
    out.append("};")
    return "\n".join(out)


def sub_offset(p, L, K, j, top_bits=29):
    """K p written with limbs 0..L-2 in [j 2^29, (j + 1) 2^29) and whatever is left in the top limb: a + offset - b needs no
    borrow for any b whose limbs 0..L-2 are below j 2^29 (and whose top limb is below the offset's).  top_bits: the width the
    caller's arithmetic allows the top limb (29 for the Fr butterflies; 31 only where the caller has shown its sums keep inside
    32 bits: OFF3 of the 753-bit field)."""
    t = K * p - sum((j << B) << (B * i) for i in range(L - 1))
    assert t >= 0
    lo = [((t >> (B * i)) & ((1 << B) - 1)) + (j << B) for i in range(L - 1)]
    c = lo + [t >> (B * (L - 1))]
    assert sum(v << (B * i) for i, v in enumerate(c)) == K * p and c[-1] < (1 << top_bits), (K, hex(c[-1]))
    return c


def fr_lazy():
    """Constants of the lazy Fr domain of the NTT butterflies (frlazy.cuh): 2^261 / r = 438.8 leaves 8.8 bits of head room."""
    L = 9
    out = ["struct FrLazy {"]
    out.append(arr("RC", (1 << (B * L)) - R_MOD, L, "2^261 - r: a + q RC = (a - q r) + q 2^261"))
    out.append("    static constexpr uint32_t MQ = %du;      // floor(2^264 / r): q = (top limb * MQ) >> 32 <= floor(a / r)" % ((1 << 264) // R_MOD))
    for name, K, j in (("OFF2", 2, 1), ("OFF3", 3, 1), ("OFF5", 5, 2)):
        c = sub_offset(R_MOD, L, K, j)
        out.append("    static constexpr uint32_t %s[%d] = {%s};  // %d r, limbs 0..7 in [%d * 2^29, %d * 2^29)" % (
            name, L, ", ".join("0x%08xu" % v for v in c), K, j, j + 1))
    out.append("};")
    return "\n".join(out)


def fq753_lazy():
    """Constants of the lazy domain of the SHE butterflies (she.hip): 26 limbs hold 754 bits, q = 0.4427 * 2^754, so values below
    2.2 q keep every limb below 2^29 and a Montgomery product (RI = 2^754) of x < 2.2 q by y < q lands below 0.4427 x + q < 2 q."""
    L, q = 26, Q753_MOD
    out = ["struct Fq753Lazy {"]
    out.append(arr("RC", (1 << (B * L)) - q, L, "2^754 - q: a + k RC = (a - k q) + k 2^754"))
    d = (q >> (B * (L - 1))) + 1
    out.append("    static constexpr uint32_t MQ = %du;      // floor(2^56 / ((q >> 725) + 1)): k = (top limb * MQ) >> 56 <= floor(a / q)" % ((1 << 56) // d))
    for name, K in (("OFF2", 2), ("OFF3", 3)):
        c = sub_offset(q, L, K, 1, top_bits=31)
        out.append("    static constexpr uint32_t %s[%d] = {%s};  // %d q, limbs 0..24 in [2^29, 2^30)" % (
            name, L, ", ".join("0x%08xu" % v for v in c), K))
    out.append("};")
    return "\n".join(out)


def main():
    two_adic_root = pow(FR_GENERATOR, (R_MOD - 1) >> FR_TWO_ADICITY, R_MOD)
    print("// GENERATED by gen_consts.py -- do not edit.  29-bit little-endian limbs.")
    print("#pragma once")
    print("#include <stdint.h>")
    print("namespace zk {")
    print(field("FrParams", R_MOD, 9, 8, internal=[
        ("GENERATOR", FR_GENERATOR),                       # fr.rs:75-86 (22)
        ("GENERATOR_INV", pow(FR_GENERATOR, -1, R_MOD)),
        ("TWO_ADIC_ROOT", two_adic_root),                  # fr.rs:34-41
    ], raw=[
        # mmul(hi, WIDE_HI) = hi * 2^256 * RI: the upper half of a 512-bit sample folded into Fr (rng.hip)
        ("WIDE_HI", (1 << 256) * (1 << (B * 9)) * (1 << (B * 9))),
    ]))
    print("static constexpr int FR_TWO_ADICITY = %d;" % FR_TWO_ADICITY)
    print(fr_lazy())
    # Fq: ONE extra Montgomery digit (RI = 2^406, not 2^377).  q = 0.84 * 2^377 leaves no room between q and 2^(29 * 13):
    # with 13 digits every product needs a conditional subtraction and every sum a full reduction (43 % of the instructions
    # of a bucket addition).  With 14 digits a product of operands up to 7 q lands in [0, q + 2^354) by itself.
    print(field("FqParams", Q_MOD, 13, 12, LR=14, internal=[
        ("G1_GEN_X", G1X), ("G1_GEN_Y", G1Y),             # curves/g1.rs:43-51
        ("G2_GEN_X0", G2X0), ("G2_GEN_X1", G2X1),         # curves/g2.rs:63-86
        ("G2_GEN_Y0", G2Y0), ("G2_GEN_Y1", G2Y1),
        ("G2_B_C1", (-pow(5, -1, Q_MOD)) % Q_MOD),        # curves/g2.rs:28-35 (b' = 1/u)
        ("TS_ROOT", pow(FQ_QNR, (Q_MOD - 1) >> FQ_TWO_ADICITY, Q_MOD)),   # a primitive 2^46-th root of unity (QNR^T)
        ("INV2", (Q_MOD + 1) // 2),
    ], raw=[
        ("HALF", (Q_MOD - 1) // 2),                       # y > -y  <=>  canonical(y) > (q - 1) / 2
        ("TS_T_MINUS1_DIV2", (((Q_MOD - 1) >> FQ_TWO_ADICITY) - 1) // 2),   # Tonelli-Shanks exponent (plain bits)
    ]))
    print("static constexpr int FQ_TWO_ADICITY = %d;" % FQ_TWO_ADICITY)
    root753 = pow(Q753_GENERATOR, (Q753_MOD - 1) >> Q753_TWO_ADICITY, Q753_MOD)
    assert pow(root753, 1 << (Q753_TWO_ADICITY - 1), Q753_MOD) == Q753_MOD - 1
    print(field("Fq753Params", Q753_MOD, 26, 24, internal=[
        ("TWO_ADIC_ROOT", root753),
        ("TWO_ADIC_ROOT_INV", pow(root753, -1, Q753_MOD)),
        ("INV2", (Q753_MOD + 1) // 2),
    ], raw=[
        ("HALF", (Q753_MOD - 1) // 2),                     # Encodedtext::decode threshold q/2 (src/she/encodedtext.rs:34-38)
        ("MOD_FR", Q753_MOD % R_MOD),                      # q mod p, subtracted from the upper half (ibid.)
    ]))
    print("static constexpr int FQ753_TWO_ADICITY = %d;" % Q753_TWO_ADICITY)
    print(fq753_lazy())
    # Fr constants that fold a 26-limb canonical integer into Fr: K[i] = 2^(29 i) * RE * RI mod r (RAW operands),
    # so that sum_i mmul(limb_i, K[i]) is the value mod r in external form.
    RI, RE = 1 << (B * 9), 1 << 256
    print("static constexpr uint32_t FR_FOLD29[26][9] = {")
    for i in range(26):
        print("    {%s}," % ", ".join("0x%08xu" % l for l in limbs((pow(2, B * i, R_MOD) * RE * RI) % R_MOD, 9)))
    print("};")
    print("}  // namespace zk")


if __name__ == "__main__":
    main()
