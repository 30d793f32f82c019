"""
oracle/zkref.py -- TEST INFRASTRUCTURE ONLY (CPU oracle, Python big-int).

A restatement, in plain Python integers, of the algorithms on zk-mpc's Groth16 proving hot
path over BLS12-377.  It is the checker for the HIP product path: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The product
(zk-mpc_amd/) never imports anything from oracle/.

PARITY PINNING: "parity unpinned" in the sense of the task statement -- no output of the Rust code is
compared with.  The recipe that closes the gap on a machine with cargo is tools/ref_vectors/ (its output,
tests/golden/ref_kats.json, is consumed by tests/test_ref_vectors.py when present).
The Rust reference cannot be compiled or run in the build container (no
cargo/rustc) and it ships no known-answer files for this path (its tests are property and
accept/reject tests).  This oracle is therefore pinned by
  (1) the reference's own parameter constants, restated below with their file:line and
      checked for internal consistency in tests/test_oracle_constants.py
      (R = 2^256 mod r, INV*r = -1 mod 2^64, generator on curve and of order r, ...),
  (2) the differential identities the reference's tests use (Pippenger == naive,
      ifft(fft(v)) = v, coset round trips, FFT == textbook DFT),
  (3) the verifier equation of arkworks/groth16/src/verifier.rs:41-61 evaluated with an
      independent pairing, and the known-trapdoor prediction of the proof bytes.
No recorded outputs of the Rust code exist; where a claim depends on that it says so.

All paths cited are relative to /root/reference/.
"""
from __future__ import annotations

import hashlib
from typing import List, Sequence, Tuple

# --------------------------------------------------------------------------------------
# Parameters (golden vectors).  Limbs are little-endian u64, exactly as in the reference.
# --------------------------------------------------------------------------------------


def limbs_to_int(limbs: Sequence[int]) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


def int_to_limbs(v: int, n: int) -> List[int]:
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


# arkworks/curves/bls12_377/src/fields/fr.rs:45-88
FR_MODULUS_LIMBS = [725501752471715841, 6461107452199829505, 6968279316240510977, 1345280370688173398]
FR_R_LIMBS = [9015221291577245683, 8239323489949974514, 1646089257421115374, 958099254763297437]
FR_R2_LIMBS = [2726216793283724667, 14712177743343147295, 12091039717619697043, 81024008013859129]
FR_INV = 725501752471715839
FR_GENERATOR_MONT_LIMBS = [2984901390528151251, 10561528701063790279, 5476750214495080041, 898978044469942640]
# arkworks/curves/bls12_377/src/fields/fr.rs:30-41
FR_TWO_ADICITY = 47
FR_TWO_ADIC_ROOT_MONT_LIMBS = [12646347781564978760, 6783048705277173164, 268534165941069093, 1121515446318641358]
FR_MODULUS_BITS = 253

# arkworks/curves/bls12_377/src/fields/fq.rs:26-75
FQ_MODULUS_LIMBS = [0x8508C00000000001, 0x170B5D4430000000, 0x1EF3622FBA094800,
                    0x1A22D9F300F5138F, 0xC63B05C06CA1493B, 0x1AE3A4617C510EA]
FQ_R_LIMBS = [202099033278250856, 5854854902718660529, 11492539364873682930,
              8885205928937022213, 5545221690922665192, 39800542322357402]
FQ_R2_LIMBS = [0xB786686C9400CD22, 0x329FCAAB00431B1, 0x22A5F11162D6B46D,
               0xBFDF7D03827DC3AC, 0x837E92F041790BF9, 0x6DFCCB1E914B88]
FQ_INV = 9586122913090633727
FQ_MODULUS_BITS = 377

R_MOD = limbs_to_int(FR_MODULUS_LIMBS)   # scalar field modulus r
Q_MOD = limbs_to_int(FQ_MODULUS_LIMBS)   # base field modulus q
FR_MONT_R = 1 << 256
FQ_MONT_R = 1 << 384

# arkworks/curves/bls12_377/src/curves/g1.rs:43-51
G1_GEN_X = 81937999373150964239938255573465948239988671502647976594219695644855304257327692006745978603320413799295628339695
G1_GEN_Y = 241266749859715473739788878240585681733927191168601896383759122102112907357779751001206799952863815012735208165030
# arkworks/curves/bls12_377/src/curves/g2.rs:63-86
G2_GEN_X = (233578398248691099356572568220835526895379068987715365179118596935057653620464273615301663571204657964920925606294,
            140913150380207355837477652521042157274541796891053068589147167627541651775299824604154852141315666357241556069118)
G2_GEN_Y = (63160294768292073209381361943935198908131692476676907196754037919244929611450776219210369229519898517858833747423,
            149157405641012693445398062341192467754805999074082136895788947234480009303640899064710353187729182149407503257491)
# arkworks/curves/bls12_377/src/curves/g2.rs:28-35  COEFF_B = (0, 1551986...874906)
G2_COEFF_B = (0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906)
G1_COEFF_B = 1                            # curves/g1.rs:22-23
FQ2_NONRESIDUE = Q_MOD - 5                # fields/fq2.rs:13  (u^2 = -5)
BLS_X = 0x8508C00000000001                # curves/mod.rs:16
FR_GENERATOR = 22                         # fr.rs:75 (multiplicative generator, canonical)

# --------------------------------------------------------------------------------------
# Prime fields: canonical ints mod p.  Montgomery helpers mirror ff/src/fields/macros.rs.
# --------------------------------------------------------------------------------------


def fr_to_mont(x: int) -> int:      # from_repr: x * R2 * R^-1  (macros.rs:464-474)
    return (x * FR_MONT_R) % R_MOD


def fr_from_mont(x: int) -> int:    # into_repr  (arithmetic.rs:59-83)
    return (x * pow(FR_MONT_R, -1, R_MOD)) % R_MOD


def fq_to_mont(x: int) -> int:
    return (x * FQ_MONT_R) % Q_MOD


def fq_from_mont(x: int) -> int:
    return (x * pow(FQ_MONT_R, -1, Q_MOD)) % Q_MOD


def mont_mul_cios(a: int, b: int, modulus_limbs: Sequence[int], inv: int) -> int:
    """Word-level CIOS Montgomery product, restating ff/src/fields/arithmetic.rs:7-57
    ("no-carry" variant computes the same value).  Operands/result are Montgomery residues."""
    n = len(modulus_limbs)
    al, bl = int_to_limbs(a, n), int_to_limbs(b, n)
    mask = 0xFFFFFFFFFFFFFFFF
    r = [0] * n
    for i in range(n):
        carry1 = 0
        t = r[0] + al[0] * bl[i]
        r0, carry1 = t & mask, t >> 64
        k = (r0 * inv) & mask
        t = r0 + k * modulus_limbs[0]
        carry2 = t >> 64
        for j in range(1, n):
            t = r[j] + al[j] * bl[i] + carry1
            rj, carry1 = t & mask, t >> 64
            t = rj + k * modulus_limbs[j] + carry2
            r[j - 1], carry2 = t & mask, t >> 64
        r[n - 1] = carry1 + carry2
    v = limbs_to_int(r)
    p = limbs_to_int(modulus_limbs)
    if v >= p:                       # reduce(): macros.rs:262-266
        v -= p
    return v


# --------------------------------------------------------------------------------------
# Fq2 = Fq[u]/(u^2 + 5)   (ff/src/fields/models/quadratic_extension.rs, fields/fq2.rs)
# --------------------------------------------------------------------------------------

Fq2 = Tuple[int, int]


def fq2_add(a: Fq2, b: Fq2) -> Fq2:
    return ((a[0] + b[0]) % Q_MOD, (a[1] + b[1]) % Q_MOD)


def fq2_sub(a: Fq2, b: Fq2) -> Fq2:
    return ((a[0] - b[0]) % Q_MOD, (a[1] - b[1]) % Q_MOD)


def fq2_neg(a: Fq2) -> Fq2:
    return ((-a[0]) % Q_MOD, (-a[1]) % Q_MOD)


def fq2_mul(a: Fq2, b: Fq2) -> Fq2:
    # quadratic_extension.rs:632-643 (Karatsuba) == schoolbook value
    v0, v1 = a[0] * b[0], a[1] * b[1]
    c0 = (v0 + FQ2_NONRESIDUE * v1) % Q_MOD
    c1 = ((a[0] + a[1]) * (b[0] + b[1]) - v0 - v1) % Q_MOD
    return (c0, c1)


def fq2_sqr(a: Fq2) -> Fq2:
    return fq2_mul(a, a)


def fq2_inv(a: Fq2) -> Fq2:
    # quadratic_extension.rs:309-325: 1/(c0 + c1 u) = (c0 - c1 u)/(c0^2 - beta c1^2)
    n = (a[0] * a[0] - FQ2_NONRESIDUE * a[1] * a[1]) % Q_MOD
    ni = pow(n, -1, Q_MOD)
    return ((a[0] * ni) % Q_MOD, (-a[1] * ni) % Q_MOD)


def fq2_scalar(a: Fq2, k: int) -> Fq2:
    return ((a[0] * k) % Q_MOD, (a[1] * k) % Q_MOD)


FQ2_ZERO: Fq2 = (0, 0)
FQ2_ONE: Fq2 = (1, 0)

# --------------------------------------------------------------------------------------
# Curves.  Points are None (infinity) or (x, y) affine.  Arithmetic is the textbook
# chord-and-tangent law -- the *definition* the reference's Jacobian formulas
# (ec/src/models/short_weierstrass_jacobian.rs:557-784) must agree with after into_affine.
# --------------------------------------------------------------------------------------


class FqOps:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % Q_MOD)
    sub = staticmethod(lambda a, b: (a - b) % Q_MOD)
    mul = staticmethod(lambda a, b: (a * b) % Q_MOD)
    neg = staticmethod(lambda a: (-a) % Q_MOD)
    inv = staticmethod(lambda a: pow(a, -1, Q_MOD))
    small = staticmethod(lambda a, k: (a * k) % Q_MOD)
    b = G1_COEFF_B


class Fq2Ops:
    zero, one = FQ2_ZERO, FQ2_ONE
    add = staticmethod(fq2_add)
    sub = staticmethod(fq2_sub)
    mul = staticmethod(fq2_mul)
    neg = staticmethod(fq2_neg)
    inv = staticmethod(fq2_inv)
    small = staticmethod(fq2_scalar)
    b = G2_COEFF_B


def ec_is_on_curve(P, F) -> bool:
    if P is None:
        return True
    x, y = P
    return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), F.b)


def ec_neg(P, F):
    if P is None:
        return None
    return (P[0], F.neg(P[1]))


def ec_add(P, Q, F):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if y1 == y2 and y1 != F.zero:
            lam = F.mul(F.small(F.mul(x1, x1), 3), F.inv(F.small(y1, 2)))
        else:
            return None
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def ec_mul(P, k: int, F):
    """Double-and-add, MSB first (ec/src/lib.rs:216-227 mul_bits)."""
    k %= R_MOD
    R = None
    for bit in bin(k)[2:] if k else "":
        R = ec_add(R, R, F)
        if bit == "1":
            R = ec_add(R, P, F)
    return R


def ec_mul_raw(P, k: int, F):
    """Scalar multiply by an arbitrary non-negative integer (no reduction mod r)."""
    R = None
    for bit in bin(k)[2:] if k else "":
        R = ec_add(R, R, F)
        if bit == "1":
            R = ec_add(R, P, F)
    return R


G1_GEN = (G1_GEN_X, G1_GEN_Y)
G2_GEN = (G2_GEN_X, G2_GEN_Y)


def g1_add(P, Q): return ec_add(P, Q, FqOps)
def g1_mul(P, k): return ec_mul(P, k, FqOps)
def g1_neg(P): return ec_neg(P, FqOps)
def g2_add(P, Q): return ec_add(P, Q, Fq2Ops)
def g2_mul(P, k): return ec_mul(P, k, Fq2Ops)
def g2_neg(P): return ec_neg(P, Fq2Ops)


# --------------------------------------------------------------------------------------
# Serialization
# --------------------------------------------------------------------------------------


def fr_serialize(x: int) -> bytes:
    """32 B LE canonical (ff/src/fields/macros.rs:3-55, :564-569)."""
    return int(x % R_MOD).to_bytes(32, "little")


def fq_serialize(x: int, flags: int = 0) -> bytes:
    b = bytearray(int(x % Q_MOD).to_bytes(48, "little"))
    b[47] |= flags
    return bytes(b)


def _fq_gt_neg(y: int) -> bool:
    # short_weierstrass_jacobian.rs:856-857: flags = from_y_sign(y > -y), integer compare
    return y > (Q_MOD - y) % Q_MOD


def _fq2_gt_neg(y: Fq2) -> bool:
    # quadratic_extension.rs:411-420: lexicographic, c1 first then c0
    ny = fq2_neg(y)
    if y[1] != ny[1]:
        return y[1] > ny[1]
    return y[0] > ny[0]


SW_INFINITY = 1 << 6      # serialize/src/flags.rs:110-124
SW_POSITIVE_Y = 1 << 7


def g1_serialize(P) -> bytes:
    """Compressed 48 B (short_weierstrass_jacobian.rs:847-859)."""
    if P is None:
        return fq_serialize(0, SW_INFINITY)
    return fq_serialize(P[0], SW_POSITIVE_Y if _fq_gt_neg(P[1]) else 0)


def fq_sqrt(a: int):
    """A square root of a in Fq, or None (Tonelli-Shanks: q - 1 = 2^46 t; ff/src/fields/macros.rs:389-443 `sqrt` has the same
    structure).  Either root: the caller picks by the sign flag."""
    a %= Q_MOD
    if a == 0:
        return 0
    if pow(a, (Q_MOD - 1) // 2, Q_MOD) != 1:
        return None
    s, t = 0, Q_MOD - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (Q_MOD - 1) // 2, Q_MOD) == 1:
        z += 1
    c, x, b, m = pow(z, t, Q_MOD), pow(a, (t + 1) // 2, Q_MOD), pow(a, t, Q_MOD), s
    while b != 1:
        i, bb = 0, b
        while bb != 1:
            bb, i = bb * bb % Q_MOD, i + 1
        g = pow(c, 1 << (m - i - 1), Q_MOD)
        x, c = x * g % Q_MOD, g * g % Q_MOD
        b, m = b * c % Q_MOD, i
    return x


def g1_deserialize(b: bytes):
    """GroupAffine::deserialize of a compressed G1 point (short_weierstrass_jacobian.rs:888-905 -> get_point_from_x, :171-183):
    x in the low 377 bits, bit 7 of the last byte = "y is the greater root", bit 6 = infinity.  Raises on a non-canonical x, an x
    off the curve or a point outside the prime-order subgroup, as the reference's Result does."""
    assert len(b) == 48
    flags = b[47] & (SW_INFINITY | SW_POSITIVE_Y)
    if flags & SW_INFINITY:
        assert flags == SW_INFINITY and not any(b[:47]) and not (b[47] & 0x3F), "infinity with a non-zero x"
        return None
    x = int.from_bytes(bytes(b[:47]) + bytes([b[47] & 0x3F]), "little")
    assert x < Q_MOD, "non-canonical x"
    y = fq_sqrt((x * x % Q_MOD * x + 1) % Q_MOD)              # y^2 = x^3 + 1 (curves/g1.rs:22-25)
    assert y is not None, "x is not on the curve"
    if _fq_gt_neg(y) != bool(flags & SW_POSITIVE_Y):
        y = (Q_MOD - y) % Q_MOD
    P = (x, y)
    assert g1_mul(P, R_MOD) is None, "not in the prime-order subgroup"
    return P


def g2_serialize(P) -> bytes:
    """Compressed 96 B: c0 || c1-with-flags (quadratic_extension.rs:659-669)."""
    if P is None:
        return fq_serialize(0) + fq_serialize(0, SW_INFINITY)
    x = P[0]
    return fq_serialize(x[0]) + fq_serialize(x[1], SW_POSITIVE_Y if _fq2_gt_neg(P[1]) else 0)


def g1_serialize_uncompressed(P) -> bytes:
    """x || y-with-flags, 96 B (short_weierstrass_jacobian.rs:867-877); infinity is (0, 1) with the infinity bit."""
    if P is None:
        return fq_serialize(0) + fq_serialize(1, SW_INFINITY)
    return fq_serialize(P[0]) + fq_serialize(P[1])


def g2_serialize_uncompressed(P) -> bytes:
    if P is None:
        return fq_serialize(0) + fq_serialize(0) + fq_serialize(1) + fq_serialize(0, SW_INFINITY)
    (x0, x1), (y0, y1) = P
    return fq_serialize(x0) + fq_serialize(x1) + fq_serialize(y0) + fq_serialize(y1)


def _vec(points, ser) -> bytes:
    """Vec<T>: u64 LE length, then the items (serialize/src/lib.rs:263-272)."""
    return len(points).to_bytes(8, "little") + b"".join(ser(p) for p in points)


def vk_serialize(pk, compressed=True) -> bytes:
    """VerifyingKey (arkworks/groth16/src/data_structures.rs:43-58): alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1."""
    s1, s2 = (g1_serialize, g2_serialize) if compressed else (g1_serialize_uncompressed, g2_serialize_uncompressed)
    return s1(pk.alpha_g1) + s2(pk.beta_g2) + s2(pk.gamma_g2) + s2(pk.delta_g2) + _vec(pk.gamma_abc_g1, s1)


def pk_serialize(pk, compressed=True) -> bytes:
    """ProvingKey (data_structures.rs:133-151): vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query."""
    s1, s2 = (g1_serialize, g2_serialize) if compressed else (g1_serialize_uncompressed, g2_serialize_uncompressed)
    return (vk_serialize(pk, compressed) + s1(pk.beta_g1) + s1(pk.delta_g1) + _vec(pk.a_query, s1) + _vec(pk.b_g1_query, s1)
            + _vec(pk.b_g2_query, s2) + _vec(pk.h_query, s1) + _vec(pk.l_query, s1))


def proof_serialize(A, B, C) -> bytes:
    """a || b || c = 192 B (arkworks/groth16/src/data_structures.rs:10-18)."""
    return g1_serialize(A) + g2_serialize(B) + g1_serialize(C)


# --------------------------------------------------------------------------------------
# MSM (ec/src/msm/variable_base.rs:11-106) and naive reference
# --------------------------------------------------------------------------------------


def ln_without_floats(a: int) -> int:
    # msm/mod.rs:10-13 ; ark_std::log2(x) = ceil(log2(x)) (0 for x<=1)
    lg = 0 if a <= 1 else (a - 1).bit_length()
    return (lg * 69) // 100


def msm_window_bits(size: int) -> int:
    return 3 if size < 32 else ln_without_floats(size) + 2


def msm_naive(bases, scalars, F):
    acc = None
    for P, s in zip(bases, scalars):
        acc = ec_add(acc, ec_mul_raw(P, s % R_MOD, F), F)
    return acc


def msm_pippenger(bases, scalars, F):
    """Restatement of VariableBaseMSM::multi_scalar_mul.  `scalars` are canonical ints."""
    size = min(len(bases), len(scalars))
    scalars, bases = scalars[:size], bases[:size]
    pairs = [(s, b) for s, b in zip(scalars, bases) if s != 0]
    c = msm_window_bits(size)
    num_bits = FR_MODULUS_BITS
    window_starts = list(range(0, num_bits, c))
    window_sums = []
    for w_start in window_starts:
        res = None
        buckets = [None] * ((1 << c) - 1)
        for s, base in pairs:
            if s == 1:
                if w_start == 0:
                    res = ec_add(res, base, F)
            else:
                d = (s >> w_start) % (1 << c)
                if d != 0:
                    buckets[d - 1] = ec_add(buckets[d - 1], base, F)
        running = None
        for b in reversed(buckets):
            running = ec_add(running, b, F)
            res = ec_add(res, running, F)
        window_sums.append(res)
    lowest = window_sums[0]
    total = None
    for s in reversed(window_sums[1:]):
        total = ec_add(total, s, F)
        for _ in range(c):
            total = ec_add(total, total, F)
    return ec_add(lowest, total, F)


# --------------------------------------------------------------------------------------
# Radix-2 domain + FFT  (poly/src/domain/radix2/{mod.rs:51-82, fft.rs}, domain/mod.rs:92-190)
# --------------------------------------------------------------------------------------


class Domain:
    def __init__(self, num_coeffs: int):
        size = 1
        while size < num_coeffs:
            size <<= 1
        self.size = size
        self.log_size = size.bit_length() - 1
        assert self.log_size <= FR_TWO_ADICITY
        root = fr_from_mont(limbs_to_int(FR_TWO_ADIC_ROOT_MONT_LIMBS))
        # ff/src/fields/mod.rs get_root_of_unity: omega = root^(2^(TWO_ADICITY - log_size))
        self.group_gen = pow(root, 1 << (FR_TWO_ADICITY - self.log_size), R_MOD)
        self.group_gen_inv = pow(self.group_gen, -1, R_MOD)
        self.size_inv = pow(size, -1, R_MOD)
        self.generator = FR_GENERATOR
        self.generator_inv = pow(FR_GENERATOR, -1, R_MOD)

    def element(self, i: int) -> int:
        return pow(self.group_gen, i, R_MOD)

    def _transform(self, v: List[int], root: int) -> List[int]:
        n = self.size
        v = list(v) + [0] * (n - len(v))
        return _fft_rec(v, root)

    def fft(self, v):   return self._transform(v, self.group_gen)

    def ifft(self, v):
        out = self._transform(v, self.group_gen_inv)
        return [(x * self.size_inv) % R_MOD for x in out]

    def coset_fft(self, v):
        v = list(v) + [0] * (self.size - len(v))
        return self.fft(distribute_powers(v, self.generator))

    def coset_ifft(self, v):
        return distribute_powers(self.ifft(v), self.generator_inv)

    def evaluate_vanishing_polynomial(self, tau: int) -> int:
        return (pow(tau, self.size, R_MOD) - 1) % R_MOD

    def divide_by_vanishing_poly_on_coset(self, evals):
        i = pow(self.evaluate_vanishing_polynomial(self.generator), -1, R_MOD)
        return [(e * i) % R_MOD for e in evals]

    def evaluate_all_lagrange_coefficients(self, tau: int) -> List[int]:
        """poly/src/domain/radix2/mod.rs:116-165."""
        size = self.size
        t_size = pow(tau, size, R_MOD)
        if t_size == 1:
            u = [0] * size
            omega_i = 1
            for i in range(size):
                if omega_i == tau:
                    u[i] = 1
                    break
                omega_i = (omega_i * self.group_gen) % R_MOD
            return u
        l = ((t_size - 1) * self.size_inv) % R_MOD
        r = 1
        u = [0] * size
        ls = [0] * size
        for i in range(size):
            u[i] = (tau - r) % R_MOD
            ls[i] = l
            l = (l * self.group_gen) % R_MOD
            r = (r * self.group_gen) % R_MOD
        return [(li * pow(ui, -1, R_MOD)) % R_MOD for li, ui in zip(ls, u)]


def distribute_powers(v, g, c=1):
    out, p = [], c % R_MOD
    for x in v:
        out.append((x * p) % R_MOD)
        p = (p * g) % R_MOD
    return out


def _fft_rec(v: List[int], w: int) -> List[int]:
    n = len(v)
    if n == 1:
        return v
    even = _fft_rec(v[0::2], (w * w) % R_MOD)
    odd = _fft_rec(v[1::2], (w * w) % R_MOD)
    out = [0] * n
    t = 1
    h = n // 2
    for k in range(h):
        o = (t * odd[k]) % R_MOD
        out[k] = (even[k] + o) % R_MOD
        out[k + h] = (even[k] - o) % R_MOD
        t = (t * w) % R_MOD
    return out


def dft_naive(v: List[int], w: int) -> List[int]:
    n = len(v)
    return [sum(v[j] * pow(w, j * k, R_MOD) for j in range(n)) % R_MOD for k in range(n)]


def _bitrev(a: int, log_len: int) -> int:
    return int(format(a, "0{}b".format(log_len))[::-1], 2) if log_len else 0


def fft_arkworks_io_oi(v: List[int], dom: Domain, inverse: bool) -> List[int]:
    """Literal restatement of in_order_fft_in_place / in_order_ifft_in_place
    (radix2/fft.rs:22-70,185-280,300-307): DIF 'io' + derange, or derange + DIT 'oi'."""
    n = dom.size
    x = list(v) + [0] * (n - len(v))
    log_n = dom.log_size

    def derange(xs):
        for idx in range(1, n - 1):
            r = _bitrev(idx, log_n)
            if idx < r:
                xs[idx], xs[r] = xs[r], xs[idx]

    if not inverse:
        root = dom.group_gen
        roots = [pow(root, i, R_MOD) for i in range(n // 2)]
        gap = n // 2
        while gap > 0:
            chunk = 2 * gap
            nchunks = n // chunk
            for c0 in range(0, n, chunk):
                for j in range(gap):
                    lo, hi = x[c0 + j], x[c0 + gap + j]
                    x[c0 + j] = (lo + hi) % R_MOD
                    x[c0 + gap + j] = ((lo - hi) * roots[j * nchunks]) % R_MOD
            gap //= 2
        derange(x)
        return x
    root = dom.group_gen_inv
    roots = [pow(root, i, R_MOD) for i in range(n // 2)]
    derange(x)
    gap = 1
    while gap < n:
        chunk = 2 * gap
        nchunks = n // chunk
        for c0 in range(0, n, chunk):
            for j in range(gap):
                hi = (x[c0 + gap + j] * roots[j * nchunks]) % R_MOD
                lo = x[c0 + j]
                x[c0 + j] = (lo + hi) % R_MOD
                x[c0 + gap + j] = (lo - hi) % R_MOD
        gap *= 2
    return [(e * dom.size_inv) % R_MOD for e in x]


# --------------------------------------------------------------------------------------
# R1CS, QAP witness map, Groth16 (src/groth16.rs, arkworks/groth16/src/*)
# --------------------------------------------------------------------------------------


class R1CS:
    """ConstraintMatrices (snark/relations/src/r1cs/constraint_system.rs:650-676): rows of
    (coeff, index); index addresses instance (first, instance[0] = 1) then witness."""

    def __init__(self, num_instance: int, num_witness: int, a, b, c):
        self.num_instance = num_instance
        self.num_witness = num_witness
        self.a, self.b, self.c = a, b, c
        self.num_constraints = len(a)
        assert len(b) == len(a) == len(c)


def mul_chain_r1cs(n: int, w0: int, w1: int):
    """SURVEY.md 8(d) config-2 family: w_i * w_{i+1} = w_{i+2}; public input = last value.
    Variables: [1, pub] ++ witness[w_0 .. w_n]  (so num_instance=2, num_witness=n+1) and the
    last product w_{n+1} is the public input.  Returns (r1cs, full_assignment)."""
    w = [w0 % R_MOD, w1 % R_MOD]
    for i in range(n):
        w.append((w[i] * w[i + 1]) % R_MOD)
    pub = w[n + 1]
    wit = w[: n + 1]
    # index of w_j: j <= n -> 2 + j ; w_{n+1} -> 1 (public)
    def idx(j): return 2 + j if j <= n else 1
    a = [[(1, idx(i))] for i in range(n)]
    b = [[(1, idx(i + 1))] for i in range(n)]
    c = [[(1, idx(i + 2))] for i in range(n)]
    return R1CS(2, n + 1, a, b, c), [1, pub] + wit


def evaluate_constraint(terms, assignment):
    # src/groth16.rs:205-234
    s = 0
    for coeff, index in terms:
        s += assignment[index] * coeff
    return s % R_MOD


def witness_map(r1cs: R1CS, full_assignment: List[int], fft_impl=None):
    """R1CStoQAP::witness_map  (src/groth16.rs:240-306).  Returns h (length domain_size)."""
    num_inputs = r1cs.num_instance
    nc = r1cs.num_constraints
    dom = Domain(nc + num_inputs)
    D = dom.size
    a = [0] * D
    b = [0] * D
    for i in range(nc):
        a[i] = evaluate_constraint(r1cs.a[i], full_assignment)
        b[i] = evaluate_constraint(r1cs.b[i], full_assignment)
    for i in range(num_inputs):
        a[nc + i] = full_assignment[i]
    a = dom.coset_fft(dom.ifft(a))
    b = dom.coset_fft(dom.ifft(b))
    ab = [(x * y) % R_MOD for x, y in zip(a, b)]
    c = [0] * D
    for i in range(nc):
        c[i] = evaluate_constraint(r1cs.c[i], full_assignment)
    c = dom.coset_fft(dom.ifft(c))
    ab = [(x - y) % R_MOD for x, y in zip(ab, c)]
    ab = dom.divide_by_vanishing_poly_on_coset(ab)
    return dom.coset_ifft(ab)


class Trapdoor:
    def __init__(self, alpha, beta, gamma, delta, tau, g1_k=1, g2_k=1):
        self.alpha, self.beta, self.gamma, self.delta, self.tau = (
            alpha % R_MOD, beta % R_MOD, gamma % R_MOD, delta % R_MOD, tau % R_MOD)
        # generators g1 = g1_k * G1_GEN, g2 = g2_k * G2_GEN (generator.rs:44-53 samples random ones)
        self.g1_k, self.g2_k = g1_k % R_MOD, g2_k % R_MOD


def qap_instance_map(r1cs: R1CS, t: int):
    """R1CStoQAP::instance_map_with_evaluation (arkworks/groth16/src/r1cs_to_qap.rs:47-92).
    Returns (a, b, c, zt, qap_num_variables, m_raw) as canonical ints."""
    dom = Domain(r1cs.num_constraints + r1cs.num_instance)
    zt = dom.evaluate_vanishing_polynomial(t)
    u = dom.evaluate_all_lagrange_coefficients(t)
    nvars = (r1cs.num_instance - 1) + r1cs.num_witness
    a = [0] * (nvars + 1)
    b = [0] * (nvars + 1)
    c = [0] * (nvars + 1)
    nc = r1cs.num_constraints
    for i in range(r1cs.num_instance):
        a[i] = u[nc + i]
    for i in range(nc):
        for coeff, idx in r1cs.a[i]:
            a[idx] = (a[idx] + u[i] * coeff) % R_MOD
        for coeff, idx in r1cs.b[i]:
            b[idx] = (b[idx] + u[i] * coeff) % R_MOD
        for coeff, idx in r1cs.c[i]:
            c[idx] = (c[idx] + u[i] * coeff) % R_MOD
    return a, b, c, zt, nvars, dom.size


class ProvingKeyScalars:
    """Discrete logs (w.r.t. g1/g2) of every proving-key element: the known-trapdoor view of
    generate_parameters (arkworks/groth16/src/generator.rs:44-231)."""

    def __init__(self, r1cs: R1CS, td: Trapdoor):
        a, b, c, zt, nvars, m_raw = qap_instance_map(r1cs, td.tau)
        ni = r1cs.num_instance
        gi = pow(td.gamma, -1, R_MOD)
        di = pow(td.delta, -1, R_MOD)
        self.a_query = a
        self.b_query = b
        self.gamma_abc = [((td.beta * a[i] + td.alpha * b[i] + c[i]) * gi) % R_MOD for i in range(ni)]
        l = [((td.beta * a[i] + td.alpha * b[i] + c[i]) * di) % R_MOD for i in range(nvars + 1)]
        self.l_query = l[ni:]
        self.h_query = [(zt * di % R_MOD) * pow(td.tau, i, R_MOD) % R_MOD for i in range(m_raw - 1)]
        self.td = td
        self.num_instance = ni


class ProvingKey:
    def __init__(self, pks: ProvingKeyScalars):
        td = pks.td
        g1 = g1_mul(G1_GEN, td.g1_k)
        g2 = g2_mul(G2_GEN, td.g2_k)
        self.g1, self.g2 = g1, g2
        self.alpha_g1 = g1_mul(g1, td.alpha)
        self.beta_g1 = g1_mul(g1, td.beta)
        self.beta_g2 = g2_mul(g2, td.beta)
        self.delta_g1 = g1_mul(g1, td.delta)
        self.delta_g2 = g2_mul(g2, td.delta)
        self.gamma_g2 = g2_mul(g2, td.gamma)
        self.a_query = [g1_mul(g1, s) for s in pks.a_query]
        self.b_g1_query = [g1_mul(g1, s) for s in pks.b_query]
        self.b_g2_query = [g2_mul(g2, s) for s in pks.b_query]
        self.h_query = [g1_mul(g1, s) for s in pks.h_query]
        self.l_query = [g1_mul(g1, s) for s in pks.l_query]
        self.gamma_abc_g1 = [g1_mul(g1, s) for s in pks.gamma_abc]


def calculate_coeff(initial, query, vk_param, assignment, F, msm=msm_pippenger):
    # src/groth16.rs:185-201
    acc = msm(query[1:], assignment, F)
    res = ec_add(initial, query[0], F)
    res = ec_add(res, acc, F)
    return ec_add(res, vk_param, F)


def create_proof(r1cs: R1CS, pk: ProvingKey, full_assignment: List[int], r: int, s: int,
                 msm=msm_pippenger):
    """create_proof  (src/groth16.rs:68-183; stock arkworks/groth16/src/prover.rs:44-153)."""
    ni = r1cs.num_instance
    h = witness_map(r1cs, full_assignment)
    h_acc = msm(pk.h_query, h, FqOps)                       # min(len) rule: variable_base.rs:15
    witness = full_assignment[ni:]
    l_aux_acc = msm(pk.l_query, witness, FqOps)
    r_s_delta_g1 = g1_mul(g1_mul(pk.delta_g1, r), s)
    assignment = full_assignment[1:]
    r_g1 = g1_mul(pk.delta_g1, r)
    g_a = calculate_coeff(r_g1, pk.a_query, pk.alpha_g1, assignment, FqOps, msm)
    s_g_a = g1_mul(g_a, s)
    s_g1 = g1_mul(pk.delta_g1, s)
    g1_b = calculate_coeff(s_g1, pk.b_g1_query, pk.beta_g1, assignment, FqOps, msm)
    s_g2 = g2_mul(pk.delta_g2, s)
    g2_b = calculate_coeff(s_g2, pk.b_g2_query, pk.beta_g2, assignment, Fq2Ops, msm)
    r_g1_b = g1_mul(g1_b, r)
    g_c = g1_add(s_g_a, r_g1_b)
    g_c = g1_add(g_c, g1_neg(r_s_delta_g1))
    g_c = g1_add(g_c, l_aux_acc)
    g_c = g1_add(g_c, h_acc)
    return g_a, g2_b, g_c


def predict_proof_scalars(r1cs: R1CS, pks: ProvingKeyScalars, full_assignment, r, s, h=None):
    """Known-trapdoor prediction: discrete logs of A (wrt g1), B (wrt g2), C (wrt g1).
    Needs only Fr arithmetic: pins the exact proof bytes without trusting any MSM/NTT under test
    except the witness map `h`, which may be passed in or is computed by this oracle."""
    td = pks.td
    ni = r1cs.num_instance
    if h is None:
        h = witness_map(r1cs, full_assignment)
    z = full_assignment
    za = sum(zi * ai for zi, ai in zip(z, pks.a_query)) % R_MOD
    zb = sum(zi * bi for zi, bi in zip(z, pks.b_query)) % R_MOD
    A = (td.alpha + za + r * td.delta) % R_MOD
    B = (td.beta + zb + s * td.delta) % R_MOD
    hsum = sum(hi * qi for hi, qi in zip(h, pks.h_query)) % R_MOD
    lsum = sum(wi * li for wi, li in zip(z[ni:], pks.l_query)) % R_MOD
    C = (s * A + r * B - r * s % R_MOD * td.delta + lsum + hsum) % R_MOD
    return A, B, C


def predict_proof(r1cs, pks, full_assignment, r, s, h=None):
    A, B, C = predict_proof_scalars(r1cs, pks, full_assignment, r, s, h)
    td = pks.td
    g1 = g1_mul(G1_GEN, td.g1_k)
    g2 = g2_mul(G2_GEN, td.g2_k)
    return g1_mul(g1, A), g2_mul(g2, B), g1_mul(g1, C)


# --------------------------------------------------------------------------------------
# Pairing (verifier only: arkworks/groth16/src/verifier.rs:41-61).  Fq12 = Fq[w]/(w^12 + 5),
# u = w^6 (Fq2 = Fq[u]/(u^2+5), Fq6 = Fq2[v]/(v^3-u), Fq12 = Fq6[w]/(w^2-v)); D-type twist
# (curves/mod.rs:16-19): psi(x', y') = (x' w^2, y' w^3).
# --------------------------------------------------------------------------------------


def fq12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                if bj:
                    t[i + j] += ai * bj
    for k in range(22, 11, -1):
        t[k - 12] -= 5 * t[k]
    return [x % Q_MOD for x in t[:12]]


FQ12_ONE = [1] + [0] * 11


def fq12_pow(a, e: int):
    r = FQ12_ONE
    for bit in bin(e)[2:]:
        r = fq12_mul(r, r)
        if bit == "1":
            r = fq12_mul(r, a)
    return r


def _line(T, lam, P):
    """Line through twist point T (Fq2 affine) with twist-slope lam, evaluated at P in G1:
    yP - lam*xP*w + (lam*xT - yT)*w^3."""
    xP, yP = P
    c1 = fq2_scalar(fq2_neg(lam), xP)
    c3 = fq2_sub(fq2_mul(lam, T[0]), T[1])
    f = [0] * 12
    f[0] = yP % Q_MOD
    f[1], f[7] = c1
    f[3], f[9] = c3
    return f


def miller_loop(P, Q):
    """f_{x,Q}(P) for P in G1, Q in G2 (twist coords). Returns Fq12 element."""
    if P is None or Q is None:
        return FQ12_ONE
    T = Q
    f = FQ12_ONE
    for bit in bin(BLS_X)[3:]:
        lam = fq2_mul(fq2_scalar(fq2_sqr(T[0]), 3), fq2_inv(fq2_scalar(T[1], 2)))
        f = fq12_mul(fq12_mul(f, f), _line(T, lam, P))
        T = g2_add(T, T)
        if bit == "1":
            lam = fq2_mul(fq2_sub(Q[1], T[1]), fq2_inv(fq2_sub(Q[0], T[0])))
            f = fq12_mul(f, _line(T, lam, P))
            T = g2_add(T, Q)
    return f


FINAL_EXP = (Q_MOD ** 12 - 1) // R_MOD


def pairing_product_is_one(pairs) -> bool:
    f = FQ12_ONE
    for P, Q in pairs:
        f = fq12_mul(f, miller_loop(P, Q))
    return fq12_pow(f, FINAL_EXP) == FQ12_ONE


def verify_proof(pk: ProvingKey, proof, public_inputs: List[int]) -> bool:
    """e(A,B) = e(alpha,beta) e(sum_i x_i gamma_abc_i, gamma) e(C, delta)
    (arkworks/groth16/src/verifier.rs:18-61); public_inputs excludes the leading 1."""
    A, B, C = proof
    if len(public_inputs) + 1 != len(pk.gamma_abc_g1):
        return False
    acc = pk.gamma_abc_g1[0]
    for x, base in zip(public_inputs, pk.gamma_abc_g1[1:]):
        acc = g1_add(acc, g1_mul(base, x))
    return pairing_product_is_one([
        (A, B), (g1_neg(pk.alpha_g1), pk.beta_g2), (g1_neg(acc), pk.gamma_g2), (g1_neg(C), pk.delta_g2)])


# --------------------------------------------------------------------------------------
# Dense polynomials and KZG10 (poly/src/polynomial/univariate/dense.rs, poly-commit/src/kzg10/mod.rs)
# --------------------------------------------------------------------------------------


def batch_inversion(v):
    """ark_ff::batch_inversion (ff/src/fields/mod.rs:597-659): zeros stay zero."""
    return [0 if x % R_MOD == 0 else pow(x, -1, R_MOD) for x in v]


def poly_evaluate(c, z):
    acc = 0
    for coeff in reversed(c):       # horner_evaluate, dense.rs:68-72
        acc = (acc * z + coeff) % R_MOD
    return acc


def poly_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % R_MOD
    return out


def poly_divide_with_q_and_r(num, den):
    """DenseOrSparsePolynomial::divide_with_q_and_r (poly/src/polynomial/univariate/mod.rs:133-176): schoolbook long division."""
    num, den = list(num), list(den)
    while den and den[-1] == 0:
        den.pop()
    q = [0] * max(len(num) - len(den) + 1, 0)
    r = list(num)
    dl = pow(den[-1], -1, R_MOD)
    for k in range(len(num) - len(den), -1, -1):
        coef = (r[k + len(den) - 1] * dl) % R_MOD
        q[k] = coef
        for i, d in enumerate(den):
            r[k + i] = (r[k + i] - coef * d) % R_MOD
    return q, r[:len(den) - 1]


class KzgParams:
    """KZG10::setup with explicit toxic waste (kzg10/mod.rs:44-135): g = g_k*G1, gamma_g = gg_k*G1, h = h_k*G2."""

    def __init__(self, max_degree, beta, g_k=1, gg_k=7, h_k=1):
        self.beta = beta % R_MOD
        self.g, self.gamma_g, self.h = g1_mul(G1_GEN, g_k), g1_mul(G1_GEN, gg_k), g2_mul(G2_GEN, h_k)
        self.powers_of_beta = [pow(self.beta, i, R_MOD) for i in range(max_degree + 2)]
        self.powers_of_g = [g1_mul(self.g, b) for b in self.powers_of_beta[:max_degree + 1]]
        self.powers_of_gamma_g = [g1_mul(self.gamma_g, b) for b in self.powers_of_beta]   # one extra power (:85-87)
        self.beta_h = g2_mul(self.h, self.beta)


def kzg_commit(pp, coeffs, blind=None):
    c = msm_naive(pp.powers_of_g, coeffs, FqOps)
    if blind:
        c = g1_add(c, msm_naive(pp.powers_of_gamma_g, blind, FqOps))
    return c


def kzg_open(pp, coeffs, z, blind=None):
    q, _ = poly_divide_with_q_and_r(coeffs, [(-z) % R_MOD, 1])
    w = msm_naive(pp.powers_of_g, q, FqOps)
    rv = None
    if blind:
        qb, _ = poly_divide_with_q_and_r(blind, [(-z) % R_MOD, 1])
        w = g1_add(w, msm_naive(pp.powers_of_gamma_g, qb, FqOps))
        rv = poly_evaluate(blind, z)
    return w, rv


def kzg_check(pp, comm, z, value, w, random_v=None):
    """KZG10::check (kzg10/mod.rs:320-343): e(C - v g - rv gamma_g, h) = e(w, beta_h - z h)."""
    inner = g1_add(comm, g1_neg(g1_mul(pp.g, value)))
    if random_v is not None:
        inner = g1_add(inner, g1_neg(g1_mul(pp.gamma_g, random_v)))
    rhs_g2 = g2_add(pp.beta_h, g2_neg(g2_mul(pp.h, z)))
    return pairing_product_is_one([(inner, pp.h), (g1_neg(w), rhs_g2)])


# --------------------------------------------------------------------------------------
# SHE over MNT4-753 Fq (src/she.rs, src/she/{texts,encodedtext,ciphertext,plaintext,polynomial}.rs)
# --------------------------------------------------------------------------------------

# arkworks/curves/mnt4_753/src/fields/fq.rs:51
Q753 = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888253786114353726529584385201591605722013126468931404347949840543007986327743462853720628051692141265303114721689601
Q753_R = (1 << 768) % Q753          # fq.rs:73 (Montgomery R of Fp768)
Q753_GENERATOR = 17                 # fq.rs:105
Q753_TWO_ADICITY = 15               # fq.rs:14


def q753_to_mont(vals):
    return [(v * Q753_R) % Q753 for v in vals]


def q753_from_mont(vals):
    ri = pow(Q753_R, -1, Q753)
    return [(v * ri) % Q753 for v in vals]


def _poly_mul_mod(a, b, mod):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % mod
    return out


def encodedtext_mul(a, b):
    """Encodedtext * Encodedtext (src/she/encodedtext.rs:115-134): dense product, then remainder modulo X^N + 1
    by long division (poly_remainder2, src/she/polynomial.rs:152-168), padded to N."""
    n = len(a)
    assert len(b) == n
    prod = _poly_mul_mod(a, b, Q753)
    r = list(prod)
    for k in range(len(r) - 1, n - 1, -1):       # divide_with_q_and_r by the monic X^n + 1
        c = r[k]
        r[k] = 0
        r[k - n] = (r[k - n] - c) % Q753
    return (r + [0] * n)[:n]


def texts_add(a, b, mod=Q753):
    return [(x + y) % mod for x, y in zip(a, b)]


def texts_sub(a, b, mod=Q753):
    return [(x - y) % mod for x, y in zip(a, b)]


def texts_neg(a, mod=Q753):
    return [(-x) % mod for x in a]


def encodedtext_scale(a, k):
    return [(x * k) % Q753 for x in a]


def ciphertext_mul(x, y):
    """Ciphertext * Ciphertext (src/she/ciphertext.rs:113-122); x, y = (c0, c1, c2)."""
    c0 = encodedtext_mul(x[0], y[0])
    c1 = texts_add(encodedtext_mul(x[0], y[1]), encodedtext_mul(x[1], y[0]))
    c2 = encodedtext_mul(texts_neg(x[1]), y[1])
    return (c0, c1, c2)


def ciphertext_encrypt_from(e, pk_a, pk_b, r, p=R_MOD):
    """Ciphertext::encrypt_from (src/she/ciphertext.rs:46-72): r = u | v | w."""
    n = len(e)
    u, v, w = r[:n], r[n:2 * n], r[2 * n:3 * n]
    c0 = texts_add(texts_add(encodedtext_mul(pk_b, v), encodedtext_scale(w, p)), e)
    c1 = texts_add(encodedtext_mul(pk_a, v), encodedtext_scale(u, p))
    return (c0, c1, [0] * n)


def ciphertext_decrypt(ct, sk):
    """Ciphertext::decrypt (src/she/ciphertext.rs:74-79)."""
    sc1 = encodedtext_mul(sk, ct[1])
    sc2 = encodedtext_mul(encodedtext_mul(sk, sk), ct[2])
    return texts_sub(texts_sub(ct[0], sc1), sc2)


def public_key_gen(sk, a, e, p=R_MOD):
    """SecretKey::public_key_gen (src/she.rs:73-80) with the randomness passed in: b = a*s + e*p."""
    return a, texts_add(encodedtext_mul(a, sk), encodedtext_scale(e, p))


def cyclotomic_moduli(length):
    """src/she/polynomial.rs:107-119 over Fr: odd powers of a primitive 2*length-th root of unity."""
    k = (2 * length - 1).bit_length()       # ark_std::log2 = ceil(log2(2*length))
    assert k < FR_TWO_ADICITY
    root = pow(fr_from_mont(limbs_to_int(FR_TWO_ADIC_ROOT_MONT_LIMBS)), 1 << (FR_TWO_ADICITY - k), R_MOD)
    return [pow(root, 2 * i + 1, R_MOD) for i in range(length)]


def plaintexts_encode(vals):
    """Plaintexts::encode (src/she/plaintext.rs:45-59): Lagrange interpolation through (moduli[i], vals[i]) in Fr
    (interpolate, src/she/polynomial.rs:21-69), coefficients then embedded into Fq by their canonical integers."""
    xs = cyclotomic_moduli(len(vals))
    n = len(xs)
    res = [0] * n
    for j in range(n):
        sca, lp = 1, [1]
        for k in range(n):
            if k != j:
                sca = sca * (xs[j] - xs[k]) % R_MOD
                lp = _poly_mul_mod(lp, [(-xs[k]) % R_MOD, 1], R_MOD)
        coef = pow(sca, -1, R_MOD) * vals[j] % R_MOD
        for i, c in enumerate(lp):
            res[i] = (res[i] + coef * c) % R_MOD
    return res


def encodedtext_decode(vals, s=None):
    """Encodedtext::decode (src/she/encodedtext.rs:24-52): centred lift to Fr, evaluate at the cyclotomic roots."""
    n = len(vals)
    roots = cyclotomic_moduli(n)
    lifted = []
    for bu in vals:
        if bu > Q753 // 2:
            bu -= Q753 % R_MOD
        lifted.append(bu % R_MOD)
    return [sum(c * pow(x, i, R_MOD) for i, c in enumerate(lifted)) % R_MOD for x in roots[:(s or n)]]


# --------------------------------------------------------------------------------------
# Deterministic test-vector PRNG (SHA-256 counter mode; NOT the reference's ChaCha rng)
# --------------------------------------------------------------------------------------


class Prng:
    def __init__(self, seed: int):
        self.seed = int(seed).to_bytes(8, "little")
        self.ctr = 0

    def u64(self) -> int:
        h = hashlib.sha256(self.seed + self.ctr.to_bytes(8, "little")).digest()
        self.ctr += 1
        return int.from_bytes(h[:8], "little")

    def fr(self) -> int:
        h = b"".join(hashlib.sha256(self.seed + (self.ctr + i).to_bytes(8, "little")).digest() for i in range(2))
        self.ctr += 2
        return int.from_bytes(h[:40], "little") % R_MOD

    def fq(self) -> int:
        h = b"".join(hashlib.sha256(self.seed + (self.ctr + i).to_bytes(8, "little")).digest() for i in range(2))
        self.ctr += 2
        return int.from_bytes(h[:56], "little") % Q_MOD


# --------------------------------------------------------------------------------------
# Additive sharing helpers (mpc-algebra/src/share/additive.rs:34-182, share/field.rs:97-129)
# --------------------------------------------------------------------------------------


def additive_share(x: int, n_parties: int, rng: Prng) -> List[int]:
    sh = [rng.fr() for _ in range(n_parties - 1)]
    sh.append((x - sum(sh)) % R_MOD)
    return sh


def beaver_batch_mul_local(party: int, xs, ys, sx_open, oy_open, triple=None):
    """Local part of FieldShare::batch_mul after the two opens (share/field.rs:118-128):
    out = z - sx*y - oy*x + (party==0 ? sx*oy : 0), with (x,y,z) this party's triple shares.
    Default triple = DummyFieldTripleSource (wire/field.rs:49-63): leader holds 1, others 0."""
    n = len(xs)
    one = 1 if party == 0 else 0
    tx, ty, tz = triple if triple is not None else ([one] * n, [one] * n, [one] * n)
    out = []
    for i in range(n):
        v = (tz[i] - sx_open[i] * ty[i] - oy_open[i] * tx[i]) % R_MOD
        if party == 0:   # shift(): additive.rs:147-152 adds public constants on party 0 only
            v = (v + sx_open[i] * oy_open[i]) % R_MOD
        out.append(v)
    return out
