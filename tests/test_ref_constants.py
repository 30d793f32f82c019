"""The parameter constants of the path, PINNED to the reference: tests/golden/ref_constants.json is parsed mechanically out of
the reference's Rust sources by tools/pin_reference_constants.py (every value carries its file:line).  Checked against it:
the oracle (oracle/zkref.py, oracle/zkref_consts.h), the inputs of the device's constant generator (csrc/gen_consts.py) and
the constants typed into host code (hostfield64.hpp, rng.hip) on the CPU; the library's own constants through the C ABI on
the GPU.  This pins constants, not behaviour: recorded outputs of the Rust prover still need tools/ref_vectors/ (no cargo here)."""
import importlib.util
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import zkref as O
import zk_mpc_amd as Z
import zk_mpc_amd.convert as cv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constants.json")))


def val(group, name):
    return int(REF[group][name]["value"])


def limbs(group, name):
    return [int(x, 16) for x in REF[group][name]["limbs"]]


FR, FQ, Q7 = "bls12_377_fr", "bls12_377_fq", "mnt4_753_fq"


def test_fixture_is_what_the_reference_tree_says():
    if not os.path.isdir("/root/reference/arkworks"):
        pytest.skip("reference tree absent (GPU box): the committed fixture stands")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_reference_constants.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_fixture_is_self_consistent():
    """The relations the reference's own parameter tests check (ff test-templates; curves/tests.rs:36-62)."""
    for g, w in ((FR, 4), (FQ, 6), (Q7, 12)):
        p, R = val(g, "MODULUS"), 1 << (64 * w)
        assert val(g, "R") == R % p and val(g, "R2") == R * R % p
        assert (val(g, "INV") * p + 1) % (1 << 64) == 0
        assert p.bit_length() == val(g, "MODULUS_BITS") and 64 * w - val(g, "MODULUS_BITS") == val(g, "REPR_SHAVE_BITS")
        s, t = val(g, "TWO_ADICITY"), val(g, "T")
        assert p - 1 == t << s and t & 1 and val(g, "T_MINUS_ONE_DIV_TWO") == (t - 1) // 2 and val(g, "MODULUS_MINUS_ONE_DIV_TWO") == (p - 1) // 2
        gen = val(g, "GENERATOR") * pow(R, -1, p) % p
        root = val(g, "TWO_ADIC_ROOT_OF_UNITY") * pow(R, -1, p) % p
        assert root == pow(gen, t, p) and pow(root, 1 << (s - 1), p) == p - 1
    assert val(FR, "GENERATOR") * pow(1 << 256, -1, val(FR, "MODULUS")) % val(FR, "MODULUS") == 22
    x = val("bls12_377", "X")
    assert val(FR, "MODULUS") == x ** 4 - x ** 2 + 1 and val(FQ, "MODULUS") == (x - 1) ** 2 * val(FR, "MODULUS") // 3 + x
    assert val("bls12_377_g1", "COFACTOR") == (x - 1) ** 2 // 3
    assert val("bls12_377_g1", "COFACTOR") * val("bls12_377_g1", "COFACTOR_INV") % val(FR, "MODULUS") == 1
    assert val("bls12_377_g2", "COFACTOR") * val("bls12_377_g2", "COFACTOR_INV") % val(FR, "MODULUS") == 1


def test_python_oracle_constants_equal_the_reference():
    assert O.FR_MODULUS_LIMBS == limbs(FR, "MODULUS") and O.FR_R_LIMBS == limbs(FR, "R") and O.FR_R2_LIMBS == limbs(FR, "R2")
    assert O.FR_INV == val(FR, "INV") and O.FR_GENERATOR_MONT_LIMBS == limbs(FR, "GENERATOR")
    assert O.FR_TWO_ADICITY == val(FR, "TWO_ADICITY") and O.FR_TWO_ADIC_ROOT_MONT_LIMBS == limbs(FR, "TWO_ADIC_ROOT_OF_UNITY")
    assert O.FR_MODULUS_BITS == val(FR, "MODULUS_BITS") and O.FQ_MODULUS_BITS == val(FQ, "MODULUS_BITS")
    assert O.FQ_MODULUS_LIMBS == limbs(FQ, "MODULUS") and O.FQ_R_LIMBS == limbs(FQ, "R") and O.FQ_R2_LIMBS == limbs(FQ, "R2")
    assert O.FQ_INV == val(FQ, "INV") and O.R_MOD == val(FR, "MODULUS") and O.Q_MOD == val(FQ, "MODULUS")
    assert O.FR_GENERATOR == val(FR, "GENERATOR") * pow(1 << 256, -1, O.R_MOD) % O.R_MOD
    g1, g2 = REF["bls12_377_g1"], REF["bls12_377_g2"]
    assert (O.G1_GEN_X, O.G1_GEN_Y) == (int(g1["G1_GENERATOR_X"]["value"]), int(g1["G1_GENERATOR_Y"]["value"]))
    assert tuple(O.G2_GEN_X) == (int(g2["G2_GENERATOR_X_C0"]["value"]), int(g2["G2_GENERATOR_X_C1"]["value"]))
    assert tuple(O.G2_GEN_Y) == (int(g2["G2_GENERATOR_Y_C0"]["value"]), int(g2["G2_GENERATOR_Y_C1"]["value"]))
    assert g1["COEFF_A"]["value"] == "FQ_ZERO" and g1["COEFF_B"]["value"] == "FQ_ONE" and O.G1_COEFF_B == 1
    assert g2["COEFF_B"]["value"][0] == "FQ_ZERO" and tuple(O.G2_COEFF_B) == (0, int(g2["COEFF_B"]["value"][1]))
    assert O.FQ2_NONRESIDUE == int(REF["bls12_377_fq2"]["NONRESIDUE"]["value"]) % O.Q_MOD
    assert O.BLS_X == val("bls12_377", "X") and REF["bls12_377"]["X_IS_NEGATIVE"]["value"] == "false"
    assert O.SW_INFINITY == 1 << val("serialize_sw_flags", "INFINITY_BIT") and O.SW_POSITIVE_Y == 1 << val("serialize_sw_flags", "POSITIVE_Y_BIT")
    assert O.Q753 == val(Q7, "MODULUS") and O.Q753_R == val(Q7, "R") and O.Q753_TWO_ADICITY == val(Q7, "TWO_ADICITY")
    assert O.Q753_GENERATOR == val(Q7, "GENERATOR") * pow(1 << 768, -1, O.Q753) % O.Q753


def test_c_oracle_generator_table_equals_the_reference():
    text = open(os.path.join(ROOT, "oracle", "zkref_consts.h")).read()
    R = 1 << 384
    q = val(FQ, "MODULUS")

    def arr(name):
        m = re.search(r"%s\[\d+\] = \{(.*?)\};" % name, text, re.S)
        w = [int(t.strip().rstrip("ul"), 16) for t in m.group(1).split(",")]
        return [sum(w[6 * k + i] << (64 * i) for i in range(6)) for k in range(len(w) // 6)]
    g1, g2 = REF["bls12_377_g1"], REF["bls12_377_g2"]
    assert arr("REF_G1_GEN") == [int(g1[k]["value"]) * R % q for k in ("G1_GENERATOR_X", "G1_GENERATOR_Y")]
    assert arr("REF_G2_GEN") == [int(g2[k]["value"]) * R % q for k in ("G2_GENERATOR_X_C0", "G2_GENERATOR_X_C1", "G2_GENERATOR_Y_C0", "G2_GENERATOR_Y_C1")]


def test_device_constant_generator_inputs_equal_the_reference():
    """csrc/gen_consts.py derives every device constant from a handful of primary inputs: those inputs are the reference's, and
    consts.cuh is exactly the generator's output (nothing hand-edited)."""
    path = os.path.join(ROOT, "zk-mpc_amd", "csrc", "gen_consts.py")
    spec = importlib.util.spec_from_file_location("gen_consts", path)
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    g1, g2 = REF["bls12_377_g1"], REF["bls12_377_g2"]
    assert G.R_MOD == val(FR, "MODULUS") and G.Q_MOD == val(FQ, "MODULUS") and G.Q753_MOD == val(Q7, "MODULUS")
    assert (G.G1X, G.G1Y) == (int(g1["G1_GENERATOR_X"]["value"]), int(g1["G1_GENERATOR_Y"]["value"]))
    assert (G.G2X0, G.G2X1, G.G2Y0, G.G2Y1) == tuple(int(g2[k]["value"]) for k in ("G2_GENERATOR_X_C0", "G2_GENERATOR_X_C1", "G2_GENERATOR_Y_C0", "G2_GENERATOR_Y_C1"))
    assert G.FR_GENERATOR == val(FR, "GENERATOR") * pow(1 << 256, -1, G.R_MOD) % G.R_MOD and G.FR_TWO_ADICITY == val(FR, "TWO_ADICITY")
    assert G.FQ_TWO_ADICITY == val(FQ, "TWO_ADICITY") and G.Q753_TWO_ADICITY == val(Q7, "TWO_ADICITY")
    assert G.Q753_GENERATOR == val(Q7, "GENERATOR") * pow(1 << 768, -1, G.Q753_MOD) % G.Q753_MOD
    # the 2^47-th root the device tables are built from is the reference's, not merely some primitive root
    assert pow(G.FR_GENERATOR, (G.R_MOD - 1) >> G.FR_TWO_ADICITY, G.R_MOD) == val(FR, "TWO_ADIC_ROOT_OF_UNITY") * pow(1 << 256, -1, G.R_MOD) % G.R_MOD
    assert pow(G.Q753_GENERATOR, (G.Q753_MOD - 1) >> G.Q753_TWO_ADICITY, G.Q753_MOD) == val(Q7, "TWO_ADIC_ROOT_OF_UNITY") * pow(1 << 768, -1, G.Q753_MOD) % G.Q753_MOD
    out = subprocess.run([sys.executable, path], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "zk-mpc_amd", "csrc", "consts.cuh")).read()


def test_constants_typed_into_host_code_equal_the_reference():
    hf = open(os.path.join(ROOT, "zk-mpc_amd", "csrc", "hostfield64.hpp")).read()

    def arr(text, name, n):
        m = re.search(r"%s\[%d\] = \{(.*?)\};" % (name, n), text, re.S)
        return [int(re.sub(r"ull$", "", t.strip()), 0) for t in m.group(1).split(",")]
    assert arr(hf, "P", 6) == limbs(FQ, "MODULUS") and arr(hf, "ONE", 6) == limbs(FQ, "R")
    assert int(re.search(r"INV = (\d+)ull", hf).group(1)) == val(FQ, "INV")
    rng = open(os.path.join(ROOT, "zk-mpc_amd", "csrc", "rng.hip")).read()
    assert arr(rng, "R", 4) == limbs(FR, "MODULUS")


# ---- the library's own constants, through the C ABI ---------------------------------------------------------------------------

def _from_canonical(ctx, v):
    import ctypes as C
    canon, out = np.array([(v >> (64 * i)) & (2 ** 64 - 1) for i in range(4)], dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_fr_from_canonical(canon.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    return [int(x) for x in out]


def _to_canonical(ctx, limbs4):
    import ctypes as C
    a, out = np.array(limbs4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_fr_to_canonical(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    return sum(int(x) << (64 * i) for i, x in enumerate(out))


def _fr_mont_limbs(v):
    m = v * (1 << 256) % val(FR, "MODULUS")
    return [(m >> (64 * i)) & (2 ** 64 - 1) for i in range(4)]


@pytest.mark.gpu
def test_device_field_and_domain_constants_equal_the_reference():
    p = val(FR, "MODULUS")
    Rinv = pow(1 << 256, -1, p)
    ctx = Z.Context(0)
    try:
        # Montgomery R, GENERATOR, TWO_ADIC_ROOT_OF_UNITY in the reference's in-memory form
        assert _from_canonical(ctx, 1) == limbs(FR, "R")
        assert _from_canonical(ctx, 22) == limbs(FR, "GENERATOR")
        assert _to_canonical(ctx, limbs(FR, "R")) == 1
        root = val(FR, "TWO_ADIC_ROOT_OF_UNITY") * Rinv % p
        assert _from_canonical(ctx, root) == limbs(FR, "TWO_ADIC_ROOT_OF_UNITY")
        assert [int(x) for x in ctx.fr_pow(np.array(limbs(FR, "TWO_ADIC_ROOT_OF_UNITY"), dtype=np.uint64), 1 << 46)] == _fr_mont_limbs(p - 1)
        # the DEVICE's twiddles and coset powers: the transform of the unit vector e_1 is (w^j), its coset transform (g w^j),
        # with w = TWO_ADIC_ROOT^(2^(47 - k)) (radix2/mod.rs:67-69) and g = GENERATOR (fr.rs GENERATOR = 22)
        for k in (1, 5, 11, 16):
            n = 1 << k
            e1 = np.zeros((n, 4), dtype=np.uint64)
            e1[1] = limbs(FR, "R")
            w = pow(root, 1 << (val(FR, "TWO_ADICITY") - k), p)
            for coset in (False, True):
                d = ctx.upload(e1)
                ctx.ntt_dev(d.ptr, k, False, coset)
                got = cv.fr_from_mont(ctx.download(d, (n, 4)))
                g = 22 if coset else 1
                idx = sorted({j for j in (0, 1, 2, n // 2, n - 1) if j < n})
                assert [got[j] for j in idx] == [g * pow(w, j, p) % p for j in idx], (k, coset)
                d.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_device_generators_and_flag_bits_equal_the_reference():
    q, R = val(FQ, "MODULUS"), 1 << 384
    g1, g2 = REF["bls12_377_g1"], REF["bls12_377_g2"]
    mont6 = lambda v: [((v * R % q) >> (64 * i)) & (2 ** 64 - 1) for i in range(6)]
    one = np.array(limbs(FR, "R"), dtype=np.uint64)
    ctx = Z.Context(0)
    try:
        r1cs = ctx.r1cs_mul_chain(2)
        tau = np.array(_fr_mont_limbs(7), dtype=np.uint64)                   # (not a root of unity: outside the evaluation domain)
        pk = ctx.groth16_setup(r1cs, one, one, one, one, tau, one, one)      # alpha = beta = gamma = delta = 1, generators x 1
        assert list(pk.vk_g1(0)) == mont6(int(g1["G1_GENERATOR_X"]["value"])) + mont6(int(g1["G1_GENERATOR_Y"]["value"]))
        want2 = sum([mont6(int(g2[k]["value"])) for k in ("G2_GENERATOR_X_C0", "G2_GENERATOR_X_C1", "G2_GENERATOR_Y_C0", "G2_GENERATOR_Y_C1")], [])
        assert list(pk.vk_g2(0)) == want2
        # compressed form: x little-endian, bit 7 of the last byte = "y > -y", bit 6 = infinity (flags.rs u8_bitmask)
        gx, gy = int(g1["G1_GENERATOR_X"]["value"]), int(g1["G1_GENERATOR_Y"]["value"])
        pts = np.array([mont6(gx) + mont6(gy), mont6(gx) + mont6(q - gy), [0] * 12], dtype=np.uint64)
        b = ctx.bases_upload(pts, 1)
        raw = b.serialize(compressed=True)
        pos, inf = 1 << val("serialize_sw_flags", "POSITIVE_Y_BIT"), 1 << val("serialize_sw_flags", "INFINITY_BIT")
        for k, y in ((0, gy), (1, q - gy)):
            rec = raw[48 * k:48 * k + 48]
            assert int.from_bytes(rec, "little") & ~((pos | inf) << 376) == gx
            assert bool(rec[47] & pos) == (y > q - y) and not rec[47] & inf
        assert raw[96:144] == bytes(47) + bytes([inf])
        b.free(); pk.free(); r1cs.free()
    finally:
        ctx.close()


@pytest.mark.gpu
def test_device_she_modulus_equals_the_reference():
    q, R = val(Q7, "MODULUS"), 1 << 768
    w12 = lambda v: [((v * R % q) >> (64 * i)) & (2 ** 64 - 1) for i in range(12)]
    a, b = q - 1, 12345678901234567890123456789
    ctx = Z.Context(0)
    try:
        da, db = ctx.upload(np.array([w12(a), w12(2)], dtype=np.uint64)), ctx.upload(np.array([w12(1), w12(b)], dtype=np.uint64))
        out = ctx.alloc(2 * 96)
        ctx.she_vec_op_dev(1, da.ptr, db.ptr, out.ptr, 2)                  # add: (q - 1) + 1 = 0 ; 2 + b
        got = ctx.download(out, (2, 12))
        assert list(got[0]) == [0] * 12 and list(got[1]) == w12(2 + b)
        ctx.she_vec_op_dev(0, da.ptr, db.ptr, out.ptr, 2)                  # mul: (q - 1) * 1 ; 2 b
        got = ctx.download(out, (2, 12))
        assert list(got[0]) == w12(q - 1) and list(got[1]) == w12(2 * b % q)
    finally:
        ctx.close()
