"""The two known answers of LONG ARITHMETIC CHAINS that the reference's own tests hold for this path, run through the HIP path:

  * Fq::multiplicative_generator().pow(T) == Fq::two_adic_root_of_unity(), T a literal of the test
    (arkworks/curves/bls12_377/src/fields/tests.rs:352-370); the same relation holds between fr.rs's GENERATOR, T and
    TWO_ADIC_ROOT_OF_UNITY (the ff test templates check it for every field: `field_test` -> `fft_field_test`);
  * "the point with x = 1 and the smaller y, scaled by the cofactor, is the G1 generator; x = 0 scales to zero"
    (arkworks/curves/bls12_377/src/curves/tests.rs:93-122).

Every number below comes out of tests/golden/ref_constants.json (parsed from the reference's sources by
tools/pin_reference_constants.py, file:line attached); the only Python arithmetic is one square root (y from x = 1) and the
choice of the smaller root.  ~380 dependent squarings and ~190 multiplications in Fq, ~250 + ~125 in Fr: a wrong carry, a wrong
reduction constant or a lazy-domain bound violated anywhere in fp29.cuh does not survive them."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constants.json")))


def limbs(group, name):
    return np.array([int(x, 16) for x in REF[group][name]["limbs"]], dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("lazy", [0, 1])
def test_fq_generator_to_the_T_is_the_root_of_unity(ctx, lazy):
    """fields/tests.rs:352-370 with the test's own literal exponent, as one chain of device products (exact and lazy domain)."""
    gen, root = limbs("bls12_377_fq", "GENERATOR"), limbs("bls12_377_fq", "TWO_ADIC_ROOT_OF_UNITY")     # Montgomery form, as in fq.rs
    exp = limbs("bls12_377_tests", "FQ_ROOT_OF_UNITY_EXPONENT")
    assert len(exp) == 6
    out = np.zeros(6, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_diag_fq_pow_dev(ctx.h, _p(gen), _p(exp), lazy, _p(out)))
    assert list(out) == list(root)
    # two_adic_root_of_unity().pow([1 << TWO_ADICITY]) == one (the same test, :366-369); and the half-way power is -1
    s = int(REF["bls12_377_fq"]["TWO_ADICITY"]["value"])
    e = np.zeros(6, dtype=np.uint64); e[0] = 1 << s
    ctx._ck(ctx.lib.zk_diag_fq_pow_dev(ctx.h, _p(root), _p(e), lazy, _p(out)))
    assert list(out) == list(limbs("bls12_377_fq", "R"))
    e[0] = 1 << (s - 1)
    ctx._ck(ctx.lib.zk_diag_fq_pow_dev(ctx.h, _p(root), _p(e), lazy, _p(out)))
    q = int(REF["bls12_377_fq"]["MODULUS"]["value"])
    minus_one = (q - int(REF["bls12_377_fq"]["R"]["value"])) % q
    assert sum(int(v) << (64 * i) for i, v in enumerate(out)) == minus_one


@pytest.mark.parametrize("lazy", [0, 1])
def test_fr_generator_to_the_T_is_the_root_of_unity(ctx, lazy):
    gen, root, T = limbs("bls12_377_fr", "GENERATOR"), limbs("bls12_377_fr", "TWO_ADIC_ROOT_OF_UNITY"), limbs("bls12_377_fr", "T")
    out = np.zeros(4, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_diag_fr_pow_dev(ctx.h, _p(gen), _p(T), lazy, _p(out)))
    assert list(out) == list(root)
    e = np.zeros(4, dtype=np.uint64); e[0] = 1 << int(REF["bls12_377_fr"]["TWO_ADICITY"]["value"])
    ctx._ck(ctx.lib.zk_diag_fr_pow_dev(ctx.h, _p(root), _p(e), lazy, _p(out)))
    assert list(out) == list(limbs("bls12_377_fr", "R"))


def test_fr_generator_to_the_T_through_the_vector_kernel(ctx):
    """The same chain through the PRODUCT's element-wise kernel (zk_fr_vec_op_dev: Fp256::mul_assign on device vectors): every
    lane of a 4 096-element vector walks GENERATOR^T by square-and-multiply, one launch per step; all lanes must end on
    TWO_ADIC_ROOT_OF_UNITY."""
    n = 4096
    gen, root = limbs("bls12_377_fr", "GENERATOR"), limbs("bls12_377_fr", "TWO_ADIC_ROOT_OF_UNITY")
    T = int(REF["bls12_377_fr"]["T"]["value"])
    base = ctx.upload(np.tile(gen, (n, 1)))
    acc = ctx.upload(np.tile(gen, (n, 1)))
    for bit in bin(T)[3:]:                                   # the top bit is the initial value
        ctx.fr_vec_op_dev(0, acc.ptr, acc.ptr, acc.ptr, n)
        if bit == "1":
            ctx.fr_vec_op_dev(0, acc.ptr, base.ptr, acc.ptr, n)
    got = ctx.download(acc, (n, 4))
    assert (got == root[None, :]).all()
    base.free(); acc.free()


def _g1_point_with_x(x):
    y = O.fq_sqrt((x ** 3 + 1) % O.Q_MOD)
    assert y is not None
    return (x, min(y, O.Q_MOD - y))                         # "if y < -y { y } else { -y }"


@pytest.mark.parametrize("route", ["host_slices", "resident", "window_multiples"])
def test_cofactor_times_the_point_at_x_1_is_the_g1_generator(ctx, route):
    """curves/tests.rs:93-122 as MSMs: [(x, y)] x [COFACTOR] through zk_msm_g1 (the trait-shaped entry), through a resident table,
    and -- a table long enough to carry them -- through window multiples with the point of interest among 4 999 points whose
    scalars are zero.  The points are NOT in the prime-order subgroup (that is the point of the test), so the bucket method's
    P + P, 2 P = -P (x = 0 has order 3) and P - P cases all occur."""
    g1 = REF["bls12_377_g1"]
    cof = int(g1["COFACTOR"]["value"])
    want = (int(g1["G1_GENERATOR_X"]["value"]), int(g1["G1_GENERATOR_Y"]["value"]))
    x_gen = int(REF["bls12_377_tests"]["G1_GENERATOR_RAW_X"]["value"])
    assert cof < O.R_MOD
    for x in range(x_gen + 1):
        p = _g1_point_with_x(x)
        expect = want if x == x_gen else None                # every smaller x: the cofactor sends the point to zero
        if route == "host_slices":
            got = cv.g1_projective_to_affine(ctx.multi_scalar_mul_g1(cv.g1_affine_to_array([p]), cv.fr_to_mont([cof])))
        else:
            n = 1 if route == "resident" else 5000
            pts = cv.g1_affine_to_array([p] + [O.G1_GEN] * (n - 1))
            sc = cv.fr_to_mont([cof] + [0] * (n - 1))
            b = ctx.bases_upload(pts, 1)
            if route == "window_multiples":
                b.precompute()
                assert ctx.lib.zk_bases_window_bits(b.h) > 0
            ds = ctx.upload(sc)
            got = cv.g1_projective_to_affine(ctx.msm_dev(b, 0, ds.ptr, n))
            b.free(); ds.free()
        assert got == expect, (route, x)
