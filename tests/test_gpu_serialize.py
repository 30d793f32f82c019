"""GPU parity: arkworks CanonicalSerialize of point tables and Groth16 keys against the oracle."""
import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv
from zk_mpc_amd import serialize as S
from helpers import mont1

pytestmark = pytest.mark.gpu


def test_point_tables_both_forms(ctx):
    rng = O.Prng(950)
    p1 = [O.g1_mul(O.G1_GEN, rng.fr()) for _ in range(9)] + [None]
    p1 += [O.g1_neg(p1[0])]                                    # same x, other sign bit
    p2 = [O.g2_mul(O.G2_GEN, rng.fr()) for _ in range(5)] + [None]
    p2 += [O.g2_neg(p2[0])]
    b1 = ctx.bases_upload(cv.g1_affine_to_array(p1), 1)
    b2 = ctx.bases_upload(cv.g2_affine_to_array(p2), 2)
    assert b1.serialize(True) == b"".join(O.g1_serialize(p) for p in p1)
    assert b1.serialize(False) == b"".join(O.g1_serialize_uncompressed(p) for p in p1)
    assert b2.serialize(True) == b"".join(O.g2_serialize(p) for p in p2)
    assert b2.serialize(False) == b"".join(O.g2_serialize_uncompressed(p) for p in p2)
    assert b1.serialize(True, offset=3, n=2) == b"".join(O.g1_serialize(p) for p in p1[3:5])
    # uncompressed round trip through the device
    r1 = ctx.bases_deserialize_uncompressed(b1.serialize(False), len(p1), 1)
    r2 = ctx.bases_deserialize_uncompressed(b2.serialize(False), len(p2), 2)
    assert cv.g1_array_to_affine(r1.download()) == p1
    assert cv.g2_array_to_affine(r2.download()) == p2
    # compressed round trip: one square root per point (Tonelli-Shanks in Fq, the norm method in Fq2), sign from the flag
    c1 = ctx.bases_deserialize_compressed(b1.serialize(True), len(p1), 1)
    c2 = ctx.bases_deserialize_compressed(b2.serialize(True), len(p2), 2)
    assert cv.g1_array_to_affine(c1.download()) == p1
    assert cv.g2_array_to_affine(c2.download()) == p2


def test_deserialize_rejects_garbage(ctx):
    rng = O.Prng(951)
    good = O.g1_serialize_uncompressed(O.g1_mul(O.G1_GEN, rng.fr()))
    bad = bytearray(good)
    bad[0] ^= 1                                                 # x changed: no longer on the curve
    with pytest.raises(Exception, match="not on the curve"):
        ctx.bases_deserialize_uncompressed(bytes(bad), 1, 1)
    xs = bytearray(O.g1_serialize(O.g1_mul(O.G1_GEN, 5)))
    for delta in range(1, 40):                                  # some x + delta has no y: x^3 + 1 is a non-residue
        cand = bytearray(xs)
        cand[0] = (cand[0] + delta) & 0xFF
        x = int.from_bytes(bytes(cand[:47]) + bytes([cand[47] & 0x3F]), "little")
        if pow((x ** 3 + 1) % O.Q_MOD, (O.Q_MOD - 1) // 2, O.Q_MOD) != 1:
            with pytest.raises(Exception, match="not on the curve"):
                ctx.bases_deserialize_compressed(bytes(cand), 1, 1)
            break
    else:
        raise AssertionError("no non-residue found")
    flagged = bytearray(good)
    flagged[95] |= 0x80                                         # sign flag has no place in the uncompressed form
    with pytest.raises(Exception, match="flag"):
        ctx.bases_deserialize_uncompressed(bytes(flagged), 1, 1)


def test_groth16_keys_wire_format(ctx):
    """Keys of the device's own setup serialise to the oracle's bytes; a key read back from its uncompressed bytes proves
    the same proof."""
    rng = O.Prng(952)
    n = 20
    r1cs, z = O.mul_chain_r1cs(n, rng.fr(), rng.fr())
    td = O.Trapdoor(*[rng.fr() for _ in range(7)])
    opk = O.ProvingKey(O.ProvingKeyScalars(r1cs, td))
    dr = ctx.r1cs_mul_chain(n)
    dpk = ctx.groth16_setup(dr, *[mont1(v) for v in (td.alpha, td.beta, td.gamma, td.delta, td.tau, td.g1_k, td.g2_k)])
    for compressed in (True, False):
        assert S.verifying_key_bytes(ctx, dpk, compressed) == O.vk_serialize(opk, compressed)
        assert S.proving_key_bytes(ctx, dpk, compressed) == O.pk_serialize(opk, compressed)
    pk3, _, _ = S.proving_key_from_bytes(ctx, O.pk_serialize(opk, True), compressed=True)
    pk2, gamma_g2, gamma_abc = S.proving_key_from_bytes(ctx, O.pk_serialize(opk, False))
    assert cv.g2_array_to_affine(gamma_g2.reshape(1, 24)) == [opk.gamma_g2]
    assert cv.g1_array_to_affine(gamma_abc) == opk.gamma_abc_g1
    r, s = rng.fr(), rng.fr()
    zm = cv.fr_to_mont(z)
    proof = ctx.create_proof(pk2, dr, zm, mont1(r), mont1(s))
    assert proof == ctx.create_proof(dpk, dr, zm, mont1(r), mont1(s))
    assert proof == ctx.create_proof(pk3, dr, zm, mont1(r), mont1(s))
    A, B, Cc = O.create_proof(r1cs, opk, z, r, s)
    assert proof == O.proof_serialize(A, B, Cc)


def test_kzg_srs_wire_format(ctx):
    """UniversalParams::serialize (kzg10/data_structures.rs:40-80; the file src/marlin.rs:371-376 writes) laid out by hand from
    the struct -- Vec powers_of_g | BTreeMap powers_of_gamma_g (u64 key, point) | h | beta_h | empty neg_powers_of_h -- with the
    oracle's point bytes, both forms; the SRS read back commits to the same point."""
    rng = O.Prng(953)
    max_degree = 37
    beta, g_k, gg_k, h_k = rng.fr(), rng.fr(), rng.fr(), rng.fr()
    pp = O.KzgParams(max_degree, beta, g_k=g_k, gg_k=gg_k, h_k=h_k)
    pw = ctx.alloc((max_degree + 2) * 32)
    ctx.fr_powers_dev(mont1(beta), mont1(1), max_degree + 2, pw.ptr)
    powers_g = ctx.fixed_base(pw.ptr, max_degree + 1, 1, mont1(g_k))
    powers_gamma_g = ctx.fixed_base(pw.ptr, max_degree + 2, 1, mont1(gg_k))
    h = cv.g2_affine_to_array([pp.h])[0]
    beta_h = cv.g2_affine_to_array([pp.beta_h])[0]
    u64 = lambda v: v.to_bytes(8, "little")
    for compressed in (True, False):
        s1, s2 = (O.g1_serialize, O.g2_serialize) if compressed else (O.g1_serialize_uncompressed, O.g2_serialize_uncompressed)
        want = u64(max_degree + 1) + b"".join(s1(p) for p in pp.powers_of_g)
        want += u64(max_degree + 2) + b"".join(u64(i) + s1(p) for i, p in enumerate(pp.powers_of_gamma_g))
        want += s2(pp.h) + s2(pp.beta_h) + u64(0)
        got = S.kzg_srs_bytes(ctx, powers_g, powers_gamma_g, h, beta_h, compressed)
        assert got == want
        pg, pgg, h2, bh2 = S.kzg_srs_from_bytes(ctx, got, compressed)
        assert cv.g1_array_to_affine(pg.download()) == pp.powers_of_g
        assert cv.g1_array_to_affine(pgg.download()) == pp.powers_of_gamma_g
        assert cv.g2_array_to_affine(h2.reshape(1, 24)) == [pp.h] and cv.g2_array_to_affine(bh2.reshape(1, 24)) == [pp.beta_h]
        coeffs = [rng.fr() for _ in range(max_degree + 1)]
        dc = ctx.upload(cv.fr_to_mont(coeffs))
        assert cv.g1_projective_to_affine(ctx.kzg_commit_dev(pg, dc.ptr, len(coeffs))) == O.kzg_commit(pp, coeffs)
        with pytest.raises(Exception, match="truncated|trailing|neg_powers"):
            S.kzg_srs_from_bytes(ctx, got[:-3], compressed)
        pg.free(); pgg.free()


def test_key_framing_rejects_malformed_input(ctx):
    rng = O.Prng(954)
    n = 9
    r1cs, z = O.mul_chain_r1cs(n, rng.fr(), rng.fr())
    td = O.Trapdoor(*[rng.fr() for _ in range(7)])
    opk = O.ProvingKey(O.ProvingKeyScalars(r1cs, td))
    data = O.pk_serialize(opk, False)
    with pytest.raises(Exception, match="truncated"):
        S.proving_key_from_bytes(ctx, data[:-10])
    with pytest.raises(Exception, match="trailing"):
        S.proving_key_from_bytes(ctx, data + b"\\0")
    bad = bytearray(data)
    bad[5] ^= 1                                                  # alpha_g1.x: not on the curve any more
    with pytest.raises(Exception, match="not on the curve"):
        S.proving_key_from_bytes(ctx, bytes(bad))
