#!/usr/bin/env python3
"""A STAND-IN for ref_kats.json written by the ORACLE (not by the reference): same schema, same order of rng draws as
dump_kats.rs.  Its only purpose is to exercise tests/test_ref_vectors.py's consumer logic (schema, replay order, byte
formats) while no reference-generated file exists; it pins nothing and must never be committed as tests/golden/ref_kats.json.

  python tools/ref_vectors/make_standin.py /tmp/standin.json && ZK_REF_KATS=/tmp/standin.json python -m pytest tests/test_ref_vectors.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fsrng_ref as FR  # noqa: E402
import zkref as O  # noqa: E402


def hx(v, n):
    return int(v).to_bytes(n, "little").hex()


def fq_rand(r, mod, words, shave):
    while True:
        l = [r.next_u64() for _ in range(words)]
        l[-1] &= (1 << (64 - shave)) - 1
        v = sum(x << (64 * i) for i, x in enumerate(l))
        if v < mod:
            return v * pow(1 << (64 * words), -1, mod) % mod      # the words are the Montgomery form


def main():
    out = {}
    r0 = FR.test_rng()
    out["test_rng_u64"] = ["%016x" % r0.next_u64() for _ in range(8)]
    r = FR.test_rng()
    p, q = O.R_MOD, O.Q_MOD
    rows = []
    for _ in range(8):
        a, b = r.next_fr(), r.next_fr()
        rows.append([hx(a, 32), hx(b, 32), hx(a * b % p, 32), hx((a + b) % p, 32), hx((a - b) % p, 32), hx(pow(a, -1, p), 32)])
    out["fr_ops"] = rows
    rows = []
    for _ in range(8):
        a, b = fq_rand(r, q, 6, 7), fq_rand(r, q, 6, 7)
        rows.append([hx(a, 48), hx(b, 48), hx(a * b % q, 48), hx((a + b) % q, 48), hx((a - b) % q, 48), hx(a * a % q, 48)])
    out["fq_ops"] = rows
    k1, k2 = r.next_fr(), r.next_fr()
    P1, Q1 = O.g1_mul(O.G1_GEN, k1), O.g1_mul(O.G1_GEN, k2)
    P2, Q2 = O.g2_mul(O.G2_GEN, k1), O.g2_mul(O.G2_GEN, k2)
    u1, u2 = O.g1_serialize_uncompressed, O.g2_serialize_uncompressed
    out["group"] = {"k1": hx(k1, 32), "k2": hx(k2, 32),
                    "g1": [u1(x).hex() for x in (P1, Q1, O.g1_add(P1, Q1), O.g1_add(P1, P1), O.g1_add(P1, O.g1_neg(Q1)))],
                    "g2": [u2(x).hex() for x in (P2, Q2, O.g2_add(P2, Q2), O.g2_add(P2, P2), O.g2_add(P2, O.g2_neg(Q2)))]}
    n = (1 << 10) - 1
    ks = [r.next_fr() for _ in range(n)]
    ss = [r.next_fr() for _ in range(n)]
    e = sum(k * s for k, s in zip(ks, ss)) % p
    out["msm"] = {"n": n, "k_first": hx(ks[0], 32), "s_first": hx(ss[0], 32), "g1": u1(O.g1_mul(O.G1_GEN, e)).hex(),
                  "g2": u2(O.g2_mul(O.G2_GEN, e)).hex()}
    v = [r.next_fr() for _ in range(64)]
    d = O.Domain(64)
    h = lambda xs: [hx(x, 32) for x in xs]
    out["fft"] = {"input": h(v), "fft": h(d.fft(v)), "ifft": h(d.ifft(v)), "coset_fft": h(d.coset_fft(v)), "coset_ifft": h(d.coset_ifft(v))}
    alpha, beta, gamma, delta, g1k, g2k, a, b, rr, ss_ = [r.next_fr() for _ in range(10)]
    tau = FR.test_rng().next_fr()
    r1cs = O.R1CS(2, 2, [[(1, 2)]] * 6, [[(1, 3)]] * 6, [[(1, 1)]] * 6)
    z = [1, a * b % p, a, b]
    pks = O.ProvingKeyScalars(r1cs, O.Trapdoor(alpha, beta, gamma, delta, tau, g1k, g2k))
    proof = O.proof_serialize(*O.predict_proof(r1cs, pks, z, rr, ss_))
    out["groth16_simple"] = dict(alpha=hx(alpha, 32), beta=hx(beta, 32), gamma=hx(gamma, 32), delta=hx(delta, 32), g1_k=hx(g1k, 32),
                                 g2_k=hx(g2k, 32), tau=hx(tau, 32), a=hx(a, 32), b=hx(b, 32), r=hx(rr, 32), s=hx(ss_, 32),
                                 proof=proof.hex(), vk=O.vk_serialize(O.ProvingKey(pks), True).hex(), pk_sha_len=0,
                                 pk=O.pk_serialize(O.ProvingKey(pks), True).hex(), pk_uncompressed=O.pk_serialize(O.ProvingKey(pks), False).hex())
    N = 4
    x = [fq_rand(r, O.Q753, 12, 15) for _ in range(N)]
    y = [fq_rand(r, O.Q753, 12, 15) for _ in range(N)]
    out["she_mul"] = {"n": N, "x": [hx(t, 96) for t in x], "y": [hx(t, 96) for t in y], "xy": [hx(t, 96) for t in O.encodedtext_mul(x, y)]}
    # (8) the Marlin transcript of MySimpleCircuit: a, b from r; the SRS and the prover's zk_rng from fresh test_rng()s
    import marlin_full_ref as MF
    import marlin_ref as M
    a, b = r.next_fr(), r.next_fr()
    r1cs = O.R1CS(2, 2, [[(1, 2)]] * 6, [[(1, 3)]] * 6, [[(1, 1)]] * 6)
    sq, zz = M.pad_and_square(r1cs, [1, a * b % p, a, b])
    index = M.Index(sq)
    srs_rng = FR.test_rng()
    pp = O.KzgParams(MF.max_degree_for(index) + 3, srs_rng.next_fr(), g_k=srs_rng.next_fr(), gg_k=srs_rng.next_fr(), h_k=srs_rng.next_fr())
    keys = MF.Keys(index, pp)
    proof = MF.prove(keys, zz, FR.test_rng())
    pub = zz[1:index.num_instance]
    assert MF.verify(keys, pub, proof)
    seed = MF.PROTOCOL_NAME + keys.ivk_bytes() + b"".join(MF.fr_bytes(v) for v in pub)
    absorb = [b"".join(MF.comm_bytes(c) for c in rnd) for rnd in proof.commitments]
    ch, _ = MF.transcript_challenges(keys.ivk_bytes(), index, pub, proof.commitments)
    out["marlin_simple"] = dict(a=hx(a, 32), b=hx(b, 32), public_input=[hx(v, 32) for v in pub], seed=seed.hex(), absorb=[x.hex() for x in absorb],
                                alpha=hx(ch["alpha"], 32), eta_a=hx(ch["eta_a"], 32), eta_b=hx(ch["eta_b"], 32), eta_c=hx(ch["eta_c"], 32),
                                beta=hx(ch["beta"], 32), gamma=hx(ch["gamma"], 32), proof=proof.serialize().hex(), ivk="", srs="",
                                domain_h=index.dom_h.size)
    # (9) MarlinKZG10::commit's draw order: three polynomials of 6 coefficients from r; the commit rng is a fresh test_rng()
    coeffs = [[r.next_fr() for _ in range(6)] for _ in range(3)]
    pp9 = O.KzgParams(16, 0x1234567, g_k=3, gg_k=7, h_k=5)
    crng = FR.test_rng()
    items = []
    for label, c, bound, hiding in (("hb", coeffs[0], 8, True), ("h", coeffs[1], None, True), ("plain", coeffs[2], None, False)):
        blind = [crng.next_fr() for _ in range(3)] if hiding else []
        comm = O.kzg_commit(pp9, c, blind or None)
        ser = O.g1_serialize(comm)
        sblind = None
        if bound is not None:
            sblind = [crng.next_fr() for _ in range(3)]
            sc = O.g1_add(O.msm_naive(pp9.powers_of_g[16 - bound:], c, O.FqOps), O.msm_naive(pp9.powers_of_gamma_g, sblind, O.FqOps))
            ser += b"\x01" + O.g1_serialize(sc)
        else:
            ser += b"\x00"
        items.append(dict(label=label, coeffs=[hx(v, 32) for v in c], commitment=ser.hex(), blind=[hx(v, 32) for v in blind],
                          shifted_blind=None if sblind is None else [hx(v, 32) for v in sblind]))
    out["marlin_pc_commit"] = dict(polys=items, rng_next_u64_after=crng.next_u64(), max_degree=16,
                                   powers=[u1(x).hex() for x in pp9.powers_of_g], shifted_powers=[u1(x).hex() for x in pp9.powers_of_g[8:]],
                                   powers_of_gamma_g=[u1(x).hex() for x in pp9.powers_of_gamma_g[:3]])
    # (14) the layouts rustc is EXPECTED to give the collaborative element types (a discriminant byte first, the payload at 8-byte
    # alignment): what dump_kats.rs finds by pattern search on the real types -- a stand-in pins nothing about the reference
    out["mpc_layouts"] = [
        dict(type="MpcField<Fr, AdditiveFieldShare<Fr>>", size=40, off_tag=0, tag_public=0, tag_shared=1, off_public=8, off_share=8, off_mac=-1),
        dict(type="MpcField<Fr, SpdzFieldShare<Fr>>", size=72, off_tag=0, tag_public=0, tag_shared=1, off_public=8, off_share=8, off_mac=40),
        dict(type="MpcG1Affine<Bls12_377, AdditivePairingShare>", size=112, off_x=8, off_y=56, off_infinity=104, public_first_bytes="00" * 8),
        dict(type="MpcG2Affine<Bls12_377, AdditivePairingShare>", size=208, off_x=8, off_y=104, off_infinity=200, public_first_bytes="00" * 8),
    ]
    json.dump(out, open(sys.argv[1], "w"))


if __name__ == "__main__":
    main()
