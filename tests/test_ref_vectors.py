"""Known-answer vectors produced BY THE REFERENCE (tools/ref_vectors/dump_kats.rs, run with cargo on a machine that has the
reference checked out) against the oracle (CPU) and the library (GPU).  Skipped while tests/golden/ref_kats.json is absent:
the reference cannot be built in the build image (no Rust toolchain) -- tools/ref_vectors/README.md is the recipe."""
import json
import os

import numpy as np
import pytest

import fsrng_ref as FR
import zkref as O
import zk_mpc_amd.convert as cv

PATH = os.environ.get("ZK_REF_KATS") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kats.json")


@pytest.fixture(scope="module")
def kats():
    if not os.path.exists(PATH):
        pytest.skip("no reference vectors: run tools/ref_vectors/dump_kats.rs with cargo (tools/ref_vectors/README.md) and "
                    "copy its output to tests/golden/ref_kats.json")
    return json.load(open(PATH))


def le(h):
    return int.from_bytes(bytes.fromhex(h), "little")


def g1_unc(h):
    b = bytes.fromhex(h)
    if b[95] & 0x40:
        return None
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little") & ((1 << 382) - 1))


def test_rng_replay(kats):
    r = FR.test_rng()
    assert ["%016x" % r.next_u64() for _ in range(8)] == kats["test_rng_u64"]


def replay(kats):
    """The inputs in the order dump_kats.rs draws them; every replayed value is compared with the dumped one."""
    r = FR.test_rng()
    out = {"fr": [], "fq": []}
    for row in kats["fr_ops"]:
        a, b = r.next_fr(), r.next_fr()
        assert (a, b) == (le(row[0]), le(row[1]))
        out["fr"].append((a, b))
    return out, r


def test_field_ops_oracle(kats):
    ins, _ = replay(kats)
    p = O.R_MOD
    for (a, b), row in zip(ins["fr"], kats["fr_ops"]):
        assert [a * b % p, (a + b) % p, (a - b) % p, pow(a, -1, p)] == [le(x) for x in row[2:]]
    q = O.Q_MOD
    for row in kats["fq_ops"]:
        a, b = le(row[0]), le(row[1])
        assert [a * b % q, (a + b) % q, (a - b) % q, a * a % q] == [le(x) for x in row[2:]]
        import zkref_c as OC
        am, bm = cv._ints_to_limbs([cv.fq_to_mont_int(a)], 6)[0], cv._ints_to_limbs([cv.fq_to_mont_int(b)], 6)[0]
        assert cv.fq_from_mont_int(cv._limbs_to_ints(OC.fq_mul(am, bm).reshape(1, 6))[0]) == le(row[2])


def test_group_law_oracle(kats):
    g = kats["group"]
    k1, k2 = le(g["k1"]), le(g["k2"])
    P, Qp = O.g1_mul(O.G1_GEN, k1), O.g1_mul(O.G1_GEN, k2)
    want = [g1_unc(h) for h in g["g1"]]
    assert [P, Qp, O.g1_add(P, Qp), O.g1_add(P, P), O.g1_add(P, O.g1_neg(Qp))] == want
    assert O.g1_serialize_uncompressed(P).hex() == g["g1"][0]
    P2 = O.g2_mul(O.G2_GEN, k1)
    assert O.g2_serialize_uncompressed(P2).hex() == g["g2"][0]
    assert O.g2_serialize_uncompressed(O.g2_add(P2, O.g2_mul(O.G2_GEN, k2))).hex() == g["g2"][2]


def _msm_inputs(kats, r=None):
    if r is None:
        _, r = replay(kats)
        for _ in kats["fq_ops"]:
            for _ in range(2):
                # Fq::rand: six u64, top 7 bits cleared (384 - 377), rejection -- not needed as values, only to advance the stream
                while True:
                    l = [r.next_u64() for _ in range(6)]
                    l[5] &= (1 << 57) - 1
                    if sum(x << (64 * i) for i, x in enumerate(l)) < O.Q_MOD:
                        break
        r.next_fr(); r.next_fr()                       # k1, k2 of the group block
    n = kats["msm"]["n"]
    ks = [r.next_fr() for _ in range(n)]
    ss = [r.next_fr() for _ in range(n)]
    assert ks[0] == le(kats["msm"]["k_first"]) and ss[0] == le(kats["msm"]["s_first"])
    return ks, ss, r


def test_msm_oracle(kats):
    ks, ss, _ = _msm_inputs(kats)
    e = sum(k * s for k, s in zip(ks, ss)) % O.R_MOD
    assert O.g1_serialize_uncompressed(O.g1_mul(O.G1_GEN, e)).hex() == kats["msm"]["g1"]
    assert O.g2_serialize_uncompressed(O.g2_mul(O.G2_GEN, e)).hex() == kats["msm"]["g2"]


def test_fft_oracle(kats):
    f = kats["fft"]
    v = [le(x) for x in f["input"]]
    d = O.Domain(64)
    assert d.fft(v) == [le(x) for x in f["fft"]]
    assert d.ifft(v) == [le(x) for x in f["ifft"]]
    assert d.coset_fft(v) == [le(x) for x in f["coset_fft"]]
    assert d.coset_ifft(v) == [le(x) for x in f["coset_ifft"]]


def _simple_circuit(g):
    a, b = le(g["a"]), le(g["b"])
    r1cs = O.R1CS(2, 2, [[(1, 2)]] * 6, [[(1, 3)]] * 6, [[(1, 1)]] * 6)
    z = [1, a * b % O.R_MOD, a, b]
    td = O.Trapdoor(*[le(g[k]) for k in ("alpha", "beta", "gamma", "delta", "tau", "g1_k", "g2_k")])
    return r1cs, z, td


def test_groth16_proof_bytes_oracle(kats):
    g = kats["groth16_simple"]
    r1cs, z, td = _simple_circuit(g)
    assert td.tau == FR.test_rng().next_fr()            # generate_parameters draws tau first from a fresh test_rng()
    pks = O.ProvingKeyScalars(r1cs, td)
    proof = O.proof_serialize(*O.predict_proof(r1cs, pks, z, le(g["r"]), le(g["s"])))
    assert proof.hex() == g["proof"]


def test_she_mul_oracle(kats):
    s = kats["she_mul"]
    x, y = [le(v) for v in s["x"]], [le(v) for v in s["y"]]
    assert O.encodedtext_mul(x, y) == [le(v) for v in s["xy"]]


def _marlin_layout(m):
    """The byte layout of what Marlin's transcript absorbs (to_bytes!: lib.rs:161-164,183,207,230): PROTOCOL_NAME | IndexInfo
    (three u64) | 12 index commitments | padded public input; a commitment is x | y | infinity | has-shift | shifted point
    (48 + 48 + 1 + 1 + 97 bytes: marlin_pc/data_structures.rs:252-263, short_weierstrass_jacobian.rs:315-322)."""
    seed = bytes.fromhex(m["seed"])
    CB = 97 + 1 + 97
    assert seed[:11] == b"MARLIN-2019"
    nv, nc, nnz = [int.from_bytes(seed[11 + 8 * i:19 + 8 * i], "little") for i in range(3)]
    assert nv == nc and len(seed) == 11 + 24 + 12 * CB + 32 * len(m["public_input"])
    assert seed[11 + 24 + 12 * CB:] == b"".join(bytes.fromhex(v) for v in m["public_input"])
    for k in range(12):
        c = seed[35 + k * CB:35 + (k + 1) * CB]
        assert c[96] == 0 and c[97] == 0 and c[98:] == bytes(48) + (1).to_bytes(48, "little") + b"\x01"    # index comms: no shift, zero()
    ab = [bytes.fromhex(x) for x in m["absorb"]]
    assert [len(x) for x in ab] == [4 * CB, 3 * CB, 2 * CB]                                # 4 + 3 + 2 oracles, empty prover messages
    assert ab[1][CB + 97] == 1 and ab[2][97] == 1                                          # g_1 and g_2 carry shifted commitments
    h = 1
    while h < nc:
        h *= 2
    return seed, ab, h


def test_marlin_transcript_oracle(kats):
    """FiatShamirRng<Blake2s> (marlin/src/rng.rs:44-67) replayed on the bytes the reference absorbed: the oracle's generator draws
    the reference's alpha, eta_a, eta_b, eta_c, beta, gamma."""
    m = kats["marlin_simple"]
    seed, ab, h = _marlin_layout(m)
    dom = O.Domain(h)
    fs = FR.FiatShamirRng(seed)
    outside = lambda: next(t for t in iter(fs.next_fr, None) if pow(t, h, O.R_MOD) != 1)
    fs.absorb(ab[0])
    got = {"alpha": outside(), "eta_a": fs.next_fr(), "eta_b": fs.next_fr(), "eta_c": fs.next_fr()}
    fs.absorb(ab[1])
    got["beta"] = outside()
    fs.absorb(ab[2])
    got["gamma"] = fs.next_fr()
    assert got == {k: le(m[k]) for k in got}
    assert dom.size == h


def test_marlin_transcript_library(kats):
    """The same replay through the PRODUCT's FiatShamirRng (csrc/fsrng.hpp behind zk_fsrng_new / zk_fsrng_absorb / zk_rng_next_fr:
    host code, no device needed) -- the generator zk_marlin_prove draws its challenges from."""
    from zk_mpc_amd.api import Rng
    m = kats["marlin_simple"]
    seed, ab, h = _marlin_layout(m)
    fs = Rng.fiat_shamir(seed)
    nxt = lambda: cv.fr_from_mont(fs.next_fr().reshape(1, 4))[0]
    outside = lambda: next(t for t in iter(nxt, None) if pow(t, h, O.R_MOD) != 1)
    fs.absorb(ab[0])
    got = {"alpha": outside(), "eta_a": nxt(), "eta_b": nxt(), "eta_c": nxt()}
    fs.absorb(ab[1])
    got["beta"] = outside()
    fs.absorb(ab[2])
    got["gamma"] = nxt()
    assert got == {k: le(m[k]) for k in got}


# ---- the library against the same file ---------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_device_against_reference_vectors(ctx, kats):
    from helpers import csr, td_mont, mont1
    # field ops through the vector kernels
    rows = kats["fr_ops"]
    a = ctx.upload(cv.fr_to_mont([le(r[0]) for r in rows])); b = ctx.upload(cv.fr_to_mont([le(r[1]) for r in rows]))
    out = ctx.alloc(len(rows) * 32)
    for op, col in ((0, 2), (1, 3), (2, 4)):
        ctx.fr_vec_op_dev(op, a.ptr, b.ptr, out.ptr, len(rows))
        assert cv.fr_from_mont(ctx.download(out, (len(rows), 4))) == [le(r[col]) for r in rows]
    # MSM through the host-slice entry points
    ks, ss, _ = _msm_inputs(kats)
    dk = ctx.upload(cv.fr_to_mont(ks))
    for group, key in ((1, "g1"), (2, "g2")):
        bases = ctx.fixed_base(dk.ptr, len(ks), group, mont1(1))
        arr = bases.download()
        res = (ctx.multi_scalar_mul_g1 if group == 1 else ctx.multi_scalar_mul_g2)(arr, cv.fr_to_mont(ss))
        got = (cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine)(res)
        ser = O.g1_serialize_uncompressed if group == 1 else O.g2_serialize_uncompressed
        assert ser(got).hex() == kats["msm"][key]
    # the four transforms
    f = kats["fft"]
    v = cv.fr_to_mont([le(x) for x in f["input"]])
    for name, fn in (("fft", ctx.fft_in_place), ("ifft", ctx.ifft_in_place), ("coset_fft", ctx.coset_fft_in_place),
                     ("coset_ifft", ctx.coset_ifft_in_place)):
        assert cv.fr_from_mont(fn(v.copy(), 6)) == [le(x) for x in f[name]]
    # Groth16: setup with the reference's toxic waste, the reference's proof bytes
    g = kats["groth16_simple"]
    r1cs, z, td = _simple_circuit(g)
    dr = ctx.r1cs_upload(2, 2, csr(r1cs.a), csr(r1cs.b), csr(r1cs.c))
    tdm = td_mont(td)
    pk = ctx.groth16_setup(dr, *[tdm[i] for i in range(7)])
    assert ctx.create_proof(pk, dr, cv.fr_to_mont(z), mont1(le(g["r"])), mont1(le(g["s"]))).hex() == g["proof"]
    pk.free()
    # SHE ring product
    s = kats["she_mul"]
    n = s["n"]
    x = ctx.upload(cv.fq753_to_mont([le(v) for v in s["x"]])); y = ctx.upload(cv.fq753_to_mont([le(v) for v in s["y"]]))
    o = ctx.alloc(n * 96)
    ctx.encodedtext_mul_dev(x.ptr, y.ptr, o.ptr, n)
    assert cv.fq753_from_mont(ctx.download(o, (n, 12))) == [le(v) for v in s["xy"]]


def test_groth16_proving_key_bytes_oracle(kats):
    """ProvingKey::serialize / serialize_uncompressed of the reference (arkworks/groth16/src/data_structures.rs:133-151) against
    the oracle's framing (zkref.pk_serialize): field order and the u64 Vec prefixes are choices, not mathematics."""
    g = kats["groth16_simple"]
    if "pk" not in g:
        pytest.skip("ref_kats.json predates the proving-key dump: regenerate it with the current tools/ref_vectors/dump_kats.rs")
    r1cs, z, td = _simple_circuit(g)
    pk = O.ProvingKey(O.ProvingKeyScalars(r1cs, td))
    assert O.pk_serialize(pk, True).hex() == g["pk"]
    assert O.pk_serialize(pk, False).hex() == g["pk_uncompressed"]
    assert O.vk_serialize(pk, True).hex() == g["vk"]


def _marlin_pc_case(kats):
    m = kats.get("marlin_pc_commit")
    if m is None:
        pytest.skip("ref_kats.json predates the MarlinKZG10::commit dump: regenerate it with the current tools/ref_vectors/dump_kats.rs")
    return m


def test_marlin_pc_commit_rng_draw_order_oracle(kats):
    """MarlinKZG10::commit (poly-commit/src/marlin/marlin_pc/mod.rs:172-243) with Some(rng): per polynomial the blinding
    polynomial of the commitment (hiding bound 1: three coefficients), then -- for a degree-bounded one -- the blinding polynomial
    of the SHIFTED commitment; nothing for a non-hiding polynomial.  The reference's rng was a fresh test_rng(): replayed here,
    the draws must BE the dumped blinding coefficients, the generator must end where the reference's did, and commitment =
    MSM(powers, p) + MSM(powers_of_gamma_g, blind) (shifted: over shifted_powers from max_degree - bound on) must give the bytes."""
    m = _marlin_pc_case(kats)
    rng = FR.test_rng()
    powers = [g1_unc(h) for h in m["powers"]]
    shifted = [g1_unc(h) for h in m["shifted_powers"]]
    gamma = [g1_unc(h) for h in m["powers_of_gamma_g"]]
    bound = 8
    for p in m["polys"]:
        coeffs = [le(c) for c in p["coeffs"]]
        hiding = p["label"] in ("hb", "h")
        blind = [rng.next_fr() for _ in range(3)] if hiding else []
        assert blind == [le(c) for c in p["blind"]], p["label"]
        comm = O.msm_naive(powers, coeffs, O.FqOps)
        if blind:
            comm = O.g1_add(comm, O.msm_naive(gamma, blind, O.FqOps))
        want = O.g1_serialize(comm)
        if p["label"] == "hb":
            sblind = [rng.next_fr() for _ in range(3)]
            assert sblind == [le(c) for c in p["shifted_blind"]]
            # shifted_powers holds the powers from max_degree - (largest enforced bound) on: with one bound, from its start
            sc = O.g1_add(O.msm_naive(shifted[:len(coeffs)], coeffs, O.FqOps), O.msm_naive(gamma, sblind, O.FqOps))
            want += b"\x01" + O.g1_serialize(sc)                # Option<Commitment>: a presence byte, then the point
            assert len(shifted) == bound + 1
        else:
            assert p["shifted_blind"] is None
            want += b"\x00"
        assert want.hex() == p["commitment"], p["label"]
    assert rng.next_u64() == m["rng_next_u64_after"]


def test_mpc_element_layouts_are_expressible_and_match_the_composer(kats):
    """(round 6) how the reference's toolchain lays MpcField / MpcG1Affine out, dumped by pattern search (dump_kats.rs section 14): every
    field the zk_mpc_* layout descriptors need was found, and the layout the composer and the tests call `tagfirst` IS rustc's."""
    if "mpc_layouts" not in kats:
        pytest.skip("vectors from before round 6: no mpc_layouts section")
    import zk_mpc_amd.api as A
    by = {e["type"]: e for e in kats["mpc_layouts"]}
    for name, spdz in (("MpcField<Fr, AdditiveFieldShare<Fr>>", False), ("MpcField<Fr, SpdzFieldShare<Fr>>", True)):
        e = by[name]
        assert min(e["off_tag"], e["off_public"], e["off_share"]) >= 0 and (e["off_mac"] >= 0) == spdz, e
        ours = A.MpcFieldLayout(spdz, tag_last=e["off_tag"] > e["off_public"])
        assert (e["size"], e["off_public"], e["off_share"]) == (ours.stride, ours.off_public, ours.off_share), (e, vars(ours))
        if spdz:
            assert e["off_mac"] == ours.off_mac
        assert e["tag_public"] != e["tag_shared"]
    for name, group in (("MpcG1Affine<Bls12_377, AdditivePairingShare>", 1), ("MpcG2Affine<Bls12_377, AdditivePairingShare>", 2)):
        e = by[name]
        assert min(e["off_x"], e["off_y"], e["off_infinity"]) >= 0, e
        ours = A.mpc_group_layout(group, False, tag_last=e["off_x"] == 0)
        assert (e["size"], e["off_x"], e["off_y"], e["off_infinity"]) == (ours.point.stride, ours.point.off_x, ours.point.off_y, ours.point.off_infinity), e
