#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/*.json from the Python big-int oracle (oracle/zkref.py).

The reference ships no known-answer vectors for this path and cannot be run here, so
these fixtures are outputs of the oracle itself: they pin the oracle (and the C restatement, and the
HIP path) against regressions, they are NOT recorded outputs of the Rust code ("parity unpinned"
in that sense).  Inputs are seeded with oracle.Prng; values are hex strings.
Run:  python3 oracle/gen_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import zkref as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def hx(v):
    return hex(v)


def pt1(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def pt2(p):
    return None if p is None else [[hx(p[0][0]), hx(p[0][1])], [hx(p[1][0]), hx(p[1][1])]]


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = O.Prng(20261002)
    # field + NTT vectors
    a = [rng.fr() for _ in range(8)]
    b = [rng.fr() for _ in range(8)]
    dom = O.Domain(8)
    field = {
        "seed": 20261002,
        "fr_a": [hx(x) for x in a], "fr_b": [hx(x) for x in b],
        "fr_mul": [hx(x * y % O.R_MOD) for x, y in zip(a, b)],
        "fr_mont_a": [hx(O.fr_to_mont(x)) for x in a],
        "fft8": [hx(x) for x in dom.fft(a)], "ifft8": [hx(x) for x in dom.ifft(a)],
        "coset_fft8": [hx(x) for x in dom.coset_fft(a)], "coset_ifft8": [hx(x) for x in dom.coset_ifft(a)],
        "group_gen8": hx(dom.group_gen),
    }
    json.dump(field, open(os.path.join(OUT, "field_ntt.json"), "w"), indent=1)
    # MSM vectors
    ks = [rng.fr() for _ in range(6)]
    g1b = [O.g1_mul(O.G1_GEN, k) for k in ks]
    g2b = [O.g2_mul(O.G2_GEN, k) for k in ks[:4]]
    sc = [rng.fr() for _ in range(6)]
    sc[1], sc[2] = 0, 1
    msm = {
        "base_scalars": [hx(k) for k in ks], "g1_bases": [pt1(p) for p in g1b], "g2_bases": [pt2(p) for p in g2b],
        "scalars": [hx(s) for s in sc],
        "msm_g1": pt1(O.msm_pippenger(g1b, sc, O.FqOps)), "msm_g2": pt2(O.msm_pippenger(g2b, sc[:4], O.Fq2Ops)),
        "g1_compressed": [O.g1_serialize(p).hex() for p in g1b] + [O.g1_serialize(None).hex()],
        "g2_compressed": [O.g2_serialize(p).hex() for p in g2b] + [O.g2_serialize(None).hex()],
    }
    json.dump(msm, open(os.path.join(OUT, "msm.json"), "w"), indent=1)
    # Groth16: BASELINE config 1 (MySimpleCircuit: a*b=c six times, domain 8) and a 5-constraint mul chain
    out = {}
    for name, (r1cs, z) in {
        "my_simple_circuit": (lambda a_, b_: (O.R1CS(2, 2, [[(1, 2)]] * 6, [[(1, 3)]] * 6, [[(1, 1)]] * 6),
                                              [1, a_ * b_ % O.R_MOD, a_, b_]))(rng.fr(), rng.fr()),
        "mul_chain_5": O.mul_chain_r1cs(5, rng.fr(), rng.fr()),
    }.items():
        td = O.Trapdoor(rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr())
        pks = O.ProvingKeyScalars(r1cs, td)
        pk = O.ProvingKey(pks)
        r, s = rng.fr(), rng.fr()
        proof = O.create_proof(r1cs, pk, z, r, s)
        assert proof == O.predict_proof(r1cs, pks, z, r, s)
        assert O.verify_proof(pk, proof, z[1:r1cs.num_instance])
        out[name] = {
            "num_instance": r1cs.num_instance, "num_witness": r1cs.num_witness,
            "a": [[[hx(c), i] for c, i in row] for row in r1cs.a],
            "b": [[[hx(c), i] for c, i in row] for row in r1cs.b],
            "c": [[[hx(c), i] for c, i in row] for row in r1cs.c],
            "z": [hx(v) for v in z],
            "trapdoor": {k: hx(getattr(td, k)) for k in ("alpha", "beta", "gamma", "delta", "tau", "g1_k", "g2_k")},
            "r": hx(r), "s": hx(s),
            "h": [hx(v) for v in O.witness_map(r1cs, z)],
            "pk": {"alpha_g1": pt1(pk.alpha_g1), "beta_g1": pt1(pk.beta_g1), "delta_g1": pt1(pk.delta_g1),
                   "beta_g2": pt2(pk.beta_g2), "delta_g2": pt2(pk.delta_g2), "gamma_g2": pt2(pk.gamma_g2),
                   "a_query": [pt1(p) for p in pk.a_query], "b_g1_query": [pt1(p) for p in pk.b_g1_query],
                   "b_g2_query": [pt2(p) for p in pk.b_g2_query], "h_query": [pt1(p) for p in pk.h_query],
                   "l_query": [pt1(p) for p in pk.l_query], "gamma_abc_g1": [pt1(p) for p in pk.gamma_abc_g1]},
            "proof": O.proof_serialize(*proof).hex(),
        }
    json.dump(out, open(os.path.join(OUT, "groth16.json"), "w"), indent=1)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
