"""Shared test helpers (oracle <-> ABI array conversions, fixtures)."""
import json
import os

import numpy as np

import zkref as O
import zk_mpc_amd.convert as cv

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def ih(x):
    return int(x, 16)


def g1_from_json(p):
    return None if p is None else (ih(p[0]), ih(p[1]))


def g2_from_json(p):
    return None if p is None else ((ih(p[0][0]), ih(p[0][1])), (ih(p[1][0]), ih(p[1][1])))


def csr(rows):
    rp, col, coeff = [0], [], []
    for r in rows:
        for c, i in r:
            col.append(i)
            coeff.append(c)
        rp.append(len(col))
    return (np.array(rp, dtype=np.uint32), np.array(col, dtype=np.uint32),
            cv.fr_to_mont(coeff) if coeff else np.zeros((0, 4), dtype=np.uint64))


def r1cs_from_json(j):
    rows = lambda m: [[(ih(c), i) for c, i in row] for row in m]
    return O.R1CS(j["num_instance"], j["num_witness"], rows(j["a"]), rows(j["b"]), rows(j["c"]))


def trapdoor_from_json(t):
    return O.Trapdoor(*[ih(t[k]) for k in ("alpha", "beta", "gamma", "delta", "tau", "g1_k", "g2_k")])


def td_mont(td):
    return cv.fr_to_mont([td.alpha, td.beta, td.gamma, td.delta, td.tau, td.g1_k, td.g2_k])


def mont1(v):
    return cv.fr_to_mont([v])[0]


def circuit_system(spec, seed):
    """A circuit-shaped R1CS with its assignment (tools/synth_r1cs.py: 3-5 / 1-2 / 2-4 terms per row of A / B / C, non-unit
    coefficients, several public inputs, |K| = 4 |H| under Marlin).  spec: log2 |H| of the padded system, or (rows, n_pub, n_free)."""
    import synth_r1cs as S
    ni, nw, a, b, c, z = S.sized_for_domain(spec, seed) if isinstance(spec, int) else S.circuit_shaped(*spec, seed)
    return O.R1CS(ni, nw, a, b, c), z


def marlin_test_system(n, rng):
    """n: int -> the mul-chain of that many constraints; "dense<k>" -> a circuit-shaped system filling |H| = 2^k;
    "tiny<r>" -> r circuit-shaped rows with 3 public inputs and 2 free witnesses."""
    if isinstance(n, str) and n.startswith("dense"):
        return circuit_system(int(n[5:]), rng.u64() & 0xffffffff)
    if isinstance(n, str) and n.startswith("tiny"):
        return circuit_system((int(n[4:]), 3, 2), rng.u64() & 0xffffffff)
    return O.mul_chain_r1cs(n, rng.fr(), rng.fr())


def free_port():
    """A rendezvous port on 127.0.0.1 below the kernel's range for outgoing connections (32768 .. 60999): a port handed out by
    bind(0) can be taken again -- as the SOURCE port of some connection -- between the probe and the listener's bind (seen once:
    EADDRINUSE in torch's TCPStore)."""
    import random
    import socket
    for _ in range(64):
        cand = random.randint(20000, 32000)
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", cand))
                return cand
            except OSError:
                continue
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
