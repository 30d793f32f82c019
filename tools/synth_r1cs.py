"""Circuit-shaped synthetic R1CS (test and bench INPUT generator; pure Python big integers, no product code, no oracle code).

The mul-chain of SURVEY.md 8(d) has one term per row, every coefficient 1 and a single public input, so under Marlin
|K| = |H| and `val` is constant -- the easiest possible index.  Real circuits (docs/benchmark.md:40-58 of the reference: range
checks, Pedersen / Poseidon rounds) have several terms per row, non-unit coefficients, several public inputs and matrices of
different density; AHPForR1CS::index sizes K by the densest matrix BEFORE balancing (arkworks/marlin/src/ahp/indexer.rs:138-141,
constraint_systems.rs:41-49), so |K| = 2 - 4 |H| there.  This generator makes such a system together with a satisfying assignment:

  variables    [1, p_1 .. p_npub] (instance) ++ [f_1 .. f_nfree, o_0 .. o_{rows-1}] (witness)
  row i        <A_i, z> * <B_i, z> = <C_i, z>;  A_i: 3-5 terms, B_i: 1-2 terms, C_i: c0 * o_i + 1-3 more terms, every term over
               variables that come BEFORE o_i (the constant, the inputs, free witnesses, earlier outputs -- two thirds of them
               recent ones, as a circuit's wires are), so the rows are solved in order for o_i
  coefficients half of them 1, a quarter small signed integers, a quarter uniform field elements; c0 in {1, -1, 2, 3, 5}

With rows = ~0.87 |H| the densest matrix (A) holds ~3.5 |H| entries: |K| = 4 |H|, the interpolation domain of round 3 16 |H|.
"""
from __future__ import annotations

import hashlib

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041      # BLS12-377 Fr


class _Rng:
    """SHA-256 in counter mode: reproducible everywhere, independent of numpy's generator versions."""

    def __init__(self, seed: int):
        self.key = b"synth-r1cs" + int(seed).to_bytes(8, "little")
        self.ctr = 0
        self.buf = b""

    def _bytes(self, n: int) -> bytes:
        while len(self.buf) < n:
            self.buf += hashlib.sha256(self.key + self.ctr.to_bytes(8, "little")).digest()
            self.ctr += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def below(self, n: int) -> int:
        return int.from_bytes(self._bytes(8), "little") % n

    def fr(self) -> int:
        return int.from_bytes(self._bytes(40), "little") % R_MOD


_C0 = (1, R_MOD - 1, 2, 3, 5)
_C0_INV = {c: pow(c, -1, R_MOD) for c in _C0}


def circuit_shaped(rows: int, n_pub: int, n_free: int, seed: int):
    """(num_instance, num_witness, a_rows, b_rows, c_rows, full_assignment): rows of (coefficient, variable index) pairs with
    canonical integer coefficients, the assignment as canonical integers (instance first)."""
    rng = _Rng(seed)
    ni = 1 + n_pub
    z = [1] + [rng.fr() for _ in range(n_pub)] + [rng.fr() for _ in range(n_free)]
    first_out = len(z)

    def coeff():
        k = rng.below(4)
        if k < 2:
            return 1
        if k == 2:
            v = 1 + rng.below(16)
            return v if rng.below(2) else R_MOD - v
        return rng.fr() or 1

    def terms(count, limit):
        out, seen = [], set()
        for _ in range(count):
            if rng.below(3) and limit > 16:
                j = limit - 1 - rng.below(16)           # a recent wire
            else:
                j = rng.below(limit)
            if j in seen:
                continue
            seen.add(j)
            out.append((coeff(), j))
        return out

    a_rows, b_rows, c_rows = [], [], []
    for i in range(rows):
        limit = first_out + i
        ra, rb = terms(3 + rng.below(3), limit), terms(1 + rng.below(2), limit)
        rc = terms(1 + rng.below(3), limit)
        av = sum(c * z[j] for c, j in ra) % R_MOD
        bv = sum(c * z[j] for c, j in rb) % R_MOD
        cv_ = sum(c * z[j] for c, j in rc) % R_MOD
        c0 = _C0[rng.below(len(_C0))]
        z.append((av * bv - cv_) * _C0_INV[c0] % R_MOD)
        rc.append((c0, limit))
        a_rows.append(ra); b_rows.append(rb); c_rows.append(rc)
    return ni, len(z) - ni, a_rows, b_rows, c_rows, z


def sized_for_domain(log_h: int, seed: int, n_pub: int = 7):
    """A system whose padded square form fills a domain H of 2^log_h with |K| = 4 |H| (log_h >= 5): 7 public inputs (8 instance
    variables: already a power of two), |H| / 8 free witnesses, three constraints short of square (so that the padding rule of
    constraint_systems.rs:90-113 adds dummy constraints)."""
    h = 1 << log_h
    n_free = h // 8
    rows = h - (1 + n_pub) - n_free - 3
    return circuit_shaped(rows, n_pub, n_free, seed)


def is_satisfied(a_rows, b_rows, c_rows, z) -> bool:
    ev = lambda row: sum(c * z[j] for c, j in row) % R_MOD
    return all(ev(a) * ev(b) % R_MOD == ev(c) for a, b, c in zip(a_rows, b_rows, c_rows))
