"""arkworks wire formats of the Groth16 key material: `CanonicalSerialize` layouts of
VerifyingKey / ProvingKey (arkworks/groth16/src/data_structures.rs:43-58,133-151), so that keys made or held by
this library interchange with the Rust prover / verifier.  Point bytes come from the device (`zk_bases_serialize`);
this module only orders the fields and writes the `Vec` length prefixes (serialize/src/lib.rs:263-272).

Proofs are already emitted in wire form by `create_proof` (192 B: a | b | c compressed)."""
from __future__ import annotations

import numpy as np

from .api import Bases, Context, ProvingKey

_G1_QUERIES = ("a_query", "b_g1_query", "h_query", "l_query", "gamma_abc_g1")


def _points(ctx: Context, arr: np.ndarray, group: int, compressed: bool) -> bytes:
    b = ctx.bases_upload(np.asarray(arr, dtype=np.uint64).reshape(-1, 12 if group == 1 else 24), group)
    try:
        return b.serialize(compressed)
    finally:
        b.free()


def _vec(bases: Bases, compressed: bool) -> bytes:
    return len(bases).to_bytes(8, "little") + bases.serialize(compressed)


def verifying_key_bytes(ctx: Context, pk: ProvingKey, compressed: bool = True) -> bytes:
    """alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | gamma_abc_g1 (Vec)."""
    return (_points(ctx, pk.vk_g1(0), 1, compressed)
            + _points(ctx, np.stack([pk.vk_g2(0), pk.vk_g2(2), pk.vk_g2(1)]), 2, compressed)
            + _vec(pk.query_bases("gamma_abc_g1"), compressed))


def proving_key_bytes(ctx: Context, pk: ProvingKey, compressed: bool = True) -> bytes:
    """vk | beta_g1 | delta_g1 | a_query | b_g1_query | b_g2_query | h_query | l_query."""
    out = [verifying_key_bytes(ctx, pk, compressed), _points(ctx, np.stack([pk.vk_g1(1), pk.vk_g1(2)]), 1, compressed)]
    for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
        out.append(_vec(pk.query_bases(name), compressed))
    return b"".join(out)


class _Reader:
    def __init__(self, data: bytes):
        self.data, self.pos = memoryview(data), 0

    def take(self, n: int) -> bytes:
        if self.pos + n > len(self.data):
            raise ValueError("truncated key")
        out = self.data[self.pos:self.pos + n]
        self.pos += n
        return bytes(out)

    def u64(self) -> int:
        return int.from_bytes(self.take(8), "little")


def proving_key_from_bytes(ctx: Context, data: bytes, compressed: bool = False):
    """Load a ProvingKey written with `serialize_uncompressed` or, with compressed=True, with `serialize` (one square
    root per point on the device).  Returns (ProvingKey, gamma_g2, gamma_abc_g1): the prover does not use the last two,
    the verifier does."""
    r = _Reader(data)

    def pts(n: int, group: int) -> np.ndarray:
        raw = r.take(n * (48 if group == 1 else 96) * (1 if compressed else 2))
        b = (ctx.bases_deserialize_compressed if compressed else ctx.bases_deserialize_uncompressed)(raw, n, group)
        try:
            return b.download()
        finally:
            b.free()

    alpha_g1 = pts(1, 1)[0]
    beta_g2, gamma_g2, delta_g2 = pts(3, 2)
    gamma_abc = pts(r.u64(), 1)
    beta_g1, delta_g1 = pts(2, 1)
    a_query = pts(r.u64(), 1)
    b_g1_query = pts(r.u64(), 1)
    b_g2_query = pts(r.u64(), 2)
    h_query = pts(r.u64(), 1)
    l_query = pts(r.u64(), 1)
    if r.pos != len(r.data):
        raise ValueError("trailing bytes after the proving key")
    pk = ctx.pk_upload(alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_query, b_g1_query, b_g2_query, h_query, l_query)
    return pk, gamma_g2, gamma_abc
