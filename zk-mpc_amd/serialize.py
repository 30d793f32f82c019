"""arkworks wire formats of the Groth16 key material and of a KZG10 SRS: `CanonicalSerialize` layouts of
VerifyingKey / ProvingKey (arkworks/groth16/src/data_structures.rs:43-58,133-151) and UniversalParams
(poly-commit/src/kzg10/data_structures.rs:40-80), so that keys made or held by this library interchange with the Rust
prover / verifier.  Framing and point bytes are the library's (zk_pk_serialize, zk_pk_deserialize, zk_kzg_srs_*): this
module is the ctypes mirror.

Proofs are already emitted in wire form by `create_proof` (192 B: a | b | c compressed)."""
from __future__ import annotations

import numpy as np

from .api import Bases, Context, ProvingKey

_G1_QUERIES = ("a_query", "b_g1_query", "h_query", "l_query", "gamma_abc_g1")


def _buf(n: int):
    import ctypes as C
    return (C.c_uint8 * n)()


def verifying_key_bytes(ctx: Context, pk: ProvingKey, compressed: bool = True) -> bytes:
    """VerifyingKey::serialize[_uncompressed] (zk_vk_serialize): alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | gamma_abc_g1 (Vec)."""
    n = ctx.lib.zk_vk_serialized_size(pk.h, int(compressed))
    out = _buf(n)
    ctx._ck(ctx.lib.zk_vk_serialize(ctx.h, pk.h, int(compressed), out, n))
    return bytes(out)


def proving_key_bytes(ctx: Context, pk: ProvingKey, compressed: bool = True) -> bytes:
    """ProvingKey::serialize[_uncompressed] (zk_pk_serialize): vk | beta_g1 | delta_g1 | a_query | b_g1_query | b_g2_query |
    h_query | l_query."""
    n = ctx.lib.zk_pk_serialized_size(pk.h, int(compressed))
    out = _buf(n)
    ctx._ck(ctx.lib.zk_pk_serialize(ctx.h, pk.h, int(compressed), out, n))
    return bytes(out)


def proving_key_from_bytes(ctx: Context, data: bytes, compressed: bool = False):
    """ProvingKey::deserialize (compressed: one square root per point on the device) or deserialize_uncompressed through
    zk_pk_deserialize: the tables go straight to the device and the key is resident like one from zk_pk_upload.
    Returns (ProvingKey, gamma_g2, gamma_abc_g1): the prover does not use the last two, the verifier does."""
    import ctypes as C
    h = C.c_void_p()
    ctx._ck(ctx.lib.zk_pk_deserialize(ctx.h, data, len(data), int(compressed), C.byref(h)))
    pk = ProvingKey(ctx, h)
    return pk, pk.vk_g2(2), pk.download("gamma_abc_g1")


def kzg_srs_bytes(ctx: Context, powers_g: Bases, powers_gamma_g: Bases, h24, beta_h24, compressed: bool = True) -> bytes:
    """UniversalParams::serialize (poly-commit/src/kzg10/data_structures.rs:40-80; src/marlin.rs:371-376 writes it to a file):
    powers_of_g (Vec) | powers_of_gamma_g (BTreeMap 0..n-1) | h | beta_h | neg_powers_of_h (empty)."""
    n = ctx.lib.zk_kzg_srs_serialized_size(len(powers_g), len(powers_gamma_g), int(compressed))
    out = _buf(n)
    h24 = np.ascontiguousarray(h24, dtype=np.uint64)
    beta_h24 = np.ascontiguousarray(beta_h24, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_kzg_srs_serialize(ctx.h, powers_g.h, powers_gamma_g.h, h24.ctypes.data, beta_h24.ctypes.data, int(compressed), out, n))
    return bytes(out)


def kzg_srs_from_bytes(ctx: Context, data: bytes, compressed: bool = True):
    """-> (powers_g Bases, powers_gamma_g Bases, h (24,) uint64, beta_h (24,) uint64)."""
    import ctypes as C
    pg, pgg = C.c_void_p(), C.c_void_p()
    h, bh = np.zeros(24, dtype=np.uint64), np.zeros(24, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_kzg_srs_deserialize(ctx.h, data, len(data), int(compressed), C.byref(pg), C.byref(pgg), h.ctypes.data, bh.ctypes.data))
    return Bases(ctx, pg, 1), Bases(ctx, pgg, 1), h, bh
