"""GPU parity: SHE ring arithmetic over the MNT4-753 base field against the oracle.

Mirrors the reference's own tests: src/she/encodedtext.rs:139-160, src/she/ciphertext.rs:151-181 and the
encode / encrypt / multiply / decrypt / decode acceptance test of src/she.rs:122-208."""
import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv
from zk_mpc_amd import _lib

pytestmark = pytest.mark.gpu


def rq(rng):
    return (rng.fr() * rng.fr() * rng.fr() + rng.fr()) % O.Q753


def upq(ctx, vals):
    return ctx.upload(cv.fq753_to_mont(vals))


def downq(ctx, buf, n):
    return cv.fq753_from_mont(ctx.download(buf, (n, 12)))


def test_fq753_vector_ops(ctx):
    rng = O.Prng(700)
    n = 300
    a, b = [rq(rng) for _ in range(n)], [rq(rng) for _ in range(n)]
    a[0], b[0], a[1], b[1], a[2], b[2] = 0, 0, O.Q753 - 1, O.Q753 - 1, O.Q753 - 1, 1
    da, db, out = upq(ctx, a), upq(ctx, b), ctx.alloc(n * 96)
    for op, f in ((_lib.OP_ADD, lambda x, y: (x + y) % O.Q753), (_lib.OP_SUB, lambda x, y: (x - y) % O.Q753),
                  (_lib.OP_MUL, lambda x, y: x * y % O.Q753)):
        ctx.she_vec_op_dev(op, da.ptr, db.ptr, out.ptr, n)
        assert downq(ctx, out, n) == [f(x, y) for x, y in zip(a, b)]
    ctx.she_vec_op_dev(_lib.OP_NEG, da.ptr, None, out.ptr, n)
    assert downq(ctx, out, n) == O.texts_neg(a)
    k = rq(rng)
    ctx.she_vec_scale_dev(da.ptr, cv.fq753_to_mont([k])[0], out.ptr, n)
    assert downq(ctx, out, n) == O.encodedtext_scale(a, k)


@pytest.mark.parametrize("n,batch", [(1, 3), (2, 2), (3, 2), (4, 1), (4, 5), (8, 3), (64, 1), (64, 17), (100, 2), (512, 3), (1024, 1)])
def test_encodedtext_mul(ctx, n, batch):
    rng = O.Prng(710 + n)
    a = [[rq(rng) for _ in range(n)] for _ in range(batch)]
    b = [[rq(rng) for _ in range(n)] for _ in range(batch)]
    da, db, out = upq(ctx, sum(a, [])), upq(ctx, sum(b, [])), ctx.alloc(batch * n * 96)
    ctx.encodedtext_mul_dev(da.ptr, db.ptr, out.ptr, n, batch)
    got = downq(ctx, out, batch * n)
    for i in range(batch):
        assert got[i * n:(i + 1) * n] == O.encodedtext_mul(a[i], b[i]), (n, i)


def test_encodedtext_mul_reference_vectors(ctx):
    """The hand-checkable cases of the reference's unit tests (src/she/encodedtext.rs:139-146: (1+2X+3X^2) mod (X+2) = 9)
    restated for the ring product: (1 + 2X)(2 + 3X) mod X^2 + 1 = -4 + 7X."""
    da, db, out = upq(ctx, [1, 2]), upq(ctx, [2, 3]), ctx.alloc(2 * 96)
    ctx.encodedtext_mul_dev(da.ptr, db.ptr, out.ptr, 2, 1)
    assert downq(ctx, out, 2) == [O.Q753 - 4, 7]


@pytest.mark.parametrize("n", [2048, 16384])
def test_encodedtext_mul_large(ctx, n):
    """Above one LDS tile the top levels run in global memory.  Checked through ring identities the oracle affords
    at this size: multiplication by X^k (a signed rotation) and by a sparse polynomial."""
    rng = O.Prng(720 + n)
    a = [rq(rng) for _ in range(n)]
    k = 5
    xk = [0] * n
    xk[k] = 1
    sparse = [0] * n
    sparse[0], sparse[1], sparse[n - 1] = 3, O.Q753 - 2, 7
    da, dxk, dsp, out = upq(ctx, a), upq(ctx, xk), upq(ctx, sparse), ctx.alloc(n * 96)
    ctx.encodedtext_mul_dev(da.ptr, dxk.ptr, out.ptr, n, 1)
    assert downq(ctx, out, n) == [(-x) % O.Q753 for x in a[n - k:]] + a[:n - k]
    ctx.encodedtext_mul_dev(da.ptr, dsp.ptr, out.ptr, n, 1)
    want = [(3 * a[i] - 2 * (a[i - 1] if i else -a[n - 1]) + 7 * (-a[i + 1] if i + 1 < n else a[0])) % O.Q753 for i in range(n)]
    # a * 7 X^(n-1): coefficient i gets -7 a[i+1] (wrap) for i < n-1, and 7 a[0] at i = n-1
    assert downq(ctx, out, n) == want
    # commutativity on dense random operands
    b = [rq(rng) for _ in range(n)]
    db, out2 = upq(ctx, b), ctx.alloc(n * 96)
    ctx.encodedtext_mul_dev(da.ptr, db.ptr, out.ptr, n, 1)
    ctx.encodedtext_mul_dev(db.ptr, da.ptr, out2.ptr, n, 1)
    g1 = downq(ctx, out, n)
    assert g1 == downq(ctx, out2, n)
    # and one coefficient of the dense product against the definition
    kk = 1234
    assert g1[kk] == (sum(a[i] * b[kk - i] for i in range(kk + 1)) - sum(a[i] * b[n + kk - i] for i in range(kk + 1, n))) % O.Q753


@pytest.mark.parametrize("n,batch", [(2, 1), (3, 2), (8, 4), (64, 3), (256, 2)])
def test_ciphertext_mul_encrypt_decrypt(ctx, n, batch):
    rng = O.Prng(730 + n)
    small = lambda k: [rng.u64() % 11 for _ in range(k)]
    sk = small(n)
    pk_a, pk_b = O.public_key_gen(sk, [rq(rng) for _ in range(n)], small(n))
    e = [[rq(rng) for _ in range(n)] for _ in range(batch)]
    r = [small(3 * n) for _ in range(batch)]
    dsk, dpa, dpb = upq(ctx, sk), upq(ctx, pk_a), upq(ctx, pk_b)
    de, dr = upq(ctx, sum(e, [])), upq(ctx, sum(r, []))
    ct = ctx.alloc(batch * 3 * n * 96)
    ctx.ciphertext_encrypt_from_dev(de.ptr, dpa.ptr, dpb.ptr, dr.ptr, cv.fq753_to_mont([O.R_MOD])[0], ct.ptr, n, batch)
    got = downq(ctx, ct, batch * 3 * n)
    want_ct = [O.ciphertext_encrypt_from(e[i], pk_a, pk_b, r[i]) for i in range(batch)]
    for i in range(batch):
        assert got[i * 3 * n:(i + 1) * 3 * n] == sum(want_ct[i], []), i
    # decrypt gives e back (noise-free comparison against the oracle's decrypt)
    dec = ctx.alloc(batch * n * 96)
    ctx.ciphertext_decrypt_dev(ct.ptr, dsk.ptr, dec.ptr, n, batch)
    gd = downq(ctx, dec, batch * n)
    for i in range(batch):
        assert gd[i * n:(i + 1) * n] == O.ciphertext_decrypt(want_ct[i], sk)
    # ciphertext product: ct[i] * ct[(i+1) % batch]
    rot = sum([sum(want_ct[(i + 1) % batch], []) for i in range(batch)], [])
    drot, prod = upq(ctx, rot), ctx.alloc(batch * 3 * n * 96)
    ctx.ciphertext_mul_dev(ct.ptr, drot.ptr, prod.ptr, n, batch)
    gp = downq(ctx, prod, batch * 3 * n)
    for i in range(batch):
        assert gp[i * 3 * n:(i + 1) * 3 * n] == sum(O.ciphertext_mul(want_ct[i], want_ct[(i + 1) % batch]), []), i
    # decrypt of the degree-2 ciphertext (uses c2)
    ctx.ciphertext_decrypt_dev(prod.ptr, dsk.ptr, dec.ptr, n, batch)
    gd = downq(ctx, dec, batch * n)
    for i in range(batch):
        assert gd[i * n:(i + 1) * n] == O.ciphertext_decrypt(O.ciphertext_mul(want_ct[i], want_ct[(i + 1) % batch]), sk)


@pytest.mark.parametrize("n,batch", [(1, 2), (2, 1), (8, 3), (64, 2)])
def test_encode_decode(ctx, n, batch):
    rng = O.Prng(740 + n)
    pt = [[rng.fr() for _ in range(n)] for _ in range(batch)]
    dpt = ctx.upload(cv.fr_to_mont(sum(pt, [])))
    enc, back = ctx.alloc(batch * n * 96), ctx.alloc(batch * n * 32)
    ctx.plaintexts_encode_dev(dpt.ptr, enc.ptr, n, batch)
    ge = downq(ctx, enc, batch * n)
    for i in range(batch):
        assert ge[i * n:(i + 1) * n] == O.plaintexts_encode(pt[i]), i
    ctx.encodedtext_decode_dev(enc.ptr, back.ptr, n, batch)
    assert cv.fr_from_mont(ctx.download(back, (batch * n, 4))) == sum(pt, [])
    # decode of arbitrary Fq coefficients (exercises the centred lift of the upper half)
    anyq = [rq(rng) for _ in range(batch * n)]
    anyq[0] = O.Q753 - 1
    dany = upq(ctx, anyq)
    ctx.encodedtext_decode_dev(dany.ptr, back.ptr, n, batch)
    got = cv.fr_from_mont(ctx.download(back, (batch * n, 4)))
    for i in range(batch):
        assert got[i * n:(i + 1) * n] == O.encodedtext_decode(anyq[i * n:(i + 1) * n])


def test_she_acceptance_roundtrip(ctx):
    """src/she.rs:122-208 on the device: Plain -> Encoded -> Cipher -> (x, +) -> Encoded -> Plain."""
    rng = O.Prng(750)
    n = 64
    small = lambda k: [rng.u64() % 7 for _ in range(k)]
    sk = small(n)
    pk_a, pk_b = O.public_key_gen(sk, [rq(rng) for _ in range(n)], small(n))
    pts = [[rng.fr() for _ in range(n)] for _ in range(3)]
    dpt = ctx.upload(cv.fr_to_mont(sum(pts, [])))
    enc = ctx.alloc(3 * n * 96)
    ctx.plaintexts_encode_dev(dpt.ptr, enc.ptr, n, 3)
    dr = upq(ctx, small(9 * n))
    ct = ctx.alloc(9 * n * 96)
    dpa, dpb, dsk = upq(ctx, pk_a), upq(ctx, pk_b), upq(ctx, sk)
    ctx.ciphertext_encrypt_from_dev(enc.ptr, dpa.ptr, dpb.ptr, dr.ptr, cv.fq753_to_mont([O.R_MOD])[0], ct.ptr, n, 3)
    prod, dec, out = ctx.alloc(3 * n * 96), ctx.alloc(n * 96), ctx.alloc(n * 32)
    ctx.ciphertext_mul_dev(ct.ptr, ct.ptr + 3 * n * 96, prod.ptr, n, 1)                 # ct * ct_2
    ctx.she_vec_op_dev(_lib.OP_ADD, prod.ptr, ct.ptr + 6 * n * 96, prod.ptr, 3 * n)      # + ct_3
    ctx.ciphertext_decrypt_dev(prod.ptr, dsk.ptr, dec.ptr, n, 1)
    ctx.encodedtext_decode_dev(dec.ptr, out.ptr, n, 1)
    want = [(x * y + z) % O.R_MOD for x, y, z in zip(*pts)]
    assert cv.fr_from_mont(ctx.download(out, (n, 4))) == want
