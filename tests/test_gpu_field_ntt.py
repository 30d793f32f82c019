"""GPU parity: Fr vector ops and NTT (through the C ABI) against the Python big-int oracle."""
import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv

pytestmark = pytest.mark.gpu


def rand_fr(rng, n):
    return [rng.fr() for _ in range(n)]


@pytest.mark.parametrize("n", [1, 7, 256, 1000])
def test_vec_ops(ctx, n):
    rng = O.Prng(100 + n)
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    a[0], b[0] = 0, O.R_MOD - 1
    if n > 2:
        a[1], b[1] = O.R_MOD - 1, O.R_MOD - 1
    da, db = ctx.upload(cv.fr_to_mont(a)), ctx.upload(cv.fr_to_mont(b))
    out = ctx.alloc(n * 32)
    for op, f in ((0, lambda x, y: x * y), (1, lambda x, y: x + y), (2, lambda x, y: x - y)):
        ctx.fr_vec_op_dev(op, da.ptr, db.ptr, out.ptr, n)
        got = cv.fr_from_mont(ctx.download(out, (n, 4)))
        assert got == [f(x, y) % O.R_MOD for x, y in zip(a, b)]
        raw = ctx.download(out, (n, 4))  # fully reduced Montgomery residues
        assert all(v < O.R_MOD for v in cv._limbs_to_ints(raw))


def test_batch_product_in_place_host(ctx):
    rng = O.Prng(5)
    a, b = rand_fr(rng, 300), rand_fr(rng, 300)
    am, bm = cv.fr_to_mont(a), cv.fr_to_mont(b)
    ctx.batch_product_in_place(am, bm)
    assert cv.fr_from_mont(am) == [x * y % O.R_MOD for x, y in zip(a, b)]
    ctx.batch_product_in_place(am[:0], bm[:0])  # empty input


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15, 17])
def test_ntt_variants(ctx, log_n):
    rng = O.Prng(200 + log_n)
    N = 1 << log_n
    v = rand_fr(rng, N)
    dom = O.Domain(N)
    vm = cv.fr_to_mont(v)
    assert cv.fr_from_mont(ctx.fft_in_place(vm, log_n)) == dom.fft(v)
    assert cv.fr_from_mont(ctx.ifft_in_place(vm, log_n)) == dom.ifft(v)
    assert cv.fr_from_mont(ctx.coset_fft_in_place(vm, log_n)) == dom.coset_fft(v)
    assert cv.fr_from_mont(ctx.coset_ifft_in_place(vm, log_n)) == dom.coset_ifft(v)


def test_ntt_matches_arkworks_schedule(ctx):
    """The oracle's literal io/oi restatement (radix2/fft.rs) and the device agree."""
    rng = O.Prng(31)
    v = rand_fr(rng, 64)
    dom = O.Domain(64)
    assert cv.fr_from_mont(ctx.fft_in_place(cv.fr_to_mont(v), 6)) == O.fft_arkworks_io_oi(v, dom, False)
    assert cv.fr_from_mont(ctx.ifft_in_place(cv.fr_to_mont(v), 6)) == O.fft_arkworks_io_oi(v, dom, True)


def test_ntt_zero_padding(ctx):
    """fft_in_place resizes the input with zeros (radix2/mod.rs:98-101)."""
    rng = O.Prng(32)
    v = rand_fr(rng, 37)
    dom = O.Domain(64)
    assert cv.fr_from_mont(ctx.fft_in_place(cv.fr_to_mont(v), 6)) == dom.fft(v)


@pytest.mark.parametrize("log_n", [16, 18, 19, 20, 21, 22, 23, 24])
def test_ntt_large_properties(ctx, log_n):
    """Size-independent properties at the benchmark sizes: round trips, linearity, a spot DFT value.  21 ... 24 are the
    three-pass sizes (1 024-element tiles); 2^24 = 4|H| of the Marlin leg at 2^22 constraints (BASELINE config 5)."""
    N = 1 << log_n
    rs = np.random.RandomState(log_n)
    a = rs.randint(0, 1 << 62, size=(N, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)   # < 2^252 < r: valid (arbitrary) Montgomery residues
    b = rs.randint(0, 1 << 62, size=(N, 4), dtype=np.uint64)
    b[:, 3] &= np.uint64((1 << 60) - 1)
    da, db, dc = ctx.upload(a), ctx.upload(b), ctx.alloc(N * 32)
    ctx.fr_vec_op_dev(1, da.ptr, db.ptr, dc.ptr, N)           # c = a + b
    for coset in (False, True):
        for buf in (da, db, dc):
            ctx.ntt_dev(buf.ptr, log_n, False, coset)
        # linearity: F(a+b) == F(a) + F(b)
        t = ctx.alloc(N * 32)
        ctx.fr_vec_op_dev(1, da.ptr, db.ptr, t.ptr, N)
        assert np.array_equal(ctx.download(t, (N, 4)), ctx.download(dc, (N, 4)))
        # spot check one output against the definition on a sparse probe is done below; round trip:
        for buf in (da, db, dc):
            ctx.ntt_dev(buf.ptr, log_n, True, coset)
        assert np.array_equal(ctx.download(da, (N, 4)), a)
        assert np.array_equal(ctx.download(db, (N, 4)), b)
        t.free()
    # delta at position j -> F[k] = w^(jk): checks twiddles / ordering at full size
    j = 12345 % N
    e = np.zeros((N, 4), dtype=np.uint64)
    e[j] = cv.fr_to_mont([1])[0]
    de = ctx.upload(e)
    ctx.ntt_dev(de.ptr, log_n, False, False)
    out = ctx.download(de, (N, 4))
    dom = O.Domain(N)
    for k in (0, 1, 2, N // 2 + 3, N - 1):
        assert cv.fr_from_mont(out[k:k + 1]) == [pow(dom.group_gen, j * k, O.R_MOD)]
