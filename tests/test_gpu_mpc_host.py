"""The trait surface for the collaborative element types (csrc/mpc_host.hip), entry point by entry point, against the semantics of
the reference's operator impls: MpcField = enum { Public(Fr), Shared(S) } (mpc-algebra/src/wire/field.rs:37-40), Public(x) next to
shared values = the share "x on the leader, 0 elsewhere" in BOTH lanes (wire/field.rs:339-362,414-437; share/additive.rs:145-152,
share/spdz.rs:214-218), a product with a public value scales every lane (wire/field.rs:463-476), batch_product_in_place of two shared
slices = FieldShare::batch_mul over a Beaver triple (wire/field.rs:917-958, share/field.rs:97-129), MpcG1Affine::multi_scalar_mul =
multi_scale_pub_group over the lanes (wire/pairing.rs:714-777).  Elements live in host byte buffers laid out as rustc lays the enums
out -- both share types (40 / 72-byte elements), both discriminant positions -- and are read and written IN PLACE.  Expected values
come from the PLAIN entry points on contiguous vectors (held to the oracle by test_gpu_field_ntt.py / test_gpu_msm.py) and from the
discrete-log identity."""
import ctypes as C

import numpy as np
import pytest

import zkref as O
import zk_mpc_amd as Z
import zk_mpc_amd.api as A
import zk_mpc_amd.convert as cv
from zk_mpc_amd import mpc
from test_gpu_mpc import run_parties

pytestmark = pytest.mark.gpu

LAYOUTS = [(False, False), (True, False), (False, True), (True, True)]        # (spdz, tag_last)


def fr_list(rng, n):
    return [rng.fr() for _ in range(n)]


def mont(vals):
    return cv.fr_to_mont(vals) if len(vals) else np.zeros((0, 4), dtype=np.uint64)


def unmont(arr):
    return cv.fr_from_mont(arr)


def party_view(leader, shared, pub, lane):
    """the lane vector a linear operation sees: Public(x) -> (leader ? x : 0)"""
    return [lane[i] if shared[i] else (pub[i] if leader else 0) for i in range(len(shared))]


@pytest.mark.parametrize("spdz,tag_last", LAYOUTS)
@pytest.mark.parametrize("inverse,coset", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_mpc_fft_in_place(spdz, tag_last, inverse, coset):
    rng = O.Prng(1000 + 2 * inverse + coset + 10 * spdz + 100 * tag_last)
    lay = A.MpcFieldLayout(spdz, tag_last)
    log_n, n = 9, 300
    N = 1 << log_n
    pub = fr_list(rng, n)
    cases = {
        "mixed": [i % 3 != 1 for i in range(n)],
        "hidden_share": [i == 17 for i in range(n)],           # the three sampled discriminants say Public: the gather must still see element 17
        "all_shared": [True] * n,
        "all_public": [False] * n,
    }
    for name, shared in cases.items():
        for party in (0, 1):
            ctx = Z.Context(0, party, 2)
            try:
                sh, mc = fr_list(rng, n), fr_list(rng, n)
                v = A.MpcVec(lay, N).set(shared + [False] * (N - n), mont([sh[i] if shared[i] else pub[i] for i in range(n)] + [0] * (N - n)),
                                         mont(mc + [0] * (N - n)))
                v.raw[n:, lay.off_tag] = 0x77                           # (beyond n the caller's Vec is not even read: the transform appends Public(0))
                ctx.mpc_fft_in_place(v, n, log_n, inverse, coset)
                if not any(shared):                                    # all Public: the transform of the values, on every party; still Public
                    want = ctx._fft_host(mont(pub + [0] * (N - n)), log_n, inverse, coset)
                    assert not v.shared().any(), name
                    assert np.array_equal(v.lane(0), want), (name, party)
                    continue
                assert v.shared().all(), name                           # N outputs, every one Shared
                for k, lane in enumerate([sh, mc][:2 if spdz else 1]):
                    x = party_view(party == 0, shared, pub, lane)
                    want = ctx._fft_host(mont(x + [0] * (N - n)), log_n, inverse, coset)
                    assert np.array_equal(v.lane(k), want), (name, party, k)
            finally:
                ctx.close()


@pytest.mark.parametrize("spdz,tag_last", LAYOUTS)
def test_mpc_divide_by_vanishing_keeps_every_variant(spdz, tag_last):
    rng = O.Prng(77 + spdz + 2 * tag_last)
    lay = A.MpcFieldLayout(spdz, tag_last)
    log_n = 8
    N = 1 << log_n
    for shared in ([i % 4 != 0 for i in range(N)], [False] * N, [True] * N):
        for party in (0, 1):
            ctx = Z.Context(0, party, 2)
            try:
                pub, sh, mc = fr_list(rng, N), fr_list(rng, N), fr_list(rng, N)
                v = A.MpcVec(lay, N).set(shared, mont([sh[i] if shared[i] else pub[i] for i in range(N)]), mont(mc))
                ctx.mpc_divide_by_vanishing_on_coset_in_place(v, log_n)

                def scaled(vals):
                    a = mont(vals)
                    ctx._ck(ctx.lib.zk_fr_divide_by_vanishing_on_coset_in_place(ctx.h, a.ctypes.data_as(C.c_void_p), log_n))
                    return a
                assert list(v.shared()) == shared                          # element-wise `*e *= &i`: nobody changes variant
                want0 = scaled([sh[i] if shared[i] else pub[i] for i in range(N)])     # a Public value is scaled on EVERY party
                assert np.array_equal(v.lane(0), want0)
                if spdz and any(shared):
                    want1 = scaled(mc)
                    idx = np.array(shared)
                    assert np.array_equal(v.lane(1)[idx], want1[idx])
            finally:
                ctx.close()


@pytest.mark.parametrize("spdz,tag_last", LAYOUTS)
def test_mpc_batch_product_local_cases(spdz, tag_last):
    """public x public, shared x public, public x shared (wire/field.rs:463-476): no exchange; a slice that mixes variants is refused."""
    rng = O.Prng(4242 + spdz + 2 * tag_last)
    lay = A.MpcFieldLayout(spdz, tag_last)
    n = 700
    for party in (0, 1):
        ctx = Z.Context(0, party, 2)
        try:
            a, am, b, bm = (fr_list(rng, n) for _ in range(4))
            prod = lambda x, y: [xi * yi % O.R_MOD for xi, yi in zip(x, y)]
            for a_sh, b_sh in ((False, False), (True, False), (False, True)):
                va = A.MpcVec(lay, n).set([a_sh] * n, mont(a), mont(am))
                vb = A.MpcVec(lay, n).set([b_sh] * n, mont(b), mont(bm))
                assert ctx.mpc_batch_product_in_place(va, vb, n) == 0
                assert va.shared().all() == (a_sh or b_sh) and va.shared().any() == (a_sh or b_sh)
                assert unmont(va.lane(0)) == prod(a, b)
                if spdz and (a_sh or b_sh):
                    assert unmont(va.lane(1)) == prod(am if a_sh else a, b if a_sh else bm)
            mixed = A.MpcVec(lay, n).set([i != 5 for i in range(n)], mont(a), mont(am))
            with pytest.raises(Z.ZkError, match="heterogenously"):
                ctx.mpc_batch_product_in_place(mixed, A.MpcVec(lay, n).set([True] * n, mont(b), mont(bm)), n)
        finally:
            ctx.close()


def _share_out(rng, vals, parties):
    """additive shares of vals: parties - 1 random vectors and the rest"""
    sh = [[rng.fr() for _ in vals] for _ in range(parties - 1)]
    last = [(v - sum(col)) % O.R_MOD for v, col in zip(vals, zip(*sh))] if parties > 1 else list(vals)
    return sh + [last]


@pytest.mark.parametrize("spdz,tag_last,parties,real_triple", [(False, False, 1, False), (True, False, 1, True), (False, True, 3, False),
                                                                (True, True, 3, False), (False, False, 3, True), (True, False, 2, True)])
def test_mpc_batch_product_beaver(spdz, tag_last, parties, real_triple):
    """both slices Shared: S::batch_mul (share/field.rs:97-129) through the party's transport -- the product shares sum to a o b,
    the MAC lane to the same (key 1), with DummyFieldTripleSource (what the wire passes) and with real triples from host vectors."""
    rng = O.Prng(9000 + spdz + 2 * tag_last + 4 * parties + 8 * real_triple)
    lay = A.MpcFieldLayout(spdz, tag_last)
    n = 1500
    a, b = fr_list(rng, n), fr_list(rng, n)
    lanes = 2 if spdz else 1
    sa = [_share_out(rng, a, parties) for _ in range(lanes)]          # [lane][party]: the MAC lane is an independent sharing of the same values
    sb = [_share_out(rng, b, parties) for _ in range(lanes)]
    tr = None
    if real_triple:
        x, y = fr_list(rng, n), fr_list(rng, n)
        z = [xi * yi % O.R_MOD for xi, yi in zip(x, y)]
        tr = [[_share_out(rng, t, parties) for t in (x, y, z)] for _ in range(lanes)]     # [lane][x|y|z][party]

    def work(p, ctx, net):
        party = mpc.Party(ctx=ctx, net=net)
        vt, errors, _keep = party._net_vtable()
        va = A.MpcVec(lay, n).set([True] * n, mont(sa[0][p]), mont(sa[-1][p]))
        vb = A.MpcVec(lay, n).set([True] * n, mont(sb[0][p]), mont(sb[-1][p]))
        triple = [mont(tr[l][k][p]) for l in range(lanes) for k in range(3)] if tr else None
        sent = ctx.mpc_batch_product_in_place(va, vb, n, vt if parties > 1 else None, triple)
        assert not errors, errors
        assert va.shared().all()
        assert sent == (4 if spdz else 2) * 32 * n                       # two masked operands (SPDZ: and their MAC checks)
        return [unmont(va.lane(l)) for l in range(lanes)]

    outs = run_parties(parties, work)
    want = [ai * bi % O.R_MOD for ai, bi in zip(a, b)]
    for l in range(lanes):
        got = [sum(outs[p][l][i] for p in range(parties)) % O.R_MOD for i in range(n)]
        assert got == want, "lane %d" % l


def test_mpc_batch_product_mac_check_fails_on_a_tampered_share():
    """SpdzFieldShare::batch_open asserts the MAC relation (share/spdz.rs:188-195): a party whose MAC lane does not belong to its share
    makes every party's call fail with ZK_ERR_MAC."""
    rng = O.Prng(31337)
    lay = A.MpcFieldLayout(True, False)
    n, parties = 600, 2
    a, b = fr_list(rng, n), fr_list(rng, n)
    sa = [_share_out(rng, a, parties) for _ in range(2)]
    sb = [_share_out(rng, b, parties) for _ in range(2)]
    sa[1][1][3] = (sa[1][1][3] + 1) % O.R_MOD                              # party 1's MAC share of element 3 is off by one

    def work(p, ctx, net):
        party = mpc.Party(ctx=ctx, net=net)
        vt, errors, _keep = party._net_vtable()
        va = A.MpcVec(lay, n).set([True] * n, mont(sa[0][p]), mont(sa[1][p]))
        vb = A.MpcVec(lay, n).set([True] * n, mont(sb[0][p]), mont(sb[1][p]))
        with pytest.raises(Z.ZkError, match="error -5"):
            ctx.mpc_batch_product_in_place(va, vb, n, vt)
        return True

    assert run_parties(parties, work) == [True, True]


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("spdz,tag_last", LAYOUTS)
def test_mpc_msm(group, spdz, tag_last):
    """multi_scale_pub_group on the party's lanes, mixed scalars forced to shares (from_public), every scalar Public -> the plain MSM,
    min(len), the empty sum, tiny (uncached) and cached tables, a table found again by content in another buffer, and the reference's
    assertion that every base is Public."""
    rng = O.Prng(555 + group + 2 * spdz + 4 * tag_last)
    flay = A.MpcFieldLayout(spdz, tag_last)
    glay = A.mpc_group_layout(group, spdz, tag_last)
    gen, gmul = (O.G1_GEN, O.g1_mul) if group == 1 else (O.G2_GEN, O.g2_mul)
    to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
    nb = 640 if group == 1 else 300
    for party in (0, 1):
        ctx = Z.Context(0, party, 2)
        try:
            ks = fr_list(rng, nb)
            dk = ctx.upload(mont(ks))
            tab = ctx.fixed_base(dk.ptr, nb, group, mont([1])[0])
            pts = np.ascontiguousarray(tab.download())
            bases = A.mpc_wrap_points(pts, glay)
            leader = party == 0
            expect = lambda sc, m: gmul(gen, sum(s * k for s, k in zip(sc[:m], ks[:m])) % O.R_MOD)
            for ns, shared in ((nb, [i % 5 != 2 for i in range(nb)]), (nb - 37, [True] * (nb - 37)), (100, [i == 50 for i in range(100)])):
                pub, sh, mc = fr_list(rng, ns), fr_list(rng, ns), fr_list(rng, ns)
                sv = A.MpcVec(flay, ns).set(shared, mont([sh[i] if shared[i] else pub[i] for i in range(ns)]), mont(mc))
                for rep in range(2):                                       # a miss, then a hit verified against the caller's wrappers
                    l0, l1, all_pub = ctx.mpc_msm(group, bases, nb, glay, sv, ns)
                    assert not all_pub
                    assert to_aff(l0) == expect(party_view(leader, shared, pub, sh), ns)
                    assert to_aff(l1) == expect(party_view(leader, shared, pub, mc if spdz else sh), ns)
            # every scalar Public: the plain MSM on every party (the wire wraps it with from_public)
            pub = fr_list(rng, nb)
            sv = A.MpcVec(flay, nb).set([False] * nb, mont(pub))
            l0, _, all_pub = ctx.mpc_msm(group, bases.copy(), nb, glay, sv, nb)        # (the table in ANOTHER buffer: found again by content)
            assert all_pub and to_aff(l0) == expect(pub, nb)
            # fewer bases than scalars, and none
            l0, _, _ = ctx.mpc_msm(group, bases, 300, glay, A.MpcVec(flay, nb).set([True] * nb, mont(pub), mont(pub)), nb)
            assert to_aff(l0) == expect(pub, 300)
            l0, l1, _ = ctx.mpc_msm(group, bases, 0, glay, sv, nb)
            assert to_aff(l0) == to_aff(l1) == gmul(gen, 0)
            # a Shared base: the reference asserts (wire/pairing.rs:716); at a sampled position and at one only the full pass sees
            for row in (0, 5):
                bad = A.mpc_wrap_points(pts, glay, shared_rows=(row,))
                with pytest.raises(Z.ZkError, match="not Public"):
                    ctx.mpc_msm(group, bad, nb, glay, sv, nb)
            tab.free(); dk.free()
        finally:
            ctx.close()


def test_arguments_that_do_not_fit_are_refused_before_anything_is_sized_by_them():
    """EvaluationDomain::new refuses a size beyond the field's two-adicity (radix2/mod.rs:51-57: None); the host-slice transforms are
    handed log_n by the override and must refuse a value they cannot serve BEFORE a buffer is sized by it -- and a layout whose fields
    do not fit its stride, a vector longer than its domain."""
    ctx = Z.Context(0, 0, 1)
    try:
        lay = A.MpcFieldLayout(False, False)
        v = A.MpcVec(lay, 8).set([False] * 8, mont([1] * 8))
        before = v.raw.copy()
        for log_n in (29, 40, 64, 1 << 31):
            ARG = -2                                                # ZK_ERR_ARG
            raw = v.raw.ctypes.data_as(C.c_void_p)
            assert ctx.lib.zk_mpc_fft_in_place(ctx.h, raw, 8, C.byref(lay.c), log_n, 0, 0) == ARG
            assert "log_n" in ctx.lib.zk_last_error(ctx.h).decode()
            assert ctx.lib.zk_mpc_divide_by_vanishing_on_coset_in_place(ctx.h, raw, C.byref(lay.c), log_n) == ARG
            plain = mont([1] * 8)
            assert ctx.lib.zk_fr_fft_in_place(ctx.h, plain.ctypes.data_as(C.c_void_p), 8, log_n, 0, 0) == ARG
            assert ctx.lib.zk_fr_divide_by_vanishing_on_coset_in_place(ctx.h, plain.ctypes.data_as(C.c_void_p), log_n) == ARG
        with pytest.raises(Z.ZkError, match="exceeds the domain"):
            ctx.mpc_fft_in_place(v, 8, 2, 0, 0)
        assert np.array_equal(v.raw, before)                        # nothing was touched
    finally:
        ctx.close()
