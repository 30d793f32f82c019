"""GPU parity: Groth16 setup + prove through the C ABI against the oracle: proving-key elements,
proof bytes (bit-exact), known-trapdoor prediction and the pairing verifier."""
import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv

pytestmark = pytest.mark.gpu


def trapdoor(rng):
    return O.Trapdoor(rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr())


def td_args(td):
    m = cv.fr_to_mont([td.alpha, td.beta, td.gamma, td.delta, td.tau, td.g1_k, td.g2_k])
    return [m[i] for i in range(7)]


def csr(rows):
    rp, col, coeff = [0], [], []
    for r in rows:
        for c, i in r:
            col.append(i)
            coeff.append(c)
        rp.append(len(col))
    return (np.array(rp, dtype=np.uint32), np.array(col, dtype=np.uint32),
            cv.fr_to_mont(coeff) if coeff else np.zeros((0, 4), dtype=np.uint64))


def upload_r1cs(ctx, r1cs):
    return ctx.r1cs_upload(r1cs.num_instance, r1cs.num_witness, csr(r1cs.a), csr(r1cs.b), csr(r1cs.c))


def my_simple_circuit(a, b):
    """MySimpleCircuit (src/circuits/circuit.rs:85-111): c = a*b public, constraint a*b=c six times."""
    c = a * b % O.R_MOD
    A = [[(1, 2)] for _ in range(6)]
    B = [[(1, 3)] for _ in range(6)]
    Cm = [[(1, 1)] for _ in range(6)]
    return O.R1CS(2, 2, A, B, Cm), [1, c, a, b]


def general_r1cs(rng, nc, ni, nw):
    """Random satisfiable R1CS with non-unit coefficients and multi-term rows."""
    nv = ni + nw
    z = [1] + [rng.fr() for _ in range(nv - 1)]
    A, B, Cm = [], [], []
    for i in range(nc):
        ra = [(rng.fr(), rng.u64() % nv) for _ in range(1 + rng.u64() % 3)]
        rb = [(rng.fr(), rng.u64() % nv) for _ in range(1 + rng.u64() % 3)]
        va, vb = O.evaluate_constraint(ra, z), O.evaluate_constraint(rb, z)
        # c row: k * z[j] with k chosen so the constraint holds (z[j] != 0)
        j = ni + (i % nw)
        k = va * vb * pow(z[j], -1, O.R_MOD) % O.R_MOD
        A.append(ra); B.append(rb); Cm.append([(k, j)])
    return O.R1CS(ni, nw, A, B, Cm), z


def check_pk(ctx, pk, opk):
    assert cv.g1_array_to_affine(pk.download("a_query")) == opk.a_query
    assert cv.g1_array_to_affine(pk.download("b_g1_query")) == opk.b_g1_query
    assert cv.g2_array_to_affine(pk.download("b_g2_query")) == opk.b_g2_query
    assert cv.g1_array_to_affine(pk.download("h_query")) == opk.h_query
    assert cv.g1_array_to_affine(pk.download("l_query")) == opk.l_query
    assert cv.g1_array_to_affine(pk.download("gamma_abc_g1")) == opk.gamma_abc_g1
    assert cv.g1_array_to_affine([pk.vk_g1(i) for i in range(3)]) == [opk.alpha_g1, opk.beta_g1, opk.delta_g1]
    assert cv.g2_array_to_affine([pk.vk_g2(i) for i in range(3)]) == [opk.beta_g2, opk.delta_g2, opk.gamma_g2]


def test_config1_my_simple_circuit(ctx):
    """BASELINE config 1 (bin_test_groth16: MySimpleCircuit, 6 constraints, domain 8), local prove."""
    rng = O.Prng(0x5EED0001)
    r1cs, z = my_simple_circuit(rng.fr(), rng.fr())
    td = trapdoor(rng)
    pks = O.ProvingKeyScalars(r1cs, td)
    opk = O.ProvingKey(pks)
    dr = upload_r1cs(ctx, r1cs)
    assert dr.domain_log == 3
    pk = ctx.groth16_setup(dr, *td_args(td))
    check_pk(ctx, pk, opk)
    r, s = rng.fr(), rng.fr()
    proof = ctx.create_proof(pk, dr, cv.fr_to_mont(z), cv.fr_to_mont([r])[0], cv.fr_to_mont([s])[0])
    oproof = O.create_proof(r1cs, opk, z, r, s)
    assert proof == O.proof_serialize(*oproof)
    assert oproof == O.predict_proof(r1cs, pks, z, r, s)
    assert O.verify_proof(opk, oproof, z[1:2])
    assert not O.verify_proof(opk, oproof, [(z[1] + 1) % O.R_MOD])
    # no-zk variant (create_proof_no_zk, src/groth16.rs:52-64)
    zero = cv.fr_to_mont([0])[0]
    assert ctx.create_proof(pk, dr, cv.fr_to_mont(z), zero, zero) == O.proof_serialize(*O.create_proof(r1cs, opk, z, 0, 0))


def test_pk_upload_path(ctx):
    """A key produced elsewhere (the oracle) uploaded through zk_pk_upload gives the same proof."""
    rng = O.Prng(77)
    r1cs, z = O.mul_chain_r1cs(9, rng.fr(), rng.fr())
    td = trapdoor(rng)
    opk = O.ProvingKey(O.ProvingKeyScalars(r1cs, td))
    dr = upload_r1cs(ctx, r1cs)
    pk = ctx.pk_upload(cv.g1_affine_to_array([opk.alpha_g1])[0], cv.g1_affine_to_array([opk.beta_g1])[0],
                       cv.g1_affine_to_array([opk.delta_g1])[0], cv.g2_affine_to_array([opk.beta_g2])[0],
                       cv.g2_affine_to_array([opk.delta_g2])[0], cv.g1_affine_to_array(opk.a_query),
                       cv.g1_affine_to_array(opk.b_g1_query), cv.g2_affine_to_array(opk.b_g2_query),
                       cv.g1_affine_to_array(opk.h_query), cv.g1_affine_to_array(opk.l_query))
    r, s = rng.fr(), rng.fr()
    proof = ctx.create_proof(pk, dr, cv.fr_to_mont(z), cv.fr_to_mont([r])[0], cv.fr_to_mont([s])[0])
    assert proof == O.proof_serialize(*O.create_proof(r1cs, opk, z, r, s))


def test_general_r1cs(ctx):
    rng = O.Prng(78)
    r1cs, z = general_r1cs(rng, nc=50, ni=3, nw=20)
    td = trapdoor(rng)
    pks = O.ProvingKeyScalars(r1cs, td)
    opk = O.ProvingKey(pks)
    dr = upload_r1cs(ctx, r1cs)
    pk = ctx.groth16_setup(dr, *td_args(td))
    check_pk(ctx, pk, opk)
    # witness map alone
    dz = ctx.upload(cv.fr_to_mont(z))
    D = 1 << dr.domain_log
    dh = ctx.alloc(D * 32)
    ctx.witness_map_dev(dr, dz.ptr, dh.ptr)
    assert cv.fr_from_mont(ctx.download(dh, (D, 4))) == O.witness_map(r1cs, z)
    r, s = rng.fr(), rng.fr()
    proof = ctx.create_proof(pk, dr, cv.fr_to_mont(z), cv.fr_to_mont([r])[0], cv.fr_to_mont([s])[0])
    oproof = O.predict_proof(r1cs, pks, z, r, s)
    assert proof == O.proof_serialize(*oproof)
    assert O.verify_proof(opk, oproof, z[1:3])


@pytest.mark.parametrize("n", [1, 62, 1000])
def test_mul_chain_device_builder(ctx, n):
    """zk_r1cs_mul_chain / zk_mul_chain_assignment_dev equal the oracle's mul-chain family, and the
    proof equals the known-trapdoor prediction (exact bytes)."""
    rng = O.Prng(500 + n)
    w0, w1 = rng.fr(), rng.fr()
    r1cs, z = O.mul_chain_r1cs(n, w0, w1)
    dr = ctx.r1cs_mul_chain(n)
    dz = ctx.mul_chain_assignment_dev(n, cv.fr_to_mont([w0])[0], cv.fr_to_mont([w1])[0])
    assert cv.fr_from_mont(ctx.download(dz, (n + 3, 4))) == z
    td = trapdoor(rng)
    pks = O.ProvingKeyScalars(r1cs, td)
    pk = ctx.groth16_setup(dr, *td_args(td))
    r, s = rng.fr(), rng.fr()
    proof = ctx.create_proof_dev(pk, dr, dz.ptr, cv.fr_to_mont([r])[0], cv.fr_to_mont([s])[0])
    assert proof == O.proof_serialize(*O.predict_proof(r1cs, pks, z, r, s))


def test_many_public_inputs_with_window_multiples(ctx):
    """5 instance variables, 2^16 + 300 constraints: the proving key carries window multiples and the L job reads the
    padded l_query through the sort shared with A / B (the instance part meets points at infinity).  Proof bytes equal
    the C oracle's known-trapdoor prediction."""
    import zkref_c as OC
    rng = O.Prng(909)
    ni, nc = 5, (1 << 16) + 300
    nw = nc + 1
    mont = lambda v: cv.fr_to_mont([v])[0]
    r = O.R_MOD
    xs = [rng.fr() for _ in range(ni - 1)]
    w = [rng.fr()]
    for i in range(nc):                                  # (w_i + x_{i mod 4}) * w_i = w_{i+1}
        w.append((w[i] + xs[i % 4]) * w[i] % r)
    z = [1] + xs + w
    one = mont(1)
    i = np.arange(nc, dtype=np.int64)
    rp1 = np.arange(nc + 1, dtype=np.uint32)
    rp2 = (2 * np.arange(nc + 1)).astype(np.uint32)
    a_col = np.stack([1 + (i % 4), ni + i], axis=1).reshape(-1).astype(np.uint32)
    A = (rp2, a_col, np.tile(one, (2 * nc, 1)))
    B = (rp1, (ni + i).astype(np.uint32), np.tile(one, (nc, 1)))
    Cm = (rp1, (ni + i + 1).astype(np.uint32), np.tile(one, (nc, 1)))
    dr = ctx.r1cs_upload(ni, nw, A, B, Cm)
    td = [mont(rng.fr()) for _ in range(7)]
    rr, ss = mont(rng.fr()), mont(rng.fr())
    pk = ctx.groth16_setup(dr, *td)
    assert ctx.lib.zk_bases_window_bits(pk.query_bases("a_query").h) >= 15
    zarr = cv.fr_to_mont(z)
    proof = ctx.create_proof(pk, dr, zarr, rr, ss)
    cr = OC.R1cs(ni, nw, A, B, Cm)
    assert proof == OC.groth16_predict(cr, np.stack(td), zarr, OC.witness_map(cr, zarr), rr, ss)
    pk.free()


@pytest.mark.parametrize("n", [(1 << 16) + 50, (1 << 15) + 7])
def test_hint_next_front_prefetch(ctx, n):
    """zk_groth16_hint_next_dev: a proof enqueues the announced next proof's front (z-sort, witness map, H-sort) behind its
    own kernels.  Proof bytes must not change: a queue of three different assignments proved with hints equals the same
    queue without; a hint that does not come true (another assignment follows) is dropped; a hint followed by a batch MSM
    on the same context is dropped too.  D = 2^17: the accumulate kernels queue on the accumulate stream; D = 2^16: every job's
    kernel runs on the stream of its own reduce chain (groth16_pipeline.hip: small_jobs) and the front's ordering rests on events."""
    rng = O.Prng(4711)
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    dr = ctx.r1cs_mul_chain(n)
    pk = ctx.groth16_setup(dr, *td)
    zs = [ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr())) for _ in range(3)]
    rs = [(mont(rng.fr()), mont(rng.fr())) for _ in range(3)]
    plain = [ctx.create_proof_dev(pk, dr, z.ptr, r, s) for z, (r, s) in zip(zs, rs)]
    assert len(set(plain)) == 3
    hinted = []
    for k in range(3):
        ctx.groth16_hint_next_dev(zs[k + 1].ptr if k + 1 < 3 else None)
        hinted.append(ctx.create_proof_dev(pk, dr, zs[k].ptr, *rs[k]))
    assert hinted == plain
    # the announced assignment does not follow
    ctx.groth16_hint_next_dev(zs[2].ptr)
    assert ctx.create_proof_dev(pk, dr, zs[0].ptr, *rs[0]) == plain[0]
    assert ctx.create_proof_dev(pk, dr, zs[1].ptr, *rs[1]) == plain[1]
    # a front in flight and then other work on the MSM scratch
    ctx.groth16_hint_next_dev(zs[1].ptr)
    assert ctx.create_proof_dev(pk, dr, zs[0].ptr, *rs[0]) == plain[0]
    q = pk.query_bases("a_query")
    one = ctx.msm_batch_dev([(q, 1, zs[2].ptr + 32, n + 2)])[0]
    assert cv.g1_projective_to_affine(one) == cv.g1_projective_to_affine(ctx.msm_dev(q, 1, zs[2].ptr + 32, n + 2))
    assert ctx.create_proof_dev(pk, dr, zs[1].ptr, *rs[1]) == plain[1]
    # steady state: the same assignment announced again and again
    for _ in range(3):
        ctx.groth16_hint_next_dev(zs[2].ptr)
        assert ctx.create_proof_dev(pk, dr, zs[2].ptr, *rs[2]) == plain[2]
    ctx.groth16_hint_next_dev(None)
    assert ctx.create_proof_dev(pk, dr, zs[2].ptr, *rs[2]) == plain[2]
    pk.free()


@pytest.mark.parametrize("n", [1000, 5000, (1 << 14) - 2])
def test_chained_fronts_of_small_proofs(ctx, n):
    """Round 5: the front of an announced SMALL proof carries its whole device chain (groth16_pipeline.hip: chained; accumulate
    launches and reduce chains of all five jobs enqueued behind the current proof's, results in alternating pinned buffers).  A
    queue of four assignments proved twelve times with hints equals the proofs without; the same with zk_groth16_chain_fronts off
    and toggled in between; an announced proof that does not follow leaves a whole chain to drain."""
    rng = O.Prng(1234 + n)
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    dr = ctx.r1cs_mul_chain(n)
    pk = ctx.groth16_setup(dr, *td)
    zs = [ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr())) for _ in range(4)]
    rs = [(mont(rng.fr()), mont(rng.fr())) for _ in range(4)]
    plain = [ctx.create_proof_dev(pk, dr, z.ptr, r, s) for z, (r, s) in zip(zs, rs)]
    assert len(set(plain)) == 4
    for mode in (True, False, True):
        ctx.groth16_chain_fronts(mode)
        for i in range(12):
            ctx.groth16_hint_next_dev(zs[(i + 1) % 4].ptr)
            assert ctx.create_proof_dev(pk, dr, zs[i % 4].ptr, *rs[i % 4]) == plain[i % 4], (mode, i)
    # the chain of zs[1] is in flight; zs[3] is proved instead, then zs[1] after all
    assert ctx.create_proof_dev(pk, dr, zs[3].ptr, *rs[3]) == plain[3]
    assert ctx.create_proof_dev(pk, dr, zs[1].ptr, *rs[1]) == plain[1]
    # toggled while a chained front is pending
    ctx.groth16_hint_next_dev(zs[2].ptr)
    assert ctx.create_proof_dev(pk, dr, zs[0].ptr, *rs[0]) == plain[0]
    ctx.groth16_chain_fronts(False)
    ctx.groth16_hint_next_dev(zs[3].ptr)
    assert ctx.create_proof_dev(pk, dr, zs[2].ptr, *rs[2]) == plain[2]
    assert ctx.create_proof_dev(pk, dr, zs[3].ptr, *rs[3]) == plain[3]
    ctx.groth16_chain_fronts(True)
    ctx.groth16_hint_next_dev(None)
    pk.free()


@pytest.mark.parametrize("n,label", [((1 << 20) - 2, "D=2^20 (BASELINE config 2, the benched shape)"),
                                     (1 << 20, "D=2^21 (the reference's natural sizing, src/groth16.rs:256-257)")])
def test_headline_size_matches_known_trapdoor_prediction(ctx, n, label):
    """The benched configuration itself: mul-chain Groth16 at n = 2^20 - 2 (domain 2^20) and n = 2^20 (domain 2^21), key with
    window multiples (c = 20: G2 accumulate / reduce on lane pairs with 16 virtual windows), proof bytes against the C
    oracle's known-trapdoor prediction (Fr arithmetic on the CPU + three scalar multiplications) -- for isolated proofs,
    for a queue with the next assignment announced (zk_groth16_hint_next_dev) and through the host-slice entry point."""
    import zkref_c as OC
    rng = O.Prng(20200 + (n & 3))
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    dr = ctx.r1cs_mul_chain(n)
    assert dr.domain_log == (20 if n < (1 << 20) - 1 else 21)
    pk = ctx.groth16_setup(dr, *td)
    assert ctx.lib.zk_bases_window_bits(pk.query_bases("b_g2_query").h) == 20
    zs = [ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr())) for _ in range(2)]
    rs = [(mont(rng.fr()), mont(rng.fr())) for _ in range(2)]
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    want = []
    zarrs = []
    for z, (r, s) in zip(zs, rs):
        zarr = ctx.download(z, (n + 3, 4))
        zarrs.append(zarr)
        want.append(OC.groth16_predict(cr, np.stack(td), zarr, OC.witness_map(cr, zarr, OC.num_threads()), r, s))
    assert want[0] != want[1]
    assert [ctx.create_proof_dev(pk, dr, z.ptr, r, s) for z, (r, s) in zip(zs, rs)] == want          # isolated
    got = []
    for k in (0, 1, 0, 1):                                                                            # announced queue
        ctx.groth16_hint_next_dev(zs[1 - k].ptr)
        got.append(ctx.create_proof_dev(pk, dr, zs[k].ptr, *rs[k]))
    ctx.groth16_hint_next_dev(None)
    assert got == [want[0], want[1], want[0], want[1]]
    assert ctx.create_proof(pk, dr, zarrs[1], *rs[1]) == want[1]                                      # host witness -> host bytes
    pk.free()
    for z in zs:
        z.free()
    dr.free()


def test_hint_next_with_split_buckets_and_plain_tables(ctx):
    """The next proof's H-sort rewrites the sort scratch of slot 5 (ctr / heavy descriptors) that the CURRENT proof's H
    reduce chain (k_fold) still reads when it runs on the other stream: it has to wait for that chain.  The race only
    matters when H's buckets are split, so: tables without window multiples (ZK_PRECOMP=0: seg = 32, 16 bucket sets) and an
    assignment whose h has many repeated scalars -- w0 = 1, w1 = 1 makes every w_i = 1, so a, b, c are constant on the domain
    and z is all ones (every digit of every scalar in one bucket per window: maximally split buckets for the z jobs too).
    A queue of such proofs with hints must give the bytes of the isolated proofs, repeatedly."""
    import os
    n = (1 << 16) + 9
    rng = O.Prng(60606)
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    dr = ctx.r1cs_mul_chain(n)
    os.environ["ZK_PRECOMP"] = "0"
    try:
        pk = ctx.groth16_setup(dr, *td)
    finally:
        del os.environ["ZK_PRECOMP"]
    assert ctx.lib.zk_bases_window_bits(pk.query_bases("h_query").h) == 0
    # all-ones, a 2-cycle (w0 = 1, w1 = -1 -> 1, -1, -1, 1, -1, -1 ...: three distinct values), and a random chain
    zs = [ctx.mul_chain_assignment_dev(n, mont(1), mont(1)), ctx.mul_chain_assignment_dev(n, mont(1), mont(O.R_MOD - 1)),
          ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr()))]
    rs = [(mont(rng.fr()), mont(rng.fr())) for _ in range(3)]
    plain = [ctx.create_proof_dev(pk, dr, z.ptr, r, s) for z, (r, s) in zip(zs, rs)]
    import zkref_c as OC
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    for z, (r, s), p in zip(zs, rs, plain):
        zarr = ctx.download(z, (n + 3, 4))
        assert p == OC.groth16_predict(cr, np.stack(td), zarr, OC.witness_map(cr, zarr), r, s)
    for rep in range(4):
        for k in range(3):
            ctx.groth16_hint_next_dev(zs[(k + 1) % 3].ptr)
            assert ctx.create_proof_dev(pk, dr, zs[k].ptr, *rs[k]) == plain[k], (rep, k)
    ctx.groth16_hint_next_dev(None)
    pk.free()


def test_pending_front_does_not_outlive_its_key(ctx):
    """hint -> prove -> free the key -> a new key (very likely at the same address) and the same z buffer: the pending front
    of the old key (matched by address) must have been dropped with it (zk_pk_free calls zk_presort_free)."""
    n = (1 << 16) + 3
    rng = O.Prng(70707)
    mont = lambda v: cv.fr_to_mont([v])[0]
    dr = ctx.r1cs_mul_chain(n)
    z = ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr()))
    r, s = mont(rng.fr()), mont(rng.fr())
    proofs = []
    for k in range(3):
        td = [mont(rng.fr()) for _ in range(7)]
        pk = ctx.groth16_setup(dr, *td)
        want = ctx.create_proof_dev(pk, dr, z.ptr, r, s)
        ctx.groth16_hint_next_dev(z.ptr)                      # leaves a front for (pk, z) pending after this proof
        assert ctx.create_proof_dev(pk, dr, z.ptr, r, s) == want
        proofs.append(want)
        pk.free()                                             # the next key reuses the address; the front must be gone
    assert len(set(proofs)) == 3


def test_prove_queued_host_assignments(ctx):
    """zk_groth16_prove_queued: a queue of different host assignments (page-locked and ordinary numpy memory), each call
    announcing the next one; bytes equal the device-resident prover's for every element of the queue, also when an
    announcement does not come true and when the queue ends (NULL)."""
    n = (1 << 16) + 21
    rng = O.Prng(80808)
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    dr = ctx.r1cs_mul_chain(n)
    pk = ctx.groth16_setup(dr, *td)
    Q = 4
    dev = [ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr())) for _ in range(Q)]
    rs = [(mont(rng.fr()), mont(rng.fr())) for _ in range(Q)]
    want = [ctx.create_proof_dev(pk, dr, z.ptr, r, s) for z, (r, s) in zip(dev, rs)]
    assert len(set(want)) == Q
    pinned = [ctx.host_alloc((n + 3) * 32) for _ in range(Q)]
    host = []
    for k in range(Q):
        a = pinned[k].array((n + 3, 4)) if k % 2 == 0 else np.empty((n + 3, 4), dtype=np.uint64)
        a[:] = ctx.download(dev[k], (n + 3, 4))
        host.append(a)
    for rep in range(2):
        got = [ctx.create_proof_queued(pk, dr, host[k], *rs[k], z_next_host=host[k + 1] if k + 1 < Q else None) for k in range(Q)]
        assert got == want
    # an announcement that does not come true, then a plain call
    assert ctx.create_proof_queued(pk, dr, host[0], *rs[0], z_next_host=host[1]) == want[0]
    assert ctx.create_proof_queued(pk, dr, host[2], *rs[2], z_next_host=host[3]) == want[2]
    assert ctx.create_proof(pk, dr, host[1], *rs[1]) == want[1]
    assert ctx.create_proof_queued(pk, dr, host[3], *rs[3]) == want[3]
    pk.free()
    for p in pinned:
        p.free()


def test_msms_begin_then_finish_and_its_misuse(ctx):
    """zk_groth16_msms_begin_dev: the four MSMs over z enqueued ahead, zk_groth16_msms_dev adds the H job -- same five sums as the
    one-call form.  A begin that is not followed by its own finish (another z, another entry point in between, a proof, the
    key freed) must be dropped without a trace."""
    n = (1 << 16) + 5
    rng = O.Prng(81818)
    mont = lambda v: cv.fr_to_mont([v])[0]
    dr = ctx.r1cs_mul_chain(n)
    pk = ctx.groth16_setup(dr, *[mont(rng.fr()) for _ in range(7)])
    D = 1 << 17
    z1 = ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr()))
    z2 = ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr()))
    h1, h2 = ctx.alloc(D * 32), ctx.alloc(D * 32)
    ctx.witness_map_dev(dr, z1.ptr, h1.ptr)
    ctx.witness_map_dev(dr, z2.ptr, h2.ptr)
    same = lambda a, b: np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    want1, want2 = ctx.groth16_msms_dev(pk, dr, z1.ptr, h1.ptr), ctx.groth16_msms_dev(pk, dr, z2.ptr, h2.ptr)
    assert not same(want1, want2)
    tmp = ctx.alloc(D * 32)
    for _ in range(2):                                   # the plain use, twice in a row
        ctx.groth16_msms_begin_dev(pk, dr, z1.ptr)
        ctx.fr_vec_op_dev(1, h2.ptr, h2.ptr, tmp.ptr, D)                    # the caller's own kernels on the context stream
        assert same(ctx.groth16_msms_dev(pk, dr, z1.ptr, h1.ptr), want1)
    ctx.groth16_msms_begin_dev(pk, dr, z1.ptr)           # begun for z1, finished for z2: the begun jobs are dropped
    assert same(ctx.groth16_msms_dev(pk, dr, z2.ptr, h2.ptr), want2)
    ctx.groth16_msms_begin_dev(pk, dr, z2.ptr)           # another entry point in between (it rotates over the same scratch slots)
    r, s = mont(rng.fr()), mont(rng.fr())
    proof = ctx.create_proof_dev(pk, dr, z1.ptr, r, s)
    assert same(ctx.groth16_msms_dev(pk, dr, z2.ptr, h2.ptr), want2)
    assert ctx.create_proof_dev(pk, dr, z1.ptr, r, s) == proof
    ctx.groth16_msms_begin_dev(pk, dr, z1.ptr)           # the SAME z, but a whole proof instead of the finishing call
    assert ctx.create_proof_dev(pk, dr, z1.ptr, r, s) == proof
    assert same(ctx.groth16_msms_dev(pk, dr, z1.ptr, h1.ptr), want1)
    ctx.groth16_msms_begin_dev(pk, dr, z1.ptr)           # begun and never finished: the key goes first
    pk.free()


@pytest.mark.parametrize("log_n", [10, 16, 20])
def test_boolean_heavy_assignment_matches_prediction(ctx, log_n):
    """A witness shaped like the reference's circuits (90 % of the variables are bits: bench.py::bool_chain_system; the reference:
    docs/benchmark.md:45-58 and arkworks' unit-scalar fast path, ec/src/msm/variable_base.rs:45-49): heavy "0" and "1" digits in
    every z-MSM (dropped zero digits, one bucket with 45 % of all points, folded in two levels), proof bytes against the
    known-trapdoor prediction -- isolated, over an announced queue, and through the host-slice entry."""
    import importlib.util
    import os
    import zkref_c as OC
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = (1 << log_n) - 2
    rng = O.Prng(5150 + log_n)
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    systems = [bench.bool_chain_system(n, 40 + q + log_n) for q in range(2)]
    a, b, c = systems[0][:3]
    dr = ctx.r1cs_upload(2, n + 1, a, b, c)
    pk = ctx.groth16_setup(dr, *td)
    cr = OC.R1cs(2, n + 1, a, b, c)
    rs = [(mont(rng.fr()), mont(rng.fr())) for _ in range(2)]
    want = [OC.groth16_predict(cr, np.stack(td), sy[3], OC.witness_map(cr, sy[3], OC.num_threads()), r, s) for sy, (r, s) in zip(systems, rs)]
    assert want[0] != want[1]
    zs = [ctx.upload(sy[3]) for sy in systems]
    assert [ctx.create_proof_dev(pk, dr, z.ptr, r, s) for z, (r, s) in zip(zs, rs)] == want
    got = []
    for k in (0, 1, 0, 1):
        ctx.groth16_hint_next_dev(zs[1 - k].ptr)
        got.append(ctx.create_proof_dev(pk, dr, zs[k].ptr, *rs[k]))
    ctx.groth16_hint_next_dev(None)
    assert got == [want[0], want[1], want[0], want[1]]
    assert ctx.create_proof(pk, dr, systems[1][3], *rs[1]) == want[1]
    pk.free(); dr.free()
    for z in zs:
        z.free()


@pytest.mark.parametrize("n,n_ctx", [(300, 2), ((1 << 14) - 2, 2), ((1 << 16) - 2, 3), ((1 << 16) - 2, 4), ((1 << 18) - 2, 5)])
def test_one_prover_over_several_contexts(ctx, n, n_ctx):
    """zk_groth16_prove_multi (second level): the five MSMs of src/groth16.rs:106-160 cut by cost into base ranges
    and dealt to n_ctx contexts -- here all on this one device, each with its own copy of the key and the constraint system --
    the partial sums added on the host: the same 192 bytes as zk_groth16_prove_dev on one context, and as the prediction.  The
    plan covers every term of every job exactly once."""
    import zk_mpc_amd as Z
    import zkref_c as OC
    rng = O.Prng(8100 + n_ctx + (n & 0xff))
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    others = [Z.Context(0) for _ in range(n_ctx - 1)]
    ctxs = [ctx] + others
    try:
        drs = [c.r1cs_mul_chain(n) for c in ctxs]
        pks = [c.groth16_setup(dr, *td) for c, dr in zip(ctxs, drs)]
        z = ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr()))
        r, s = mont(rng.fr()), mont(rng.fr())
        single = ctx.create_proof_dev(pks[0], drs[0], z.ptr, r, s)
        cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
        zarr = ctx.download(z, (n + 3, 4))
        assert single == OC.groth16_predict(cr, np.stack(td), zarr, OC.witness_map(cr, zarr, OC.num_threads()), r, s)
        for _ in range(2):                                   # (twice: scratch and key state survive a call)
            assert ctx.create_proof_multi(others, pks, drs, z.ptr, r, s) == single
        plan = ctx.multi_plan(pks[0], drs[0], n_ctx)
        lens = {0: n + 2, 1: n + 2, 2: n + 2, 3: n + 1, 4: min(len(pks[0].query_bases("h_query")), 1 << drs[0].domain_log)}
        for job in range(5):
            pieces = sorted((lo, m) for (_, j, lo, m) in plan if j == job)
            assert pieces[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(pieces, pieces[1:])) and sum(m for _, m in pieces) == lens[job]
        if n >= 1 << 14:
            assert len(set(c for c, _, _, _ in plan)) == n_ctx                  # every context got work
        assert ctx.create_proof_dev(pks[0], drs[0], z.ptr, r, s) == single     # the single-context pipeline still works afterwards
        for pk in pks:
            pk.free()
        for dr in drs:
            dr.free()
        z.free()
    finally:
        for c in others:
            c.close()


def test_one_prover_deal_with_a_boundary_crumb_on_wide_window_tables(ctx):
    """zk_groth16_prove_multi's deal leaves remainder pieces of 4096 .. 5041 terms; over a key of ~2^20 points (window
    multiples with c = 20: 2^19 buckets) such a piece has fewer than 2^16 digits and used to fall to the counting sort, which cannot
    scan more than 2^16 buckets (ZK_ERR_ARG).  n = 1 031 045 constraints over 5 contexts deals context 2 a 5 041-term tail of the
    B-in-G2 job (found with a Python port of the deal; the library's own plan is asserted below).  Same bytes as one context."""
    import zk_mpc_amd as Z
    n, n_ctx = 1031045, 5
    rng = O.Prng(8191)
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    others = [Z.Context(0) for _ in range(n_ctx - 1)]
    ctxs = [ctx] + others
    try:
        drs = [c.r1cs_mul_chain(n) for c in ctxs]
        pks = [c.groth16_setup(dr, *td) for c, dr in zip(ctxs, drs)]
        assert ctx.lib.zk_bases_window_bits(pks[0].query_bases("b_g2_query").h) == 20
        plan = ctx.multi_plan(pks[0], drs[0], n_ctx)
        crumbs = [(c, j, lo, m) for (c, j, lo, m) in plan if 4096 <= m <= 5041]
        assert crumbs, plan
        z = ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr()))
        r, s = mont(rng.fr()), mont(rng.fr())
        single = ctx.create_proof_dev(pks[0], drs[0], z.ptr, r, s)
        assert ctx.create_proof_multi(others, pks, drs, z.ptr, r, s) == single
        for pk in pks:
            pk.free()
        for dr in drs:
            dr.free()
        z.free()
    finally:
        for c in others:
            c.close()
