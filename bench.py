#!/usr/bin/env python3
"""bench.py -- Groth16 (BLS12-377) prove throughput on MI355X: R1CS constraints/s.

  python bench.py --gpus N --steps K --warmup W [--log-constraints L]

N = 1: local (non-MPC) prove of the SURVEY 8(d) config-2 workload: mul-chain R1CS with
       n = 2^20 - 2 constraints (QAP domain 2^20), proving key resident on the device,
       witness resident on the device when the timed region starts.
N > 1: N-party collaborative prove (additive shares, honest backend; --spdz: the malicious one), one
       party per GPU, one process per party.  Launched by torch.distributed.run -- or by nobody: without
       a launcher this script starts the N ranks itself as a child process.  The two Beaver opens are
       collectives over RCCL.  Per-GPU work is fixed (every party runs the full-size NTTs / MSMs on its
       shares: "weak"); the parties jointly produce ONE proof, so value = n * K / T.  Before anything is
       timed both transports carry tiny opens that are checked against the host-side sum.
--one-prover: ONE local prover whose five MSMs are spread over N devices inside this process
       (zk_groth16_prove_multi): total work fixed, "strong".

A step = one proof.  Timing: W untimed proofs, then exactly K proofs bracketed by barrier +
device synchronisation; max over ranks.  One JSON line on rank 0.

Outside the timed region the run checks itself: the bytes of the timed proofs are compared with the oracle's known-trapdoor
prediction (Fr arithmetic on the CPU + three scalar multiplications; `proof_matches_prediction`), for N > 1 the revealed
proof against the prediction on the summed shares, identical on every rank.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
INT_MAD_PEAK = 32.16e12  # v_mad_u64_u32 wave-lane instr/s measured on MI355X (tools/ubench_int.hip, gpurun_out/ubench_int.txt)
# v_mad_u64_u32 per XYZZ mixed addition (8M + 2S) with Fq's 14 Montgomery digits (fp29.cuh): a product is 13 x 13 limb products
# + 14 x 12 reduction products (the modulus' low limb is 1: those 14 are plain additions) = 337, a squaring 91 + 168 = 259, the
# fused double product R (Q - X3) - PPP Y1 is 2 x 169 + 168 = 506
MADS_PER_MADD_G1 = 6 * 337 + 2 * 259 + 506
MADS_PER_MADD_G2 = 10 * 2 * 506                   # per lane PAIR: each of the 10 Fq2 products is one fused double product per lane


def seeded_fr(seed: int):
    import hashlib
    import zk_mpc_amd.convert as cv
    h = hashlib.sha256(b"zkmpc-bench" + seed.to_bytes(8, "little")).digest() + hashlib.sha256(b"x" + seed.to_bytes(8, "little")).digest()
    return int.from_bytes(h[:40], "little") % cv.R_MOD


def cpu_baseline(ctx, td, sample_log: int, threads: int, full_log: int = 20):
    """Time the oracle's C restatement of the reference prover (oracle/zkref.c: arkworks' CIOS field
    arithmetic, Jacobian formulas, Pippenger with c = ln(n)+2, in-order radix-2 FFT, src/groth16.rs
    pipeline) on a bounded sample of the same workload, on this host's cores.  The proving key is the
    device's (downloaded), and the CPU proof must equal the device's proof for the same inputs."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import zkref_c as OC
    except Exception as e:  # oracle not built: report, never substitute
        return {"value": None, "unit": "constraints/s", "cores": threads, "kind": "port", "sample": "oracle unavailable: %s" % e}
    import zk_mpc_amd.convert as cv
    n = (1 << sample_log) - 2
    mont = lambda v: cv.fr_to_mont([v])[0]
    w0, w1, r_, s_ = mont(seeded_fr(100)), mont(seeded_fr(101)), mont(seeded_fr(200)), mont(seeded_fr(201))
    r1cs = ctx.r1cs_mul_chain(n)
    pk = ctx.groth16_setup(r1cs, *td)
    z = ctx.mul_chain_assignment_dev(n, w0, w1)
    gpu_proof = ctx.create_proof_dev(pk, r1cs, z.ptr, r_, s_)
    hp = OC.Pk(pk.vk_g1(0), pk.vk_g1(1), pk.vk_g1(2), pk.vk_g2(0), pk.vk_g2(1), pk.download("a_query"),
               pk.download("b_g1_query"), pk.download("b_g2_query"), pk.download("h_query"), pk.download("l_query"))
    t_all, proof_all, ph = OC.bench_mul_chain_prove(n, w0, w1, hp, r_, s_, threads)
    ok = proof_all == gpu_proof
    for o in (z, ):
        o.free()
    pk.free(); r1cs.free()
    # single thread (the reference's actual build has no rayon feature) on a BOUNDED sample: 2^17 - 2 constraints is ~11 s of one
    # core (the full 2^20 would be ~90 s); its own device proof is the check
    one_log = min(sample_log, 17)
    n1 = (1 << one_log) - 2
    r1 = ctx.r1cs_mul_chain(n1)
    pk1 = ctx.groth16_setup(r1, *td)
    z1 = ctx.mul_chain_assignment_dev(n1, w0, w1)
    gpu1 = ctx.create_proof_dev(pk1, r1, z1.ptr, r_, s_)
    hp1 = OC.Pk(pk1.vk_g1(0), pk1.vk_g1(1), pk1.vk_g1(2), pk1.vk_g2(0), pk1.vk_g2(1), pk1.download("a_query"),
                pk1.download("b_g1_query"), pk1.download("b_g2_query"), pk1.download("h_query"), pk1.download("l_query"))
    t_one, proof_one, _ = OC.bench_mul_chain_prove(n1, w0, w1, hp1, r_, s_, 1)
    ok = ok and proof_one == gpu1
    z1.free(); pk1.free(); r1.free()
    c_ref = max(3, int(__import__("math").log(max(n, 2))) + 2)        # arkworks' window: ln(n) + 2 (msm/variable_base.rs:22-26)
    windows = (253 + c_ref - 1) // c_ref
    return {"value": round(n / t_all, 1), "unit": "constraints/s", "cores": threads, "kind": "port",
            "sample": "mul-chain prove, n=2^%d-2 constraints (%s), device's proving key; %.2f s with %d threads requested "
                      "(17-way: one thread per Pippenger window; witness map %.2f s, MSMs %.2f s); single thread (the reference's "
                      "actual build: no rayon feature) on the bounded sample n=2^%d-2: %.2f s = %.0f constraints/s" % (
                          sample_log, "the benched configuration itself" if sample_log == full_log else
                          "bounded sample of the 2^%d workload" % full_log, t_all, threads, ph[0], ph[1], one_log, t_one, n1 / t_one),
            "effective_parallel_width": {"msm": min(threads, windows), "fft": min(threads, 32),
                                         "note": "the port parallelises where arkworks' `parallel` feature does: over the %d Pippenger "
                                                 "windows of an MSM (c = %d) and over butterfly chunks; the five MSMs run one after the "
                                                 "other as in the reference" % (windows, c_ref)},
            "single_thread_value": round(n1 / t_one, 1), "single_thread_sample_log": one_log, "proof_matches_device": bool(ok)}


def pk_bases(ctx, pk, which):
    """A non-owning Bases view of one proving-key query (for the standalone MSM measurement)."""
    import ctypes as C
    from zk_mpc_amd.api import Bases
    h = C.c_void_p(ctx.lib.zk_pk_query_bases(pk.h, {"a": 0, "b_g1": 1, "b_g2": 2, "h": 3, "l": 4}[which]))
    return Bases(ctx, h, 2 if which == "b_g2" else 1, owned=False)


MADS_PER_BUTTERFLY = 153          # v_mad_u64_u32 per radix-2 butterfly equivalent of the lazy-domain NTT (k_ntt_pass)


def msm_plan_windows(n: int) -> int:
    """Digits per scalar of an MSM over a PLAIN table (msm.hip::make_plan): c = round(log2 n) - 4 (- 2 up to 2^15 terms) in
    [4, 16], W = ceil(255 / c)."""
    lg = 0
    while (3 << lg) <= 2 * n:
        lg += 1
    c = min(16, max(4, lg - (2 if lg <= 15 else 4)))
    return (255 + c - 1) // c


def micro_sweeps(ctx, max_log=24, budget_s=45.0):
    """SURVEY 8(d) 'MSM micro' and 'NTT micro', outside the timed region: variable-base MSM in G1 (2^16 .. 2^24) and G2
    (2^16 .. 2^22) over plain tables (what zk_msm_g1_dev gets from a caller's bases: bases = k_i G from the device's fixed-base
    kernel, scalars uniform below 2^252), three adversarial scalar sets at 2^20, and the four transform variants for
    log n = 10 .. 24 -- each with its roofline fractions.  MSM: integer ALU (the accumulate kernel's multiply-adds, n W x 3 046
    in G1 / 10 120 per lane pair in G2, over the WHOLE call incl. sort, reduce and the host round trip).  NTT: HBM (2 x 32 N
    algorithmic bytes) and integer ALU ((N/2) log2 N butterflies x 153)."""
    import zk_mpc_amd.convert as cv2
    t_start = time.perf_counter()
    rs = np.random.RandomState(1)
    out = {"msm": [], "ntt": [], "note": "one call at a time, result on the host; plain tables (no window multiples)"}
    n_max = 1 << max_log
    a = rs.randint(0, 1 << 62, size=(n_max, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.upload(a)
    one = cv2.fr_to_mont([1])[0]

    def timed(fn, reps):
        fn(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps

    for lg in range(10, max_log + 1, 2):
        n = 1 << lg
        for inv, cos in ((0, 0), (1, 0), (0, 1), (1, 1)):
            dt = timed(lambda: ctx.ntt_dev(d.ptr, lg, inv, cos), 5 if lg >= 20 else 20)
            mads = (n // 2) * lg * MADS_PER_BUTTERFLY
            out["ntt"].append({"log_n": lg, "inverse": inv, "coset": cos, "us": round(dt * 1e6, 1),
                               "hbm_frac": round(2 * 32 * n / dt / 1e9 / HBM_PEAK_GBS, 4), "int_alu_frac": round(mads / dt / INT_MAD_PEAK, 4)})
    # (the transforms ran in place: d now holds other residues below r, which do as scalars)
    for group, top, mads_per in ((1, max_log, MADS_PER_MADD_G1), (2, max_log - 2, MADS_PER_MADD_G2)):
        if time.perf_counter() - t_start > budget_s:
            out["truncated"] = "time budget reached before G%d" % group
            break
        bases = ctx.fixed_base(d.ptr, 1 << top, group, one)
        ctx.sync()
        for lg in reversed(range(16, top + 1, 2)):          # largest first: the scratch arena is sized once
            n = 1 << lg
            for _ in range(2):
                ctx.msm_dev(bases, 0, d.ptr, n)
            dt = timed(lambda: ctx.msm_dev(bases, 0, d.ptr, n), 3 if lg >= 22 else 5)
            W = msm_plan_windows(n)
            out["msm"].append({"group": "G%d" % group, "log_n": lg, "ms": round(dt * 1e3, 3), "mscalar_per_s": round(n / dt / 1e6, 1),
                               "windows": W, "int_alu_frac": round(n * W * mads_per / dt / INT_MAD_PEAK, 4)})
        if group == 1:
            n = 1 << min(20, top)
            sets = {"all_equal": np.tile(a[12345:12346], (n, 1)), "all_zero": np.zeros((n, 4), dtype=np.uint64)}
            z01 = a[:n].copy()
            pick = rs.rand(n)
            z01[pick < 0.45] = 0
            z01[(pick >= 0.45) & (pick < 0.9)] = one
            sets["zero_one_heavy"] = z01                      # 90 % of the scalars 0 or 1, like a real witness
            small = cv2.fr_to_mont([int(v) for v in rs.randint(0, 1 << 16, size=4096)])
            sv = small[rs.randint(0, 4096, size=n)]
            sv[pick >= 0.9] = a[:n][pick >= 0.9]
            sets["small_values"] = np.ascontiguousarray(sv)   # 16-bit range-checked values with a uniform tenth
            for name, arr in sets.items():
                dd = ctx.upload(np.ascontiguousarray(arr))
                dt = timed(lambda: ctx.msm_dev(bases, 0, dd.ptr, n), 3)
                out["msm"].append({"group": "G1", "log_n": 20, "scalars": name, "ms": round(dt * 1e3, 3), "mscalar_per_s": round(n / dt / 1e6, 1)})
                dd.free()
        bases.free()
    d.free()
    out["seconds"] = round(time.perf_counter() - t_start, 1)
    return out


def natural_domain_leg(ctx, log_constraints, td, threads):
    """The reference's natural sizing (src/groth16.rs:256-257): n = 2^L constraints => QAP domain 2^(L+1).  Three proofs over a
    queue, the last one checked against the known-trapdoor prediction."""
    import zk_mpc_amd.convert as cv2
    mont = lambda v: cv2.fr_to_mont([v])[0]
    n = 1 << log_constraints
    r1cs = ctx.r1cs_mul_chain(n)
    pk = ctx.groth16_setup(r1cs, *td)
    zs = [ctx.mul_chain_assignment_dev(n, mont(seeded_fr(300 + q)), mont(seeded_fr(310 + q))) for q in range(2)]
    r_, s_ = mont(seeded_fr(320)), mont(seeded_fr(321))
    proof = None
    for i in range(2):
        ctx.groth16_hint_next_dev(zs[(i + 1) % 2].ptr)
        ctx.create_proof_dev(pk, r1cs, zs[i % 2].ptr, r_, s_)
    ctx.sync()
    K = 6
    t0 = time.perf_counter()
    for i in range(K):
        ctx.groth16_hint_next_dev(zs[(i + 1) % 2].ptr)
        proof = ctx.create_proof_dev(pk, r1cs, zs[i % 2].ptr, r_, s_)
    ctx.sync()
    dt = (time.perf_counter() - t0) / K
    ctx.groth16_hint_next_dev(None)
    t0 = time.perf_counter()
    iso = ctx.create_proof_dev(pk, r1cs, zs[(K - 1) % 2].ptr, r_, s_)
    ctx.sync()
    t_iso = time.perf_counter() - t0
    zarr = ctx.download(zs[(K - 1) % 2], (n + 3, 4))
    want, note = predict_proof(ctx, n, zarr, td, r_, s_, threads)
    out = {"constraints": n, "domain_log": r1cs.domain_log, "ms_per_proof": round(dt * 1e3, 3), "constraints_per_s": round(n / dt, 1),
           "isolated_ms": round(t_iso * 1e3, 3), "proof_matches_prediction": None if want is None else bool(want == proof and iso == proof)}
    if want is None:
        out["note"] = note
    for z in zs:
        z.free()
    pk.free(); r1cs.free()
    return out


def bool_chain_system(n: int, seed: int, frac_bool: float = 0.9):
    """A witness shaped like the reference's circuits (boolean-heavy: BitDecomposition / SmallerThan / Pedersen gadgets,
    docs/benchmark.md:45-58): the variable layout of the mul-chain system (z = [1, public, w_0 .. w_n]), but the first 90 % of the rows
    are booleanity checks w_j * w_j = w_j over random bits and only the tail is a multiplication chain over uniform field
    elements, w_j * w_{j+1} = w_{j+2}, ending in the public input.  Returns (CSR a, b, c; the assignment in the reference's
    Montgomery form) -- plain data for zk_r1cs_upload on one side and the oracle on the other."""
    import zk_mpc_amd.convert as cv
    nb = int(n * frac_bool)
    one = cv.fr_to_mont([1])[0]
    rs = np.random.RandomState(seed)
    bits = rs.randint(0, 2, size=nb)
    w = [0] * (n + 2)
    w[nb], w[nb + 1] = seeded_fr(seed * 2 + 1), seeded_fr(seed * 2 + 2)
    for j in range(nb, n):
        w[j + 2] = w[j] * w[j + 1] % cv.R_MOD
    z = np.zeros((n + 3, 4), dtype=np.uint64)
    z[0] = one
    z[2:2 + nb][bits == 1] = one
    tail = cv.fr_to_mont(w[nb:n + 2])
    z[2 + nb:2 + n + 1] = tail[:n + 1 - nb]
    z[1] = tail[n + 1 - nb]                                   # the public input w_{n+1}
    rp = np.arange(n + 1, dtype=np.uint32)
    j = np.arange(n, dtype=np.int64)
    idx = lambda v: np.where(v <= n, 2 + v, 1).astype(np.uint32)
    ones = np.tile(one, (n, 1))
    a = (rp, idx(j), ones)
    b = (rp, idx(np.where(j < nb, j, j + 1)), ones)
    c = (rp, idx(np.where(j < nb, j, j + 2)), ones)
    return a, b, c, z, nb


def boolean_heavy_leg(ctx, log_constraints, td, threads):
    """The same prover on a boolean-heavy assignment (bool_chain_system: 90 % of the variables are 0 or 1), beside the uniform
    headline: arkworks keeps a unit-scalar fast path for exactly this (ec/src/msm/variable_base.rs:45-49).  Proofs back to back
    over two different assignments, the last one checked against the known-trapdoor prediction."""
    n = (1 << log_constraints) - 2
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    systems = [bool_chain_system(n, 7000 + q) for q in range(2)]
    a, b, c = systems[0][:3]
    r1cs = ctx.r1cs_upload(2, n + 1, a, b, c)
    pk = ctx.groth16_setup(r1cs, *td)
    import zk_mpc_amd.convert as cv2
    mont = lambda v: cv2.fr_to_mont([v])[0]
    zs = [ctx.upload(sy[3]) for sy in systems]
    r_, s_ = mont(seeded_fr(420)), mont(seeded_fr(421))
    for i in range(3):
        ctx.groth16_hint_next_dev(zs[(i + 1) % 2].ptr)
        ctx.create_proof_dev(pk, r1cs, zs[i % 2].ptr, r_, s_)
    ctx.sync()
    K = 8
    per = []
    proof = None
    t0 = time.perf_counter()
    for i in range(K):
        ts = time.perf_counter()
        ctx.groth16_hint_next_dev(zs[(i + 1) % 2].ptr)
        proof = ctx.create_proof_dev(pk, r1cs, zs[i % 2].ptr, r_, s_)
        per.append(time.perf_counter() - ts)
    ctx.sync()
    dt = (time.perf_counter() - t0) / K
    ctx.groth16_hint_next_dev(None)
    iso = []
    for i in range(5):
        t1 = time.perf_counter()
        p_iso = ctx.create_proof_dev(pk, r1cs, zs[(K - 1) % 2].ptr, r_, s_)
        ctx.sync()
        iso.append(time.perf_counter() - t1)
    out = {"workload": "bool-chain R1CS: %d booleanity rows b*b = b over random bits + a %d-row multiplication chain over uniform "
                       "elements; n=2^%d-2 constraints, QAP domain 2^%d" % (systems[0][4], n - systems[0][4], log_constraints, r1cs.domain_log),
           "constraints": n, "boolean_fraction_of_assignment": round(systems[0][4] / (n + 3), 3),
           "ms_per_proof": round(dt * 1e3, 3), "median_ms_per_proof": round(float(np.median(per)) * 1e3, 3),
           "constraints_per_s": round(n / dt, 1), "isolated_ms": round(float(np.median(iso)) * 1e3, 3)}
    try:
        import zkref_c as OC
        cr = OC.R1cs(2, n + 1, a, b, c)
        zarr = systems[(K - 1) % 2][3]
        want = OC.groth16_predict(cr, np.stack(td), zarr, OC.witness_map(cr, zarr, threads), r_, s_)
        out["proof_matches_prediction"] = bool(want == proof and p_iso == proof)
    except Exception as e:
        out["proof_matches_prediction"] = None
        out["note"] = "oracle unavailable: %r" % (e,)
    for z in zs:
        z.free()
    pk.free(); r1cs.free()
    return out


def other_workloads(ctx, log_h=20):
    """Rows a14 / a15 beside the headline metric (not part of `value`): Marlin AHP prover + KZG10 commitments / openings on
    the same mul-chain family (BASELINE config 4 shape), and the SHE ciphertext product.  Same code as
    tools/bench_marlin.py / tools/bench_she.py, fewer repetitions."""
    import numpy as np
    from zk_mpc_amd import marlin as DM
    import zk_mpc_amd.convert as cv2
    out = {}
    rng = np.random.default_rng(11)
    m = DM.HostField.m

    def rand_fr(k):
        a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    try:
        from zk_mpc_amd.api import Rng
        n = (1 << log_h) - 3
        ni, nw, a, b, c = DM.mul_chain_system(ctx, n)
        index = DM.Index(ctx, ni, nw, a, b, c)
        H = index.dom_h.size
        beta_srs, g_k, gg_k, h_k = 0x1234567, 3, 7, 5
        srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(index) + 5, beta_srs, g_k, gg_k)     # resident SRS with window multiples
        keys = DM.IndexKeys(index, srs)                                                  # Marlin::index: commitments to the 12 index polynomials
        z = ctx.mul_chain_assignment_dev(n, m(3), m(5))
        ctx.pooling = True
        seed = bytes(range(32))
        proof_bytes = DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)     # zk_marlin_prove: one C-ABI call
        ctx.sync()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            proof_bytes = DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        ctx.pooling = False
        ctx.drop_pool()
        out["marlin"] = {"workload": "Marlin::prove (AHP rounds + MarlinKZG10 commitments with hiding and degree bounds + Fiat-Shamir "
                                     "transcript + open_combinations), mul-chain R1CS, |H| = |K| = 2^%d, index key and SRS resident, "
                                     "prover randomness from a ChaCha20 rng (the mask polynomial sampled on the device)" % log_h,
                         "constraints": n, "ms_per_proof": round(dt * 1e3, 2), "constraints_per_s": round(n / dt, 1),
                         "entry_point": "zk_marlin_prove", "proof_bytes": len(proof_bytes)}
        out["marlin"].update(marlin_oracle_verdict(ctx, index, keys, srs, z, proof_bytes, beta_srs, g_k, gg_k, h_k))
        del index, srs, keys, z
    except Exception as e:  # the headline line must not depend on this leg
        ctx.pooling = False
        out["marlin"] = {"error": repr(e)}
    try:
        out["marlin_dense"] = marlin_dense_leg(ctx, min(log_h, 20) - 2)
    except Exception as e:
        ctx.pooling = False
        out["marlin_dense"] = {"error": repr(e)}
    try:
        N, batch = 1024, 2048

        def rnd753(k):
            a = rng.integers(0, 1 << 63, size=(k, 12), dtype=np.uint64)
            a[:, 11] &= np.uint64((1 << 46) - 1)
            return a
        x, y = ctx.upload(rnd753(batch * 3 * N)), ctx.upload(rnd753(batch * 3 * N))
        o = ctx.alloc(batch * 3 * N * 96)
        ctx.ciphertext_mul_dev(x.ptr, y.ptr, o.ptr, N, batch)
        ctx.sync()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctx.ciphertext_mul_dev(x.ptr, y.ptr, o.ptr, N, batch)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        mmuls = (7 * (N // 2) * 10 + 7 * N) * batch
        out["she"] = {"workload": "Ciphertext::mul in F_q[X]/(X^N+1), q = MNT4-753 base prime, N = %d, batch %d" % (N, batch),
                      "products_per_s": round(batch / dt, 1), "fq753_modmul_per_s": round(mmuls / dt, 1),
                      "frac_of_int_mad_peak": round(mmuls / dt * 2 * 26 * 26 / INT_MAD_PEAK, 3)}
    except Exception as e:
        out["she"] = {"error": repr(e)}
    return out


def marlin_dense_leg(ctx, log_h: int):
    """Marlin::prove on a CIRCUIT-SHAPED system (tools/synth_r1cs.py): 3-5 / 1-2 / 2-4 terms per row of A / B / C, non-unit
    coefficients, 7 public inputs, so that AHPForR1CS::index sizes K = 4 |H| (arkworks/marlin/src/ahp/indexer.rs:138-181) and the
    round-3 interpolation domain is 16 |H| -- what the reference's circuits look like under Marlin, where the mul-chain row above has
    |K| = |H|.  |H| = 2^log_h.  The emitted bytes go through the oracle's Marlin::verify."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synth_r1cs as S
    from zk_mpc_amd import marlin as DM
    from zk_mpc_amd.api import Rng
    import zk_mpc_amd.convert as cv2
    t0 = time.time()
    ni, nw, ra, rb, rc, zvals = S.sized_for_domain(log_h, 20260)
    nv, nc = ni + nw, len(ra)
    pad = [[]] * (nv - nc)                                     # make_matrices_square: dummy constraints 0 * 0 = 0
    a, b, c = (DM.Csr.from_rows(m + pad) for m in (ra, rb, rc))
    t_gen = time.time() - t0
    index = DM.Index(ctx, ni, nw, a, b, c)
    beta_srs, g_k, gg_k, h_k = 0x7654321, 3, 7, 5
    srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(index) + 5, beta_srs, g_k, gg_k)
    keys = DM.IndexKeys(index, srs)
    z = ctx.upload(cv2.fr_to_mont(zvals))
    ctx.pooling = True
    seed = bytes(range(32))
    proof_bytes = DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)
    ctx.sync()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        proof_bytes = DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    ctx.pooling = False
    ctx.drop_pool()
    rec = {"workload": "Marlin::prove, circuit-shaped R1CS (3-5 / 1-2 / 2-4 terms per row, non-unit coefficients, 7 public inputs): "
                       "|H| = 2^%d, |K| = 2^%d, round-3 domain 2^%d" % (log_h, index.dom_k.log, index.dom_b.log),
           "constraints": nc, "non_zero": [a.nnz, b.nnz, c.nnz], "ms_per_proof": round(dt * 1e3, 2), "constraints_per_s": round(nc / dt, 1),
           "entry_point": "zk_marlin_prove", "proof_bytes": len(proof_bytes), "generate_system_s": round(t_gen, 1)}
    rec.update(marlin_oracle_verdict(ctx, index, keys, srs, z, proof_bytes, beta_srs, g_k, gg_k, h_k))
    return rec


def trait_path_leg(log_d: int, resident_ms: float, proofs: int = 10):
    """The reference-shaped boundary, composed and timed (outside the timed region): examples/host_trait_groth16.cpp is
    src/groth16.rs:68-183,240-306 over the trait-shaped entry points only -- zk_fr_fft_in_place x7, zk_fr_batch_product_in_place,
    zk_fr_divide_by_vanishing_on_coset_in_place, zk_msm_g1 x4, zk_msm_g2 x1, host group helpers -- with the proving key in HOST
    vectors and freshly allocated host Vecs per proof.  A child process with its own context (this process keeps its key resident:
    the two do not share anything but the device).  `lib` = time inside library calls; `total` adds the caller's own single-thread
    scalar loops (evaluate_constraint, ab -= c), as the reference runs them.  First proof: every base table crosses PCIe once;
    then the window multiples are built by a builder thread in the caller's gaps (no call pays for them); the last three: steady state.  The proofs' bytes are checked against the
    prediction by tests/test_gpu_trait_path.py (2^10, 2^16, 2^20); here they must agree with each other."""
    import hashlib
    import subprocess
    exe = os.path.join(ROOT, "examples", "_bin", "host_trait_groth16")
    if not os.path.exists(exe):
        return {"error": "examples/_bin/host_trait_groth16 not built (python -c 'import __graft_entry__ as g; g.build()')"}
    out = {"entry_points": "zk_fr_fft_in_place x7, zk_fr_batch_product_in_place, zk_fr_divide_by_vanishing_on_coset_in_place, zk_msm_g1 x4, zk_msm_g2 x1",
           "log_domain": log_d, "resident_api_ms": round(resident_ms, 3)}
    for mode in (("cache", "packed"), ("cache", "strided"), ("cache", "strided", "trust"), ("nocache", "packed")):
        try:
            r = subprocess.run([exe, str(log_d), str(proofs if mode[0] == "cache" else 3), *mode], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                out["_".join(mode)] = {"error": r.stderr[-400:]}
                continue
            lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
            pr, last = lines[:-1], lines[-1]
            # steady state = the last three calls: the window multiples of large tables are built in the caller's gaps over the
            # first handful of proofs (lib_ms_every_call shows the approach)
            steady = pr[-3:] if len(pr) > 5 else pr[-1:]
            med = lambda k: round(float(np.median([p["ms"][k] for p in steady])), 3)
            rec = {"first_call_ms": pr[0]["ms"], "second_call_ms": {k: pr[1]["ms"][k] for k in ("total", "lib")} if len(pr) > 1 else None,
                   "third_call_ms": {k: pr[2]["ms"][k] for k in ("total", "lib")} if len(pr) > 2 else None,
                   "lib_ms_every_call": [p["ms"]["lib"] for p in pr],
                   "steady_state_ms": {k: med(k) for k in pr[0]["ms"]}, "steady_state_over": len(steady),
                   "same_bytes_every_proof": len(set(p["proof"] for p in pr)) == 1, "proof_sha": hashlib.sha256(bytes.fromhex(pr[0]["proof"])).hexdigest()[:16],
                   "cache": last["cache"]}
            rec["lib_vs_resident_api"] = round(rec["steady_state_ms"]["lib"] / resident_ms, 2) if resident_ms else None
            out["_".join(mode)] = rec
        except Exception as e:
            out["_".join(mode)] = {"error": repr(e)}
    # bytes the trait shape moves per proof (host slices in and out): 7 transforms + divide in place, the product's 2 in / 1 out, 5 scalar vectors
    D = 1 << log_d
    out["pcie_bytes_per_proof"] = {"fft_and_divide": 8 * 64 * D, "batch_product": 96 * D, "msm_scalars": 5 * 32 * D, "total": (8 * 64 + 96 + 160) * D}
    return out


def trait_path_collab_leg(log_d: int, resident_ms: float, proofs: int = 7):
    """The same boundary under E = MpcPairingEngine: examples/host_trait_collab_groth16.cpp is create_proof +
    witness_map (src/groth16.rs:68-183,240-306) over Vec<MpcField<Fr, S>> / &[MpcG1Affine] in their enum layouts -- P parties as
    threads of one process sharing ONE GPU here (one GPU per party on a node), each with its own context, key copy and shares --
    calling nothing but zk_mpc_fft_in_place x7, zk_mpc_batch_product_in_place (Beaver: two vector opens through the transport
    vtable), zk_mpc_divide_by_vanishing_on_coset_in_place, zk_mpc_msm_g1 x4 / _g2 x1 and the host group helpers.  `lib` = time inside
    those calls on party 0 (`max_lib`: the slowest party); the parties' own scalar loops over MpcField elements and the O(1) group
    tail with its nine small opens are the caller's.  The bytes are held to the prediction on the summed shares by
    tests/test_gpu_trait_path.py; here they must agree over the proofs."""
    import hashlib
    import subprocess
    exe = os.path.join(ROOT, "examples", "_bin", "host_trait_collab_groth16")
    if not os.path.exists(exe):
        return {"error": "examples/_bin/host_trait_collab_groth16 not built (python -c 'import __graft_entry__ as g; g.build()')"}
    out = {"entry_points": "zk_mpc_fft_in_place x7, zk_mpc_batch_product_in_place, zk_mpc_divide_by_vanishing_on_coset_in_place, zk_mpc_msm_g1 x4, zk_mpc_msm_g2 x1",
           "resident_api_ms_local_prove": round(resident_ms, 3), "note": "every party on cuda:0 of a one-GPU box: the parties' device work is serialised, "
           "so P-party times here are an upper bound for one GPU per party"}
    runs = [("additive_p3", log_d, 3, ("additive", "tagfirst", "verify")), ("additive_p3_trusted_hits", log_d, 3, ("additive", "tagfirst", "trust")),
            ("additive_p1", log_d, 1, ("additive", "tagfirst", "verify")), ("spdz_p2", min(log_d, 18), 2, ("spdz", "tagfirst", "verify"))]
    if log_d > 18:
        runs.append(("additive_p3_2p18", 18, 3, ("additive", "tagfirst", "verify")))
    for name, ld, parties, mode in runs:
        try:
            # (sync2: the window-multiple builds are waited for after the second proof, outside the laps -- P parties share ONE GPU
            # here and their builders find no quiet device; on a GPU per party they finish in the callers' own time, as trait_path shows)
            r = subprocess.run([exe, str(ld), str(proofs), str(parties), *mode, "sync2"], capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                out[name] = {"error": r.stderr[-400:]}
                continue
            lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
            pr, last = lines[:-1], lines[-1]
            steady = pr[2:] if len(pr) > 3 else pr[-1:]
            med = lambda k: round(float(np.median([p["ms"][k] for p in steady])), 3)
            out[name] = {"log_domain": ld, "parties": parties, "shares": last["shares"], "hits": last["hits"], "element_bytes": last["element_bytes"],
                         "first_call_ms": {k: pr[0]["ms"][k] for k in ("total", "lib")},
                         "second_call_ms": {k: pr[1]["ms"][k] for k in ("total", "lib")} if len(pr) > 1 else None,
                         "lib_ms_every_call": [p["ms"]["lib"] for p in pr],
                         "steady_state_ms": {k: med(k) for k in pr[0]["ms"]}, "steady_state_over": len(steady),
                         "steady_state_max_lib_ms": round(float(np.median([p["ms_max_lib"] for p in steady])), 3),
                         "beaver_bytes_sent_per_party": pr[0]["beaver_bytes_sent"],
                         "same_bytes_every_proof": len(set(p["proof"] for p in pr)) == 1,
                         "proof_sha": hashlib.sha256(bytes.fromhex(pr[0]["proof"])).hexdigest()[:16], "cache_party0": last["cache"]}
        except Exception as e:
            out[name] = {"error": repr(e)}
    D = 1 << log_d
    out["pcie_bytes_per_proof_per_party_additive"] = {"fft_and_divide": 8 * 64 * D, "batch_product": 96 * D, "msm_scalars": 5 * 32 * D,
                                                      "verified_hits": 96 * (2 * (D - 1) + 2 * D) + 192 * D, "total": (8 * 64 + 96 + 160) * D + 96 * (4 * D - 2) + 192 * D}
    return out


def predict_proof(ctx, n, zarr, td, r_, s_, threads):
    """The oracle's known-trapdoor prediction of the 192 proof bytes for the mul-chain system (checker only: never in the
    timed region).  Returns (bytes or None, note)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import zkref_c as OC
    except Exception as e:
        return None, "oracle unavailable: %s" % e
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    h = OC.witness_map(cr, zarr, threads)
    return OC.groth16_predict(cr, np.stack(td), zarr, h, r_, s_), "ok"


def marlin_oracle_verdict(ctx, index, keys, srs, z, proof_bytes, beta_srs, g_k, gg_k, h_k):
    """Checker (outside every timed region): the oracle's Marlin::verify on the BYTES the prover emitted -- CanonicalDeserialize of
    the proof (points through GroupAffine::deserialize: on the curve, in the subgroup), the transcript re-derived, the two
    sum-check combinations and one KZG pairing equation per query point; a wrong public input must be rejected."""
    import zk_mpc_amd.convert as cv
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import marlin_full_ref as MF
        import marlin_ref as MR
        import zkref as O

        class PP:
            pass
        pp = PP()
        pp.beta = beta_srs
        pp.g, pp.gamma_g, pp.h = O.g1_mul(O.G1_GEN, g_k), O.g1_mul(O.G1_GEN, gg_k), O.g2_mul(O.G2_GEN, h_k)
        pp.beta_h = O.g2_mul(pp.h, beta_srs)
        info = MR.IndexInfo(index.num_constraints, index.num_non_zero, index.num_instance)
        info.num_variables, info.num_constraints, info.num_non_zero = index.num_variables, index.num_constraints, index.num_non_zero
        okeys = MF.Keys(info, pp, max_degree=srs.max_degree, index_comms={l: keys.index_comms[l].comm_aff for l in MF.INDEX_LABELS})
        t1 = time.time()
        as_oracle = MF.proof_deserialize(proof_bytes)
        assert as_oracle.serialize() == proof_bytes
        pub = cv.fr_from_mont(ctx.download(z.ptr + 32, (index.num_instance - 1, 4)))
        return {"oracle_verifier_accepts": bool(MF.verify(okeys, pub, as_oracle)),
                "oracle_verifier_rejects_wrong_input": bool(not MF.verify(okeys, [(pub[0] + 1) % O.R_MOD] + pub[1:], as_oracle)),
                "verified": "the emitted bytes, deserialised by the oracle", "verify_seconds": round(time.time() - t1, 1)}
    except Exception as e:
        return {"oracle_verifier_accepts": None, "verifier_error": repr(e)}


def marlin_bench(args, ctx, dist, rank, world, real_stdout):
    """--marlin: Marlin::prove on the mul-chain system with |H| = |K| = 2^L (BASELINE config 4: one GPU, zk_marlin_prove;
    config 5's shape with --gpus N [--spdz]: the N-party collaborative prover, every party the full-size rounds and MSMs on its
    shares).  A step = one proof; the last proof is checked by the oracle's Marlin::verify outside the timed region."""
    import types
    import torch
    import zk_mpc_amd.convert as cv
    from zk_mpc_amd import marlin as DM
    from zk_mpc_amd.api import Rng
    m = DM.HostField.m
    n = (1 << args.log_constraints) - 3
    ni, nw, a, b, c = DM.mul_chain_system(ctx, n)
    index = DM.Index(ctx, ni, nw, a, b, c)
    beta_srs, g_k, gg_k, h_k = 0x1234567, 3, 7, 5
    t0 = time.time()
    srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(index) + 5, beta_srs, g_k, gg_k)
    keys = DM.IndexKeys(index, srs)
    t_setup = time.time() - t0
    z = ctx.mul_chain_assignment_dev(n, m(3), m(5))
    seed = bytes((rank * 17 + i) & 0xff for i in range(32))
    ctx.pooling = True
    if dist is None:
        def step():
            return DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)
    else:
        from zk_mpc_amd import mpc
        party = make_party(mpc.SpdzParty if args.spdz else mpc.Party, mpc, ctx, dist, torch, getattr(args, 'data_group', None))
        shape = types.SimpleNamespace(num_instance=ni, num_witness=nw)
        z0 = party.share_assignment_dev(z, shape, seed=1234)
        keep = [party._keep]
        zl = [z0]
        if args.spdz:
            zl.append(party.share_assignment_dev(z, shape, seed=4321))
            keep.append(party._keep)
        zb = [types.SimpleNamespace(ptr=p) for p in zl]

        def step():      # one library call per proof (zk_marlin_prove_shared[_spdz])
            if args.spdz:
                return party.marlin_prove_shared_spdz_native(keys, (zb[0], zb[1]), Rng.from_seed(seed, 20), mask_on_device=True)
            return party.marlin_prove_shared_native(keys, zb[0], Rng.from_seed(seed, 20), mask_on_device=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()
    preflight = ranks_seen = None
    if dist is not None and world > 1:
        preflight = preflight_opens(ctx, dist, party, torch, args.transport)
        ranks_seen = comm_report(ctx, dist, party, torch)
    for _ in range(2 + args.warmup):
        proof = step()
    sent0 = int(party.bytes_sent) if dist is not None else 0
    if dist is not None:
        party.be.open_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    barrier()
    dt = time.perf_counter() - t0
    opens_timed = party.be.open_stats(args.steps) if dist is not None else None
    sent_timed = (int(party.bytes_sent) - sent0) if dist is not None else 0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)          # over the control plane (gloo)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    proof_bytes = proof
    verdict = marlin_oracle_verdict(ctx, index, keys, srs, z, proof_bytes, beta_srs, g_k, gg_k, h_k)
    K = args.steps
    out = {"metric": "R1CS constraints/sec (prove), Marlin/KZG10 BLS12-377", "value": round(n * K / dt, 1), "unit": "constraints/s",
           "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32x13 / u32x9 (29-bit limbs, int64 accumulate)", "data": "synthetic",
           "config": {"workload": "Marlin::prove, mul-chain R1CS, |H| = |K| = 2^%d, %s" % (
               args.log_constraints, "local prove (zk_marlin_prove)" if dist is None else "%d-party %s collaborative prove" % (
                   world, "SPDZ" if args.spdz else "additive-share")), "constraints": n, "parties": world},
           "proof_constraints_per_s": round(n * K / dt, 1), "proof_bytes": len(proof_bytes),
           "setup_s": round(t_setup, 2), "proof_sha": __import__("hashlib").sha256(proof_bytes).hexdigest()[:16], **verdict}
    out["hbm_in_use_gb"] = hbm_in_use_gb()
    if dist is not None:
        out["bytes_sent_per_party"] = int(party.bytes_sent)
        out["bytes_sent_per_party_per_proof"] = int(sent_timed // max(K, 1))
        out["aggregate_constraint_shares_per_s"] = round(n * K / dt * world, 1)
        out["opens_in_timed_proofs"] = opens_timed
        out["prover_entry"] = "zk_marlin_prove_shared_spdz" if args.spdz else "zk_marlin_prove_shared"
        out["preflight_opens"] = preflight
        out["ranks"] = ranks_seen
        out["rccl_ranks_seen"] = None if ranks_seen is None else {
            "torch_distributed": dist.get_world_size() if args.transport == "nccl" else 0,
            "zk_comm": max([r.get("zk_comm", {}).get("n_ranks", 0) for r in ranks_seen] +
                           [q["zk_comm"].get("n_ranks", 0) for q in (preflight or []) if "zk_comm" in q]),
            "devices": sorted(set((r.get("device_uuid") or r["device"]) for r in ranks_seen))}
        out["transport"] = ("RCCL (torch.distributed nccl)" if args.transport == "nccl" else
                            "gloo, opens staged through host memory" + (", every party on cuda:0 (functional run, not a measurement)" if args.one_gpu else ""))
        out["transport_fallback"] = getattr(args, "transport_fallback", None)
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def one_prover_bench(args, real_stdout):
    """--one-prover: ONE local prover whose five MSMs are spread over --gpus N devices (zk_groth16_prove_multi; SURVEY 8e, second
    level), all inside this one process -- no process group, no collective: the assignment goes out by peer copies, partial sums
    come back as points.  Total work is fixed as N grows ("strong").  --one-gpu puts every context on cuda:0 (a functional run)."""
    import hashlib
    import torch
    import zk_mpc_amd as Z
    import zk_mpc_amd.convert as cv
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libzkmpc_hip has no CPU path")
    N = args.gpus
    ndev = torch.cuda.device_count()
    if not args.one_gpu and ndev < N:
        sys.exit("bench.py --one-prover --gpus %d: only %d device(s) visible (add --one-gpu for a functional run on one)" % (N, ndev))
    mont = lambda v: cv.fr_to_mont([v])[0]
    n = (1 << args.log_constraints) - 2
    ctxs = [Z.Context(0 if args.one_gpu else d) for d in range(N)]
    td = [mont(seeded_fr(i)) for i in range(1, 8)]
    t0 = time.time()
    r1css = [c.r1cs_mul_chain(n) for c in ctxs]
    pks = [c.groth16_setup(r, *td) for c, r in zip(ctxs, r1css)]
    t_setup = time.time() - t0
    Q = max(1, args.queue)
    zs = [ctxs[0].mul_chain_assignment_dev(n, mont(seeded_fr(100 + 10 * q)), mont(seeded_fr(101 + 10 * q))) for q in range(Q)]
    rs = [(mont(seeded_fr(200 + 10 * q)), mont(seeded_fr(201 + 10 * q))) for q in range(Q)]
    last = {}

    def step(i):
        q = i % Q
        last[q] = ctxs[0].create_proof_multi(ctxs[1:], pks, r1css, zs[q].ptr, *rs[q])
        return last[q]
    it = 0
    for _ in range(2 + args.warmup):
        step(it); it += 1
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        proof = step(it); it += 1
        per.append(time.perf_counter() - ts)
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    single = [ctxs[0].create_proof_dev(pks[0], r1css[0], zs[q].ptr, *rs[q]) for q in sorted(last)]
    t1 = time.perf_counter()
    for q in sorted(last):
        ctxs[0].create_proof_dev(pks[0], r1css[0], zs[q].ptr, *rs[q])
    ctxs[0].sync()
    t_single = (time.perf_counter() - t1) / len(last)
    pred = None
    if not args.no_predict:
        q = sorted(last)[-1]
        want, note = predict_proof(ctxs[0], n, ctxs[0].download(zs[q], (n + 3, 4)), td, rs[q][0], rs[q][1], os.cpu_count() or 1)
        pred = None if want is None else bool(want == last[q])
    K = args.steps
    out = {"metric": "R1CS constraints/sec (prove), Groth16 BLS12-377", "value": round(n * K / dt, 1), "unit": "constraints/s",
           "n_gpus": N, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
           "ms_per_step_median": round(float(np.median(per)) * 1e3, 3), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u32x13 (29-bit limbs, int64 accumulate)", "data": "synthetic",
           "config": {"workload": "mul-chain R1CS, n=2^%d-2 constraints, ONE local prover over %d contexts (zk_groth16_prove_multi)%s"
                                  % (args.log_constraints, N, ", every context on cuda:0 (functional run, not a measurement)" if args.one_gpu else ""),
                      "constraints": n, "parties": 1, "contexts": N},
           "plan": ctxs[0].multi_plan(pks[0], r1css[0], N),
           "plan_columns": ["context", "job (0 = B in G2, 1 = A, 2 = B in G1, 3 = L, 4 = H)", "first term", "terms"],
           "equals_single_context_proof": bool(all(a == last[q] for a, q in zip(single, sorted(last)))),
           "single_context_isolated_ms": round(t_single * 1e3, 3),
           "proof_matches_prediction": pred, "setup_s": round(t_setup, 2), "proof_sha": hashlib.sha256(proof).hexdigest()[:16],
           "devices": [int(c.device) for c in ctxs], "hbm_in_use_gb": hbm_in_use_gb()}
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def hbm_in_use_gb():
    """Device memory held by this process when the line is written.  Scratch arenas and key tables are grow-only, so this is the
    high-water mark of the run up to freed temporaries."""
    try:
        import torch
        free, total = torch.cuda.mem_get_info()
        return round((total - free) / 1e9, 2)
    except Exception:
        return None


def launch_ranks(n: int, real_stdout: int) -> int:
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child process (the launch line
    the driver uses for N > 1), relay the one JSON line rank 0 prints, return the launcher's exit code.  The rendezvous is on
    127.0.0.1 at a port the kernel has just handed out."""
    import random
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's driver only supports dmabuf IPC (RCCL across processes)

    def pick_port():
        # below the kernel's range for outgoing connections (32768 .. 60999): a port handed out by bind(0) can be taken again, as a
        # SOURCE port of some connection, between the probe and the child's listen (seen once: EADDRINUSE in the TCPStore)
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
    for attempt in range(3):
        port = pick_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        out, err = proc.communicate()
        sys.stderr.write(err.decode(errors="replace"))
        if proc.returncode != 0 and b"EADDRINUSE" in err and not any(ln.startswith(b"{") for ln in out.splitlines()):
            sys.stderr.write("bench.py: rendezvous port %d was taken; another one (attempt %d)\n" % (port, attempt + 2))
            continue
        break
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        try:
            rec = json.loads(lines[-1])
            rec["launched_by"] = "bench.py itself (child torch.distributed.run, %d ranks, 127.0.0.1:%d)" % (n, port)
            os.write(real_stdout, (json.dumps(rec) + "\n").encode())
        except ValueError:
            os.write(real_stdout, (lines[-1] + "\n").encode())
    elif proc.returncode == 0:
        sys.stderr.write("bench.py: the %d-rank child printed no JSON line\n" % n)
        return 1
    return proc.returncode


def open_transport(args, torch, dist, rank, world, local_rank):
    """Process groups of an N-party run.  The DEFAULT group is always gloo: the control plane (barriers, the agreement on
    verdicts, object broadcasts, the max over ranks of the timed region).  With --transport nccl the share vectors travel over a
    second group on RCCL; it is created and made to carry one all-reduce HERE, before anything else, and the ranks agree on the
    outcome over gloo: if RCCL does not come up on every rank (an exception, a wrong sum), the run goes on with the gloo group
    as its data plane too and says so in its line (`transport_fallback`) instead of ending without a number.  A rank that HANGS
    inside RCCL cannot be rescued from in here: the group's timeout ends the job.
    Returns (data_group or None, reason or None); sets args.transport to what is actually used."""
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.transport != "nccl":
        return None, None
    err, grp = None, None
    try:
        grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180), device_id=torch.device("cuda", local_rank))
        t = torch.full((4,), float(rank + 1), device="cuda", dtype=torch.float64)
        dist.all_reduce(t, group=grp)
        torch.cuda.synchronize()
        if float(t[0].item()) != world * (world + 1) / 2:
            err = "RCCL all-reduce over %d ranks returned %r on rank %d" % (world, float(t[0].item()), rank)
    except Exception as e:
        err = "rank %d: %r" % (rank, e)
    box = [None] * world
    dist.all_gather_object(box, err)
    bad = [e for e in box if e]
    if not bad:
        return grp, None
    args.transport = "gloo"
    sys.stderr.write("bench.py: RCCL did not come up (%s): continuing over gloo\n" % bad[0])
    return None, "RCCL group failed on %d of %d ranks (first: %s); the opens are staged through host memory over gloo" % (len(bad), world, bad[0][:300])


def make_party(cls, mpc, ctx, dist, torch, data_group):
    """The party over the run's data plane: RCCL group (device buffers) when there is one, else the gloo default group."""
    if data_group is not None:
        return cls(ctx, net=mpc.DistNet(dist, torch.device("cuda", ctx.device), group=data_group))
    return cls(ctx, dist)


def comm_report(ctx, dist, party, torch):
    """What actually carried the ranks of an N > 1 run, gathered over the process group: one entry per rank with its device and
    -- when the library's own communicator is up (ZK_TRANSPORT=native) -- the rank count RCCL itself reports."""
    grp = getattr(party.net, "group", None)
    info = {"rank": dist.get_rank(), "device": int(ctx.device), "dist_world_size": dist.get_world_size(grp),
            "dist_backend": dist.get_backend(grp), "control_plane": dist.get_backend()}
    try:
        if torch.cuda.is_available():
            info["device_name"] = torch.cuda.get_device_name(ctx.device)
            info["device_uuid"] = str(getattr(torch.cuda.get_device_properties(ctx.device), "uuid", ""))
    except Exception:
        pass
    if getattr(party.be, "native_open", False):
        try:
            info["zk_comm"] = ctx.comm_info()
        except Exception as e:
            info["zk_comm"] = {"error": repr(e)}
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, info)
    return box


def preflight_opens(ctx, dist, party, torch, transport: str):
    """Before the timed loop of an N > 1 run: tiny opens through every transport this launch can use, each compared with the
    host-side sum mod r of the vectors every rank is known to hold.  Sizes: 1 (fewer elements than parties: padding only), a
    size not divisible by the party count, and 4096.  mpc-net/src/multi.rs:469-525 semantics: every party ends up with all
    payloads in party order; here already summed.
    The ranks AGREE on every verdict (one small all-reduce per check) so that nobody leaves a collective sequence alone.  A wrong
    result on the transport the timed proofs use raises on every rank -- a wrong collective must not become a timed number; a
    failure of the OTHER transport (the one this run does not time) is recorded in the line with rank, pattern and size, and the
    run goes on: the first multi-GPU run must not lose its number to a path it does not measure."""
    import zk_mpc_amd.convert as cv
    rank, P = dist.get_rank(), dist.get_world_size()
    report = []
    def agree(ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)      # over the control plane (gloo)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def vec(p, n):        # party p's vector: canonical residues p * 2^200 + i * (p + 3) + 1, in the library's Montgomery form
        return cv.fr_to_mont([((p << 200) + i * (p + 3) + 1) % cv.R_MOD for i in range(n)]) if n else np.zeros((0, 4), np.uint64)

    def check(label, pattern, n, fn):
        err = None
        try:
            mine = ctx.upload(np.ascontiguousarray(vec(rank, n)))
            out = ctx.alloc(max(n, 1) * 32)
            fn(mine.ptr, out.ptr, n)
            ctx.sync()
            got = cv.fr_from_mont(ctx.download(out, (n, 4)))
            want = [sum(((p << 200) + i * (p + 3) + 1) for p in range(P)) % cv.R_MOD for i in range(n)]
            mine.free(); out.free()
            if list(got) != want:
                err = "wrong sum on rank %d, first wrong element %d" % (rank, next(i for i in range(n) if got[i] != want[i]))
        except Exception as e:               # (an error code from the library: the collective itself returned on every rank)
            err = "rank %d: %r" % (rank, e)
        ok = agree(err is None)
        entry = {"transport": label, "pattern": pattern, "n": n, "ok": ok}
        if not ok:
            entry["error"] = err or "another rank failed this check"
        report.append(entry)
        return ok, entry

    def section(label, pattern, fn, fatal):
        for n in sizes:
            ok, entry = check(label, pattern, n, fn)
            if not ok:
                if fatal:
                    raise RuntimeError("pre-flight open FAILED: transport %s, pattern %s, n = %d: %s" % (label, pattern, n, entry["error"]))
                return False                  # skip the rest of a section that does not work (every rank takes this branch)
        return True

    sizes = (1, 1000 + 1, 4096)
    be = party.be
    native = getattr(be, "native_open", False)
    native_label = "native (zk_open_sum_fr_dev, RCCL inside the library)"
    # (a) the transport the timed proofs will use: fatal
    section(native_label if native else "torch.distributed/%s" % transport, "by party count", lambda v, o, m: be._open_vec(v, o, m), True)
    if transport == "nccl":
        # (b) the other RCCL path, so that both have carried P ranks before either is trusted: DistNet's collectives when the
        # timed path is native, the library's own communicator when it is DistNet's.  Recorded, not fatal.
        if native:
            be.native_open = False
            try:
                section("torch.distributed/nccl", "by party count", lambda v, o, m: be._open_vec(v, o, m), False)
            finally:
                be.native_open = True
        else:
            up = None
            try:
                box = [ctx.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ctx.comm_init(box[0], rank, P)
            except Exception as e:
                up = "rank %d: zk_comm_init: %r" % (rank, e)
            if agree(up is None):
                try:
                    for pat, name in ((0, "by party count"), (1, "all-gather"), (2, "all-to-all of slices")):
                        ctx.comm_set_open_pattern(pat)
                        if not section(native_label, name, lambda v, o, m: (ctx.open_sum_fr_dev(v, m, o), ctx.sync()), False):
                            break
                    report.append({"zk_comm": ctx.comm_info()})
                finally:
                    ctx.comm_destroy()
            else:
                report.append({"transport": native_label, "ok": False, "error": up or "zk_comm_init failed on another rank"})
                try:
                    ctx.comm_destroy()
                except Exception:
                    pass
        # (c) both exchange patterns of DistNet, whatever the party count picks by default
        keep = party.net.open_pattern
        native_keep = getattr(be, "native_open", False)
        be.native_open = False
        try:
            for pat in ("allgather", "a2a"):
                party.net.open_pattern = pat
                section("torch.distributed/nccl", pat, lambda v, o, m: be._open_vec(v, o, m), not native_keep and (pat == ("allgather" if P < 3 else "a2a")))
        finally:
            party.net.open_pattern = keep
            be.native_open = native_keep
    dist.barrier()
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-constraints", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hint", action="store_true", help="do not announce the next assignment (isolated proofs)")
    ap.add_argument("--force-mpc", action="store_true", help="run the collaborative code path even with one rank (1-party: exercises transport + share plumbing)")
    ap.add_argument("--cpu-sample-log", type=int, default=None,
                    help="log2 of the CPU baseline's constraint count (default: the benched configuration itself; 16 = bounded sample)")
    ap.add_argument("--no-extras", action="store_true", help="skip the Marlin / SHE side measurements")
    ap.add_argument("--no-micro", action="store_true", help="skip the SURVEY 8(d) MSM / NTT sweeps and the natural-domain proof (~60 s)")
    ap.add_argument("--no-predict", action="store_true", help="skip the known-trapdoor check of the timed proofs (CPU, ~10 s per proof at 2^20)")
    ap.add_argument("--queue", type=int, default=4, help="number of DIFFERENT assignments the timed proofs cycle through")
    ap.add_argument("--natural-domain", action="store_true",
                    help="n = 2^L constraints, so that the QAP domain is 2^(L+1) (the reference's natural sizing, src/groth16.rs:256-257)")
    ap.add_argument("--spdz", action="store_true", help="N > 1: SPDZ (malicious) shares instead of additive ones")
    ap.add_argument("--transport", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: torch.distributed backend of the parties' opens (nccl = RCCL over xGMI, the default; gloo = staged through host memory)")
    ap.add_argument("--one-gpu", action="store_true",
                    help="N > 1: every rank on cuda:0 (a functional run of the N-party path on a one-GPU box; needs --transport gloo: RCCL "
                         "refuses two ranks on one device)")
    ap.add_argument("--one-prover", action="store_true",
                    help="--gpus N: ONE local prover whose MSMs are spread over N devices inside this process (zk_groth16_prove_multi), instead "
                         "of N parties; total work fixed (strong scaling); add --one-gpu for a functional run on one device")
    ap.add_argument("--marlin", action="store_true",
                    help="prove with Marlin/KZG instead of Groth16 (BASELINE configs 4 and 5: --gpus 1, or --gpus 8 --spdz --log-constraints 22)")
    args = ap.parse_args()

    # stdout must carry exactly ONE JSON line: libraries (RCCL prints a version banner on stdout at communicator
    # creation) are redirected to stderr for the whole run and the JSON is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.one_prover:
        if rank == 0:                       # one process drives every device; under a launcher the other ranks have nothing to do
            one_prover_bench(args, real_stdout)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has not touched the GPU (no
        # torch import, no HIP call so far) and never will: the ranks are CHILD processes of torch.distributed.run, rank 0's one
        # JSON line is relayed, the exit code is the launcher's.
        sys.exit(launch_ranks(args.gpus, real_stdout))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d inside a launch of WORLD_SIZE=%d: the two must agree" % (args.gpus, world))

    import hashlib
    import torch
    import zk_mpc_amd as Z
    import zk_mpc_amd.convert as cv

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libzkmpc_hip has no CPU path")
    if args.one_gpu:
        local_rank = 0          # (with --transport nccl RCCL refuses two ranks on one device: the run falls back to gloo and says so)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_mpc:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        args.data_group, args.transport_fallback = open_transport(args, torch, dist, rank, world, local_rank)

    if args.marlin:
        ctx = Z.Context(local_rank, rank, world)
        marlin_bench(args, ctx, dist, rank, world, real_stdout)
        if dist is not None:
            dist.destroy_process_group()
        return
    n = (1 << args.log_constraints) - (0 if args.natural_domain else 2)      # + 2 instance variables -> domain 2^L exactly
    ctx = Z.Context(local_rank, rank, world)
    mont = lambda v: cv.fr_to_mont([v])[0]
    r1cs = ctx.r1cs_mul_chain(n)
    D = 1 << r1cs.domain_log
    td = [mont(seeded_fr(i)) for i in range(1, 8)]   # alpha beta gamma delta tau g1_k g2_k
    t0 = time.time()
    pk = ctx.groth16_setup(r1cs, *td)
    t_setup = time.time() - t0
    # a queue of DIFFERENT assignments (same circuit): the timed proofs cycle through them
    Q = max(1, args.queue) if dist is None else 1
    zs = [ctx.mul_chain_assignment_dev(n, mont(seeded_fr(100 + 10 * q)), mont(seeded_fr(101 + 10 * q))) for q in range(Q)]
    rs = [(mont(seeded_fr(200 + 10 * q)), mont(seeded_fr(201 + 10 * q))) for q in range(Q)]
    last_proof = {}                                   # assignment index -> bytes of its most recent proof

    pinned, hz = [], []
    if dist is None:
        # The timed region is SURVEY 8(d)'s / BASELINE.md 3's definition of t: "from witness vector on host to 192 proof bytes on
        # host, PK resident on device" -- the host-slice entry point zk_groth16_prove_queued on a queue of DIFFERENT assignments
        # that sit in page-locked host memory (zk_host_alloc); the next one is announced, so its upload runs on a copy stream and
        # its front (z-sort, witness map, H-sort) behind this proof's kernels.  Every timed proof still contains one full front:
        # the one it runs for its successor.  --no-hint times isolated proofs.
        for q in range(Q):
            pb = ctx.host_alloc((n + 3) * 32)
            a = pb.array((n + 3, 4))
            a[:] = ctx.download(zs[q], (n + 3, 4))
            pinned.append(pb)
            hz.append(a)

        def step(i):
            q = i % Q
            last_proof[q] = ctx.create_proof_queued(pk, r1cs, hz[q], *rs[q], z_next_host=None if args.no_hint else hz[(i + 1) % Q])
            return last_proof[q]

        def step_dev(i):
            # the same queue with the assignments already resident in HBM (zk_groth16_hint_next_dev + zk_groth16_prove_dev): rounds
            # 1-5 reported this as `value`; now the named extra `device_resident_leg`
            q = i % Q
            if not args.no_hint:
                ctx.groth16_hint_next_dev(zs[(i + 1) % Q].ptr)
            return ctx.create_proof_dev(pk, r1cs, zs[q].ptr, *rs[q])
    else:
        from zk_mpc_amd import mpc
        party = make_party(mpc.SpdzParty if args.spdz else mpc.Party, mpc, ctx, dist, torch, getattr(args, 'data_group', None))
        r_plain, s_plain = seeded_fr(200), seeded_fr(201)
        if args.spdz:
            # SPDZ shares: (share, mac) lanes, MAC key alpha = 1 held by the leader (share/spdz.rs:31-37): the mac lane is an
            # independent additive sharing of the same values
            zshare = (party.share_assignment_dev(zs[0], r1cs, seed=1234), None)
            keep0 = party._keep
            zshare = (zshare[0], party.share_assignment_dev(zs[0], r1cs, seed=4321))
            keep1 = party._keep
            ra, rb = party.share_scalars([r_plain, s_plain], seed=99), party.share_scalars([r_plain, s_plain], seed=77)
            rsh, ssh = (ra[0], rb[0]), (ra[1], rb[1])

            def step(i):
                last_proof[0] = party.create_proof_shared_spdz_native(pk, r1cs, zshare, rsh, ssh)
                return last_proof[0]
        else:
            zshare = party.share_assignment_dev(zs[0], r1cs, seed=1234)
            sc = party.share_scalars([r_plain, s_plain], seed=99)

            def step(i):
                # one library call per proof (zk_groth16_prove_shared; the transport reached through callbacks)
                last_proof[0] = party.create_proof_shared_native(pk, r1cs, zshare, sc[0], sc[1])
                return last_proof[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    proof = None
    # priming, part of set-up like the key generation above: the first few proofs of a process pay one-off costs (pinned
    # staging buffers, scratch arenas growing to their final size, RCCL's lazy channel set-up: the 3rd collaborative proof
    # of a process takes 65 ms instead of 28) that must not land in the timed region when the caller asks for W < 3
    preflight = None
    ranks_seen = None
    if dist is not None and world > 1:
        # N > 1: both transports carry tiny opens and are checked against the host-side sum before the first proof
        preflight = preflight_opens(ctx, dist, party, torch, args.transport)
        ranks_seen = comm_report(ctx, dist, party, torch)
    it = 0
    for _ in range(4 if dist is not None else 2):
        step(it); it += 1
    for _ in range(args.warmup):
        proof = step(it); it += 1
    # the timed region runs the PRODUCTION path: phase timers off (they also switch the captured sort graphs off: core.hip); the
    # per-kernel averages of `roofline` come from a separate short profiled loop of the same step right after it (N = 1)
    if dist is not None:
        ctx.set_profiling(True)               # (N > 1: the per-open wall times are part of the line)
        party.be.open_stats()                 # reset the per-open wall-time counters
    barrier()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        proof = step(it); it += 1
        step_s.append(time.perf_counter() - ts)          # (a proof call returns with the proof's bytes: no extra synchronisation)
    barrier()
    dt = time.perf_counter() - t0
    profiled_steps = args.steps
    profiled_ms = None
    if dist is None:
        profiled_steps = max(4, min(args.steps, 8))
        ctx.set_profiling(True)
        barrier()
        tp0 = time.perf_counter()
        for _ in range(profiled_steps):
            step(it); it += 1
        barrier()
        profiled_ms = (time.perf_counter() - tp0) / profiled_steps * 1e3
    timers = ctx.timers()
    ctx.set_profiling(False)
    opens_timed = party.be.open_stats(args.steps) if dist is not None else None
    isolated_ms = None
    isolated_all = None
    host_leg = None
    dev_leg = None
    if dist is None:
        try:
            host_ok = True
            # isolated host proofs (no announcement): upload + proof + bytes back, each on its own
            ctx.create_proof_queued(pk, r1cs, hz[0], *rs[0])
            barrier()
            hi_ = []
            for i in range(5):
                t1 = time.perf_counter()
                p = ctx.create_proof_queued(pk, r1cs, hz[i % Q], *rs[i % Q])
                ctx.sync()
                hi_.append(time.perf_counter() - t1)
                if (i % Q) in last_proof and p != last_proof[i % Q]:
                    host_ok = False
            th_iso = float(np.median(hi_))
            host_leg = {"entry_point": "zk_groth16_prove_queued (host assignment in page-locked memory -> 192 proof bytes on the host; "
                                       "the next assignment announced and uploaded on a copy stream under the current proof)",
                        "note": "the queued figures of this entry point ARE the headline (`value`, `ms_per_step`) since round 6",
                        "ms_per_proof": round(dt / args.steps * 1e3, 3), "constraints_per_s": round(n * args.steps / dt, 1),
                        "isolated_ms_per_proof": round(th_iso * 1e3, 3), "isolated_constraints_per_s": round(n / th_iso, 1),
                        "isolated_ms_all": [round(x * 1e3, 3) for x in hi_], "statistic": "median of 5 isolated",
                        "isolated_proofs_equal_queued": bool(host_ok),
                        "profiled_loop_ms_per_proof": round(profiled_ms, 3), "profiled_loop_steps": profiled_steps}
        except Exception as e:          # the headline line must not depend on this leg
            host_leg = {"error": repr(e)}
        try:
            # the device-resident queue (what rounds 1-5 reported as value), same number of steps, timers off
            dev_ok = True
            for i in range(2):
                step_dev(it); it += 1
            barrier()
            t1 = time.perf_counter()
            dq = []
            for _ in range(args.steps):
                ts = time.perf_counter()
                p = step_dev(it)
                dq.append(time.perf_counter() - ts)
                if (it % Q) in last_proof and p != last_proof[it % Q]:
                    dev_ok = False
                it += 1
            barrier()
            td_ = (time.perf_counter() - t1) / args.steps
            ctx.groth16_hint_next_dev(None)
            dev_leg = {"entry_point": "zk_groth16_hint_next_dev + zk_groth16_prove_dev (assignments resident in HBM -> 192 proof bytes on the host)",
                       "ms_per_proof": round(td_ * 1e3, 3), "constraints_per_s": round(n / td_, 1),
                       "median_ms_per_proof": round(float(np.median(dq)) * 1e3, 3), "max_ms_per_proof": round(max(dq) * 1e3, 3),
                       "proofs_equal_host_leg": bool(dev_ok), "steps": args.steps}
            if not args.no_hint:
                # the same proofs without the announcement (latency of one isolated proof)
                ctx.create_proof_dev(pk, r1cs, zs[0].ptr, *rs[0])
                barrier()
                iso = []
                for i in range(7):
                    t1 = time.perf_counter()
                    ctx.create_proof_dev(pk, r1cs, zs[i % Q].ptr, *rs[i % Q])
                    ctx.sync()
                    iso.append((time.perf_counter() - t1) * 1e3)
                isolated_ms = float(np.median(iso))
                isolated_all = [round(x, 3) for x in iso]
        except Exception as e:
            dev_leg = {"error": repr(e)}
        for pb in pinned:
            pb.free()
    open_probe = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)          # over the control plane (gloo)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank must hold the same revealed proof
        hsh = np.frombuffer(hashlib.sha256(proof).digest()[:8], dtype=np.uint64).copy()
        allh = party.net.all_gather_small(hsh)
        same_on_all_ranks = all(int(x[0]) == int(hsh[0]) for x in allh)
        # cost of one share-vector open (the data-path collective) on its own: D elements, outside the timed region
        try:
            be = party.be
            va, vo = be.vec("probe_a", D), be.vec("probe_o", D)
            ctx.dev_zero(va, D * 32)
            be.open_vec(va, vo, D)
            barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                be.open_vec(va, vo, D)
            barrier()
            t_open = (time.perf_counter() - t1) / 5
            pattern = "all-gather + sum" if world < 3 else "all-to-all of slices + sum + all-gather of the summed slices"
            in_bytes = (world - 1) * D * 32 if world < 3 else 2 * (world - 1) * ((D + world - 1) // world) * 32
            open_probe = {"elements": D, "ms": round(t_open * 1e3, 3), "pattern": pattern, "bytes_in_per_gpu": in_bytes,
                          "gb_per_s_in_per_gpu": round(in_bytes / t_open / 1e9, 2) if t_open > 0 else None,
                          "opens_per_proof": 4 if args.spdz else 2}
        except Exception as e:
            open_probe = {"error": repr(e)}

    if rank == 0:
        K = args.steps
        per_proof = n * K / dt
        # ---- self-check: the timed proofs against the known-trapdoor prediction (CPU, outside the timed region) ----
        pred = {"checked": 0}
        if not args.no_predict:
            try:
                threads = os.cpu_count() or 1
                ok = True
                t1 = time.time()
                check = sorted(last_proof)[-2:] if dist is None else [0]
                for q in check:
                    zarr = ctx.download(zs[q], (n + 3, 4))
                    r_q, s_q = (rs[q] if dist is None else (mont(seeded_fr(200)), mont(seeded_fr(201))))
                    want, note = predict_proof(ctx, n, zarr, td, r_q, s_q, threads)
                    if want is None:
                        pred = {"checked": 0, "note": note}
                        ok = None
                        break
                    ok = ok and (want == last_proof[q])
                    pred["checked"] += 1
                pred["ok"] = ok
                pred["seconds"] = round(time.time() - t1, 1)
                pred["what"] = ("bytes of the last proof of %d different assignments of the timed queue" % len(check)) if dist is None else \
                               "revealed %d-party proof vs the prediction for z = sum of the shares, r = sum r_i, s = sum s_i" % world
            except Exception as e:
                pred = {"checked": 0, "ok": None, "error": repr(e)}
        # dominant kernel: G1 bucket accumulation (k_accum<G1>), 4 launches per proof.
        acc_ms, acc_cnt = timers.get("msm_g1.accum", (0.0, 0))
        roof = None
        if acc_cnt:
            avg_s = acc_ms / acc_cnt * 1e-3
            n_msm = n  # every G1 MSM of this workload has ~n terms (h: D-1, l: n+1, a/b: n+2)
            alg_bytes = 128.0 * n_msm            # 32 B scalar + 96 B base per term
            achieved = alg_bytes / avg_s / 1e9
            c = ctx.lib.zk_bases_window_bits(pk_bases(ctx, pk, "a").h)
            W = (255 + c - 1) // c if c else msm_plan_windows(n_msm)      # digits per scalar (13 with the key's window multiples, c = 20)
            madds = n_msm * W                     # mixed additions in the accumulate kernel
            mads_per_madd = MADS_PER_MADD_G1
            mads = madds * mads_per_madd
            traffic, traffic_src = None, None
            try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), not measured live; a file collected on
                # OTHER kernel sources than the ones this run executes is refused (tools/pmc_traffic.py records their hash)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import pmc_traffic as PT
                for f in ("r6_pmc_traffic.json", "r5_pmc_traffic.json"):
                    pth = os.path.join(ROOT, "profiles", f)
                    if os.path.exists(pth) and args.log_constraints == 20 and not args.natural_domain:
                        doc = json.load(open(pth))
                        if doc.get("kernel_sources_sha256") != PT.sources_sha256():
                            traffic_src = "profiles/%s REFUSED: collected on other kernel sources than this run's (hash %s)" % (f, str(doc.get("kernel_sources_sha256"))[:12])
                            continue
                        ks = doc["kernels"]
                        traffic = (ks.get("k_accum<G1, true>") or ks["k_accum<G1>"])["hbm_bytes"]      # (limb-form tables: the <F, true> instance)
                        traffic_src = "profiles/" + f
                        break
            except Exception:
                pass
            # the roof that binds: SURVEY 8(d) prices an MSM kernel against the v_mad_u64_u32 issue rate, measured on THIS device in
            # THIS run (zk_diag_int_mad_peak: 12 launches of a pure multiply-add kernel after the timed loop; median launch = the roof)
            peak, peak_src = INT_MAD_PEAK, "constant from profiles/r1_ubench_int.txt (the live measurement failed)"
            try:
                pk_meas = ctx.int_mad_peak(12)
                peak = pk_meas["median"]
                peak_src = ("measured in this run: zk_diag_int_mad_peak, median of %d launches of ~4.5 ms each, ~50 ms in all (the sustained "
                            "rate; fastest launch %.2f T/s)" % (pk_meas["launches"], pk_meas["best"] / 1e12))
            except Exception as e:
                peak_src += ": %r" % (e,)
            roof = {"bound": "int_alu", "kernel": "k_accum<G1> (MSM bucket accumulation), 4 launches per proof",
                    "achieved": round(mads / avg_s / 1e12, 3), "peak": round(peak / 1e12, 3), "unit": "T v_mad_u64_u32 lane-ops/s",
                    "frac": round(mads / avg_s / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                    "peak_source": peak_src, "avg_launch_ms": round(avg_s * 1e3, 3), "launches": acc_cnt,
                    "work_per_launch": {"mixed_additions": madds, "mads_per_mixed_add": mads_per_madd, "terms": n_msm, "digits_per_scalar": W,
                                        "window_bits": c},
                    "traffic_source": traffic_src,
                    "traffic_note": "2*FETCH_SIZE+WRITE_SIZE from a SEPARATE rocprofv3 --pmc run committed under profiles/ (not measured in "
                                    "this run); above the 128 B/term algorithmic figure because the bucket method reads every base once per "
                                    "digit in 128-B lines; the kernel is ALU-bound, the HBM view is nested under `hbm`",
                    "hbm": {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                            "algorithmic_bytes_per_launch": alg_bytes,
                            "note": "128 B/term (32 B scalar + 96 B base) over the kernel's launch time: not the binding roof"}}
            roof["int_alu"] = {"achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"], "frac": roof["frac"],
                               "mads_per_mixed_add": mads_per_madd}          # (the place earlier rounds' parsers looked)
            try:  # instruction mix of the kernel's main path (static, from the ISA: tools/isa_hist.py) -> mix-weighted ceiling
                isa = next(f for f in ("r4_isa_k_accum_g1.json", "r3_isa_k_accum_g1.json", "r2_isa_k_accum_g1.json")
                           if os.path.exists(os.path.join(ROOT, "profiles", f)))
                ih = json.load(open(os.path.join(ROOT, "profiles", isa)))
                roof["int_alu"]["mix_weighted_ceiling"] = {
                    "source": "profiles/" + isa + " (tools/isa_hist.py over hipcc -S; priced with profiles/r1_ubench_int.txt: "
                              "v_mad_u64_u32 and the other ~35 T/s classes = 1 issue slot, the ~67 T/s classes = 0.5)",
                    "mad_share_of_issue_slots": round(ih["main_path"]["mix_ceiling"], 4)}
            except Exception:
                pass
            try:  # the instruction-issue view: VALU wave-instructions per launch (SQ_INSTS_VALU, a separate --pmc run) over the
                # launch time, against the issue rate of the pure multiply-add kernel measured in this run (lane-ops / 64)
                if args.log_constraints != 20 or args.natural_domain:
                    raise KeyError("the committed counter run is of the 2^20 workload")
                pvf = next(f for f in ("r6_pmc_valu.json", "r5_pmc_valu.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
                pvdoc = json.load(open(os.path.join(ROOT, "profiles", pvf)))
                if pvdoc.get("kernel_sources_sha256") != PT.sources_sha256():      # (a stale instruction count must not feed the fraction)
                    roof["int_alu"]["issue"] = {"refused": "profiles/%s was collected on other kernel sources than this run's" % pvf}
                    raise KeyError("stale counter file")
                pv = pvdoc["kernels"]
                kk = next(k for k in pv if k.startswith("k_accum<G1"))
                insts = float(pv[kk]["SQ_INSTS_VALU"])
                peak_insts = roof["peak"] * 1e12 / 64
                roof["int_alu"]["issue"] = {
                    "valu_wave_insts_per_launch": insts, "achieved_wave_insts_per_s": round(insts / avg_s, 1),
                    "peak_wave_insts_per_s": round(peak_insts, 1), "frac": round(insts / avg_s / peak_insts, 4),
                    "source": "profiles/%s (SQ_INSTS_VALU per launch of %s, not measured in this run; the count does "
                              "not depend on what runs beside the kernel)" % (pvf, kk),
                    "note": "every VALU instruction priced as a multiply-add slot: the share of the chip's issue rate this kernel "
                            "takes while it shares the chip with the other jobs' reduce chains, sorts and transforms"}
            except Exception:
                pass
            # the single longest kernel: the G2 accumulate on lane pairs (one launch per proof); every one of its 10 Fq2
            # products per mixed addition is two fused double products
            g2_ms, g2_cnt = timers.get("msm_g2.accum", (0.0, 0))
            if g2_cnt:
                g2_s = g2_ms / g2_cnt * 1e-3
                roof["g2_accum"] = {"kernel": "k_accum_g2pair (B-in-G2 bucket accumulation, two lanes per addition)",
                                    "avg_launch_ms": round(g2_s * 1e3, 3), "launches": g2_cnt,
                                    "int_alu": {"achieved": round(madds * MADS_PER_MADD_G2 / g2_s / 1e12, 3), "peak": round(peak / 1e12, 3),
                                                "unit": "T v_mad_u64_u32 lane-ops/s", "frac": round(madds * MADS_PER_MADD_G2 / g2_s / peak, 4)}}
        share_kind = "SPDZ (share + MAC lanes, MAC-checked opens)" if args.spdz else "additive-share"
        out = {
            "metric": "R1CS constraints/sec (prove), Groth16 BLS12-377",
            "value": round(per_proof, 1),
            "unit": "constraints/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x13 (29-bit limbs, int64 accumulate)", "data": "synthetic",
            "config": {"workload": "mul-chain R1CS, n=2^%d%s constraints, QAP domain 2^%d, %s" % (
                args.log_constraints, "" if args.natural_domain else "-2", r1cs.domain_log,
                "local prove" if dist is None else "%d-party %s collaborative prove" % (world, share_kind)),
                "constraints": n, "parties": world,
                "queue": ("isolated proofs" if (dist is not None or args.no_hint) else
                          "host witness -> host proof bytes (zk_groth16_prove_queued), proofs back to back over a queue of %d DIFFERENT "
                          "assignments in page-locked host memory, the next one announced: its upload runs under the current proof and "
                          "each timed proof also runs the front of its successor; phase timers off" % Q)},
            "proof_constraints_per_s": round(per_proof, 1),
            "proof_matches_prediction": pred.get("ok"),
            "prediction_check": pred,
            "ms_per_step_median": round(float(np.median(step_s)) * 1e3, 3), "ms_per_step_max": round(max(step_s) * 1e3, 3),
            "isolated_proof_ms": None if isolated_ms is None else round(isolated_ms, 3),
            "isolated_proof_ms_all": isolated_all,
            "host_witness_leg": host_leg,
            "device_resident_leg": dev_leg,
            "phases_ms_per_proof": {k: round(v[0] / profiled_steps, 3) for k, v in sorted(timers.items())},
            "phases_from": ("the timed steps" if dist is not None else
                            "a separate loop of %d steps of the same call with the phase timers on, right after the timed region" % profiled_steps),
            "setup_s": round(t_setup, 2),
            "proof_sha": hashlib.sha256(proof).hexdigest()[:16],
            "roofline": roof,
            "hbm_in_use_gb": hbm_in_use_gb(),
        }
        try:      # which tables carry window multiples, in which layout -- or why not (a skipped table costs 16 digits per scalar instead of 13)
            wm = {}
            for which in ("a", "b_g1", "b_g2", "h", "l"):
                qb = pk_bases(ctx, pk, which)
                wm[which] = {"window_bits": int(ctx.lib.zk_bases_window_bits(qb.h)), "layout_or_reason": qb.precompute_note()}
            out["window_multiples"] = wm
        except Exception as e:
            out["window_multiples"] = {"error": repr(e)}
        if dist is not None:
            out["value_note"] = ("value = constraints of ONE proof x proofs / time: the N parties jointly produce one proof, so the job's "
                                 "output does not grow with N although every party runs the full-size prover on its shares (per-GPU work "
                                 "fixed: 'weak'); value(N) / value(1) is the north star's '3-party within 2x of 1-GPU' ratio.  N x value "
                                 "(constraint-shares/s, the figure rounds 1-2 reported as value) is aggregate_constraint_shares_per_s")
            out["aggregate_constraint_shares_per_s"] = round(per_proof * world, 1)
            out["opens_in_timed_proofs"] = opens_timed
            out["same_proof_on_all_ranks"] = bool(same_on_all_ranks)
            out["prover_entry"] = "zk_groth16_prove_shared%s (one C-ABI call per proof)" % ("_spdz" if args.spdz else "")
            out["open_probe"] = open_probe
            out["preflight_opens"] = preflight
            out["ranks"] = ranks_seen
            out["rccl_ranks_seen"] = None if ranks_seen is None else {
                "torch_distributed": dist.get_world_size() if args.transport == "nccl" else 0,
                "zk_comm": max([r.get("zk_comm", {}).get("n_ranks", 0) for r in ranks_seen] +
                               [p["zk_comm"].get("n_ranks", 0) for p in (preflight or []) if "zk_comm" in p]),
                "devices": sorted(set((r.get("device_uuid") or r["device"]) for r in ranks_seen))}
            out["transport"] = ("RCCL (torch.distributed nccl), one GPU per party" if args.transport == "nccl" else
                                "gloo, opens staged through host memory" + (", every party on cuda:0 (functional run, not a measurement)" if args.one_gpu else ""))
            out["transport_fallback"] = getattr(args, "transport_fallback", None)
            out["bytes_sent_per_party"] = int(party.bytes_sent)
        if dist is None:
            # second half of the headline metric: standalone variable-base MSM throughput (resident bases = the
            # proving key's A / B-in-G2 queries, scalars = the assignment already in HBM), outside the timed region
            msm = {}
            for name, q, grp in (("g1", pk_bases(ctx, pk, "a"), 1), ("g2", pk_bases(ctx, pk, "b_g2"), 2)):
                m = n                       # terms
                warm = []
                for _ in range(2):          # the first calls of a new shape after the proofs (reported, not part of the median)
                    t1 = time.perf_counter()
                    ctx.msm_dev(q, 1, zs[0].ptr + 32, m)
                    warm.append(time.perf_counter() - t1)
                ctx.sync()
                reps = []
                for _ in range(9):
                    t1 = time.perf_counter()
                    ctx.msm_dev(q, 1, zs[0].ptr + 32, m)      # returns with the result on the host
                    reps.append(time.perf_counter() - t1)
                med = float(np.median(reps))
                msm[name] = round(m / med / 1e6, 1)
                msm[name + "_ms"] = {"median": round(med * 1e3, 3), "min": round(min(reps) * 1e3, 3), "max": round(max(reps) * 1e3, 3),
                                     "all": [round(x * 1e3, 3) for x in reps], "first_two_calls": [round(x * 1e3, 3) for x in warm]}
            out["msm_mscalar_per_s"] = dict(msm, n=n, note="single MSM per call incl. host round trip, bases resident (window multiples); median of 9 calls")
        if dist is None and not args.no_extras:
            out["other_workloads"] = other_workloads(ctx, min(args.log_constraints, 20))
        if dist is None and not args.no_micro and not args.no_extras:
            # SURVEY 8(d)'s other measurement rows, outside the timed region (each guarded: the headline line must not depend on them)
            try:
                out["micro"] = micro_sweeps(ctx)
            except Exception as e:
                out["micro"] = {"error": repr(e)}
            try:
                out["boolean_heavy"] = boolean_heavy_leg(ctx, args.log_constraints, td, os.cpu_count() or 1)
            except Exception as e:
                out["boolean_heavy"] = {"error": repr(e)}
            try:
                if not args.natural_domain and args.log_constraints <= 20:
                    out["natural_domain"] = natural_domain_leg(ctx, args.log_constraints, td, os.cpu_count() or 1)
            except Exception as e:
                out["natural_domain"] = {"error": repr(e)}
        if dist is None and not args.no_extras and not args.natural_domain:
            try:
                out["trait_path"] = trait_path_leg(r1cs.domain_log, dt / K * 1e3)
            except Exception as e:
                out["trait_path"] = {"error": repr(e)}
            try:
                out["trait_path_collab"] = trait_path_collab_leg(r1cs.domain_log, dt / K * 1e3)
            except Exception as e:
                out["trait_path_collab"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            sample_log = args.cpu_sample_log if args.cpu_sample_log is not None else args.log_constraints
            out["cpu_baseline"] = cpu_baseline(ctx, td, sample_log, os.cpu_count() or 1, args.log_constraints)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
