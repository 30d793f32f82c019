#!/usr/bin/env python3
"""bench.py -- Groth16 (BLS12-377) prove throughput on MI355X: R1CS constraints/s.

  python bench.py --gpus N --steps K --warmup W [--log-constraints L]

N = 1: local (non-MPC) prove of the SURVEY 8(d) config-2 workload: mul-chain R1CS with
       n = 2^20 - 2 constraints (QAP domain 2^20), proving key resident on the device,
       witness resident on the device when the timed region starts.
N > 1: N-party collaborative prove (additive shares, honest backend), one party per GPU,
       launched by torch.distributed.run; the two Beaver opens are all-gathers over RCCL.
       Per-GPU work is fixed (every party runs the full-size NTTs/MSMs on its shares), so
       scaling is "weak": value = N * n * K / T (constraint-shares proved per second);
       `proof_constraints_per_s` = n * K / T is the per-proof rate.

A step = one proof.  Timing: W untimed proofs, then exactly K proofs bracketed by barrier +
device synchronisation; max over ranks.  One JSON line on rank 0.
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


def seeded_fr(seed: int):
    import hashlib
    import zk_mpc_amd.convert as cv
    h = hashlib.sha256(b"zkmpc-bench" + seed.to_bytes(8, "little")).digest() + hashlib.sha256(b"x" + seed.to_bytes(8, "little")).digest()
    return int.from_bytes(h[:40], "little") % cv.R_MOD


def cpu_baseline(ctx, td, sample_log: int, threads: int):
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
    t_one, proof_one, _ = OC.bench_mul_chain_prove(n, w0, w1, hp, r_, s_, 1)
    ok = ok and proof_one == gpu_proof
    for o in (z, ):
        o.free()
    pk.free(); r1cs.free()
    return {"value": round(n / t_all, 1), "unit": "constraints/s", "cores": threads, "kind": "port",
            "sample": "mul-chain prove, n=2^%d-2 constraints (bounded sample of the 2^20 workload), device's proving key; "
                      "%.2f s on %d threads (witness map %.2f s, MSMs %.2f s); single thread (the reference's build: no rayon): "
                      "%.2f s = %.0f constraints/s" % (sample_log, t_all, threads, ph[0], ph[1], t_one, n / t_one),
            "single_thread_value": round(n / t_one, 1), "proof_matches_device": bool(ok)}


def pk_bases(ctx, pk, which):
    """A non-owning Bases view of one proving-key query (for the standalone MSM measurement)."""
    import ctypes as C
    from zk_mpc_amd.api import Bases
    h = C.c_void_p(ctx.lib.zk_pk_query_bases(pk.h, {"a": 0, "b_g1": 1, "b_g2": 2, "h": 3, "l": 4}[which]))
    return Bases(ctx, h, 2 if which == "b_g2" else 1, owned=False)


def other_workloads(ctx, log_h=20):
    """Rows a14 / a15 beside the headline metric (not part of `value`): Marlin AHP prover + KZG10 commitments / openings on
    the same mul-chain family (BASELINE config 4 shape), and the SHE ciphertext product.  Same code as
    tools/bench_marlin.py / tools/bench_she.py, fewer repetitions."""
    import numpy as np
    from zk_mpc_amd import marlin as DM
    out = {}
    rng = np.random.default_rng(11)
    m = DM.HostField.m

    def rand_fr(k):
        a = rng.integers(0, 1 << 63, size=(k, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    try:
        n = (1 << log_h) - 3
        ni, nw, a, b, c = DM.mul_chain_system(ctx, n)
        index = DM.Index(ctx, ni, nw, a, b, c)
        H = index.dom_h.size
        deg = 3 * max(H, index.dom_k.size) + 2
        pw = ctx.alloc(deg * 32)
        ctx.fr_powers_dev(m(0x1234567), m(1), deg, pw.ptr)
        powers_g = ctx.fixed_base(pw.ptr, deg, 1, m(1))
        powers_g.precompute()          # resident SRS: window multiples, 13 digits per scalar instead of 16
        z = ctx.mul_chain_assignment_dev(n, m(3), m(5))
        rnd = ctx.upload(rand_fr(3 + 3 * H))
        ch = {k: int(rng.integers(2, 1 << 62)) for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma", "xi")}
        ctx.pooling = True

        def prove():
            st = DM.prover_init(index, z)
            polys = dict(DM.prover_first_round(st, rnd))
            comms = DM.commit(ctx, powers_g, polys)
            r2 = DM.prover_second_round(st, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"])
            comms.update(DM.commit(ctx, powers_g, r2))
            r3 = DM.prover_third_round(st, ch["beta"])
            comms.update(DM.commit(ctx, powers_g, r3))
            polys.update(r2)
            polys.update(r3)
            for l, pt in (("g_1", "beta"), ("z_b", "beta"), ("t", "beta"), ("g_2", "gamma")):
                ctx.poly_evaluate_dev(polys[l].ptr, polys[l].n, m(ch[pt]))
            ixp = index.polynomials()
            DM.batch_open(ctx, powers_g, [([polys[l] for l in ("g_1", "z_b", "t", "mask_poly", "z_a", "w", "h_1")], ch["beta"]),
                                          ([polys["g_2"], polys["h_2"]] + [ixp[l] for l in sorted(ixp)], ch["gamma"])], ch["xi"])
        prove()
        ctx.sync()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            prove()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        ctx.pooling = False
        ctx.drop_pool()
        out["marlin"] = {"workload": "Marlin AHP prover + 9 KZG10 commitments + 2 batched openings, mul-chain R1CS, |H| = |K| = 2^%d, "
                                     "index and SRS resident, challenges supplied by the caller" % log_h,
                         "constraints": n, "ms_per_proof": round(dt * 1e3, 2), "constraints_per_s": round(n / dt, 1)}
        del index, powers_g, pw, z, rnd
    except Exception as e:  # the headline line must not depend on this leg
        ctx.pooling = False
        out["marlin"] = {"error": repr(e)}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-constraints", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hint", action="store_true", help="do not announce the next assignment (isolated proofs)")
    ap.add_argument("--force-mpc", action="store_true", help="run the collaborative code path even with one rank (1-party: exercises transport + share plumbing)")
    ap.add_argument("--cpu-sample-log", type=int, default=16)
    ap.add_argument("--no-extras", action="store_true", help="skip the Marlin / SHE side measurements")
    args = ap.parse_args()

    # stdout must carry exactly ONE JSON line: libraries (RCCL prints a version banner on stdout at communicator
    # creation) are redirected to stderr for the whole run and the JSON is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))

    import torch
    import zk_mpc_amd as Z
    import zk_mpc_amd.convert as cv

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libzkmpc_hip has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_mpc:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n = (1 << args.log_constraints) - 2      # + 2 instance variables -> domain 2^L exactly
    ctx = Z.Context(local_rank, rank, world)
    mont = lambda v: cv.fr_to_mont([v])[0]
    r1cs = ctx.r1cs_mul_chain(n)
    D = 1 << r1cs.domain_log
    td = [mont(seeded_fr(i)) for i in range(1, 8)]   # alpha beta gamma delta tau g1_k g2_k
    t0 = time.time()
    pk = ctx.groth16_setup(r1cs, *td)
    t_setup = time.time() - t0
    z = ctx.mul_chain_assignment_dev(n, mont(seeded_fr(100)), mont(seeded_fr(101)))
    r_, s_ = mont(seeded_fr(200)), mont(seeded_fr(201))

    if dist is None:
        def step():
            # a prover working through a queue of assignments announces the next one: the proof then enqueues the next
            # proof's front (z-sort, witness map, H-sort) behind its own kernels (zk_groth16_hint_next_dev).  Every timed
            # proof still contains one full front: the one it runs for its successor.  --no-hint times isolated proofs.
            if not args.no_hint:
                ctx.groth16_hint_next_dev(z.ptr)
            return ctx.create_proof_dev(pk, r1cs, z.ptr, r_, s_)
    else:
        from zk_mpc_amd import mpc
        party = mpc.Party(ctx, dist)
        zshare = party.share_assignment_dev(z, r1cs, seed=1234)
        rs = party.share_scalars([seeded_fr(200), seeded_fr(201)], seed=99)

        def step():
            return party.create_proof_shared(pk, r1cs, zshare, rs[0], rs[1])

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    proof = None
    # priming, part of set-up like the key generation above: the first few proofs of a process pay one-off costs (pinned
    # staging buffers, scratch arenas growing to their final size, RCCL's lazy channel set-up: the 3rd collaborative proof
    # of a process takes 65 ms instead of 28) that must not land in the timed region when the caller asks for W < 3
    for _ in range(4 if dist is not None else 2):
        step()
    for _ in range(args.warmup):
        proof = step()
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = step()
    barrier()
    dt = time.perf_counter() - t0
    timers = ctx.timers()
    ctx.set_profiling(False)
    isolated_ms = None
    if dist is None and not args.no_hint:
        # the same proof without the announcement (latency of one isolated proof), outside the timed region
        ctx.groth16_hint_next_dev(None)
        ctx.create_proof_dev(pk, r1cs, z.ptr, r_, s_)
        barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            ctx.create_proof_dev(pk, r1cs, z.ptr, r_, s_)
        barrier()
        isolated_ms = (time.perf_counter() - t1) / 3 * 1e3
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        K = args.steps
        per_proof = n * K / dt
        # dominant kernel: G1 bucket accumulation (k_accum<G1>), 4 launches per proof.
        acc_ms, acc_cnt = timers.get("msm_g1.accum", (0.0, 0))
        roof = None
        if acc_cnt:
            avg_s = acc_ms / acc_cnt * 1e-3
            n_msm = n  # every G1 MSM of this workload has ~n terms (h: D-1, l: n+1, a/b: n+2)
            alg_bytes = 128.0 * n_msm            # 32 B scalar + 96 B base per term
            achieved = alg_bytes / avg_s / 1e9
            c = ctx.lib.zk_bases_window_bits(pk_bases(ctx, pk, "a").h) or max(4, min(16, n_msm.bit_length() - 1 - 4))
            W = (255 + c - 1) // c                # digits per scalar (13 with the key's precomputed window multiples, c = 20)
            madds = n_msm * W                     # mixed additions in the accumulate kernel
            mads = madds * (6 * 325 + 494 + 2 * 260)   # 8M + 2S with R(Q - X3) - Y1 PPP as one fused double product (494 mads)
            traffic = None
            try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), not measured live
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
                if args.log_constraints == 20:
                    traffic = pmc["kernels"]["k_accum<G1>"]["hbm_bytes"]
            except Exception:
                pass
            roof = {"bound": "hbm", "kernel": "k_accum<G1> (MSM bucket accumulation)", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_note": "bytes/launch, 2*FETCH_SIZE+WRITE_SIZE from profiles/r1_pmc_traffic.json; ~29x the 128 B/term "
                                    "algorithmic figure because the bucket method re-reads every base once per window (16x) in "
                                    "128-B lines (96-B points); the kernel is ALU-bound at ~1.2 TB/s of gather traffic",
                    "avg_launch_ms": round(avg_s * 1e3, 3), "launches": acc_cnt,
                    "note": "kernel is integer-ALU bound, not HBM bound (SURVEY 8d); see int_alu",
                    "int_alu": {"achieved": round(mads / avg_s / 1e12, 3), "peak": round(INT_MAD_PEAK / 1e12, 2),
                                "unit": "T v_mad_u64_u32 lane-ops/s", "frac": round(mads / avg_s / INT_MAD_PEAK, 4)}}
            try:  # VALU issue utilisation from the committed rocprofv3 PMC passes (tools/pmc_valu.py), not measured live
                pv = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_valu.json")))["kernels"]
                roof["valu_issue"] = {
                    "note": "SQ_INSTS_VALU per launch / kernel duration (kernel alone on the chip, under counter collection) vs the "
                            "measured integer issue peak of 32.16 T lane-ops/s = 0.5025 T wave-instructions/s: both accumulate "
                            "kernels fill every VALU issue slot; what int_alu reports below 1 is the share of v_mad_u64_u32 in "
                            "their instruction mix",
                    "k_accum<G1>": {"wave_insts_per_launch": round(pv["k_accum<G1>"]["SQ_INSTS_VALU"]),
                                    "frac_of_issue_peak": round(pv["k_accum<G1>"]["frac_of_int_issue_peak"], 4),
                                    "mean_waves_per_cu": round(pv["k_accum<G1>"]["MeanOccupancyPerCU"], 2)},
                    "k_accum_g2pair": {"wave_insts_per_launch": round(pv["k_accum_g2pair<2>"]["SQ_INSTS_VALU"]),
                                       "frac_of_issue_peak": round(pv["k_accum_g2pair<2>"]["frac_of_int_issue_peak"], 4),
                                       "mean_waves_per_cu": round(pv["k_accum_g2pair<2>"]["MeanOccupancyPerCU"], 2)}}
            except Exception:
                pass
            # the single longest kernel: the G2 accumulate on lane pairs (one launch per proof); every one of its 10 Fq2
            # products per mixed addition is two fused double products of 507 mads
            g2_ms, g2_cnt = timers.get("msm_g2.accum", (0.0, 0))
            if g2_cnt:
                g2_s = g2_ms / g2_cnt * 1e-3
                roof["g2_accum"] = {"kernel": "k_accum_g2pair (B-in-G2 bucket accumulation, two lanes per addition)",
                                    "avg_launch_ms": round(g2_s * 1e3, 3), "launches": g2_cnt,
                                    "int_alu": {"achieved": round(madds * 10 * 1014 / g2_s / 1e12, 3), "peak": round(INT_MAD_PEAK / 1e12, 2),
                                                "unit": "T v_mad_u64_u32 lane-ops/s", "frac": round(madds * 10 * 1014 / g2_s / INT_MAD_PEAK, 4)}}
        out = {
            "metric": "R1CS constraints/sec (prove), Groth16 BLS12-377",
            "value": round(per_proof * world, 1),
            "unit": "constraints/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x13 (29-bit limbs, int64 accumulate)", "data": "synthetic",
            "config": {"workload": "mul-chain R1CS, n=2^%d-2 constraints, QAP domain 2^%d, %s" % (
                args.log_constraints, r1cs.domain_log,
                "local prove" if dist is None else "%d-party additive-share collaborative prove" % world),
                "constraints": n, "parties": world,
                "queue": ("isolated proofs" if (dist is not None or args.no_hint) else
                          "proofs back to back, the next assignment announced (zk_groth16_hint_next_dev): each timed proof "
                          "also runs the front of its successor")},
            "proof_constraints_per_s": round(per_proof, 1),
            "isolated_proof_ms": None if isolated_ms is None else round(isolated_ms, 3),
            "phases_ms_per_proof": {k: round(v[0] / K, 3) for k, v in sorted(timers.items())},
            "setup_s": round(t_setup, 2),
            "proof_sha": __import__("hashlib").sha256(proof).hexdigest()[:16],
            "roofline": roof,
        }
        if dist is None:
            # second half of the headline metric: standalone variable-base MSM throughput (resident bases = the
            # proving key's A / B-in-G2 queries, scalars = the assignment already in HBM), outside the timed region
            msm = {}
            for name, q, grp in (("g1", pk_bases(ctx, pk, "a"), 1), ("g2", pk_bases(ctx, pk, "b_g2"), 2)):
                m = n                       # terms
                ctx.msm_dev(q, 1, z.ptr + 32, m)
                ctx.sync()
                t1 = time.perf_counter()
                reps = 5
                for _ in range(reps):
                    ctx.msm_dev(q, 1, z.ptr + 32, m)
                ctx.sync()
                msm[name] = round(m * reps / (time.perf_counter() - t1) / 1e6, 1)
            out["msm_mscalar_per_s"] = dict(msm, n=n, note="single MSM per call incl. host round trip, bases resident")
        if dist is None and not args.no_extras:
            out["other_workloads"] = other_workloads(ctx, min(args.log_constraints, 20))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ctx, td, args.cpu_sample_log, os.cpu_count() or 1)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
