"""bench.py's one-line contract on the GPU box, at small sizes: the N = 1 line carries `roofline` and the self-checks, and the
N-party path (one process per party, launched as the driver launches it) runs end to end -- here over gloo with every party on
cuda:0, because RCCL refuses two ranks on one device; with one GPU per rank the same code runs over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _line(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_small():
    d = _line([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--log-constraints", "14", "--no-extras", "--no-micro",
               "--cpu-sample-log", "12"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "constraints/s" and d["value"] > 0
    assert d["proof_matches_prediction"] is True
    # the top-level roofline is the roof that binds (an MSM kernel is judged on the integer-ALU roof), its peak measured
    # in the run; the HBM view is nested
    roof = d["roofline"]
    assert roof["bound"] == "int_alu" and 0 < roof["frac"] < 1 and roof["peak"] > 20 and "measured in this run" in roof["peak_source"]
    assert roof["hbm"]["frac"] < roof["frac"] and roof["int_alu"]["frac"] == roof["frac"]
    assert len(d["msm_mscalar_per_s"]["g2_ms"]["all"]) >= 7 and d["msm_mscalar_per_s"]["g2_ms"]["max"] < 4 * d["msm_mscalar_per_s"]["g2_ms"]["median"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("port", "reference")


@pytest.mark.parametrize("world,extra", [(2, []), (3, ["--spdz"]), (2, ["--marlin"])])
def test_n_party_line_over_gloo_on_one_gpu(world, extra):
    from helpers import free_port
    port = free_port()
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--log-constraints", "12",
               "--transport", "gloo", "--one-gpu"] + extra)
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["value"] > 0
    assert d["opens_in_timed_proofs"]["opens_per_proof"] > 0
    if "--marlin" in extra:
        assert d["oracle_verifier_accepts"] is True and d["oracle_verifier_rejects_wrong_input"] is True
    else:
        assert d["same_proof_on_all_ranks"] is True and d["proof_matches_prediction"] is True


@pytest.mark.parametrize("extra", [[], ["--marlin"]])
def test_bench_starts_its_own_ranks(extra):
    """`python3 bench.py --gpus 2 ...` with NO launcher on the command line (how the driver calls N = 1, extended to N > 1): the
    parent must spawn torch.distributed.run as a child before touching the GPU, relay one JSON line and the exit code; the line
    names the ranks it saw and the pre-flight opens it checked."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-constraints", "12",
                        "--transport", "gloo", "--one-gpu"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(env_clean, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "bench.py itself" in d["launched_by"]
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and all(x["dist_world_size"] == 2 for x in d["ranks"])
    assert d["preflight_opens"] and all(x["ok"] for x in d["preflight_opens"] if "ok" in x)
    assert d["rccl_ranks_seen"]["torch_distributed"] == 0          # gloo on one GPU: no RCCL rank, and the line says so


def test_rccl_that_does_not_come_up_falls_back_to_gloo_and_says_so():
    """Two ranks asked to run over RCCL on ONE device: RCCL refuses the duplicate device, the ranks agree on that over the gloo
    control plane, the run goes on with the opens staged through host memory and the line carries the reason -- the same path a
    multi-GPU box takes if its RCCL group fails to come up."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-constraints", "12",
                        "--transport", "nccl", "--one-gpu"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(env_clean, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["same_proof_on_all_ranks"] is True and d["proof_matches_prediction"] is True
    assert d["transport_fallback"] and "RCCL" in d["transport_fallback"] and d["transport"].startswith("gloo")
    assert d["rccl_ranks_seen"]["torch_distributed"] == 0


def test_one_rccl_rank_carries_the_data_plane_beside_a_gloo_control_plane():
    """--force-mpc with one rank over --transport nccl: the RCCL group (one rank) is the data plane, gloo the control plane."""
    d = _line([sys.executable, "bench.py", "--gpus", "1", "--force-mpc", "--steps", "2", "--warmup", "1", "--log-constraints", "12",
               "--transport", "nccl", "--no-cpu-baseline"])
    assert d["value"] > 0 and d["transport_fallback"] is None and d["transport"].startswith("RCCL")


def test_bench_rejects_mismatched_launch():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=ROOT, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "must agree" in r.stderr


# ---- the BASELINE configurations at their OWN sizes (the small-size parity of every one of them is in the other test files) ----
def _own_size(cmd, timeout=1100):
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py"] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                       env=dict(env_clean, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_config3_three_party_groth16_at_2p18():
    """BASELINE config 3 at its own size: 3 parties (three OS processes, here sharing cuda:0 with the opens over gloo), additive
    shares, n = 2^18 - 2: the revealed proof equals the known-trapdoor prediction on the summed shares, on every rank
    (src/groth16.rs:68-183 over MpcField; reveal = sum, share/additive.rs:81-83)."""
    d = _own_size(["--gpus", "3", "--transport", "gloo", "--one-gpu", "--log-constraints", "18", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 3 and d["config"]["constraints"] == (1 << 18) - 2 and d["config"]["parties"] == 3
    assert d["proof_matches_prediction"] is True and d["same_proof_on_all_ranks"] is True
    assert d["opens_in_timed_proofs"]["elements_per_open"] == 1 << 18 and d["opens_in_timed_proofs"]["opens_per_proof"] == 2
    assert all(x["ok"] for x in d["preflight_opens"] if "ok" in x)


def test_config4_marlin_at_2p20():
    """BASELINE config 4 at its own size: Marlin::prove (arkworks/marlin/src/lib.rs:152-319) with |H| = |K| = 2^20 through
    zk_marlin_prove; the oracle's Marlin::verify accepts the proof and rejects a wrong public input."""
    d = _own_size(["--marlin", "--log-constraints", "20", "--steps", "2", "--warmup", "1"])
    assert d["config"]["constraints"] == (1 << 20) - 3 and d["n_gpus"] == 1
    assert d["oracle_verifier_accepts"] is True and d["oracle_verifier_rejects_wrong_input"] is True
    assert d["verified"].startswith("the emitted bytes") and d["proof_bytes"] > 900


def test_config5_spdz_marlin_at_2p22():
    """BASELINE config 5's per-party workload at its own size: SPDZ (malicious-backend) collaborative Marlin, |H| = |K| = 2^22,
    through zk_marlin_prove_shared_spdz -- two parties here (two OS processes on cuda:0, 2 x 64 GB of HBM, opens of 2^23 elements
    over gloo; eight parties of this size need eight GPUs), MAC-checked opens, the oracle's verifier accepts."""
    d = _own_size(["--gpus", "2", "--transport", "gloo", "--one-gpu", "--marlin", "--spdz", "--log-constraints", "22", "--steps", "1",
                   "--warmup", "0"])
    assert d["config"]["constraints"] == (1 << 22) - 3 and d["n_gpus"] == 2 and "SPDZ" in d["config"]["workload"]
    assert d["oracle_verifier_accepts"] is True and d["oracle_verifier_rejects_wrong_input"] is True
    assert d["prover_entry"] == "zk_marlin_prove_shared_spdz" and d["opens_in_timed_proofs"]["opens_per_proof"] >= 8


def test_one_prover_line_over_three_contexts():
    """`bench.py --gpus 3 --one-prover --one-gpu`: one local prover over three contexts (here on one device), no launcher and no
    process group; the line says so, carries the plan, and the proofs equal the single-context ones and the prediction."""
    d = _own_size(["--gpus", "3", "--one-prover", "--one-gpu", "--log-constraints", "16", "--steps", "3", "--warmup", "1"], timeout=600)
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["contexts"] == 3 and d["value"] > 0
    assert d["equals_single_context_proof"] is True and d["proof_matches_prediction"] is True
    assert sorted(set(p[0] for p in d["plan"])) == [0, 1, 2]


def test_single_gpu_line_carries_the_composed_trait_paths():
    """The default command's side legs at a small size: `trait_path` (examples/host_trait_groth16 = create_proof over the trait-shaped
    entry points, plain element types) and `trait_path_collab` (the same over MpcField / MpcG1Affine, P parties) run, every proof of a
    run has the same bytes, and the steady state says how it compares with the resident API."""
    d = _line([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--log-constraints", "12", "--no-micro", "--no-cpu-baseline",
               "--no-predict"])
    tp = d["trait_path"]
    assert "error" not in tp, tp
    for mode in ("cache_packed", "cache_strided", "cache_strided_trust", "nocache_packed"):
        leg = tp[mode]
        assert "error" not in leg and leg["same_bytes_every_proof"] is True and leg["lib_vs_resident_api"] > 0, (mode, leg)
    assert len({tp[m]["proof_sha"] for m in ("cache_packed", "cache_strided", "cache_strided_trust", "nocache_packed")}) == 1
    tc = d["trait_path_collab"]
    assert "error" not in tc, tc
    legs = [k for k, v in tc.items() if isinstance(v, dict) and "parties" in v]
    assert len(legs) >= 4, tc.keys()
    for k in legs:
        assert tc[k]["same_bytes_every_proof"] is True and tc[k]["steady_state_max_lib_ms"] > 0, (k, tc[k])
