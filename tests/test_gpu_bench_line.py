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
    assert d["roofline"]["bound"] in ("hbm", "mfma") and d["roofline"]["int_alu"]["frac"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("port", "reference")


@pytest.mark.parametrize("world,extra", [(2, []), (3, ["--spdz"]), (2, ["--marlin"])])
def test_n_party_line_over_gloo_on_one_gpu(world, extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--log-constraints", "12",
               "--transport", "gloo", "--one-gpu"] + extra)
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["value"] > 0
    assert d["opens_in_timed_proofs"]["opens_per_proof"] > 0
    if "--marlin" in extra:
        assert d["oracle_verifier_accepts"] is True and d["equals_python_sequence"] is True
    else:
        assert d["same_proof_on_all_ranks"] is True and d["proof_matches_prediction"] is True
