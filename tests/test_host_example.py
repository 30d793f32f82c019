"""A compiled host above the C ABI (examples/host_groth16.cpp: the flow of arkworks/groth16/src/test.rs:14-77 written against
include/zkmpc_hip.h alone).  CPU: it builds with g++ and, without a GPU, fails loudly instead of computing anything.  GPU: the
192 bytes it prints are the oracle's known-trapdoor prediction for the same circuit, toxic waste and prover randomness."""
import os
import shutil
import subprocess

import pytest

import zkref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    import zk_mpc_amd as Z
    Z.load()                                                 # the library is there (built in-tree)
    libdir = os.path.join(ROOT, "zk-mpc_amd", "lib")
    exe = str(tmp_path / "host_groth16")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "host_groth16.cpp"), "-L", libdir, "-lzkmpc_hip", "-Wl,-rpath," + libdir,
                        "-Wl,--allow-shlib-undefined", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_compiled_host_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the gpu test runs it")
    r = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "proof" not in r.stdout and "zk_ctx_create" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("nc", [1, 100, 5000])
def test_compiled_host_prints_the_predicted_proof(tmp_path, nc):
    exe = build(tmp_path)
    r = subprocess.run([exe, str(nc)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = bytes.fromhex(r.stdout.split("proof ")[1].split()[0])
    a, b = 3, 5
    one = [(1, 2)], [(1, 3)], [(1, 1)]
    r1cs = O.R1CS(2, 2, [one[0]] * nc, [one[1]] * nc, [one[2]] * nc)
    td = O.Trapdoor(2, 3, 5, 7, 11)
    want = O.proof_serialize(*O.predict_proof(r1cs, O.ProvingKeyScalars(r1cs, td), [1, a * b, a, b], 13, 17))
    assert got == want
