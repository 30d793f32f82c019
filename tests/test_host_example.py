"""A compiled host above the C ABI (examples/host_groth16.cpp: the flow of arkworks/groth16/src/test.rs:14-77 written against
include/zkmpc_hip.h alone).  CPU: it builds with g++ and, without a GPU, fails loudly instead of computing anything.  GPU: the
192 bytes it prints are the oracle's known-trapdoor prediction for the same circuit, toxic waste and prover randomness."""
import os
import shutil
import subprocess

import pytest

import zkref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path, name="host_groth16"):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    import zk_mpc_amd as Z
    Z.load()                                                 # the library is there (built in-tree)
    libdir = os.path.join(ROOT, "zk-mpc_amd", "lib")
    exe = str(tmp_path / name)
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", name + ".cpp"), "-L", libdir, "-lzkmpc_hip", "-Wl,-rpath," + libdir,
                        "-Wl,--allow-shlib-undefined", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_compiled_host_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    collab = build(tmp_path, "host_collab_groth16")
    trait = build(tmp_path, "host_trait_groth16")          # create_proof over the trait-shaped entry points (tests/test_gpu_trait_path.py)
    tcollab = build(tmp_path, "host_trait_collab_groth16")  # ... over MpcPairingEngine's element types
    if torch.cuda.is_available():
        pytest.skip("GPU present: the gpu tests run them")
    r = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "proof" not in r.stdout and "zk_ctx_create" in r.stderr
    r = subprocess.run([collab, "2", "8"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "proof" not in r.stdout and "error" in r.stderr
    r = subprocess.run([trait, "4", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "proof" not in r.stdout and "zk_ctx_create" in r.stderr
    r = subprocess.run([tcollab, "4", "1", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "proof" not in r.stdout and "zk_ctx_create" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("nc", [1, 100, 5000])
def test_compiled_host_prints_the_predicted_proof(tmp_path, nc):
    """(nc = 5000: the H query has 8192 points and gets window multiples; their three memory layouts are exercised one by one in
    tests/test_gpu_msm.py::test_window_multiple_layouts.)"""
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


@pytest.mark.gpu
@pytest.mark.parametrize("parties,n", [(1, 20), (2, 37), (3, 1000), (8, 300)])
def test_compiled_host_drives_the_collaborative_prover_over_its_own_transport(tmp_path, parties, n):
    """examples/host_collab_groth16.cpp: parties as C++ threads, zk_net_vtable callbacks written in C++ (shared memory + a
    barrier), zk_groth16_prove_shared per party.  The program itself checks that every party ends with the same bytes and that
    they equal zk_groth16_prove on the summed inputs; here they are compared with the oracle's prediction for the printed r, s."""
    exe = build(tmp_path, "host_collab_groth16")
    r = subprocess.run([exe, str(parties), str(n)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines() if line.split(" ", 1)[0] in ("r", "s", "proof"))
    rr, ss = int.from_bytes(bytes.fromhex(out["r"]), "little"), int.from_bytes(bytes.fromhex(out["s"]), "little")
    r1cs, z = O.mul_chain_r1cs(n, 3, 5)
    td = O.Trapdoor(2, 3, 5, 7, 11)
    want = O.proof_serialize(*O.predict_proof(r1cs, O.ProvingKeyScalars(r1cs, td), z, rr, ss))
    assert bytes.fromhex(out["proof"].strip()) == want
