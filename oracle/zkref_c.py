"""oracle/zkref_c.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libzkref.so (zkref.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Arrays are numpy uint64 in the product's C-ABI layouts (Fr: (n,4) Montgomery; G1 affine: (n,12);
G2 affine: (n,24); projective: 18 / 36 words).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzkref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libzkref.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.ref_bench_mul_chain_prove.restype = C.c_double
        _lib.ref_witness_map.restype = C.c_uint32
        _lib.ref_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class R1csT(C.Structure):
    _fields_ = [("nc", C.c_size_t), ("ni", C.c_size_t), ("nw", C.c_size_t),
                ("a_rp", C.c_void_p), ("a_col", C.c_void_p), ("a_coeff", C.c_void_p),
                ("b_rp", C.c_void_p), ("b_col", C.c_void_p), ("b_coeff", C.c_void_p),
                ("c_rp", C.c_void_p), ("c_col", C.c_void_p), ("c_coeff", C.c_void_p)]


class PkT(C.Structure):
    _fields_ = [("alpha_g1", C.c_void_p), ("beta_g1", C.c_void_p), ("delta_g1", C.c_void_p),
                ("beta_g2", C.c_void_p), ("delta_g2", C.c_void_p),
                ("a_query", C.c_void_p), ("a_len", C.c_size_t),
                ("b_g1_query", C.c_void_p), ("b_g1_len", C.c_size_t),
                ("b_g2_query", C.c_void_p), ("b_g2_len", C.c_size_t),
                ("h_query", C.c_void_p), ("h_len", C.c_size_t),
                ("l_query", C.c_void_p), ("l_len", C.c_size_t)]


class R1cs:
    """CSR triple (row_ptr uint32, col uint32, coeff (nnz,4) uint64 Montgomery) x 3."""

    def __init__(self, num_instance, num_witness, a, b, c):
        self.keep = []
        self.t = R1csT()
        self.t.nc = len(a[0]) - 1
        self.t.ni, self.t.nw = num_instance, num_witness
        for name, (rp, col, coeff) in zip("abc", (a, b, c)):
            rp = np.ascontiguousarray(rp, dtype=np.uint32)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            coeff = np.ascontiguousarray(coeff, dtype=np.uint64)
            self.keep += [rp, col, coeff]
            setattr(self.t, name + "_rp", rp.ctypes.data)
            setattr(self.t, name + "_col", col.ctypes.data)
            setattr(self.t, name + "_coeff", coeff.ctypes.data)
        self.domain_log = max(0, (self.t.nc + self.t.ni - 1).bit_length())


class Pk:
    def __init__(self, alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_query, b_g1_query, b_g2_query, h_query, l_query):
        self.keep = []
        self.t = PkT()
        for name, arr in (("alpha_g1", alpha_g1), ("beta_g1", beta_g1), ("delta_g1", delta_g1), ("beta_g2", beta_g2),
                          ("delta_g2", delta_g2)):
            arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
            self.keep.append(arr)
            setattr(self.t, name, arr.ctypes.data)
        for name, arr in (("a_query", a_query), ("b_g1_query", b_g1_query), ("b_g2_query", b_g2_query),
                          ("h_query", h_query), ("l_query", l_query)):
            arr = np.ascontiguousarray(arr, dtype=np.uint64)
            self.keep.append(arr)
            setattr(self.t, name, arr.ctypes.data)
            setattr(self.t, name.replace("_query", "_len"), arr.shape[0])


def fr_vec_op(op, a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().ref_fr_vec_op(op, _p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def fq_mul(a6, b6):
    a6 = np.ascontiguousarray(a6, dtype=np.uint64); b6 = np.ascontiguousarray(b6, dtype=np.uint64)
    out = np.zeros(6, dtype=np.uint64)
    lib().ref_fq_mul(_p(a6), _p(b6), _p(out))
    return out


def fft(v, log_n, inverse, coset, threads=1):
    v = np.array(v, dtype=np.uint64, copy=True)
    assert v.shape[0] == 1 << log_n
    lib().ref_fft(_p(v), C.c_uint32(log_n), int(inverse), int(coset), threads)
    return v


def msm_g1(bases, scalars, threads=1):
    bases = np.ascontiguousarray(bases, dtype=np.uint64); scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.zeros(18, dtype=np.uint64)
    lib().ref_msm_g1(_p(bases), _p(scalars), C.c_size_t(n), _p(out), threads)
    return out


def msm_g2(bases, scalars, threads=1):
    bases = np.ascontiguousarray(bases, dtype=np.uint64); scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.zeros(36, dtype=np.uint64)
    lib().ref_msm_g2(_p(bases), _p(scalars), C.c_size_t(n), _p(out), threads)
    return out


def g1_mul(base12, k4):
    base12 = np.ascontiguousarray(base12, dtype=np.uint64); k4 = np.ascontiguousarray(k4, dtype=np.uint64)
    out = np.zeros(18, dtype=np.uint64)
    lib().ref_g1_mul(_p(base12), _p(k4), _p(out))
    return out


def g2_mul(base24, k4):
    base24 = np.ascontiguousarray(base24, dtype=np.uint64); k4 = np.ascontiguousarray(k4, dtype=np.uint64)
    out = np.zeros(36, dtype=np.uint64)
    lib().ref_g2_mul(_p(base24), _p(k4), _p(out))
    return out


def witness_map(r1cs: R1cs, z, threads=1):
    z = np.ascontiguousarray(z, dtype=np.uint64)
    D = 1 << r1cs.domain_log
    h = np.zeros((D, 4), dtype=np.uint64)
    lib().ref_witness_map(C.byref(r1cs.t), _p(z), _p(h), threads)
    return h


def groth16_prove(r1cs: R1cs, pk: Pk, z, r4, s4, threads=1):
    z = np.ascontiguousarray(z, dtype=np.uint64)
    r4 = np.ascontiguousarray(r4, dtype=np.uint64); s4 = np.ascontiguousarray(s4, dtype=np.uint64)
    proof = np.zeros(192, dtype=np.uint8)
    ph = (C.c_double * 2)()
    lib().ref_groth16_prove(C.byref(r1cs.t), C.byref(pk.t), _p(z), _p(r4), _p(s4), _p(proof), threads, ph)
    return proof.tobytes()


def groth16_predict(r1cs: R1cs, trapdoor7x4, z, h, r4, s4):
    """Known-trapdoor prediction of the 192 proof bytes (Fr arithmetic + 3 scalar muls)."""
    td = np.ascontiguousarray(trapdoor7x4, dtype=np.uint64).reshape(-1)
    z = np.ascontiguousarray(z, dtype=np.uint64); h = np.ascontiguousarray(h, dtype=np.uint64)
    r4 = np.ascontiguousarray(r4, dtype=np.uint64); s4 = np.ascontiguousarray(s4, dtype=np.uint64)
    proof = np.zeros(192, dtype=np.uint8)
    lib().ref_groth16_predict(C.byref(r1cs.t), _p(td), _p(z), _p(h), _p(r4), _p(s4), _p(proof))
    return proof.tobytes()


def mul_chain_csr(n: int):
    """The SURVEY 8(d) mul-chain R1CS as CSR arrays (coeffs = 1 in Montgomery form)."""
    import zkref as O
    one = np.array([(O.FR_MONT_R % O.R_MOD >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
    rp = np.arange(n + 1, dtype=np.uint32)
    idx = lambda j: np.where(j <= n, 2 + j, 1).astype(np.uint32)
    i = np.arange(n, dtype=np.int64)
    ones = np.tile(one, (n, 1))
    return (rp, idx(i), ones), (rp, idx(i + 1), ones), (rp, idx(i + 2), ones)


def bench_mul_chain_prove(n, w0_4, w1_4, pk: Pk, r4, s4, threads):
    w0_4 = np.ascontiguousarray(w0_4, dtype=np.uint64); w1_4 = np.ascontiguousarray(w1_4, dtype=np.uint64)
    r4 = np.ascontiguousarray(r4, dtype=np.uint64); s4 = np.ascontiguousarray(s4, dtype=np.uint64)
    proof = np.zeros(192, dtype=np.uint8)
    ph = (C.c_double * 2)()
    t = lib().ref_bench_mul_chain_prove(C.c_size_t(n), _p(w0_4), _p(w1_4), C.byref(pk.t), _p(r4), _p(s4), _p(proof), threads, ph)
    return float(t), proof.tobytes(), (ph[0], ph[1])


def num_threads():
    return lib().ref_num_threads()
