#!/usr/bin/env python3
"""Soak of the MSMs started ahead (run on an MI355X from the repo root): 1 500 rounds of a host-slice transform followed by the MSM over its
output (the H pattern) and three MSMs over one scalar vector (A, B in G1, B in G2), every fiftieth round with one scalar changed between two
calls; the free device memory must not move between round 100 and round 1 500, and the counters say what was started, taken and dropped."""
import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np, ctypes as C
import zk_mpc_amd as Z, zk_mpc_amd.convert as cv
import zkref as O
import torch
ctx = Z.Context(0)
rng = O.Prng(5)
n, log_n = 4095, 12
tabs = []
for g in (1, 1, 1, 2):
    dk = ctx.upload(cv.fr_to_mont([rng.fr() for _ in range(n)]))
    tb = ctx.fixed_base(dk.ptr, n, g, cv.fr_to_mont([1])[0]); tabs.append((g, np.ascontiguousarray(tb.download()))); tb.free(); dk.free()
rs = np.random.RandomState(1)
def vec(m):
    a = rs.randint(0, 1 << 62, size=(m, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1); return a
def free_mem():
    ctx.sync(); return torch.cuda.mem_get_info(0)[0]
marks = {}
for it in range(1501):
    h = ctx.coset_ifft_in_place(vec(n + 1), log_n)
    ctx.multi_scalar_mul_g1(tabs[0][1], h)                 # H after the transform
    z = vec(n)
    if it % 50 == 7: z2 = z.copy(); z2[5, 0] ^= np.uint64(1)
    ctx.multi_scalar_mul_g1(tabs[1][1], z)                 # A, B1, B2 over one vector
    ctx.multi_scalar_mul_g1(tabs[2][1], z2 if it % 50 == 7 else z)
    ctx.multi_scalar_mul_g2(tabs[3][1], z)
    if it in (100, 500, 1000, 1500): marks[it] = free_mem()
out = np.zeros(3, dtype=np.uint64); ctx.lib.zk_msm_speculate_stats(ctx.h, out.ctypes.data_as(C.c_void_p))
print("free bytes at iterations", marks, "stats started/taken/dropped", out.tolist())
assert marks[1500] >= marks[100] - (8 << 20), "device memory shrinks with the number of calls"
print("LEAK CHECK ok")
