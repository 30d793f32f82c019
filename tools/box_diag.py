#!/usr/bin/env python3
"""What kind of box is this?  (run on an MI355X from the repo root)  The headline differs by up to 8 % between boxes of the pool at an
unchanged multiply-add peak; this prints, for one box: the multiply-add peak, HBM streaming and random 128-byte-line gather rates
(torch kernels: independent of this library), the clocks rocm-smi reports under load, and the G1 / G2 accumulate kernels alone and
in the queue of proofs -- one JSON line to lay beside another box's."""
import json, os, subprocess, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import torch
import zk_mpc_amd as Z, zk_mpc_amd.convert as cv
import zkref as O

out = {}
dev = torch.device("cuda", 0)
out["device"] = torch.cuda.get_device_name(0)
ctx = Z.Context(0)
out["mad_peak"] = ctx.int_mad_peak(8)
# HBM streaming: 4 GiB device-to-device copies
a = torch.empty(1 << 30, dtype=torch.int32, device=dev); b = torch.empty_like(a)
torch.cuda.synchronize()
def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: b.copy_(a), 5)
out["hbm_copy_GBps"] = round(2 * a.numel() * 4 / t / 1e9, 1)
# random gather of 128-byte lines out of 4 GiB (32 x int32 per row)
rows = a.view(-1, 32)
idx = torch.randint(0, rows.shape[0], (1 << 24,), device=dev)
dst = torch.empty((1 << 24, 32), dtype=torch.int32, device=dev)
t = timed(lambda: torch.index_select(rows, 0, idx, out=dst), 5)
out["gather128_Glines_per_s"] = round((1 << 24) / t / 1e9, 3)
out["gather128_GBps"] = round((1 << 24) * 128 / t / 1e9, 1)
del a, b, rows, idx, dst
torch.cuda.empty_cache()
try:
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20)
    out["rocm_smi"] = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power", "Temperature (Sensor junction)", "Temperature (Sensor memory)"))][:10]
except Exception as e:
    out["rocm_smi"] = repr(e)
# the accumulate kernels: a G1 MSM over a resident 2^20 table with window multiples, alone (per-kernel timers)
rng = O.Prng(1)
n = (1 << 20) - 1
k = np.random.RandomState(3).randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); k[:, 3] &= np.uint64((1 << 60) - 1)
dk = ctx.upload(k)
one = cv.fr_to_mont([1])[0]
for group in (1, 2):
    tb = ctx.fixed_base(dk.ptr, n, group, one); tb.precompute()
    for _ in range(2): ctx.msm_dev(tb, 0, dk.ptr, n)
    ctx.set_profiling(True); ctx.timers()
    for _ in range(5): ctx.msm_dev(tb, 0, dk.ptr, n)
    tm = ctx.timers(); ctx.set_profiling(False)
    out["msm_g%d_alone_ms" % group] = {kk: round(v[0] / max(1, v[1]), 3) for kk, v in tm.items() if kk.startswith("msm_")}
    tb.free()
print(json.dumps(out))
