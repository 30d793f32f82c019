import sys, time, json, os
sys.path.insert(0, '/root/repo')
import numpy as np
import zk_mpc_amd as Z
ctx = Z.Context(0)
rs = np.random.RandomState(1)
out = {"threads": os.environ.get("ZK_XFER_THREADS"), "div": os.environ.get("ZK_XFER_DIV")}
for log in (14, 16, 18, 20, 22):
    n = 1 << log
    a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    ts = []
    for rep in range(9):
        t0 = time.perf_counter()
        ctx._ck(ctx.lib.zk_fr_fft_in_place(ctx.h, a.ctypes.data, n, log, 0, 0))
        ts.append((time.perf_counter() - t0) * 1e3)
    d = ctx.upload(a)
    tk = []
    for rep in range(5):
        t0 = time.perf_counter(); ctx.ntt_dev(d.ptr, log, False, False); ctx.sync(); tk.append((time.perf_counter() - t0) * 1e3)
    d.free()
    out["log%d" % log] = {"fft_host_ms": round(float(np.median(ts[2:])), 3), "kernel_ms": round(float(np.median(tk[1:])), 3)}
print(json.dumps(out))
