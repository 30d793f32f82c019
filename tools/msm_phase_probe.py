import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import zk_mpc_amd as Z, zk_mpc_amd.convert as cv
ctx = Z.Context(0)
rs = np.random.RandomState(1)
for lg in (16, 18, 19, 20):
    n = 1 << lg
    a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.upload(a)
    bases = ctx.fixed_base(d.ptr, n, 1, cv.fr_to_mont([1])[0])
    ctx.msm_dev(bases, 0, d.ptr, n); ctx.sync()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(3): ctx.msm_dev(bases, 0, d.ptr, n)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 3
    print(lg, round(dt * 1e3, 3), {k: round(v[0] / v[1], 3) for k, v in ctx.timers().items()})
    ctx.set_profiling(False)
