"""G2 MSM phase times (plain table and table with window multiples).  Compare ZK_G2_PAIR=0 / 1 and ZK_G2PAIR_WAVES=1 / 2."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import zk_mpc_amd as Z, zk_mpc_amd.convert as cv
ctx = Z.Context(0)
rs = np.random.RandomState(1)
for lg, pre in ((18, 0), (20, 0), (20, 1)):
    n = 1 << lg
    a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.upload(a)
    bases = ctx.fixed_base(d.ptr, n, 2, cv.fr_to_mont([1])[0])
    if pre: bases.precompute()
    ctx.msm_dev(bases, 0, d.ptr, n); ctx.sync()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(3): r = ctx.msm_dev(bases, 0, d.ptr, n)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 3
    print("G2", lg, "pre" if pre else "plain", round(dt * 1e3, 3), {k: round(v[0] / v[1], 3) for k, v in ctx.timers().items()},
          "result", hex(int(np.asarray(r).view(np.uint64).ravel()[0])), flush=True)
    ctx.set_profiling(False)
    bases.free()
