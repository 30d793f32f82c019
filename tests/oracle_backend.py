"""TEST-ONLY arithmetic backend for zk_mpc_amd.mpc.Party built on the oracle (oracle/zkref.py, zkref_c.py).

It lets the CPU suite run the N-party protocol + transport code (the product's mpc.py) over gloo
without a GPU.  It is never importable from the product package."""
import numpy as np

import zkref as O
import zkref_c as OC
import zk_mpc_amd.convert as cv


def _proj1(p):   # affine tuple -> (18,) uint64 Jacobian Z=1 / (1,1,0)
    one = cv.fq_to_mont_int(1)
    vals = [one, one, 0] if p is None else [cv.fq_to_mont_int(p[0]), cv.fq_to_mont_int(p[1]), one]
    return cv._ints_to_limbs(vals, 6).reshape(-1)


def _proj2(p):
    one = cv.fq_to_mont_int(1)
    if p is None:
        vals = [one, 0, one, 0, 0, 0]
    else:
        vals = [cv.fq_to_mont_int(p[0][0]), cv.fq_to_mont_int(p[0][1]), cv.fq_to_mont_int(p[1][0]), cv.fq_to_mont_int(p[1][1]), one, 0]
    return cv._ints_to_limbs(vals, 6).reshape(-1)


class OraclePk:
    def __init__(self, pk: O.ProvingKey):
        self.pk = pk
        self.a = cv.g1_affine_to_array(pk.a_query)
        self.b1 = cv.g1_affine_to_array(pk.b_g1_query)
        self.b2 = cv.g2_affine_to_array(pk.b_g2_query)
        self.h = cv.g1_affine_to_array(pk.h_query)
        self.l = cv.g1_affine_to_array(pk.l_query)


class OracleBackend:
    def __init__(self, net, r1cs: O.R1CS):
        self.net, self.r1cs = net, r1cs
        self.store = {}
        self.dom = O.Domain(r1cs.num_constraints + r1cs.num_instance)

    # vectors are keys into self.store holding (n,4) Montgomery arrays
    def vec(self, name, n):
        key = (name, n)
        self.store.setdefault(key, np.zeros((n, 4), dtype=np.uint64))
        return key

    def put(self, name, arr):
        key = (name, arr.shape[0])
        self.store[key] = np.array(arr, dtype=np.uint64, copy=True)
        return key

    def const_vec(self, value_mont4, n):
        key = ("const", tuple(int(x) for x in value_mont4), n)
        self.store[key] = np.tile(np.asarray(value_mont4, dtype=np.uint64), (n, 1))
        return key

    def add(self, a, b, out, n): self.store[out] = OC.fr_vec_op(1, self.store[a], self.store[b])
    def sub(self, a, b, out, n): self.store[out] = OC.fr_vec_op(2, self.store[a], self.store[b])

    def king_share(self, values, n, key32=None):
        """Test-side mirror of GpuBackend.king_share (numpy shares, the product's transport scatter)."""
        import torch
        net = self.net
        parts = None
        if net.is_leader():
            rs = np.random.RandomState(int.from_bytes((key32 or b'\x05' * 32)[:4], 'little') & 0x7FFFFFFF)
            last = np.array(self.store[values], copy=True)
            parts = []
            for _ in range(net.n - 1):
                t = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
                t[:, 3] &= np.uint64((1 << 60) - 1)
                last = OC.fr_vec_op(2, last, t)
                parts.append(t)
            parts.append(last)
        if hasattr(net, "dist"):
            tensors = [torch.from_numpy(np.ascontiguousarray(p).view(np.int64).reshape(-1).copy()) for p in parts] if parts else None
            mine = net.scatter(tensors, n * 32).numpy().view(np.uint64).reshape(n, 4)
        else:
            mine = net.scatter(parts, n * 32)
        return self.put("king_share_%d" % len(self.store), mine)

    def open_vec(self, v, out, n):
        parts = self.net.all_gather_small(self.store[v])
        acc = parts[0]
        for q in parts[1:]:
            acc = OC.fr_vec_op(1, acc, q)
        self.store[out] = acc

    def beaver_combine(self, sx, oy, tx, ty, tz, out, n):
        S = self.store
        z = OC.fr_vec_op(2, S[tz], OC.fr_vec_op(0, S[sx], S[ty]))
        z = OC.fr_vec_op(2, z, OC.fr_vec_op(0, S[oy], S[tx]))
        if self.net.is_leader():
            z = OC.fr_vec_op(1, z, OC.fr_vec_op(0, S[sx], S[oy]))
        S[out] = z

    def domain_size(self, r1cs): return self.dom.size

    def witness_map_pre(self, r1cs, z, a, b, c):
        R, D, lg = self.r1cs, self.dom.size, self.dom.log_size
        zz = cv.fr_from_mont(self.store[z])
        rows = lambda M: [O.evaluate_constraint(row, zz) for row in M] + [0] * (D - R.num_constraints)
        va, vb, vc = rows(R.a), rows(R.b), rows(R.c)
        for i in range(R.num_instance):
            va[R.num_constraints + i] = zz[i]
        for key, v in ((a, va), (b, vb), (c, vc)):
            m = OC.fft(cv.fr_to_mont(v), lg, 1, 0)
            self.store[key] = OC.fft(m, lg, 0, 1)

    def witness_map_post(self, r1cs, ab, c):
        zinv = pow(self.dom.evaluate_vanishing_polynomial(O.FR_GENERATOR), -1, O.R_MOD)
        d = OC.fr_vec_op(2, self.store[ab], self.store[c])
        d = OC.fr_vec_op(0, d, np.tile(cv.fr_to_mont([zinv])[0], (self.dom.size, 1)))
        self.store[ab] = OC.fft(d, self.dom.log_size, 1, 1)

    def msms_presort(self, pk, r1cs, z):
        pass                                   # a scheduling hint of the device backend

    def msms_begin(self, pk, r1cs, z):
        pass

    def msms(self, pk: OraclePk, r1cs, z, h):
        zz, hh = self.store[z], self.store[h]
        ni = self.r1cs.num_instance
        g1 = np.stack([OC.msm_g1(pk.h, hh), OC.msm_g1(pk.l, zz[ni:]), OC.msm_g1(pk.a[1:], zz[1:]), OC.msm_g1(pk.b1[1:], zz[1:])])
        return g1, OC.msm_g2(pk.b2[1:], zz[1:])

    # group / field helpers on projective arrays
    def _a1(self, p): return cv.g1_projective_to_affine(p)
    def _a2(self, p): return cv.g2_projective_to_affine(p)
    def g1_add(self, a, b): return _proj1(O.g1_add(self._a1(a), self._a1(b)))
    def g2_add(self, a, b): return _proj2(O.g2_add(self._a2(a), self._a2(b)))
    def g1_neg(self, a): return _proj1(O.g1_neg(self._a1(a)))
    def g2_neg(self, a): return _proj2(O.g2_neg(self._a2(a)))
    def g1_mul(self, a, k): return _proj1(O.g1_mul(self._a1(a), cv.fr_from_mont(np.asarray(k).reshape(1, 4))[0]))
    def g2_mul(self, a, k): return _proj2(O.g2_mul(self._a2(a), cv.fr_from_mont(np.asarray(k).reshape(1, 4))[0]))
    def g1_zero(self): return _proj1(None)
    def g2_zero(self): return _proj2(None)
    def g1_serialize(self, a): return O.g1_serialize(self._a1(a))
    def g2_serialize(self, a): return O.g2_serialize(self._a2(a))
    def fr_add(self, a, b): return OC.fr_vec_op(1, np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
    def fr_sub(self, a, b): return OC.fr_vec_op(2, np.asarray(a).reshape(1, 4), np.asarray(b).reshape(1, 4))[0]
    def fr_one(self): return cv.fr_to_mont([1])[0]
    def is_zero_vec(self, v, n): return not np.any(self.store[v])

    def pk_points(self, pk: OraclePk):
        k = pk.pk
        return dict(alpha_g1=_proj1(k.alpha_g1), beta_g1=_proj1(k.beta_g1), delta_g1=_proj1(k.delta_g1),
                    beta_g2=_proj2(k.beta_g2), delta_g2=_proj2(k.delta_g2), a0=_proj1(k.a_query[0]),
                    b0_g1=_proj1(k.b_g1_query[0]), b0_g2=_proj2(k.b_g2_query[0]))


def additive_shares(vals, n_parties, rng, public_prefix=0):
    """Per-party share vectors of `vals`; the first `public_prefix` entries are public (leader holds them)."""
    shares = [[0] * len(vals) for _ in range(n_parties)]
    for i, v in enumerate(vals):
        if i < public_prefix:
            shares[0][i] = v
        else:
            sh = O.additive_share(v, n_parties, rng)
            for p in range(n_parties):
                shares[p][i] = sh[p]
    return shares
