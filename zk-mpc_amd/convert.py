"""Host-side data-format helpers: Python ints <-> the reference's in-memory layouts.

Layouts: Fr = 4 x u64 LE limbs, Montgomery R = 2^256; Fq = 6 x u64 LE
limbs, Montgomery R = 2^384 (arkworks/algebra/ff/src/fields/macros.rs:107-112).  These helpers
only re-encode integers; all field arithmetic of the product runs in the HIP library.
"""
from __future__ import annotations

import numpy as np

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
Q_MOD = 258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177
_FR_R = (1 << 256) % R_MOD
_FR_RINV = pow(_FR_R, -1, R_MOD)
_FQ_R = (1 << 384) % Q_MOD
_FQ_RINV = pow(_FQ_R, -1, Q_MOD)
_M64 = (1 << 64) - 1


def _ints_to_limbs(vals, nlimbs: int) -> np.ndarray:
    out = np.empty((len(vals), nlimbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(nlimbs):
            out[i, j] = (v >> (64 * j)) & _M64
    return out


def _limbs_to_ints(arr: np.ndarray):
    arr = np.asarray(arr, dtype=np.uint64)
    n, k = arr.shape
    return [sum(int(arr[i, j]) << (64 * j) for j in range(k)) for i in range(n)]


def fr_to_mont(vals) -> np.ndarray:
    """canonical ints -> (n,4) uint64 array in Montgomery form."""
    return _ints_to_limbs([(int(v) % R_MOD) * _FR_R % R_MOD for v in vals], 4)


def fr_from_mont(arr) -> list:
    return [(v * _FR_RINV) % R_MOD for v in _limbs_to_ints(np.asarray(arr).reshape(-1, 4))]


# MNT4-753 base field, the SHE ciphertext modulus (arkworks/curves/mnt4_753/src/fields/fq.rs:51); 12 x u64, R = 2^768
Q753_MOD = 41898490967918953402344214791240637128170709919953949071783502921025352812571106773058893763790338921418070971888253786114353726529584385201591605722013126468931404347949840543007986327743462853720628051692141265303114721689601
_Q753_R = (1 << 768) % Q753_MOD
_Q753_RINV = pow(_Q753_R, -1, Q753_MOD)


def fq753_to_mont(vals) -> np.ndarray:
    """canonical ints -> (n,12) uint64 array in Montgomery form (ark_mnt4_753::Fq layout)."""
    return _ints_to_limbs([(int(v) % Q753_MOD) * _Q753_R % Q753_MOD for v in vals], 12)


def fq753_from_mont(arr) -> list:
    return [(v * _Q753_RINV) % Q753_MOD for v in _limbs_to_ints(np.asarray(arr).reshape(-1, 12))]


def fr_raw(vals) -> np.ndarray:
    """ints -> (n,4) uint64, no Montgomery factor (canonical BigInteger256)."""
    return _ints_to_limbs([int(v) for v in vals], 4)


def fq_to_mont_int(v: int) -> int:
    return (int(v) % Q_MOD) * _FQ_R % Q_MOD


def fq_from_mont_int(v: int) -> int:
    return (int(v) * _FQ_RINV) % Q_MOD


def g1_affine_to_array(points) -> np.ndarray:
    """[(x,y) | None] -> (n,12) uint64 (x|y Montgomery limbs; None = all zero)."""
    out = np.zeros((len(points), 12), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is None:
            continue
        x, y = fq_to_mont_int(p[0]), fq_to_mont_int(p[1])
        for j in range(6):
            out[i, j] = (x >> (64 * j)) & _M64
            out[i, 6 + j] = (y >> (64 * j)) & _M64
    return out


def g2_affine_to_array(points) -> np.ndarray:
    """[((x0,x1),(y0,y1)) | None] -> (n,24) uint64."""
    out = np.zeros((len(points), 24), dtype=np.uint64)
    for i, p in enumerate(points):
        if p is None:
            continue
        vals = [p[0][0], p[0][1], p[1][0], p[1][1]]
        for k, v in enumerate(vals):
            m = fq_to_mont_int(v)
            for j in range(6):
                out[i, 6 * k + j] = (m >> (64 * j)) & _M64
    return out


def _fq_at(row, k) -> int:
    return fq_from_mont_int(sum(int(row[6 * k + j]) << (64 * j) for j in range(6)))


def g1_array_to_affine(arr) -> list:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 12)
    out = []
    for row in arr:
        out.append(None if not row.any() else (_fq_at(row, 0), _fq_at(row, 1)))
    return out


def g2_array_to_affine(arr) -> list:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 24)
    out = []
    for row in arr:
        out.append(None if not row.any() else ((_fq_at(row, 0), _fq_at(row, 1)), (_fq_at(row, 2), _fq_at(row, 3))))
    return out


def g1_projective_to_affine(arr):
    """(18,) uint64 Jacobian (X,Y,Z) Montgomery -> affine tuple or None."""
    row = np.asarray(arr, dtype=np.uint64).reshape(-1)
    x, y, z = _fq_at(row, 0), _fq_at(row, 1), _fq_at(row, 2)
    if z == 0:
        return None
    zi = pow(z, -1, Q_MOD)
    return (x * zi * zi % Q_MOD, y * zi * zi * zi % Q_MOD)


def g2_projective_to_affine(arr):
    row = np.asarray(arr, dtype=np.uint64).reshape(-1)
    f = [_fq_at(row, k) for k in range(6)]
    x, y, z = (f[0], f[1]), (f[2], f[3]), (f[4], f[5])
    if z == (0, 0):
        return None

    def mul(a, b):
        return ((a[0] * b[0] - 5 * a[1] * b[1]) % Q_MOD, (a[0] * b[1] + a[1] * b[0]) % Q_MOD)

    n = pow((z[0] * z[0] + 5 * z[1] * z[1]) % Q_MOD, -1, Q_MOD)
    zi = (z[0] * n % Q_MOD, (-z[1]) * n % Q_MOD)
    zi2 = mul(zi, zi)
    return (mul(x, zi2), mul(y, mul(zi2, zi)))
