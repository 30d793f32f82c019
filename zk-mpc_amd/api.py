"""Host-side mirror of the reference's operator interface for the Groth16 proving path.

Method names follow the trait methods the reference dispatches through:
  Field::batch_product_in_place              (arkworks/algebra/ff/src/fields/mod.rs:216)
  EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place, divide_by_vanishing_poly_on_coset_in_place
                                             (arkworks/algebra/poly/src/domain/mod.rs:78-190)
  AffineCurve::multi_scalar_mul              (arkworks/algebra/ec/src/lib.rs:305)
  create_proof / generate_parameters         (src/groth16.rs:68, arkworks/groth16/src/generator.rs:44)
All arithmetic happens in libzkmpc_hip.so; this file only moves buffers and checks return codes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ZkError


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _fr_struct(limbs4) -> _lib.Fr:
    f = _lib.Fr()
    for i in range(4):
        f.l[i] = int(limbs4[i])
    return f


class HostBuf:
    """Page-locked host memory owned by a Context; `.array(shape)` is a numpy uint64 view of it."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._ck(ctx.lib.zk_host_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p.value

    def array(self, shape, dtype=np.uint64) -> np.ndarray:
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if n > self.nbytes:
            raise ValueError("HostBuf too small")
        buf = (C.c_uint8 * n).from_address(self.ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx.lib.zk_host_free(self.ctx.h, C.c_void_p(self.ptr))
        self.ptr = None


class Rng:
    """Byte-exact host generators (zk_rng): FiatShamirRng<Blake2s> (Marlin's transcript), ChaChaRng, rand's StdRng."""

    def __init__(self, h):
        self.lib = _lib.load()
        self.h = h

    @classmethod
    def fiat_shamir(cls, seed_bytes: bytes) -> "Rng":
        lib = _lib.load()
        h = C.c_void_p()
        if lib.zk_fsrng_new(seed_bytes, len(seed_bytes), C.byref(h)) != 0:
            raise ZkError("zk_fsrng_new failed")
        return cls(h)

    @classmethod
    def from_seed(cls, seed32: bytes, rounds: int = 20) -> "Rng":
        lib = _lib.load()
        h = C.c_void_p()
        if len(seed32) != 32 or lib.zk_rng_from_seed(seed32, rounds, C.byref(h)) != 0:
            raise ZkError("zk_rng_from_seed failed")
        return cls(h)

    @classmethod
    def test_rng(cls) -> "Rng":
        """ark_std::test_rng() (arkworks/std/src/rand_helper.rs:31-39): StdRng (ChaCha12) from the fixed seed."""
        seed = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
        return cls.from_seed(seed, 12)

    def absorb(self, data: bytes):
        if self.lib.zk_fsrng_absorb(self.h, data, len(data)) != 0:
            raise ZkError("zk_fsrng_absorb: not a Fiat-Shamir generator")

    def next_u64(self) -> int:
        v = C.c_uint64()
        self.lib.zk_rng_next_u64(self.h, C.byref(v))
        return v.value

    def next_u128(self) -> int:
        v = (C.c_uint64 * 2)()
        self.lib.zk_rng_next_u128(self.h, v)
        return v[0] | (v[1] << 64)

    def next_fr(self) -> np.ndarray:
        """Fr::rand: the element in the reference's in-memory (Montgomery) form, (4,) uint64."""
        out = np.zeros(4, dtype=np.uint64)
        self.lib.zk_rng_next_fr(self.h, _ptr(out))
        return out

    def fill_fr(self, n: int) -> np.ndarray:
        """n draws of Fr::rand: (n, 4) uint64 in the reference's in-memory form."""
        out = np.zeros((n, 4), dtype=np.uint64)
        if n and self.lib.zk_rng_fill_fr(self.h, _ptr(out), n) != 0:
            raise ZkError("zk_rng_fill_fr failed")
        return out

    def fill_bytes(self, n: int) -> bytes:
        buf = (C.c_uint8 * n)()
        self.lib.zk_rng_fill_bytes(self.h, buf, n)
        return bytes(buf)

    def __del__(self):
        try:
            if self.h:
                self.lib.zk_rng_free(self.h)
                self.h = None
        except Exception:
            pass


class DevBuf:
    """A device allocation owned by a Context."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = nbytes
        pooled = ctx._pool.get(nbytes) if ctx.pooling else None
        if pooled:
            self.ptr = pooled.pop()
            return
        p = C.c_void_p()
        ctx._ck(ctx.lib.zk_dev_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p.value

    def at(self, byte_offset: int) -> int:
        return self.ptr + byte_offset

    def free(self):
        if self.ptr:
            if self.ctx.pooling and self.ctx.h:
                # all work of a context is ordered on its stream (the MSM entry points synchronise before they return),
                # so a buffer handed to a later call is only touched after every earlier kernel that used it
                self.ctx._pool.setdefault(self.nbytes, []).append(self.ptr)
            else:
                self.ctx.lib.zk_dev_free(self.ctx.h, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_SIZE_MAX = (1 << 64) - 1


class _CMpcFieldLayout(C.Structure):
    _fields_ = [("stride", C.c_size_t), ("off_tag", C.c_size_t), ("off_public", C.c_size_t), ("off_share", C.c_size_t), ("off_mac", C.c_size_t),
                ("tag_public", C.c_uint8), ("tag_shared", C.c_uint8)]


class _CAffineLayout(C.Structure):
    _fields_ = [("stride", C.c_size_t), ("off_x", C.c_size_t), ("off_y", C.c_size_t), ("off_infinity", C.c_size_t)]


class CMpcGroupLayout(C.Structure):
    _fields_ = [("point", _CAffineLayout), ("off_tag", C.c_size_t), ("tag_public", C.c_uint8)]


class MpcFieldLayout:
    """zk_mpc_field_layout: MpcField<Fr, S> = enum { Public(Fr), Shared(S) } as rustc lays it out -- a discriminant byte and the
    payload at 8-byte alignment; tag_last puts the discriminant behind the payload (and swaps its values): nothing may depend on it."""

    def __init__(self, spdz: bool = False, tag_last: bool = False):
        pay = 64 if spdz else 32
        base = 0 if tag_last else 8
        self.spdz, self.stride = spdz, pay + 8
        self.off_tag = pay if tag_last else 0
        self.off_public = self.off_share = base
        self.off_mac = base + 32 if spdz else _SIZE_MAX
        self.tag_public, self.tag_shared = (1, 0) if tag_last else (0, 1)
        self.c = _CMpcFieldLayout(self.stride, self.off_tag, self.off_public, self.off_share, self.off_mac, self.tag_public, self.tag_shared)


class MpcVec:
    """A Vec<MpcField<Fr, S>> in host memory: raw bytes (n, stride) in the given layout (padding bytes deliberately non-zero)."""

    def __init__(self, lay: MpcFieldLayout, n: int):
        self.lay, self.n = lay, n
        self.raw = np.full((n, lay.stride), 0xA5, dtype=np.uint8)

    def set(self, shared, lane0, lane1=None):
        """shared: (n,) bool; lane0: (n,4) uint64 -- the public value or the share; lane1: the MAC share (SPDZ)."""
        L = self.lay
        shared = np.asarray(shared, dtype=bool)
        self.raw[:, L.off_tag] = np.where(shared, L.tag_shared, L.tag_public).astype(np.uint8)
        b0 = np.ascontiguousarray(lane0, dtype=np.uint64).view(np.uint8).reshape(self.n, 32)
        self.raw[:, L.off_public:L.off_public + 32] = b0                  # (Public and Shared payloads start at the same offset here)
        if L.spdz:
            b1 = np.ascontiguousarray(lane1 if lane1 is not None else lane0, dtype=np.uint64).view(np.uint8).reshape(self.n, 32)
            self.raw[shared, L.off_mac:L.off_mac + 32] = b1[shared]
        return self

    def shared(self):
        return self.raw[:, self.lay.off_tag] != self.lay.tag_public

    def lane(self, k: int = 0):
        off = self.lay.off_share if k == 0 else self.lay.off_mac
        return np.ascontiguousarray(self.raw[:, off:off + 32]).view(np.uint64).reshape(self.n, 4)


def mpc_group_layout(group: int, spdz: bool = False, tag_last: bool = False) -> CMpcGroupLayout:
    """zk_mpc_group_layout of MpcG1Affine / MpcG2Affine: GroupAffine {x, y, infinity: bool} (2 fe + 8 bytes; a SPDZ share holds two)."""
    fe = 48 if group == 1 else 96
    aff = 2 * fe + 8
    pay = 2 * aff if spdz else aff
    gb = 0 if tag_last else 8
    return CMpcGroupLayout(_CAffineLayout(pay + 8, gb, gb + fe, gb + 2 * fe), pay if tag_last else 0, 1 if tag_last else 0)


def mpc_wrap_points(points: np.ndarray, lay: CMpcGroupLayout, shared_rows=()) -> np.ndarray:
    """(n, 12 | 24) uint64 packed points -> (n, stride) bytes of Public(GroupAffine) wrappers; rows in shared_rows get the other tag."""
    n, words = points.shape
    fe = words * 4
    raw = np.full((n, lay.point.stride), 0x5A, dtype=np.uint8)
    b = np.ascontiguousarray(points, dtype=np.uint64).view(np.uint8).reshape(n, 2 * fe)
    raw[:, lay.point.off_x:lay.point.off_x + fe] = b[:, :fe]
    raw[:, lay.point.off_y:lay.point.off_y + fe] = b[:, fe:]
    raw[:, lay.point.off_infinity] = (~b.any(axis=1)).astype(np.uint8)
    raw[:, lay.off_tag] = lay.tag_public
    for r in shared_rows:
        raw[r, lay.off_tag] = lay.tag_public ^ 1
    return raw


class Context:
    """One MPC party / one GPU (zk_ctx)."""

    def __init__(self, device: int = 0, party_id: int = 0, n_parties: int = 1):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.zk_ctx_create(device, party_id, n_parties, C.byref(h))
        if rc != 0:
            raise ZkError("zk_ctx_create failed (rc=%d): no usable HIP device %d -- this library has no CPU path" % (rc, device))
        self.h = h
        self.device, self.party_id, self.n_parties = device, party_id, n_parties
        self.pooling = False        # keep freed DevBufs for same-size reuse (no hipMalloc / hipFree on a proving path)
        self._pool = {}

    def drop_pool(self):
        for ptrs in self._pool.values():
            for ptr in ptrs:
                self.lib.zk_dev_free(self.h, C.c_void_p(ptr))
        self._pool = {}

    def close(self):
        if self.h:
            self.drop_pool()
            self.lib.zk_ctx_destroy(self.h)
            self.h = None

    def _ck(self, rc: int):
        if rc != 0:
            raise ZkError("libzkmpc_hip error %d: %s" % (rc, (self.lib.zk_last_error(self.h) or b"").decode()))

    # ---- buffers ----
    def alloc(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def upload(self, arr: np.ndarray) -> DevBuf:
        arr = np.ascontiguousarray(arr)
        b = DevBuf(self, max(arr.nbytes, 16))
        if arr.nbytes:
            self._ck(self.lib.zk_memcpy_h2d(self.h, C.c_void_p(b.ptr), _ptr(arr), arr.nbytes))
        return b

    def download(self, buf, shape, dtype=np.uint64, byte_offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        ptr = buf.ptr if isinstance(buf, DevBuf) else int(buf)
        if out.nbytes:
            self._ck(self.lib.zk_memcpy_d2h(self.h, _ptr(out), C.c_void_p(ptr + byte_offset), out.nbytes))
        return out

    def sync(self):
        self._ck(self.lib.zk_ctx_sync(self.h))

    def stream(self) -> int:
        return self.lib.zk_ctx_stream(self.h)

    def int_mad_peak(self, launches: int = 12) -> dict:
        """zk_diag_int_mad_peak: v_mad_u64_u32 lane-ops/s of this device, measured now (best and median launch)."""
        best, med = C.c_double(0), C.c_double(0)
        self._ck(self.lib.zk_diag_int_mad_peak(self.h, launches, C.byref(best), C.byref(med)))
        return {"best": best.value, "median": med.value, "launches": launches}

    def set_profiling(self, on: bool):
        self._ck(self.lib.zk_set_profiling(self.h, int(on)))

    def timers(self) -> dict:
        """{phase: (total device ms, launches)} since the previous call."""
        names = C.create_string_buffer(64 * 64)
        ms = (C.c_float * 64)()
        cnt = (C.c_int * 64)()
        k = self.lib.zk_last_timers(self.h, names, 64, ms, cnt, 64)
        return {names.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode(): (float(ms[i]), int(cnt[i])) for i in range(k)}

    # ---- Fr vectors ----
    def fr_vec_op_dev(self, op: int, a, b, out, n: int):
        self._ck(self.lib.zk_fr_vec_op_dev(self.h, op, C.c_void_p(int(a)), C.c_void_p(int(b)), C.c_void_p(int(out)), n))

    def fr_vec_scale_dev(self, a, k_mont4, out, n: int):
        k = _fr_struct(k_mont4)
        self._ck(self.lib.zk_fr_vec_scale_dev(self.h, C.c_void_p(int(a)), C.byref(k), C.c_void_p(int(out)), n))

    def batch_product_in_place(self, selfs: np.ndarray, others: np.ndarray):
        """Field::batch_product_in_place on host slices ((n,4) uint64 Montgomery)."""
        assert selfs.dtype == np.uint64 and selfs.flags.c_contiguous
        others = np.ascontiguousarray(others, dtype=np.uint64)
        n = min(selfs.shape[0], others.shape[0])
        self._ck(self.lib.zk_fr_batch_product_in_place(self.h, _ptr(selfs), _ptr(others), n))

    # ---- EvaluationDomain ----
    def ntt_dev(self, buf, log_n: int, inverse: bool, coset: bool):
        self._ck(self.lib.zk_fr_ntt_dev(self.h, C.c_void_p(int(buf)), log_n, int(inverse), int(coset)))

    def _fft_host(self, vec: np.ndarray, log_n: int, inverse: int, coset: int) -> np.ndarray:
        n = vec.shape[0]
        N = 1 << log_n
        out = np.zeros((N, 4), dtype=np.uint64)
        out[:n] = vec
        self._ck(self.lib.zk_fr_fft_in_place(self.h, _ptr(out), n, log_n, inverse, coset))
        return out

    def fft_in_place(self, vec, log_n):        return self._fft_host(vec, log_n, 0, 0)
    def ifft_in_place(self, vec, log_n):       return self._fft_host(vec, log_n, 1, 0)
    def coset_fft_in_place(self, vec, log_n):  return self._fft_host(vec, log_n, 0, 1)
    def coset_ifft_in_place(self, vec, log_n): return self._fft_host(vec, log_n, 1, 1)

    def divide_by_vanishing_poly_on_coset_in_place_dev(self, buf, log_n: int):
        self._ck(self.lib.zk_fr_divide_by_vanishing_on_coset_dev(self.h, C.c_void_p(int(buf)), log_n))

    # ---- the same dispatch points on MpcField / MpcG1Affine elements in the caller's enum layout (csrc/mpc_host.hip) ----
    def mpc_fft_in_place(self, vec: "MpcVec", n: int, log_n: int, inverse: bool, coset: bool):
        """EvaluationDomain::*fft_in_place(&mut Vec<MpcField>) (src/groth16.rs:278-303): n elements read, 2^log_n written in place."""
        assert vec.raw.shape[0] >= (1 << log_n)
        self._ck(self.lib.zk_mpc_fft_in_place(self.h, _ptr(vec.raw), n, C.byref(vec.lay.c), log_n, int(inverse), int(coset)))

    def mpc_divide_by_vanishing_on_coset_in_place(self, vec: "MpcVec", log_n: int):
        self._ck(self.lib.zk_mpc_divide_by_vanishing_on_coset_in_place(self.h, _ptr(vec.raw), C.byref(vec.lay.c), log_n))

    def mpc_batch_product_in_place(self, selfs: "MpcVec", others: "MpcVec", n: int, net_vtable=None, triple=None) -> int:
        """MpcField::batch_product_in_place (mpc-algebra/src/wire/field.rs:917-958).  triple: None (DummyFieldTripleSource) or a
        list of 3 (additive) / 6 (SPDZ) (n,4) uint64 arrays.  Returns the payload bytes this party sent to opens."""
        sent = C.c_uint64(0)
        tp = None
        if triple is not None:
            triple = [np.ascontiguousarray(t, dtype=np.uint64) for t in triple]
            tp = (C.c_void_p * len(triple))(*[t.ctypes.data for t in triple])
        self._ck(self.lib.zk_mpc_batch_product_in_place(self.h, _ptr(selfs.raw), _ptr(others.raw), n, C.byref(selfs.lay.c), tp,
                                                        C.byref(net_vtable) if net_vtable is not None else None, C.byref(sent)))
        return int(sent.value)

    def mpc_msm(self, group: int, bases_raw: np.ndarray, n_bases: int, base_layout, scalars: "MpcVec", n_scalars: int):
        """MpcG1Affine / MpcG2Affine::multi_scalar_mul (wire/pairing.rs:714-777): (lane 0, lane 1, every scalar public?)."""
        w = 18 if group == 1 else 36
        out = np.zeros(2 * w, dtype=np.uint64)
        pub = C.c_int(0)
        fn = self.lib.zk_mpc_msm_g1 if group == 1 else self.lib.zk_mpc_msm_g2
        self._ck(fn(self.h, _ptr(bases_raw), n_bases, C.byref(base_layout), _ptr(scalars.raw), n_scalars, C.byref(scalars.lay.c), _ptr(out), C.byref(pub)))
        return out[:w].copy(), out[w:].copy(), bool(pub.value)

    # ---- AffineCurve::multi_scalar_mul ----
    def multi_scalar_mul_g1(self, bases: np.ndarray, scalars: np.ndarray) -> np.ndarray:
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(18, dtype=np.uint64)
        self._ck(self.lib.zk_msm_g1(self.h, _ptr(bases), bases.shape[0], _ptr(scalars), scalars.shape[0], _ptr(out)))
        return out

    def multi_scalar_mul_g2(self, bases: np.ndarray, scalars: np.ndarray) -> np.ndarray:
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(36, dtype=np.uint64)
        self._ck(self.lib.zk_msm_g2(self.h, _ptr(bases), bases.shape[0], _ptr(scalars), scalars.shape[0], _ptr(out)))
        return out

    def bases_upload(self, arr: np.ndarray, group: int) -> "Bases":
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        h = C.c_void_p()
        fn = self.lib.zk_bases_upload_g1 if group == 1 else self.lib.zk_bases_upload_g2
        self._ck(fn(self.h, _ptr(arr), arr.shape[0], C.byref(h)))
        return Bases(self, h, group)

    def bases_deserialize_uncompressed(self, data: bytes, n: int, group: int) -> "Bases":
        return self._bases_deserialize(data, n, group, 0)

    def bases_deserialize_compressed(self, data: bytes, n: int, group: int) -> "Bases":
        return self._bases_deserialize(data, n, group, 1)

    def _bases_deserialize(self, data: bytes, n: int, group: int, compressed: int) -> "Bases":
        buf = np.frombuffer(data, dtype=np.uint8)
        assert buf.size == n * self.lib.zk_point_serialized_size(group, compressed)
        out = C.c_void_p()
        fn = self.lib.zk_bases_deserialize_compressed if compressed else self.lib.zk_bases_deserialize_uncompressed
        self._ck(fn(self.h, group, _ptr(np.ascontiguousarray(buf)) if n else None, n, C.byref(out)))
        return Bases(self, out, group)

    def fixed_base(self, scalars_dev, n: int, group: int, gen_k_mont4) -> "Bases":
        h = C.c_void_p()
        k = _fr_struct(gen_k_mont4)
        fn = self.lib.zk_fixed_base_g1_dev if group == 1 else self.lib.zk_fixed_base_g2_dev
        self._ck(fn(self.h, C.byref(k), C.c_void_p(int(scalars_dev)), n, C.byref(h)))
        return Bases(self, h, group)

    def msm_dev(self, bases: "Bases", base_offset: int, scalars_dev, n: int) -> np.ndarray:
        out = np.zeros(18 if bases.group == 1 else 36, dtype=np.uint64)
        fn = self.lib.zk_msm_g1_dev if bases.group == 1 else self.lib.zk_msm_g2_dev
        self._ck(fn(self.h, bases.h, base_offset, C.c_void_p(int(scalars_dev)), n, _ptr(out)))
        return out

    # ---- dense polynomials / KZG10 (names follow DensePolynomial / KZG10 in the reference) ----
    def msm_batch_dev(self, jobs):
        """jobs: list of (Bases, base_offset, scalars_dev, n).  Returns one projective array per job (18 or 36 u64)."""
        k = len(jobs)
        bases = (C.c_void_p * k)(*[j[0].h for j in jobs])
        offs = (C.c_size_t * k)(*[j[1] for j in jobs])
        scal = (C.c_void_p * k)(*[int(j[2]) for j in jobs])
        lens = (C.c_size_t * k)(*[j[3] for j in jobs])
        outs = [np.zeros(18 if j[0].group == 1 else 36, dtype=np.uint64) for j in jobs]
        outp = (C.c_void_p * k)(*[o.ctypes.data for o in outs])
        self._ck(self.lib.zk_msm_batch_dev(self.h, k, bases, offs, scal, lens, outp))
        return outs

    def fr_powers_dev(self, base4, start4, n: int, out):
        b, s_ = _fr_struct(base4), _fr_struct(start4)
        self._ck(self.lib.zk_fr_powers_dev(self.h, C.byref(b), C.byref(s_), n, C.c_void_p(int(out))))

    def batch_inversion_dev(self, v, n: int):
        self._ck(self.lib.zk_fr_batch_inverse_dev(self.h, C.c_void_p(int(v)), n))

    def poly_evaluate_dev(self, coeffs, n: int, point4) -> np.ndarray:
        pt = _fr_struct(point4)
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.zk_poly_evaluate_dev(self.h, C.c_void_p(int(coeffs)), n, C.byref(pt), _ptr(out)))
        return out

    def poly_evaluate_batch_dev(self, polys, points4) -> np.ndarray:
        """polys: [(device pointer, n)], points4: (count, 4) u64 -> (count, 4) u64; two launches and one copy back for all."""
        from ._lib import PolyRef
        k = len(polys)
        refs = (PolyRef * max(k, 1))()
        for i, (ptr, n) in enumerate(polys):
            refs[i].ptr, refs[i].n = int(ptr), int(n)
        pts = np.ascontiguousarray(np.asarray(points4, dtype=np.uint64).reshape(k, 4))
        out = np.zeros((k, 4), dtype=np.uint64)
        self._ck(self.lib.zk_poly_evaluate_batch_dev(self.h, refs, _ptr(pts), k, _ptr(out)))
        return out

    def poly_divide_by_linear_dev(self, coeffs, n: int, z4, q):
        z = _fr_struct(z4)
        rem = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.zk_poly_divide_by_linear_dev(self.h, C.c_void_p(int(coeffs)), n, C.byref(z), C.c_void_p(int(q)), _ptr(rem)))
        return rem

    def poly_divide_by_vanishing_dev(self, coeffs, n: int, log_domain: int, q, r):
        self._ck(self.lib.zk_poly_divide_by_vanishing_dev(self.h, C.c_void_p(int(coeffs)), n, log_domain,
                                                          C.c_void_p(int(q)) if q else None, C.c_void_p(int(r))))

    def poly_mul_dev(self, a, na: int, b, nb: int, out):
        self._ck(self.lib.zk_poly_mul_dev(self.h, C.c_void_p(int(a)), na, C.c_void_p(int(b)), nb, C.c_void_p(int(out))))

    def kzg_commit_dev(self, powers_g: "Bases", coeffs, n: int, powers_gamma_g: "Bases" = None, blind=None, n_blind: int = 0):
        out = np.zeros(18, dtype=np.uint64)
        self._ck(self.lib.zk_kzg_commit_dev(self.h, powers_g.h, C.c_void_p(int(coeffs)), n,
                                            powers_gamma_g.h if powers_gamma_g else None,
                                            C.c_void_p(int(blind)) if blind else None, n_blind, _ptr(out)))
        return out

    def kzg_open_dev(self, powers_g: "Bases", coeffs, n: int, point4, powers_gamma_g: "Bases" = None, blind=None, n_blind: int = 0):
        pt = _fr_struct(point4)
        w = np.zeros(18, dtype=np.uint64)
        rv = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.zk_kzg_open_dev(self.h, powers_g.h, C.c_void_p(int(coeffs)), n, C.byref(pt),
                                          powers_gamma_g.h if powers_gamma_g else None,
                                          C.c_void_p(int(blind)) if blind else None, n_blind, _ptr(w), _ptr(rv)))
        return w, rv

    # ---- Marlin AHP pieces ----
    def memcpy_d2d(self, dst, src, nbytes: int):
        self._ck(self.lib.zk_memcpy_d2d(self.h, C.c_void_p(int(dst)), C.c_void_p(int(src)), nbytes))

    def dev_zero(self, dev, nbytes: int):
        self._ck(self.lib.zk_dev_zero(self.h, C.c_void_p(int(dev)), nbytes))

    def fr_inverse(self, a4):
        out = np.zeros(4, dtype=np.uint64)
        a4 = np.ascontiguousarray(a4, dtype=np.uint64)
        self._ck(self.lib.zk_fr_inverse(_ptr(a4), _ptr(out)))
        return out

    def fr_pow(self, a4, e: int):
        out = np.zeros(4, dtype=np.uint64)
        a4 = np.ascontiguousarray(a4, dtype=np.uint64)
        self._ck(self.lib.zk_fr_pow(_ptr(a4), e, _ptr(out)))
        return out

    def r1cs_matvec_dev(self, r1cs: "R1cs", which: int, z, out, out_len: int):
        self._ck(self.lib.zk_r1cs_matvec_dev(self.h, r1cs.h, which, C.c_void_p(int(z)), C.c_void_p(int(out)), out_len))

    def fr_gather_dev(self, src, idx, n: int, out):
        self._ck(self.lib.zk_fr_gather_dev(self.h, C.c_void_p(int(src)), C.c_void_p(int(idx)), n, C.c_void_p(int(out))))

    def _marlin_args(self, mats, alpha4, beta4, etas4, vv4):
        arr = (_lib.MarlinMatrixEvals * 3)()
        for i, m in enumerate(mats):
            arr[i].row, arr[i].col, arr[i].val = int(m["row"]), int(m["col"]), int(m["val"])
            arr[i].row_col = int(m["row_col"]) if m.get("row_col") else None
        eta = (_lib.Fr * 3)()
        for i in range(3):
            for j in range(4):
                eta[i].l[j] = int(etas4[i][j])
        return arr, _fr_struct(alpha4), _fr_struct(beta4), eta, _fr_struct(vv4)

    def marlin_round3_f_evals_dev(self, on_k, k_size: int, alpha4, beta4, etas4, vv4, f_out):
        arr, a, b, eta, vv = self._marlin_args(on_k, alpha4, beta4, etas4, vv4)
        self._ck(self.lib.zk_marlin_round3_f_evals_dev(self.h, arr, k_size, C.byref(a), C.byref(b), eta, C.byref(vv), C.c_void_p(int(f_out))))

    def marlin_round3_ab_evals_dev(self, on_b, b_size: int, alpha4, beta4, etas4, vv4, a_out, b_out):
        arr, a, b, eta, vv = self._marlin_args(on_b, alpha4, beta4, etas4, vv4)
        self._ck(self.lib.zk_marlin_round3_ab_evals_dev(self.h, arr, b_size, C.byref(a), C.byref(b), eta, C.byref(vv),
                                                        C.c_void_p(int(a_out)), C.c_void_p(int(b_out))))

    # ---- SHE ring arithmetic (names follow src/she: Texts / Encodedtext / Ciphertext / Plaintexts) ----
    def _fq753_struct(self, limbs12):
        f = _lib.Fq753()
        for i in range(12):
            f.l[i] = int(limbs12[i])
        return f

    def she_vec_op_dev(self, op: int, a, b, out, n: int):
        self._ck(self.lib.zk_she_vec_op_dev(self.h, op, C.c_void_p(int(a)), C.c_void_p(int(b)) if b else None, C.c_void_p(int(out)), n))

    def she_vec_scale_dev(self, a, k12, out, n: int):
        k = self._fq753_struct(k12)
        self._ck(self.lib.zk_she_vec_scale_dev(self.h, C.c_void_p(int(a)), C.byref(k), C.c_void_p(int(out)), n))

    def encodedtext_mul_dev(self, a, b, out, n: int, batch: int = 1):
        self._ck(self.lib.zk_she_negacyclic_mul_dev(self.h, C.c_void_p(int(a)), C.c_void_p(int(b)), C.c_void_p(int(out)), n, batch))

    def ciphertext_mul_dev(self, x, y, out, n: int, batch: int = 1):
        self._ck(self.lib.zk_she_ciphertext_mul_dev(self.h, C.c_void_p(int(x)), C.c_void_p(int(y)), C.c_void_p(int(out)), n, batch))

    def ciphertext_encrypt_from_dev(self, e, pk_a, pk_b, r, p12, out, n: int, batch: int = 1):
        p = self._fq753_struct(p12)
        self._ck(self.lib.zk_she_encrypt_dev(self.h, C.c_void_p(int(e)), C.c_void_p(int(pk_a)), C.c_void_p(int(pk_b)),
                                             C.c_void_p(int(r)), C.byref(p), C.c_void_p(int(out)), n, batch))

    def ciphertext_decrypt_dev(self, ct, sk, out, n: int, batch: int = 1):
        self._ck(self.lib.zk_she_decrypt_dev(self.h, C.c_void_p(int(ct)), C.c_void_p(int(sk)), C.c_void_p(int(out)), n, batch))

    def plaintexts_encode_dev(self, plain_fr, out, n: int, batch: int = 1):
        self._ck(self.lib.zk_she_encode_dev(self.h, C.c_void_p(int(plain_fr)), C.c_void_p(int(out)), n, batch))

    def encodedtext_decode_dev(self, enc, out_fr, n: int, batch: int = 1):
        self._ck(self.lib.zk_she_decode_dev(self.h, C.c_void_p(int(enc)), C.c_void_p(int(out_fr)), n, batch))

    # ---- native transport (RCCL inside the library) ----
    def comm_unique_id(self) -> bytes:
        out = np.zeros(128, dtype=np.uint8)
        self._ck(self.lib.zk_comm_unique_id(_ptr(out)))
        return out.tobytes()

    def comm_init(self, unique_id: bytes, rank: int, n_parties: int):
        buf = np.frombuffer(unique_id, dtype=np.uint8).copy()
        assert buf.size == 128
        self._ck(self.lib.zk_comm_init(self.h, _ptr(buf), rank, n_parties))

    def comm_info(self) -> dict:
        """zk_comm_info: what the context's RCCL communicator reports about itself (n_ranks 0 without one) and which RCCL copy
        the library bound."""
        n, r, d, v = C.c_int(0), C.c_int(-1), C.c_int(-1), C.c_int(0)
        path = C.create_string_buffer(1024)
        self._ck(self.lib.zk_comm_info(self.h, C.byref(n), C.byref(r), C.byref(d), C.byref(v), path, 1024))
        return {"n_ranks": n.value, "rank": r.value, "device": d.value, "rccl_version": v.value, "library": path.value.decode()}

    def comm_set_open_pattern(self, pattern: int):
        """0: by party count, 1: all-gather, 2: all-to-all of slices (every party must choose the same)."""
        self._ck(self.lib.zk_comm_set_open_pattern(self.h, pattern))

    def comm_destroy(self):
        self._ck(self.lib.zk_comm_destroy(self.h))

    def open_sum_fr_dev(self, v, n: int, out):
        self._ck(self.lib.zk_open_sum_fr_dev(self.h, C.c_void_p(int(v)), n, C.c_void_p(int(out))))

    # ---- share algebra ----
    def fr_random_dev(self, out, n: int, key32: bytes = None, stream_id: int = 0):
        """n uniform field elements (zk_fr_random_dev).  key32 = None: keyed from the operating system's CSPRNG."""
        if key32 is not None and len(key32) != 32:
            raise ValueError("key32 must be 32 bytes")
        self._ck(self.lib.zk_fr_random_dev(self.h, key32, C.c_uint64(stream_id), C.c_void_p(int(out)), n))

    def fr_sum_parties_dev(self, gathered, n_parties: int, n: int, out):
        self._ck(self.lib.zk_fr_sum_parties_dev(self.h, C.c_void_p(int(gathered)), n_parties, n, C.c_void_p(int(out))))

    def fr_vec_is_zero_dev(self, v, n: int) -> bool:
        f = C.c_int(0)
        self._ck(self.lib.zk_fr_vec_is_zero_dev(self.h, C.c_void_p(int(v)), n, C.byref(f)))
        return bool(f.value)

    def beaver_combine_dev(self, sx, oy, out, n: int, triple=None):
        tx, ty, tz = (None, None, None) if triple is None else [C.c_void_p(int(t)) for t in triple]
        self._ck(self.lib.zk_beaver_combine_dev(self.h, C.c_void_p(int(sx)), C.c_void_p(int(oy)), tx, ty, tz,
                                                C.c_void_p(int(out)), n))

    # ---- host group helpers ----
    def g1_add(self, a, b):
        out = np.zeros(18, dtype=np.uint64)
        self._ck(self.lib.zk_g1_add(_ptr(a), _ptr(b), _ptr(out)))
        return out

    def g2_add(self, a, b):
        out = np.zeros(36, dtype=np.uint64)
        self._ck(self.lib.zk_g2_add(_ptr(a), _ptr(b), _ptr(out)))
        return out

    def g1_neg(self, a):
        out = np.zeros(18, dtype=np.uint64)
        self._ck(self.lib.zk_g1_neg(_ptr(a), _ptr(out)))
        return out

    def g2_neg(self, a):
        out = np.zeros(36, dtype=np.uint64)
        self._ck(self.lib.zk_g2_neg(_ptr(a), _ptr(out)))
        return out

    def g1_mul(self, a, k_mont4):
        out = np.zeros(18, dtype=np.uint64)
        k = np.ascontiguousarray(k_mont4, dtype=np.uint64)
        self._ck(self.lib.zk_g1_mul(_ptr(a), _ptr(k), _ptr(out)))
        return out

    def g2_mul(self, a, k_mont4):
        out = np.zeros(36, dtype=np.uint64)
        k = np.ascontiguousarray(k_mont4, dtype=np.uint64)
        self._ck(self.lib.zk_g2_mul(_ptr(a), _ptr(k), _ptr(out)))
        return out

    def g1_from_affine(self, a12):
        out = np.zeros(18, dtype=np.uint64)
        a12 = np.ascontiguousarray(a12, dtype=np.uint64)
        self._ck(self.lib.zk_g1_from_affine(_ptr(a12), _ptr(out)))
        return out

    def g2_from_affine(self, a24):
        out = np.zeros(36, dtype=np.uint64)
        a24 = np.ascontiguousarray(a24, dtype=np.uint64)
        self._ck(self.lib.zk_g2_from_affine(_ptr(a24), _ptr(out)))
        return out

    def g1_serialize(self, a) -> bytes:
        out = np.zeros(48, dtype=np.uint8)
        self._ck(self.lib.zk_g1_serialize(_ptr(a), _ptr(out)))
        return out.tobytes()

    def g2_serialize(self, a) -> bytes:
        out = np.zeros(96, dtype=np.uint8)
        self._ck(self.lib.zk_g2_serialize(_ptr(a), _ptr(out)))
        return out.tobytes()

    def fr_op(self, name: str, a4, b4):
        out = np.zeros(4, dtype=np.uint64)
        a4 = np.ascontiguousarray(a4, dtype=np.uint64)
        b4 = np.ascontiguousarray(b4, dtype=np.uint64)
        self._ck(getattr(self.lib, "zk_fr_" + name)(_ptr(a4), _ptr(b4), _ptr(out)))
        return out

    # ---- R1CS / Groth16 ----
    def r1cs_upload(self, num_instance: int, num_witness: int, a, b, c) -> "R1cs":
        """a, b, c: (row_ptr uint32[nc+1], col uint32[nnz], coeff uint64[nnz,4] Montgomery)."""
        keep = []
        h = _lib.R1csHost()
        h.num_constraints = len(a[0]) - 1
        h.num_instance, h.num_witness = num_instance, num_witness
        for name, (rp, col, coeff) in zip("abc", (a, b, c)):
            rp = np.ascontiguousarray(rp, dtype=np.uint32)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            coeff = np.ascontiguousarray(coeff, dtype=np.uint64)
            keep += [rp, col, coeff]
            setattr(h, name + "_row_ptr", rp.ctypes.data)
            setattr(h, name + "_col", col.ctypes.data)
            setattr(h, name + "_coeff", coeff.ctypes.data)
        out = C.c_void_p()
        self._ck(self.lib.zk_r1cs_upload(self.h, C.byref(h), C.byref(out)))
        return R1cs(self, out, h.num_constraints, num_instance, num_witness)

    def r1cs_mul_chain(self, n: int) -> "R1cs":
        out = C.c_void_p()
        self._ck(self.lib.zk_r1cs_mul_chain(self.h, n, C.byref(out)))
        return R1cs(self, out, n, 2, n + 1)

    def mul_chain_assignment_dev(self, n: int, w0_mont4, w1_mont4) -> DevBuf:
        z = self.alloc((n + 3) * 32)
        a, b = _fr_struct(w0_mont4), _fr_struct(w1_mont4)
        self._ck(self.lib.zk_mul_chain_assignment_dev(self.h, n, C.byref(a), C.byref(b), C.c_void_p(z.ptr)))
        return z

    def groth16_setup(self, r1cs: "R1cs", alpha, beta, gamma, delta, tau, g1_k, g2_k) -> "ProvingKey":
        """generate_parameters with explicit toxic waste; all arguments (4,) uint64 Montgomery."""
        s = [_fr_struct(x) for x in (alpha, beta, gamma, delta, tau, g1_k, g2_k)]
        out = C.c_void_p()
        self._ck(self.lib.zk_groth16_setup(self.h, r1cs.h, *[C.byref(x) for x in s], C.byref(out)))
        return ProvingKey(self, out)

    def pk_upload(self, alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_query, b_g1_query, b_g2_query, h_query,
                  l_query) -> "ProvingKey":
        h = _lib.PkHost()
        keep = []

        def put(field, arr, n):
            arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1)
            assert arr.size == n
            C.memmove(C.addressof(getattr(h, field)), arr.ctypes.data, n * 8)

        put("alpha_g1", alpha_g1, 12); put("beta_g1", beta_g1, 12); put("delta_g1", delta_g1, 12)
        put("beta_g2", beta_g2, 24); put("delta_g2", delta_g2, 24)
        for name, arr in (("a_query", a_query), ("b_g1_query", b_g1_query), ("b_g2_query", b_g2_query),
                          ("h_query", h_query), ("l_query", l_query)):
            arr = np.ascontiguousarray(arr, dtype=np.uint64)
            keep.append(arr)
            setattr(h, name, arr.ctypes.data)
            setattr(h, name.replace("_query", "_len"), arr.shape[0])
        out = C.c_void_p()
        self._ck(self.lib.zk_pk_upload(self.h, C.byref(h), C.byref(out)))
        return ProvingKey(self, out)

    def witness_map_dev(self, r1cs: "R1cs", z_dev, h_dev):
        self._ck(self.lib.zk_groth16_witness_map_dev(self.h, r1cs.h, C.c_void_p(int(z_dev)), C.c_void_p(int(h_dev))))

    def witness_map_pre_dev(self, r1cs, z_dev, a, b, c, include_instance=True):
        self._ck(self.lib.zk_groth16_witness_map_pre_dev(self.h, r1cs.h, C.c_void_p(int(z_dev)), int(include_instance),
                                                         C.c_void_p(int(a)), C.c_void_p(int(b)), C.c_void_p(int(c))))

    def witness_map_post_dev(self, r1cs, ab, c):
        self._ck(self.lib.zk_groth16_witness_map_post_dev(self.h, r1cs.h, C.c_void_p(int(ab)), C.c_void_p(int(c))))

    def groth16_hint_next_dev(self, z_next_dev):
        """Announce the assignment of the next create_proof_dev call (same key and constraint system); None withdraws."""
        self._ck(self.lib.zk_groth16_hint_next_dev(self.h, C.c_void_p(int(z_next_dev)) if z_next_dev else None))

    def groth16_chain_fronts(self, on: bool):
        """zk_groth16_chain_fronts: whether an announced small proof's front carries its whole device chain (default on)."""
        self._ck(self.lib.zk_groth16_chain_fronts(self.h, int(on)))

    def host_alloc(self, nbytes: int) -> "HostBuf":
        """Page-locked host memory (zk_host_alloc) viewed as a numpy uint64 array."""
        return HostBuf(self, nbytes)

    def create_proof_queued(self, pk: "ProvingKey", r1cs: "R1cs", z_host: np.ndarray, r_mont4, s_mont4, z_next_host=None) -> bytes:
        """zk_groth16_prove_queued: host assignment -> 192 proof bytes; z_next_host announces the next call's assignment.
        The arrays are passed by address and must be C-contiguous uint64 (n, 4); keep them alive and unchanged."""
        r, s = _fr_struct(r_mont4), _fr_struct(s_mont4)
        out = np.zeros(192, dtype=np.uint8)
        for a in (z_host, z_next_host):
            if a is not None and not (a.flags["C_CONTIGUOUS"] and a.dtype == np.uint64):
                raise ValueError("assignments must be C-contiguous uint64 arrays")
        self._ck(self.lib.zk_groth16_prove_queued(self.h, pk.h, r1cs.h, _ptr(z_host), C.byref(r), C.byref(s),
                                                  _ptr(z_next_host) if z_next_host is not None else None, _ptr(out)))
        return out.tobytes()

    def groth16_msms_presort_dev(self, pk: "ProvingKey", r1cs: "R1cs", z_dev):
        """Enqueue the shared sort of z[1..] ahead of groth16_msms_dev on the same z_dev (asynchronous)."""
        self._ck(self.lib.zk_groth16_msms_presort_dev(self.h, pk.h, r1cs.h, C.c_void_p(int(z_dev))))

    def groth16_msms_begin_dev(self, pk: "ProvingKey", r1cs: "R1cs", z_dev):
        """Enqueue the four MSMs over z to the end (asynchronous); groth16_msms_dev on the same z_dev adds the H job and collects."""
        self._ck(self.lib.zk_groth16_msms_begin_dev(self.h, pk.h, r1cs.h, C.c_void_p(int(z_dev))))

    def groth16_msms_dev(self, pk: "ProvingKey", r1cs: "R1cs", z_dev, h_dev):
        g1 = np.zeros((4, 18), dtype=np.uint64)
        g2 = np.zeros(36, dtype=np.uint64)
        self._ck(self.lib.zk_groth16_msms_dev(self.h, pk.h, r1cs.h, C.c_void_p(int(z_dev)), C.c_void_p(int(h_dev)),
                                              _ptr(g1), _ptr(g2)))
        return g1, g2

    def create_proof_dev(self, pk: "ProvingKey", r1cs: "R1cs", z_dev, r_mont4, s_mont4) -> bytes:
        """create_proof (src/groth16.rs:68): 192-byte compressed proof a||b||c."""
        r, s = _fr_struct(r_mont4), _fr_struct(s_mont4)
        out = np.zeros(192, dtype=np.uint8)
        self._ck(self.lib.zk_groth16_prove_dev(self.h, pk.h, r1cs.h, C.c_void_p(int(z_dev)), C.byref(r), C.byref(s), _ptr(out)))
        return out.tobytes()

    def create_proof_multi(self, others, pks, r1css, z_dev, r_mont4, s_mont4) -> bytes:
        """zk_groth16_prove_multi: this context plus `others` (one per further device), each with its own key and constraint
        system (pks[i], r1css[i] belong to [self] + others [i]); z_dev lives on this context's device."""
        ctxs = [self] + list(others)
        n = len(ctxs)
        assert len(pks) == n and len(r1css) == n
        ca = (C.c_void_p * n)(*[c.h for c in ctxs])
        pa = (C.c_void_p * n)(*[p.h for p in pks])
        ra = (C.c_void_p * n)(*[r.h for r in r1css])
        r, s = _fr_struct(r_mont4), _fr_struct(s_mont4)
        out = np.zeros(192, dtype=np.uint8)
        self._ck(self.lib.zk_groth16_prove_multi(ca, pa, ra, n, C.c_void_p(int(z_dev)), C.byref(r), C.byref(s), _ptr(out)))
        return out.tobytes()

    def multi_plan(self, pk: "ProvingKey", r1cs: "R1cs", n_ctx: int):
        """[(ctx, job, lo, n)]: how zk_groth16_prove_multi deals this proof over n_ctx contexts."""
        buf = C.create_string_buffer(1 << 16)
        k = C.c_size_t(0)
        self._ck(self.lib.zk_groth16_multi_plan(pk.h, r1cs.h, n_ctx, buf, len(buf), C.byref(k)))
        return [tuple(int(x) for x in line.split()) for line in buf.raw[:k.value].decode().splitlines()]

    def create_proof(self, pk: "ProvingKey", r1cs: "R1cs", z_mont: np.ndarray, r_mont4, s_mont4) -> bytes:
        z = np.ascontiguousarray(z_mont, dtype=np.uint64)
        assert z.shape[0] == r1cs.num_instance + r1cs.num_witness
        r, s = _fr_struct(r_mont4), _fr_struct(s_mont4)
        out = np.zeros(192, dtype=np.uint8)
        self._ck(self.lib.zk_groth16_prove(self.h, pk.h, r1cs.h, _ptr(z), C.byref(r), C.byref(s), _ptr(out)))
        return out.tobytes()


class Bases:
    def __init__(self, ctx: Context, h, group: int, owned: bool = True):
        self.ctx, self.h, self.group, self.owned = ctx, h, group, owned

    def __len__(self):
        return self.ctx.lib.zk_bases_len(self.h)

    def precompute(self, layout: int = 0):
        """Store the window multiples of this resident table (13x memory at 2^20, one bucket set per MSM).  layout: 0 = chosen by
        the memory budget, 1 = packed, 2 = one point per 128-byte line, 3 = limbs with both signs (zk_bases_precompute_as)."""
        if layout:
            self.ctx._ck(self.ctx.lib.zk_bases_precompute_as(self.ctx.h, self.h, layout))
        else:
            self.ctx._ck(self.ctx.lib.zk_bases_precompute(self.ctx.h, self.h))

    def precompute_note(self) -> str:
        """The layout of the window multiples, or why they were skipped (zk_bases_precompute_note)."""
        return (self.ctx.lib.zk_bases_precompute_note(self.h) or b"").decode()

    def download(self, offset: int = 0, n: int = None) -> np.ndarray:
        n = len(self) - offset if n is None else n
        out = np.zeros((n, 12 if self.group == 1 else 24), dtype=np.uint64)
        fn = self.ctx.lib.zk_bases_download_g1 if self.group == 1 else self.ctx.lib.zk_bases_download_g2
        self.ctx._ck(fn(self.ctx.h, self.h, offset, n, _ptr(out)))
        return out

    def serialize(self, compressed: bool = True, offset: int = 0, n: int = None) -> bytes:
        """The points as GroupAffine::serialize / serialize_uncompressed write them, back to back."""
        n = len(self) - offset if n is None else n
        per = self.ctx.lib.zk_point_serialized_size(self.group, int(compressed))
        out = np.zeros(max(n * per, 1), dtype=np.uint8)
        self.ctx._ck(self.ctx.lib.zk_bases_serialize(self.ctx.h, self.h, offset, n, int(compressed), _ptr(out)))
        return out[: n * per].tobytes()

    def free(self):
        if self.h and self.owned:
            self.ctx.lib.zk_bases_free(self.ctx.h, self.h)
        self.h = None


class R1cs:
    def __init__(self, ctx: Context, h, nc: int, ni: int, nw: int):
        self.ctx, self.h = ctx, h
        self.num_constraints, self.num_instance, self.num_witness = nc, ni, nw

    @property
    def domain_log(self) -> int:
        return self.ctx.lib.zk_r1cs_domain_log(self.h)

    def free(self):
        if self.h:
            self.ctx.lib.zk_r1cs_free(self.ctx.h, self.h)
            self.h = None


class ProvingKey:
    QUERIES = {"a_query": 0, "b_g1_query": 1, "b_g2_query": 2, "h_query": 3, "l_query": 4, "gamma_abc_g1": 5}

    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h

    def query_len(self, name: str) -> int:
        return self.ctx.lib.zk_pk_query_len(self.h, self.QUERIES[name])

    def download(self, name: str, offset: int = 0, n: int = None) -> np.ndarray:
        which = self.QUERIES[name]
        n = self.query_len(name) - offset if n is None else n
        if which == 2:
            out = np.zeros((n, 24), dtype=np.uint64)
            self.ctx._ck(self.ctx.lib.zk_pk_download_g2(self.ctx.h, self.h, which, offset, n, _ptr(out)))
        else:
            out = np.zeros((n, 12), dtype=np.uint64)
            self.ctx._ck(self.ctx.lib.zk_pk_download_g1(self.ctx.h, self.h, which, offset, n, _ptr(out)))
        return out

    def query_bases(self, name: str) -> "Bases":
        """Borrowed handle to one query table (valid while the key lives)."""
        which = self.QUERIES[name]
        h = self.ctx.lib.zk_pk_query_bases(self.h, which)
        return Bases(self.ctx, C.c_void_p(h), 2 if which == 2 else 1, owned=False)

    def vk_g1(self, which: int) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        self.ctx._ck(self.ctx.lib.zk_pk_vk_g1(self.h, which, _ptr(out)))
        return out

    def vk_g2(self, which: int) -> np.ndarray:
        out = np.zeros(24, dtype=np.uint64)
        self.ctx._ck(self.ctx.lib.zk_pk_vk_g2(self.h, which, _ptr(out)))
        return out

    def free(self):
        if self.h:
            self.ctx.lib.zk_pk_free(self.ctx.h, self.h)
            self.h = None
