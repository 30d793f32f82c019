"""Marlin on the GPU, set-up side and entry point: AHPForR1CS::index (arkworks/marlin/src/ahp/{indexer.rs:121-208,
constraint_systems.rs:152-264}) -- the index polynomials, their evaluation tables and the integer index arrays, resident on the
device -- KZG10::setup with explicit toxic waste, Marlin::index (lib.rs:101-146), and Marlin::prove as ONE C-ABI call
(prove_native -> zk_marlin_prove, csrc/marlin_prove.hip).

Every polynomial lives in device memory (coefficients, low degree first, the reference's Fr layout).  The round-by-round Python
sequence of the same prover (prover_init / prover_*_round / commit / batch_open / prove) is test infrastructure -- the second
implementation the one-call provers are compared with -- and lives in tests/pyseq/marlin_seq.py.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from . import convert as cv
from .api import Context, DevBuf

R_MOD = cv.R_MOD
SENTINEL = np.uint32(0xFFFFFFFF)


def _log2(n: int) -> int:
    return max(n - 1, 0).bit_length()


def _next_pow2(n: int) -> int:
    return 1 << _log2(n)


class DevPoly:
    """n coefficients in a device buffer."""

    def __init__(self, buf: DevBuf, n: int, offset: int = 0):
        self.buf, self.n, self.offset = buf, n, offset

    @property
    def ptr(self) -> int:
        return self.buf.ptr + 32 * self.offset

    def slice(self, start: int, n: int) -> "DevPoly":
        return DevPoly(self.buf, n, self.offset + start)


class HostField:
    """Scalar Fr arithmetic through the library's host entry points; values are canonical Python ints."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    @staticmethod
    def m(v: int):
        return cv.fr_to_mont([v % R_MOD])[0]

    @staticmethod
    def i(limbs4) -> int:
        return cv.fr_from_mont(np.asarray(limbs4).reshape(1, 4))[0]

    def mul(self, a, b): return self.i(self.ctx.fr_op("mul", self.m(a), self.m(b)))
    def add(self, a, b): return self.i(self.ctx.fr_op("add", self.m(a), self.m(b)))
    def sub(self, a, b): return self.i(self.ctx.fr_op("sub", self.m(a), self.m(b)))
    def pow(self, a, e): return self.i(self.ctx.fr_pow(self.m(a), e))
    def inv(self, a): return self.i(self.ctx.fr_inverse(self.m(a)))


# arkworks/curves/bls12_377/src/fields/fr.rs:34-41 (TWO_ADIC_ROOT_OF_UNITY, Montgomery limbs) and :28 (TWO_ADICITY)
_FR_TWO_ADICITY = 47
_FR_TWO_ADIC_ROOT_MONT = np.array([12646347781564978760, 6783048705277173164, 268534165941069093, 1121515446318641358], dtype=np.uint64)


def reindex_by_subdomain(size_self: int, size_other: int, index: np.ndarray) -> np.ndarray:
    """EvaluationDomain::reindex_by_subdomain (poly/src/domain/mod.rs:195-217), vectorised."""
    index = np.asarray(index, dtype=np.int64)
    period = size_self // size_other
    i = index - size_other
    x = max(period - 1, 1)
    return np.where(index < size_other, index * period, i + i // x + 1)


class Csr:
    """A constraint matrix as (row_ptr, col, coeff) with coeff in the reference's Montgomery limbs."""

    def __init__(self, row_ptr, col, coeff):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.coeff = np.ascontiguousarray(coeff, dtype=np.uint64).reshape(-1, 4)

    @staticmethod
    def from_rows(rows) -> "Csr":
        rp, col, co = [0], [], []
        for row in rows:
            for c, i in row:
                col.append(i)
                co.append(c)
            rp.append(len(col))
        return Csr(rp, col, cv.fr_to_mont(co) if co else np.zeros((0, 4), dtype=np.uint64))

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[-1])

    def row_of_entry(self) -> np.ndarray:
        return np.repeat(np.arange(len(self.row_ptr) - 1, dtype=np.int64), np.diff(self.row_ptr.astype(np.int64)))

    def sorted_by_column(self) -> "Csr":
        """Rows in ascending column order (constraint_systems.rs:188-190)."""
        rows = self.row_of_entry()
        order = np.lexsort((self.col, rows))
        return Csr(self.row_ptr, self.col[order], self.coeff[order])

    def tuple(self):
        return (self.row_ptr, self.col, self.coeff)


def balance_matrices(a: Csr, b: Csr):
    """constraint_systems.rs:23-40: walk the rows, moving the denser matrix's row to the other one."""
    la, lb = np.diff(a.row_ptr.astype(np.int64)), np.diff(b.row_ptr.astype(np.int64))
    a_density, b_density = int(la.sum()), int(lb.sum())
    a_is_denser = a_density == max(a_density, b_density)
    swap = np.zeros(len(la), dtype=bool)
    for k in range(len(la)):
        if a_is_denser:
            swap[k] = True
            a_density += int(lb[k]) - int(la[k])
            b_density += int(la[k]) - int(lb[k])
            a_is_denser = a_density == max(a_density, b_density)

    def pick(first: Csr, second: Csr, take_second):
        ra, rb = first.row_of_entry(), second.row_of_entry()
        keep_a, keep_b = ~take_second[ra], take_second[rb]
        rows = np.concatenate([ra[keep_a], rb[keep_b]])
        col = np.concatenate([first.col[keep_a], second.col[keep_b]])
        coeff = np.concatenate([first.coeff[keep_a], second.coeff[keep_b]])
        order = np.argsort(rows, kind="stable")
        rp = np.zeros(len(take_second) + 1, dtype=np.int64)
        np.add.at(rp, rows + 1, 1)
        return Csr(np.cumsum(rp), col[order], coeff[order])
    return pick(a, b, swap), pick(b, a, swap)


class Domain:
    """Radix-2 evaluation domain handle: size, generator, and device helpers."""

    def __init__(self, ctx: Context, num_coeffs: int):
        self.ctx = ctx
        self.size = _next_pow2(num_coeffs)
        self.log = _log2(self.size)
        self.F = HostField(ctx)
        self.group_gen4 = ctx.fr_pow(_FR_TWO_ADIC_ROOT_MONT, 1 << (_FR_TWO_ADICITY - self.log))
        self._elems = None

    def elements(self) -> DevBuf:
        if self._elems is None:
            self._elems = self.ctx.alloc(self.size * 32)
            self.ctx.fr_powers_dev(self.group_gen4, HostField.m(1), self.size, self._elems.ptr)
        return self._elems

    def evaluate_vanishing_polynomial(self, tau: int) -> int:
        return self.F.sub(self.F.pow(tau, self.size), 1)

    def fft(self, ctx, poly: DevPoly) -> DevBuf:
        """evaluate_over_domain: zero-pad the coefficients to the domain size, forward transform."""
        assert poly.n <= self.size
        out = ctx.alloc(self.size * 32)
        if poly.n < self.size:
            ctx.dev_zero(out.ptr + 32 * poly.n, (self.size - poly.n) * 32)
        ctx.memcpy_d2d(out.ptr, poly.ptr, poly.n * 32)
        ctx.ntt_dev(out.ptr, self.log, False, False)
        return out

    def ifft_in_place(self, ctx, evals: DevBuf) -> DevPoly:
        ctx.ntt_dev(evals.ptr, self.log, True, False)
        return DevPoly(evals, self.size)


class MatrixArithmetization:
    """constraint_systems.rs:126-264 for one matrix, on the device."""

    def __init__(self, ctx: Context, m: Csr, dom_k: Domain, dom_h: Domain, dom_x: Domain, dom_b: Domain):
        m = m.sorted_by_column()
        nnz, K, n = m.nnz, dom_k.size, dom_h.size
        rows = m.row_of_entry()
        cols = reindex_by_subdomain(n, dom_x.size, m.col)
        pad = K - nnz
        idx_row = np.concatenate([cols, np.zeros(pad, dtype=np.int64)]).astype(np.uint32)       # transposed: row <- column
        idx_col = np.concatenate([rows, np.zeros(pad, dtype=np.int64)]).astype(np.uint32)
        # u_H(x, x) at x = w^j is |H| w^-j (mod.rs:362-369); its inverse is w^j / |H|
        idx_eq = np.concatenate([cols, np.full(pad, int(SENTINEL), dtype=np.int64)]).astype(np.uint32)
        elems = dom_h.elements()
        d_row, d_col, d_val, d_rc = (ctx.alloc(K * 32) for _ in range(4))
        i_row, i_col, i_eq = ctx.upload(idx_row), ctx.upload(idx_col), ctx.upload(idx_eq)
        ctx.fr_gather_dev(elems.ptr, i_row.ptr, K, d_row.ptr)
        ctx.fr_gather_dev(elems.ptr, i_col.ptr, K, d_col.ptr)
        ctx.fr_gather_dev(elems.ptr, i_eq.ptr, K, d_val.ptr)
        n_inv = dom_h.F.inv(n)
        ctx.fr_vec_scale_dev(d_val.ptr, HostField.m(n_inv), d_val.ptr, K)
        coeff = np.zeros((K, 4), dtype=np.uint64)
        coeff[:nnz] = m.coeff
        d_coeff = ctx.upload(coeff)
        ctx.fr_vec_op_dev(_lib.OP_MUL, d_val.ptr, d_coeff.ptr, d_val.ptr, K)
        ctx.fr_vec_op_dev(_lib.OP_MUL, d_row.ptr, d_col.ptr, d_rc.ptr, K)
        self.evals_on_K = {"row": d_row, "col": d_col, "val": d_val}
        polys = {}
        for name, ev in (("row", d_row), ("col", d_col), ("val", d_val), ("row_col", d_rc)):
            c = ctx.alloc(K * 32)
            ctx.memcpy_d2d(c.ptr, ev.ptr, K * 32)
            polys[name] = dom_k.ifft_in_place(ctx, c)
        self.row, self.col, self.val, self.row_col = polys["row"], polys["col"], polys["val"], polys["row_col"]
        self.evals_on_B = {name: dom_b.fft(ctx, p) for name, p in polys.items()}
        ctx.sync()      # the index arrays above may be released now


class Index:
    """AHPForR1CS::index on an already padded, square constraint system (indexer.rs:121-208)."""

    def __init__(self, ctx: Context, num_instance: int, num_witness: int, a: Csr, b: Csr, c: Csr):
        nc = len(a.row_ptr) - 1
        if num_instance + num_witness != nc:
            raise ValueError("NonSquareMatrix")
        if num_instance & (num_instance - 1):
            raise ValueError("InvalidPublicInputLength")
        self.ctx = ctx
        self.num_constraints = self.num_variables = nc
        self.num_instance, self.num_witness = num_instance, num_witness
        self.num_non_zero = max(a.nnz, b.nnz, c.nnz)
        a, b = balance_matrices(a, b)
        self.a, self.b, self.c = a, b, c
        self.dom_h, self.dom_k = Domain(ctx, nc), Domain(ctx, self.num_non_zero)
        self.dom_x, self.dom_b = Domain(ctx, num_instance), Domain(ctx, 3 * _next_pow2(self.num_non_zero) - 3)
        self.arith = {n: MatrixArithmetization(ctx, m, self.dom_k, self.dom_h, self.dom_x, self.dom_b)
                      for n, m in (("a", a), ("b", b), ("c", c))}
        self.r1cs = ctx.r1cs_upload(num_instance, num_witness, a.tuple(), b.tuple(), c.tuple())
        # transposed matrices with rows re-indexed into H, for calculate_t (prover.rs:406-423)
        H = self.dom_h.size
        tr = []
        for m in (a, b, c):
            rows_t = reindex_by_subdomain(H, self.dom_x.size, m.col)
            order = np.argsort(rows_t, kind="stable")
            rp = np.zeros(H + 1, dtype=np.int64)
            np.add.at(rp, rows_t + 1, 1)
            tr.append((np.cumsum(rp).astype(np.uint32), m.row_of_entry()[order].astype(np.uint32), m.coeff[order]))
        self.r1cs_t = ctx.r1cs_upload(1, H - 1, *tr)

    def w_evals_index(self):
        """Index maps of prover.rs:343-353: position k of H takes witness k - k/ratio - 1 (zero beyond the witness) minus
        x_evals[k], and is zero on the sub-domain X.  Built once per index."""
        if getattr(self, "_w_idx", None) is None:
            n, ratio = self.dom_h.size, self.dom_h.size // self.dom_x.size
            k = np.arange(n, dtype=np.int64)
            wi = k - k // ratio - 1
            on_x = (k % ratio) == 0
            idx_w = np.where(on_x | (wi >= self.num_witness), int(SENTINEL), wi + self.num_instance).astype(np.uint32)
            idx_x = np.where(on_x, int(SENTINEL), k).astype(np.uint32)
            self._w_idx = (self.ctx.upload(idx_w), self.ctx.upload(idx_x))
        return self._w_idx

    def polynomials(self):
        out = {}
        for m in "abc":
            ar = self.arith[m]
            out[m + "_row"], out[m + "_col"], out[m + "_val"], out[m + "_row_col"] = ar.row, ar.col, ar.val, ar.row_col
        return out


def commit(ctx: Context, powers_g, polys: dict) -> dict:
    """PC::commit without hiding (kzg10/mod.rs:142-205): one G1 MSM per polynomial."""
    labels = list(polys)
    outs = ctx.msm_batch_dev([(powers_g, 0, polys[l].ptr, polys[l].n) for l in labels])
    return dict(zip(labels, outs))


# degree bounds and hiding bounds of the prover's oracles as the reference labels them
# (LabeledPolynomial::new(label, poly, degree_bound, hiding_bound), prover.rs:383-387,548-552,708-711)
def oracle_bounds(index: "Index") -> dict:
    H, K = index.dom_h.size, index.dom_k.size
    return {"w": (None, 1), "z_a": (None, 1), "z_b": (None, 1), "mask_poly": (None, None), "t": (None, None),
            "g_1": (H - 2, 1), "h_1": (None, None), "g_2": (K - 2, None), "h_2": (None, None)}


def mul_chain_system(ctx: Context, n: int):
    """The SURVEY 8(d) mul-chain R1CS (w_i w_{i+1} = w_{i+2}, public input = the last product) already in Marlin's
    padded square form: 2 instance variables (a power of two), n + 1 witnesses, n constraints plus 3 empty rows."""
    def idx(j):
        return np.where(j <= n, 2 + j, 1)
    i = np.arange(n, dtype=np.int64)
    rp = np.concatenate([np.arange(n + 1, dtype=np.int64), np.full(3, n, dtype=np.int64)])
    ones = np.tile(HostField.m(1), (n, 1))
    mk = lambda col: Csr(rp, col, ones)
    return 2, n + 1, mk(idx(i)), mk(idx(i + 1)), mk(idx(i + 2))


def download_poly(ctx: Context, p: DevPoly) -> list:
    return cv.fr_from_mont(ctx.download(p.ptr, (p.n, 4)))


# ------------------------------------------------------------------------------------------------------------------------------
# Marlin as a proof: Marlin::{index, prove} (arkworks/marlin/src/lib.rs:100-319) over MarlinKZG10
# ------------------------------------------------------------------------------------------------------------------------------
# The Fiat-Shamir generator (FiatShamirRng<Blake2s>, marlin/src/rng.rs) is the library's zk_rng (csrc/fsrng.hpp); what is
# absorbed are the bytes the reference's to_bytes! writes: items back to back without length prefixes (ff/src/bytes.rs), a G1
# point as x | y | infinity with canonical little-endian coordinates (ec/.../short_weierstrass_jacobian.rs:315-322), a
# marlin_pc commitment as comm | shifted_exists | shifted_comm-or-zero (marlin_pc/data_structures.rs:252-263), IndexInfo as three
# u64 (ahp/indexer.rs:44-50), the empty prover messages as nothing (ahp/prover.rs:76-83).

PROTOCOL_NAME = b"MARLIN-2019"                                       # lib.rs:76
INDEX_LABELS = [m + s for m in "abc" for s in ("_row", "_col", "_val", "_row_col")]          # ahp/mod.rs:33-40
ROUND_LABELS = [["w", "z_a", "z_b", "mask_poly"], ["t", "g_1", "h_1"], ["g_2", "h_2"]]       # ahp/mod.rs:43-49
QUERY_SET = {"beta": ["g_1", "outer_sumcheck", "t", "z_b"],                                   # ahp/verifier.rs:103-170, BTree order
             "gamma": ["a_denom", "b_denom", "c_denom", "g_2", "inner_sumcheck"]}
EVAL_LABELS = ["a_denom", "b_denom", "c_denom", "g_1", "g_2", "t", "z_b"]                     # lib.rs:279-294: sorted, zero LCs left out


def _fr_bytes(v: int) -> bytes:
    return (v % R_MOD).to_bytes(32, "little")


def _g1_to_bytes(pt) -> bytes:
    if pt is None:                                                   # GroupAffine::zero() = (0, 1, infinity)
        return (0).to_bytes(48, "little") + (1).to_bytes(48, "little") + b"\x01"
    return pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little") + b"\x00"


class PcCommitment:
    """marlin_pc::Commitment: comm and, for a degree-bounded oracle, shifted_comm (Jacobian arrays from the device)."""

    def __init__(self, comm, shifted=None):
        self.comm, self.shifted = comm, shifted
        self.comm_aff = cv.g1_projective_to_affine(comm)
        self.shifted_aff = cv.g1_projective_to_affine(shifted) if shifted is not None else None

    def to_bytes(self) -> bytes:                                     # ToBytes (the transcript)
        return _g1_to_bytes(self.comm_aff) + (b"\x01" if self.shifted is not None else b"\x00") + \
            _g1_to_bytes(self.shifted_aff if self.shifted is not None else None)


def ahp_max_degree(index: "Index") -> int:
    """AHPForR1CS::max_degree (ahp/mod.rs:75-97), zk_bound = 1."""
    h, k = index.dom_h.size, index.dom_k.size
    return max(2 * h + 1 - 2, 3 * h + 2 - 3, h, 3 * k - 3)


class UniversalSrs:
    """KZG10::setup (poly-commit/src/kzg10/mod.rs:44-135) with explicit toxic waste: powers_of_g[i] = beta^i g for i <= max_degree
    (a resident table with window multiples), powers_of_gamma_g[i] = beta^i gamma_g for the three indices a hiding bound of 1
    uses (marlin_pc/mod.rs:98-102).  g = g_k G1, gamma_g = gamma_g_k G1; the G2 side (h, beta h) belongs to the verifier."""

    def __init__(self, ctx: Context, max_degree: int, beta: int, g_k: int = 1, gamma_g_k: int = 7):
        self.ctx, self.max_degree, self.beta = ctx, max_degree, beta % R_MOD
        m = HostField.m
        pw = ctx.alloc((max_degree + 1) * 32)
        ctx.fr_powers_dev(m(beta), m(1), max_degree + 1, pw.ptr)
        self.powers_g = ctx.fixed_base(pw.ptr, max_degree + 1, 1, m(g_k))
        self.powers_g.precompute()
        self.powers_gamma_g = ctx.fixed_base(pw.ptr, 3, 1, m(gamma_g_k))
        ctx.sync()
        pw.free()


class IndexKeys:
    """Marlin::index (lib.rs:101-146): trim the SRS to the index (degree bounds |H| - 2 and |K| - 2, hiding bound 1) and commit
    to the twelve index polynomials without hiding.  Holds the IndexProverKey's data and the IndexVerifierKey's bytes."""

    def __init__(self, index: "Index", srs: UniversalSrs):
        if srs.max_degree < ahp_max_degree(index):
            raise ValueError("IndexTooLarge")
        self.index, self.srs = index, srs
        self.bounds = {"g_1": index.dom_h.size - 2, "g_2": index.dom_k.size - 2}              # get_degree_bounds (ahp/mod.rs:100-110)
        self.hiding = {"w": 1, "z_a": 1, "z_b": 1, "g_1": 1}                                  # prover.rs:383-387,548-552
        polys = index.polynomials()
        comms = commit(index.ctx, srs.powers_g, {l: polys[l] for l in INDEX_LABELS})
        self.index_comms = {l: PcCommitment(comms[l]) for l in INDEX_LABELS}

    def ivk_bytes(self) -> bytes:                                    # IndexVerifierKey::write: index_info | index_comms
        ix = self.index
        head = ix.num_variables.to_bytes(8, "little") + ix.num_constraints.to_bytes(8, "little") + ix.num_non_zero.to_bytes(8, "little")
        return head + b"".join(self.index_comms[l].to_bytes() for l in INDEX_LABELS)


def native_index(keys: IndexKeys):
    """The index as the C ABI's zk_marlin_index (include/zkmpc_hip.h): (struct, objects that must outlive the call)."""
    index = keys.index
    d = _lib.MarlinIndex()
    d.num_constraints, d.num_variables = index.num_constraints, index.num_variables
    d.num_non_zero, d.num_instance = index.num_non_zero, index.num_instance
    d.r1cs, d.r1cs_t = index.r1cs.h, index.r1cs_t.h
    polys = index.polynomials()
    for i, l in enumerate(INDEX_LABELS):
        d.index_polys[i].ptr, d.index_polys[i].n = polys[l].ptr, polys[l].n
    for i, m in enumerate("abc"):
        ek, eb = index.arith[m].evals_on_K, index.arith[m].evals_on_B
        d.on_k[i].row, d.on_k[i].col, d.on_k[i].val, d.on_k[i].row_col = ek["row"].ptr, ek["col"].ptr, ek["val"].ptr, None
        d.on_b[i].row, d.on_b[i].col, d.on_b[i].val, d.on_b[i].row_col = eb["row"].ptr, eb["col"].ptr, eb["val"].ptr, eb["row_col"].ptr
    d_iw, d_ix = index.w_evals_index()
    d.w_idx, d.x_idx = d_iw.ptr, d_ix.ptr
    ivk = keys.ivk_bytes()
    d.ivk_bytes, d.ivk_len = ivk, len(ivk)
    return d, (polys, d_iw, d_ix, ivk)


def prove_native(keys: IndexKeys, assignment_dev: DevBuf, zk_rng, mask_on_device: bool = False) -> bytes:
    """Marlin::prove through the library's single entry point zk_marlin_prove (csrc/marlin_prove.hip): the sequence of
    tests/pyseq/marlin_seq.py::prove, on the host in C++.  Returns the CanonicalSerialize bytes of the proof."""
    import ctypes as C
    srs = keys.srs
    ctx = keys.index.ctx
    d, _keep = native_index(keys)
    cap = ctx.lib.zk_marlin_proof_max_size()
    out = (C.c_uint8 * cap)()
    n = C.c_size_t()
    ctx._ck(ctx.lib.zk_marlin_prove(ctx.h, C.byref(d), srs.powers_g.h, srs.powers_gamma_g.h, C.c_void_p(assignment_dev.ptr), zk_rng.h,
                                    int(mask_on_device), out, cap, C.byref(n)))
    return bytes(out[:n.value])
