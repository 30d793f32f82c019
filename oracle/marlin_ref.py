"""CPU restatement of the Marlin AHP for R1CS (index, the three prover rounds, the verifier's two sum-check
equations) over BLS12-377 Fr.  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product.

Follows /root/reference/arkworks/marlin/src/ahp/{constraint_systems.rs, indexer.rs, prover.rs, verifier.rs, mod.rs};
each function cites the lines it restates.  Polynomials are coefficient lists, low degree first, kept at their
nominal length (the reference truncates leading zeros; compare through `strip`).

Parity note: the reference holds no recorded Marlin proofs.  What pins this restatement is (i) the AHP's own
soundness equations -- outer_sumcheck(beta) = 0 and inner_sumcheck(gamma) = 0 exactly as the reference's verifier
builds them (mod.rs:134-290), with the debug assertions of the reference at the same places, (ii) the degree
bounds asserted in prover.rs:545-546,707, and (iii) rejection when the witness is wrong.
"""
from typing import List

import zkref as O

P = O.R_MOD


def strip(p):
    p = list(p)
    while p and p[-1] == 0:
        p.pop()
    return p


def next_pow2(n):
    return 1 if n <= 1 else 1 << (n - 1).bit_length()


def padd(a, b):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % P for i in range(n)]


def psub(a, b):
    n = max(len(a), len(b))
    return [((a[i] if i < len(a) else 0) - (b[i] if i < len(b) else 0)) % P for i in range(n)]


def pscale(a, k):
    return [x * k % P for x in a]


def pmul(a, b):
    """Exact product via one FFT (values are unique, so the method is immaterial)."""
    a, b = strip(a), strip(b)
    if not a or not b:
        return []
    n = len(a) + len(b) - 1
    d = O.Domain(n)
    ea, eb = d.fft(a + [0] * (d.size - len(a))), d.fft(b + [0] * (d.size - len(b)))
    return d.ifft([x * y % P for x, y in zip(ea, eb)])[:n]


def divide_by_vanishing(p, n):
    """DensePolynomial::divide_by_vanishing_poly (poly/src/polynomial/univariate/dense.rs:166-173) for X^n - 1."""
    p = list(p)
    if len(p) <= n:
        return [], p + [0] * (n - len(p))
    q = [0] * (len(p) - n)
    for k in range(len(p) - 1, n - 1, -1):
        c = p[k]
        q[k - n] = c
        p[k - n] = (p[k - n] + c) % P
    return q, p[:n]


def evaluate(p, x):
    return O.poly_evaluate(p, x)


def reindex_by_subdomain(size_self, size_other, index):
    """EvaluationDomain::reindex_by_subdomain (poly/src/domain/mod.rs:195-217)."""
    period = size_self // size_other
    if index < size_other:
        return index * period
    i = index - size_other
    x = period - 1
    return i + (i // x) + 1


def eval_unnormalized_bivariate_lagrange_poly(dom, x, y):
    """mod.rs:343-350."""
    if x != y:
        return (dom.evaluate_vanishing_polynomial(x) - dom.evaluate_vanishing_polynomial(y)) * pow((x - y) % P, -1, P) % P
    return dom.size * pow(x, dom.size - 1, P) % P


# ---- constraint_systems.rs ----

def pad_and_square(r1cs: O.R1CS, assignment: List[int]):
    """pad_input_for_indexer_and_prover (:74-88) then make_matrices_square (:90-113).
    Returns (R1CS, full assignment) with num_instance a power of two and num_variables == num_constraints."""
    ni, nw = r1cs.num_instance, r1cs.num_witness
    extra = next_pow2(ni) - ni

    def shift(rows):
        return [[(c, i if i < ni else i + extra) for c, i in row] for row in rows]
    a, b, c = shift(r1cs.a), shift(r1cs.b), shift(r1cs.c)
    inst = list(assignment[:ni]) + [0] * extra
    wit = list(assignment[ni:])
    ni += extra
    nv, nc = ni + nw, len(a)
    if nv > nc:
        for _ in range(nv - nc):        # dummy constraints 0 * 0 = 0
            a.append([]); b.append([]); c.append([])
    else:
        wit += [1] * (nc - nv)          # dummy unconstrained variables, value one
        nw += nc - nv
    return O.R1CS(ni, nw, a, b, c), inst + wit


def balance_matrices(a, b):
    """constraint_systems.rs:23-40."""
    a_density, b_density = sum(len(r) for r in a), sum(len(r) for r in b)
    max_density = max(a_density, b_density)
    a_is_denser = a_density == max_density
    for k in range(len(a)):
        if a_is_denser:
            ra, rb = len(a[k]), len(b[k])
            a[k], b[k] = b[k], a[k]
            a_density = a_density - ra + rb
            b_density = b_density - rb + ra
            max_density = max(a_density, b_density)
            a_is_denser = a_density == max_density


class MatrixArith:
    """MatrixArithmetization (constraint_systems.rs:126-150) of M^*."""

    def __init__(self, rows, dom_k, dom_h, dom_x, dom_b):
        elems = [dom_h.element(i) for i in range(dom_h.size)]
        # batch_eval_unnormalized_bivariate_lagrange_poly_with_same_inputs (mod.rs:362-369)
        eq = [e * dom_h.size % P for e in elems]
        eq[1:] = eq[1:][::-1]
        eq_of = dict(zip(elems, eq))
        row_vec, col_vec, val_vec, inv = [], [], [], []
        for r, row in enumerate(rows):
            for val, i in sorted(row, key=lambda t: t[1]):
                row_val = elems[r]
                col_val = elems[reindex_by_subdomain(dom_h.size, dom_x.size, i)]
                row_vec.append(col_val)          # transpose
                col_vec.append(row_val)
                val_vec.append(val % P)
                inv.append(eq_of[col_val])
        val_vec = [v * pow(u, -1, P) % P for v, u in zip(val_vec, inv)]
        pad = dom_k.size - len(val_vec)
        row_vec += [elems[0]] * pad
        col_vec += [elems[0]] * pad
        val_vec += [0] * pad
        row_col_vec = [x * y % P for x, y in zip(row_vec, col_vec)]
        self.evals_on_K = {"row": row_vec, "col": col_vec, "val": val_vec}
        self.row, self.col, self.val, self.row_col = (dom_k.ifft(v) for v in (row_vec, col_vec, val_vec, row_col_vec))
        ext = lambda p: dom_b.fft(p + [0] * (dom_b.size - len(p)))
        self.evals_on_B = {"row": ext(self.row), "col": ext(self.col), "val": ext(self.val), "row_col": ext(self.row_col)}


class Index:
    """AHPForR1CS::index (indexer.rs:121-208) on an already padded, square R1CS."""

    def __init__(self, r1cs: O.R1CS):
        assert r1cs.num_instance + r1cs.num_witness == r1cs.num_constraints, "NonSquareMatrix"
        assert r1cs.num_instance & (r1cs.num_instance - 1) == 0, "InvalidPublicInputLength"
        self.num_variables = self.num_constraints = r1cs.num_constraints
        self.num_instance = r1cs.num_instance
        self.num_non_zero = max(sum(len(r) for r in m) for m in (r1cs.a, r1cs.b, r1cs.c))
        self.a, self.b, self.c = [list(r) for r in r1cs.a], [list(r) for r in r1cs.b], [list(r) for r in r1cs.c]
        balance_matrices(self.a, self.b)
        self.dom_h = O.Domain(self.num_constraints)
        self.dom_k = O.Domain(self.num_non_zero)
        self.dom_x = O.Domain(self.num_instance)
        self.dom_b = O.Domain(3 * self.dom_k.size - 3)
        self.arith = {n: MatrixArith(m, self.dom_k, self.dom_h, self.dom_x, self.dom_b)
                      for n, m in (("a", self.a), ("b", self.b), ("c", self.c))}

    def polynomials(self):
        """Index::iter (indexer.rs:100-118): a_row, a_col, a_val, a_row_col, b_..., c_..."""
        out = {}
        for m in "abc":
            ar = self.arith[m]
            out[m + "_row"], out[m + "_col"], out[m + "_val"], out[m + "_row_col"] = ar.row, ar.col, ar.val, ar.row_col
        return out


# ---- prover.rs ----

class ProverState:
    pass


def prover_init(index: Index, full_assignment):
    """AHPForR1CS::prover_init (prover.rs:216-309): z_A = A z, z_B = B z with the balanced matrices."""
    st = ProverState()
    st.index = index
    st.x = list(full_assignment[:index.num_instance])
    st.w = list(full_assignment[index.num_instance:])
    assert len(st.x) + len(st.w) == index.num_variables, "InstanceDoesNotMatchIndex"
    z = st.x + st.w
    st.z_a = [O.evaluate_constraint(row, z) for row in index.a]
    st.z_b = [O.evaluate_constraint(row, z) for row in index.b]
    st.zk_bound = 1
    return st


def mask_poly_degree(index):
    return 3 * index.dom_h.size + 2 * 1 - 3


def prover_first_round(st, r_w, r_za, r_zb, mask_coeffs):
    """prover.rs:311-404.  The four rng draws are passed in, in the reference's order: F::rand for w, z_a, z_b, then
    DensePolynomial::rand(mask_poly_degree) = mask_poly_degree + 1 coefficients."""
    ix = st.index
    H, X = ix.dom_h, ix.dom_x
    n = H.size
    x_poly = X.ifft(st.x)
    x_evals = H.fft(x_poly + [0] * (n - len(x_poly)))
    ratio = n // X.size
    w_ext = st.w + [0] * (n - X.size - len(st.w))
    w_evals = [0 if k % ratio == 0 else (w_ext[k - k // ratio - 1] - x_evals[k]) % P for k in range(n)]
    blind = lambda p, r: psub(p + [r % P], [r % P])          # + r * (X^n - 1)
    w_poly = blind(H.ifft(w_evals), r_w)
    w_poly, rem = divide_by_vanishing(w_poly, X.size)
    assert not strip(rem)
    st.w_poly = w_poly
    st.z_a_poly = blind(H.ifft(st.z_a), r_za)
    st.z_b_poly = blind(H.ifft(st.z_b), r_zb)
    assert len(mask_coeffs) == mask_poly_degree(ix) + 1
    mask = [c % P for c in mask_coeffs]
    mask[0] = (mask[0] - divide_by_vanishing(mask, n)[1][0]) % P
    st.mask_poly = mask
    st.x_poly = x_poly
    return {"w": st.w_poly, "z_a": st.z_a_poly, "z_b": st.z_b_poly, "mask_poly": st.mask_poly}


def r_alpha_x_evals(dom, alpha):
    """batch_eval_unnormalized_bivariate_lagrange_poly_with_diff_inputs (mod.rs:352-360)."""
    v = dom.evaluate_vanishing_polynomial(alpha)
    return [v * pow((alpha - dom.element(i)) % P, -1, P) % P for i in range(dom.size)]


def calculate_t(index, etas, r_alpha):
    """prover.rs:406-423."""
    H, X = index.dom_h, index.dom_x
    t = [0] * H.size
    for mat, eta in zip((index.a, index.b, index.c), etas):
        for r, row in enumerate(mat):
            for coeff, c in row:
                j = reindex_by_subdomain(H.size, X.size, c)
                t[j] = (t[j] + eta * coeff % P * r_alpha[r]) % P
    return H.ifft(t)


def prover_second_round(st, alpha, eta_a, eta_b, eta_c):
    """prover.rs:438-565."""
    ix = st.index
    H, X = ix.dom_h, ix.dom_x
    z_c = pmul(st.z_a_poly, st.z_b_poly)
    summed = pscale(z_c, eta_c)
    for i in range(min(len(summed), len(st.z_a_poly), len(st.z_b_poly))):
        summed[i] = (summed[i] + eta_a * st.z_a_poly[i] + eta_b * st.z_b_poly[i]) % P
    ra = r_alpha_x_evals(H, alpha)
    r_alpha_poly = H.ifft(ra)
    t_poly = calculate_t(ix, (eta_a, eta_b, eta_c), ra)
    z_poly = psub([0] * X.size + st.w_poly, st.w_poly)        # mul_by_vanishing_poly(domain_x)
    for i, xv in enumerate(st.x_poly):
        z_poly[i] = (z_poly[i] + xv) % P
    assert len(strip(z_poly)) - 1 < H.size + st.zk_bound
    rhs = psub(pmul(r_alpha_poly, summed), pmul(t_poly, z_poly))
    q_1 = padd(st.mask_poly, rhs)
    h_1, x_g_1 = divide_by_vanishing(q_1, H.size)
    assert x_g_1[0] == 0, "sum over H is not zero"
    g_1 = x_g_1[1:]
    assert len(strip(g_1)) - 1 <= H.size - 2
    assert len(strip(h_1)) - 1 <= 2 * H.size + 2 * st.zk_bound - 2
    st.first_msg = (alpha, eta_a, eta_b, eta_c)
    st.t_poly, st.z_poly = t_poly, z_poly
    return {"t": t_poly, "g_1": g_1, "h_1": h_1}


def prover_third_round(st, beta):
    """prover.rs:583-716."""
    ix = st.index
    H, K, B = ix.dom_h, ix.dom_k, ix.dom_b
    alpha, eta_a, eta_b, eta_c = st.first_msg
    vv = H.evaluate_vanishing_polynomial(alpha) * H.evaluate_vanishing_polynomial(beta) % P
    A, Bm, C = ix.arith["a"], ix.arith["b"], ix.arith["c"]
    f_vals = []
    for i in range(K.size):
        t = 0
        for eta, M in ((eta_a, A), (eta_b, Bm), (eta_c, C)):
            den = (beta - M.evals_on_K["row"][i]) * (alpha - M.evals_on_K["col"][i]) % P
            t += eta * M.evals_on_K["val"][i] % P * (pow(den, -1, P) if den else 0)
        f_vals.append(vv * (t % P) % P)
    f = K.ifft(f_vals)
    g_2 = f[1:]
    den = {}
    for name, M in (("a", A), ("b", Bm), ("c", C)):
        e = M.evals_on_B
        den[name] = [(beta * alpha - r * alpha - beta * c + rc) % P for r, c, rc in zip(e["row"], e["col"], e["row_col"])]
    a_on_B = [vv * ((eta_a * A.evals_on_B["val"][i] % P * den["b"][i] % P * den["c"][i]
                     + eta_b * Bm.evals_on_B["val"][i] % P * den["a"][i] % P * den["c"][i]
                     + eta_c * C.evals_on_B["val"][i] % P * den["a"][i] % P * den["b"][i]) % P) % P for i in range(B.size)]
    b_on_B = [den["a"][i] * den["b"][i] % P * den["c"][i] % P for i in range(B.size)]
    a_poly, b_poly = B.ifft(a_on_B), B.ifft(b_on_B)
    h_2, rem = divide_by_vanishing(psub(a_poly, pmul(b_poly, f)), K.size)
    assert not strip(rem), "inner sumcheck does not divide"
    assert len(strip(g_2)) - 1 <= K.size - 2
    return {"g_2": g_2, "h_2": h_2}


# ---- verifier: mod.rs:134-290 ----

def sumcheck_equations(index: Index, public_input, polys, alpha, eta_a, eta_b, eta_c, beta, gamma):
    """Evaluate the verifier's outer_sumcheck at beta and inner_sumcheck at gamma from the polynomials themselves
    (the `Vec<LabeledPolynomial>` EvaluationsProvider, mod.rs:303-331).  public_input excludes the leading one.
    Both results must be zero for an honest prover.  `polys` is a dict label -> coefficients or a callable
    (label, point) -> evaluation; `index` needs dom_h, dom_k and num_instance only."""
    H, K = index.dom_h, index.dom_k
    x = [1] + list(public_input)
    x = x + [0] * (index.num_instance - len(x))       # the indexer's zero padding of the formatted input
    Xd = O.Domain(len(x))
    ev = polys if callable(polys) else (lambda label, pt: evaluate(polys[label], pt))
    r_alpha_at_beta = eval_unnormalized_bivariate_lagrange_poly(H, alpha, beta)
    v_H_alpha, v_H_beta = H.evaluate_vanishing_polynomial(alpha), H.evaluate_vanishing_polynomial(beta)
    v_X_beta = Xd.evaluate_vanishing_polynomial(beta)
    z_b_beta, t_beta, g_1_beta = ev("z_b", beta), ev("t", beta), ev("g_1", beta)
    x_beta = sum(l * xv for l, xv in zip(Xd.evaluate_all_lagrange_coefficients(beta), x)) % P
    outer = (ev("mask_poly", beta)
             + r_alpha_at_beta * (eta_a + eta_c * z_b_beta) % P * ev("z_a", beta)
             + r_alpha_at_beta * eta_b % P * z_b_beta
             - t_beta * v_X_beta % P * ev("w", beta)
             - t_beta * x_beta
             - v_H_beta * ev("h_1", beta)
             - beta * g_1_beta) % P
    ba = beta * alpha % P
    den = {m: (ba - alpha * ev(m + "_row", gamma) - beta * ev(m + "_col", gamma) + ev(m + "_row_col", gamma)) % P for m in "abc"}
    g_2_gamma = ev("g_2", gamma)
    v_K_gamma = K.evaluate_vanishing_polynomial(gamma)
    a_lc = (eta_a * den["b"] % P * den["c"] % P * ev("a_val", gamma)
            + eta_b * den["a"] % P * den["c"] % P * ev("b_val", gamma)
            + eta_c * den["b"] % P * den["a"] % P * ev("c_val", gamma)) % P * (v_H_alpha * v_H_beta % P) % P
    b_at_gamma = den["a"] * den["b"] % P * den["c"] % P
    b_expr = b_at_gamma * ((gamma * g_2_gamma + t_beta * pow(K.size, -1, P)) % P) % P
    inner = (a_lc - b_expr - v_K_gamma * ev("h_2", gamma)) % P
    return outer, inner


def ahp_prove(index, full_assignment, rng, challenges=None):
    """All three rounds with randomness and verifier messages from `rng` (an O.Prng); returns (polys, challenges)."""
    st = prover_init(index, full_assignment)
    r = [rng.fr() for _ in range(3)]
    mask = [rng.fr() for _ in range(mask_poly_degree(index) + 1)]
    polys = dict(index.polynomials())
    polys.update(prover_first_round(st, r[0], r[1], r[2], mask))
    ch = challenges or {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma")}
    polys.update(prover_second_round(st, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]))
    polys.update(prover_third_round(st, ch["beta"]))
    return polys, ch, st


class IndexInfo:
    """The sizes the verifier needs (indexer.rs:29-41), enough for sumcheck_equations."""

    def __init__(self, num_constraints, num_non_zero, num_instance):
        self.dom_h, self.dom_k, self.num_instance = O.Domain(num_constraints), O.Domain(num_non_zero), num_instance
