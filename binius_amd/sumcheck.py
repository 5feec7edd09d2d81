"""Host-side mirror of the reference's v3 bivariate sumcheck provers over a ComputeLayer.

  BivariateSumcheckProver   crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:27-254
  calculate_round_evals     crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:303-408
  round coeffs from evals   crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:410-424
  eq_ind_partial_eval       crates/compute/src/ops.rs:26-50
  MLE-check round evals     crates/core/src/protocols/sumcheck/v3/bivariate_mlecheck.rs:391-520

Only protocol bookkeeping lives here (which buffers, which challenge, three field additions per
round).  Every field multiplication over the hypercube happens in HIP kernels behind
`binius_amd.Context`.  The two scalar multiplications the host needs per round
(evaluate_univariate of the degree-2 round polynomial) go through `HostField` = bn_scalar_mul.
"""
from ._ffi import BN_ERR_INPUT_VALIDATION, BnError, HostField


def bivariate_product_expr(hal, i, j):
    """CompositionPoly::expression of IndexComposition<BivariateProduct, 2> {indices [i, j]}
    (crates/core/src/composition/product_composition.rs:30-32 remapped by index.rs:50-55)."""
    return hal.compile_expr([("var", i), ("var", j), ("mul", 0, 1)])


def round_eval_kernel(n_vars, batch_coeffs, multilins, compositions, eq_ind=None):
    """The kernel-spec closure and memory mappings of calculate_round_evals
    (v3/bivariate_product.rs:317-402; with eq_ind: v3/bivariate_mlecheck.rs:409-515)."""
    split_n_vars = n_vars - 1
    mem_maps = []
    for ml in multilins:
        lo, hi = ml.split_half()
        mem_maps += [("chunked", lo, 0), ("chunked", hi, 0), ("local", split_n_vars)]
    if eq_ind is not None:
        mem_maps.append(("chunked", eq_ind, 0))
    m = len(multilins)

    def kernel(local_exec, log_chunks, buffers):
        log_chunk_size = split_n_vars - log_chunks
        eq = buffers[-1] if eq_ind is not None else None
        acc_1 = local_exec.decl_value(0)
        eval_1s = [buffers[i * 3 + 1].to_ref() for i in range(m)]
        if eq is not None:
            eval_1s.append(eq.to_ref())
        for coeff, evaluator in zip(batch_coeffs, compositions):
            local_exec.sum_composition_evals(eval_1s, evaluator, coeff, acc_1)
        for i in range(m):
            local_exec.add(log_chunk_size, buffers[3 * i], buffers[3 * i + 1], buffers[3 * i + 2])
        acc_inf = local_exec.decl_value(0)
        eval_infs = [buffers[i * 3 + 2].to_ref() for i in range(m)]
        if eq is not None:
            eval_infs.append(eq.to_ref())
        for coeff, evaluator in zip(batch_coeffs, compositions):
            local_exec.sum_composition_evals(eval_infs, evaluator, coeff, acc_inf)
        return [acc_1, acc_inf]

    return kernel, mem_maps


def calculate_round_evals(hal, n_vars, batch_coeffs, multilins, compositions, eq_ind=None):
    """Returns [y_1, y_inf].  `compositions` are compiled exprs over the m (+1 with eq_ind) rows;
    `batch_coeffs[c]` = batch_coeff ** c (powers(), computed by the caller)."""
    kernel, mem_maps = round_eval_kernel(n_vars, batch_coeffs, multilins, compositions, eq_ind)
    return hal.accumulate_kernels(kernel, mem_maps)


def calculate_round_coeffs_from_evals(batched_sum, evals):
    y_1, y_inf = evals
    y_0 = batched_sum ^ y_1
    c_0 = y_0
    c_2 = y_inf
    c_1 = y_1 ^ c_0 ^ c_2
    return [c_0, c_1, c_2]


def eq_ind_partial_eval(hal, dev_alloc, point):
    n_vars = len(point)
    out = dev_alloc.alloc(1 << n_vars)
    hal.fill(out.slice(0, 1), 1)
    hal.tensor_expand(0, point, out)
    return out


class BivariateSumcheckProver:
    """State machine execute -> fold -> ... -> finish, High-to-Low binding order."""

    def __init__(self, hal, dev_alloc, n_vars, multilins, composition_indices, sums, field=HostField):
        """field: mul(a,b) for the O(1)-per-round host scalars (binius_amd.HostField, i.e.
        bn_scalar_mul); it never touches hypercube-sized data."""
        for ml in multilins:
            if ml.len != 1 << n_vars:
                raise BnError(BN_ERR_INPUT_VALIDATION, "NumberOfVariablesMismatch")
        self.hal = hal
        self.dev_alloc = dev_alloc
        self.n_vars_initial = n_vars
        self.n_vars_remaining = n_vars
        self.multilins = [("pre", ml) for ml in multilins]
        self.compositions = [bivariate_product_expr(hal, i, j) for (i, j) in composition_indices]
        self.field = field
        self.state = ("initial_sums", list(sums))

    def _evaluate_univariate(self, coeffs, x):
        e = 0
        for c in reversed(coeffs):
            e = self.field.mul(e, x) ^ c
        return e

    def execute(self, batch_coeff):
        coeffs = []
        p = 1
        for _ in self.compositions:
            coeffs.append(p)
            p = self.field.mul(p, batch_coeff)
        evals = calculate_round_evals(
            self.hal, self.n_vars_remaining, coeffs, [ml for _, ml in self.multilins], self.compositions
        )
        kind, val = self.state
        if kind == "coeffs":
            raise RuntimeError("ExpectedFold")
        batched_sum = self._evaluate_univariate(val, batch_coeff) if kind == "initial_sums" else val
        rc = calculate_round_coeffs_from_evals(batched_sum, evals)
        self.state = ("coeffs", rc)
        return rc

    def fold(self, challenge):
        if self.n_vars_remaining == 0:
            raise RuntimeError("ExpectedFinish")
        kind, val = self.state
        if kind != "coeffs":
            raise RuntimeError("ExpectedExecution")
        self.state = ("batched_sum", self._evaluate_univariate(val, challenge))
        # exec.map over the multilinears (v3/bivariate_product.rs:217-228): one fold batch
        e0s, e1s = [], []
        for kind, evals in self.multilins:
            evals_0, evals_1 = evals.split_half()
            if kind == "pre":
                folded = self.dev_alloc.alloc(1 << (self.n_vars_remaining - 1))
                self.hal.copy_d2d(evals_0, folded)
                evals_0 = folded
            e0s.append(evals_0)
            e1s.append(evals_1)
        self.hal.extrapolate_line_batch(e0s, e1s, challenge)
        new = [("post", e) for e in e0s]
        self.multilins = new
        self.n_vars_remaining -= 1

    def finish(self):
        if self.state[0] == "coeffs":
            raise RuntimeError("ExpectedFold")
        if self.n_vars_remaining != 0:
            raise RuntimeError("ExpectedExecution")
        out = []
        for _, ml in self.multilins:
            h = self.hal.copy_d2h(ml)
            out.append(int(h[0, 0]) | (int(h[0, 1]) << 64))
        return out


def bivariate_product_eq_expr(hal, i, j, m):
    """IndexComposition<BivariateProduct, 2>::expression() * var(m) -- the eq indicator is the last
    composition variable (v3/bivariate_mlecheck.rs:399-407)."""
    return hal.compile_expr([("var", i), ("var", j), ("mul", 0, 1), ("var", m), ("mul", 2, 3)])


class BivariateMLEcheckProver:
    """BivariateMLEcheckProver (v3/bivariate_mlecheck.rs:27-372): eq-indicator sumcheck for bivariate
    products, High-to-Low.  eq_ind_partial_evals is the tensor expansion of
    eq_ind_challenges[0 .. n_vars-1) (2^(n_vars-1) elements)."""

    def __init__(self, hal, dev_alloc, n_vars, multilins, composition_indices, sums, eq_ind_partial_evals, eq_ind_challenges,
                 field=HostField):
        for ml in multilins:
            if ml.len != 1 << n_vars:
                raise BnError(BN_ERR_INPUT_VALIDATION, "NumberOfVariablesMismatch")
        if eq_ind_partial_evals.len != 1 << max(n_vars - 1, 0):
            raise BnError(BN_ERR_INPUT_VALIDATION, "IncorrectEqIndPartialEvalsSize")
        self.hal, self.dev_alloc, self.field = hal, dev_alloc, field
        self.n_vars_initial = self.n_vars_remaining = n_vars
        self.multilins = [("pre", ml) for ml in multilins]
        m = len(multilins)
        self.compositions = [bivariate_product_eq_expr(hal, i, j, m) for (i, j) in composition_indices]
        self.state = ("initial_sums", list(sums))
        self.eq_ind_prefix_eval = 1
        self.eq_ind = ("pre", eq_ind_partial_evals)
        self.eq_ind_challenges = list(eq_ind_challenges)

    def _evaluate_univariate(self, coeffs, x):
        e = 0
        for c in reversed(coeffs):
            e = self.field.mul(e, x) ^ c
        return e

    def execute(self, batch_coeff):
        f = self.field
        coeffs, p = [], 1
        for _ in self.compositions:
            coeffs.append(p)
            p = f.mul(p, batch_coeff)
        y_1, y_inf = calculate_round_evals(
            self.hal, self.n_vars_remaining, coeffs, [ml for _, ml in self.multilins], self.compositions, eq_ind=self.eq_ind[1]
        )
        kind, val = self.state
        if kind == "coeffs":
            raise RuntimeError("ExpectedFold")
        batched_sum = self._evaluate_univariate(val, batch_coeff) if kind == "initial_sums" else val
        alpha = self.eq_ind_challenges[self.n_vars_remaining - 1]
        # calculate_round_coeffs_from_evals (:375-389)
        y_0 = f.mul(batched_sum ^ f.mul(y_1, alpha), f.invert(1 ^ alpha))
        prime = [y_0, y_1 ^ y_0 ^ y_inf, y_inf]
        self.state = ("coeffs", prime)
        # v(X) = v'(X) * ((1 - alpha) + (2 alpha - 1) X) * prefix; 2 alpha = 0 in characteristic 2 (:303-313)
        k0 = 1 ^ alpha
        out = []
        for d in range(4):
            v = 0
            if d < 3:
                v ^= f.mul(prime[d], k0)
            if d >= 1:
                v ^= prime[d - 1]
            out.append(f.mul(v, self.eq_ind_prefix_eval))
        return out

    def fold(self, challenge):
        if self.n_vars_remaining == 0:
            raise RuntimeError("ExpectedFinish")
        kind, val = self.state
        if kind != "coeffs":
            raise RuntimeError("ExpectedExecution")
        self.state = ("batched_sum", self._evaluate_univariate(val, challenge))
        # eq(alpha, z) = alpha + z + 1 in characteristic 2 (field/src/util.rs:74-81)
        alpha = self.eq_ind_challenges[self.n_vars_remaining - 1]
        self.eq_ind_prefix_eval = self.field.mul(self.eq_ind_prefix_eval, alpha ^ challenge ^ 1)
        # fold_multilinears (:145-193): one map scope = one fold batch
        e0s, e1s = [], []
        for kind, evals in self.multilins:
            evals_0, evals_1 = evals.split_half()
            if kind == "pre":
                folded = self.dev_alloc.alloc(1 << (self.n_vars_remaining - 1))
                self.hal.copy_d2d(evals_0, folded)
                evals_0 = folded
            e0s.append(evals_0)
            e1s.append(evals_1)
        self.hal.extrapolate_line_batch(e0s, e1s, challenge)
        self.multilins = [("post", e) for e in e0s]
        if self.n_vars_remaining - 1 != 0:
            # fold_eq_ind (:195-254): map_kernels { add_assign(evals_1 -> evals_0) }
            kind, evals = self.eq_ind
            evals_0, evals_1 = evals.split_half()
            if kind == "pre":
                buf = self.dev_alloc.alloc(evals_0.len)
                self.hal.copy_d2d(evals_0, buf)
                evals_0 = buf
            split_n_vars = self.n_vars_remaining - 2

            def kernel(local_exec, log_chunks, buffers):
                local_exec.add_assign(split_n_vars - log_chunks, buffers[1], buffers[0])

            self.hal.map_kernels(kernel, [("chunked_mut", evals_0, 0), ("chunked", evals_1, 0)])
            self.eq_ind = ("post", evals_0)
        self.n_vars_remaining -= 1

    def finish(self):
        kind, _ = self.state
        if kind == "coeffs":
            raise RuntimeError("ExpectedFold")
        if self.n_vars_remaining != 0:
            raise RuntimeError("ExpectedExecution")
        out = []
        for _, ml in self.multilins:
            h = self.hal.copy_d2h(ml)
            out.append(int(h[0, 0]) | (int(h[0, 1]) << 64))
        out.append(self.eq_ind_prefix_eval)
        return out

